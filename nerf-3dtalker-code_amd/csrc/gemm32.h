// Generic fp32 MFMA GEMM used by the training path (forward-with-saved-activations and backward).
//   C[M,N] (+)= epi( sum_k A(m,k) * B(k,n) )
// Operands are described by (pointer, leading dimension, k_major):
//   k_major = 0 : element (row r, k) at ptr[r*ld + k]   (e.g. activations X[M,K], weights W[N,K])
//   k_major = 1 : element (row r, k) at ptr[k*ld + r]   (e.g. dY[M_pts, N] used as A(n, m_pt) in dW = dY^T X)
// so the three products of a linear layer are one kernel:
//   forward  y  = x W^T      : A = x (0), B = W (0)
//   backward dx = dy W       : A = dy (0), B = W (1)   [k = n]
//   backward dW = dy^T x     : A = dy (1), B = x (1)   [k = point index, split over grid.z with atomics]
// v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain, so the training path has fp32 numerics.
#pragma once
#include <hip/hip_runtime.h>

#define G32_ACT_NONE 0
#define G32_ACT_RELU 1
#define G32_ACT_LRELU 2  // slope 0.2

struct Gemm32 {
    int M, N, K;
    const float* A;
    long lda;
    int a_kmajor;
    const float* B;
    long ldb;
    int b_kmajor;
    float* C;
    long ldc;
    // bias[(m / bias_group_rows) * bias_ld + n]; bias_group_rows == 0 -> bias[n]; bias == nullptr -> none
    const float* bias;
    int bias_group_rows;
    long bias_ld;
    int act;  // applied to (acc + bias)
    // multiply the result by act'(gate[m*ldgate + n]) (gate = the forward OUTPUT of the layer being
    // differentiated: relu' = gate > 0, lrelu' = gate > 0 ? 1 : 0.2); gate_act == G32_ACT_NONE -> no gating
    const float* gate;
    long ldgate;
    int gate_act;
    int accumulate;  // C += result instead of C = result
    int split_k;     // >1: grid.z slices of K, combined with atomicAdd (C must be pre-initialised)
    // bf16 STORAGE (gemm16 only; the pointers above then address bf16 elements and are passed reinterpret_cast'ed):
    // the neural renderer's mixed-precision training keeps its maps and their gradients as bf16 in HBM
    int a16, b16, c16, gate16;
};

void n3dt_gemm32(const Gemm32& g, hipStream_t stream);
// bf16 != 0: same product on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 while staging, fp32 accumulate/output)
void n3dt_gemm(const Gemm32& g, int bf16, hipStream_t stream);
