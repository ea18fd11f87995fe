// fp32 MFMA GEMM (see gemm32.h): 128x128x16 tile, 4 waves x (2x2) 32x32 accumulators,
// operands staged k-major in LDS so that the A/B fragment reads are conflict-free ds_read_b32.
#include "gemm32.h"

#include "n3dt_device.h"

#define G_BM 128
#define G_BN 128
#define G_BK 16
#define G_LD (G_BM + 4)

// stage a [128 rows x 16 k] operand tile into LDS as T[k][row]
__device__ __forceinline__ void g32_stage(float (*T)[G_LD], const float* __restrict__ P, long ld, int kmajor, int row0, int nrows,
                                          int k0, int kend, int tid) {
    if (!kmajor) {
        // rows are K-contiguous: thread -> (row, 8 consecutive k)
        const int row = tid >> 1, kq = (tid & 1) * 8;
        const int r = row0 + row;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        if (r < nrows) {
            const float* src = P + (long)r * ld + k0 + kq;
            if (k0 + kq + 8 <= kend && ((((size_t)src) & 15) == 0)) {
                f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + kq + j < kend) v[j] = src[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) T[kq + j][row] = v[j];
    } else {
        // k-major: thread -> (k row, 8 consecutive operand rows)
        const int kr = tid >> 4, c8 = (tid & 15) * 8;
        const int k = k0 + kr;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        if (k < kend) {
            const float* src = P + (long)k * ld + row0 + c8;
            if (row0 + c8 + 8 <= nrows && ((((size_t)src) & 15) == 0)) {
                f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (row0 + c8 + j < nrows) v[j] = src[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) T[kr][c8 + j] = v[j];
    }
}

__global__ __launch_bounds__(256) void gemm32_kernel(Gemm32 g) {
    __shared__ float As[G_BK][G_LD];
    __shared__ float Bs[G_BK][G_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // blockIdx.x walks the N tiles: the few column tiles that share one 128-row slab of A are dispatched back to
    // back, so the slab is fetched from HBM once and re-read from the Infinity Cache (A is the big operand here)
    const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;
    int kbeg = 0, kend = g.K;
    if (g.split_k > 1) {
        const int per = ((g.K + g.split_k - 1) / g.split_k + G_BK - 1) / G_BK * G_BK;
        kbeg = blockIdx.z * per;
        kend = min(g.K, kbeg + per);
        if (kbeg >= kend) return;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int k0 = kbeg; k0 < kend; k0 += G_BK) {
        __syncthreads();
        g32_stage(As, g.A, g.lda, g.a_kmajor, m0, g.M, k0, kend, tid);
        g32_stage(Bs, g.B, g.ldb, g.b_kmajor, n0, g.N, k0, kend, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < G_BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[kk + (lane >> 5)][wr * 64 + t * 32 + (lane & 31)];
                b[t] = Bs[kk + (lane >> 5)][wc * 64 + t * 32 + (lane & 31)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    const bool first_slice = (g.split_k <= 1) || blockIdx.z == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = acc[i][j][reg];
                if (g.bias && first_slice) v += g.bias[(g.bias_group_rows ? (long)(m / g.bias_group_rows) * g.bias_ld : 0) + n];
                if (g.act == G32_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (g.act == G32_ACT_LRELU) v = v > 0.0f ? v : 0.2f * v;
                if (g.gate_act != G32_ACT_NONE) {
                    const float y = g.gate[(long)m * g.ldgate + n];
                    if (g.gate_act == G32_ACT_RELU) v = y > 0.0f ? v : 0.0f;
                    else v = y > 0.0f ? v : 0.2f * v;
                }
                float* dst = g.C + (long)m * g.ldc + n;
                if (g.split_k > 1) atomicAdd(dst, v);
                else if (g.accumulate) *dst += v;
                else *dst = v;
            }
    }
}

void n3dt_gemm32(const Gemm32& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return;
    dim3 grid((g.N + G_BN - 1) / G_BN, (g.M + G_BM - 1) / G_BM, g.split_k > 1 ? g.split_k : 1);
    hipLaunchKernelGGL(gemm32_kernel, grid, dim3(256), 0, stream, g);
}


// ---------------------------------------------------------------------------------------------
// bf16-MFMA variant: same descriptor, fp32 operands in HBM converted to bf16 while staging, fp32
// accumulate and fp32 output (v_mfma_f32_32x32x16_bf16, 128x128x32 tiles).  Used by the training path
// when HeadNeRFNet.train_precision == "bf16": 16x the matrix rate of the exact-fp32 kernel, at bf16
// input rounding (gradients ~1e-2 relative).
// ---------------------------------------------------------------------------------------------
typedef __bf16 g16_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short g16_u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short g16_u16x4 __attribute__((ext_vector_type(4)));

#define H_BK 32
#define H_LD (H_BK + 8)

__device__ __forceinline__ unsigned short g16_cvt(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float g16_up(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// stage a [128 rows x 32 k] operand tile into LDS as T[row][k] (k contiguous), converting to bf16
__device__ __forceinline__ void g16_stage(unsigned short* T, const float* __restrict__ P, int is16, long ld, int kmajor, int row0,
                                          int nrows, int k0, int kend, int tid) {
    if (is16) {
        // bf16 storage: the same index maps, 8-byte loads, no conversion
        const unsigned short* P16 = reinterpret_cast<const unsigned short*>(P);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g16_u16x4 o = {0, 0, 0, 0};
            if (!kmajor) {
                const int idx = tid + 256 * i, row = idx >> 3, kq = (idx & 7) * 4;
                const int r = row0 + row;
                if (r < nrows) {
                    const unsigned short* src = P16 + (long)r * ld + k0 + kq;
                    if (k0 + kq + 4 <= kend && ((((size_t)src) & 7) == 0)) {
                        o = *reinterpret_cast<const g16_u16x4*>(src);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k0 + kq + j < kend) o[j] = src[j];
                    }
                }
                *reinterpret_cast<g16_u16x4*>(T + row * H_LD + kq) = o;
            } else {
                const int idx = tid + 256 * i, kr = idx >> 5, c4 = (idx & 31) * 4;
                const int k = k0 + kr;
                if (k < kend) {
                    const unsigned short* src = P16 + (long)k * ld + row0 + c4;
                    if (row0 + c4 + 4 <= nrows && ((((size_t)src) & 7) == 0)) {
                        o = *reinterpret_cast<const g16_u16x4*>(src);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (row0 + c4 + j < nrows) o[j] = src[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) T[(c4 + j) * H_LD + kr] = o[j];
            }
        }
        return;
    }
    if (!kmajor) {
        // memory rows are K-contiguous: thread -> 4 x (row, 4 consecutive k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, kq = (idx & 7) * 4;
            const int r = row0 + row;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < nrows) {
                const float* src = P + (long)r * ld + k0 + kq;
                if (k0 + kq + 4 <= kend && ((((size_t)src) & 15) == 0)) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + kq + j < kend) v[j] = src[j];
                }
            }
            g16_u16x4 o = {g16_cvt(v[0]), g16_cvt(v[1]), g16_cvt(v[2]), g16_cvt(v[3])};
            *reinterpret_cast<g16_u16x4*>(T + row * H_LD + kq) = o;
        }
    } else {
        // k-major memory: thread -> 4 x (k, 4 consecutive operand rows), scattered into T[row][k]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, kr = idx >> 5, c4 = (idx & 31) * 4;
            const int k = k0 + kr;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < kend) {
                const float* src = P + (long)k * ld + row0 + c4;
                if (row0 + c4 + 4 <= nrows && ((((size_t)src) & 15) == 0)) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (row0 + c4 + j < nrows) v[j] = src[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) T[(c4 + j) * H_LD + kr] = g16_cvt(v[j]);
        }
    }
}

__global__ __launch_bounds__(256) void gemm16_kernel(Gemm32 g) {
    __shared__ __attribute__((aligned(16))) unsigned short As[G_BM * H_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[G_BN * H_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;  // N tiles fastest, see gemm32_kernel
    int kbeg = 0, kend = g.K;
    if (g.split_k > 1) {
        const int per = ((g.K + g.split_k - 1) / g.split_k + H_BK - 1) / H_BK * H_BK;
        kbeg = blockIdx.z * per;
        kend = min(g.K, kbeg + per);
        if (kbeg >= kend) return;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int k0 = kbeg; k0 < kend; k0 += H_BK) {
        __syncthreads();
        g16_stage(As, g.A, g.a16, g.lda, g.a_kmajor, m0, g.M, k0, kend, tid);
        g16_stage(Bs, g.B, g.b16, g.ldb, g.b_kmajor, n0, g.N, k0, kend, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < H_BK / 16; ++ks) {
            g16_u16x8 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const g16_u16x8*>(As + (wr * 64 + t * 32 + (lane & 31)) * H_LD + 16 * ks + 8 * (lane >> 5));
                bf[t] = *reinterpret_cast<const g16_u16x8*>(Bs + (wc * 64 + t * 32 + (lane & 31)) * H_LD + 16 * ks + 8 * (lane >> 5));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g16_bf16x8, af[i]),
                                                                        __builtin_bit_cast(g16_bf16x8, bf[j]), acc[i][j], 0, 0, 0);
        }
    }
    const bool first_slice = (g.split_k <= 1) || blockIdx.z == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + (lane & 31);
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = acc[i][j][reg];
                if (g.bias && first_slice) v += g.bias[(g.bias_group_rows ? (long)(m / g.bias_group_rows) * g.bias_ld : 0) + n];
                if (g.act == G32_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (g.act == G32_ACT_LRELU) v = v > 0.0f ? v : 0.2f * v;
                if (g.gate_act != G32_ACT_NONE) {
                    const float y = g.gate16 ? g16_up(reinterpret_cast<const unsigned short*>(g.gate)[(long)m * g.ldgate + n])
                                             : g.gate[(long)m * g.ldgate + n];
                    if (g.gate_act == G32_ACT_RELU) v = y > 0.0f ? v : 0.0f;
                    else v = y > 0.0f ? v : 0.2f * v;
                }
                if (g.c16) {  // bf16 output (never split: a split product is a parameter gradient, fp32)
                    unsigned short* dst = reinterpret_cast<unsigned short*>(g.C) + (long)m * g.ldc + n;
                    if (g.accumulate) v += g16_up(*dst);
                    *dst = g16_cvt(v);
                } else {
                    float* dst = g.C + (long)m * g.ldc + n;
                    if (g.split_k > 1) atomicAdd(dst, v);
                    else if (g.accumulate) *dst += v;
                    else *dst = v;
                }
            }
    }
}

void n3dt_gemm(const Gemm32& g, int bf16, hipStream_t stream) {
    if (!bf16) {
        n3dt_gemm32(g, stream);
        return;
    }
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return;
    dim3 grid((g.N + G_BN - 1) / G_BN, (g.M + G_BM - 1) / G_BM, g.split_k > 1 ? g.split_k : 1);
    hipLaunchKernelGGL(gemm16_kernel, grid, dim3(256), 0, stream, g);
}
