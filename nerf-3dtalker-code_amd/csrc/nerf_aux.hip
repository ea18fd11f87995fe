// Small kernels around the fused render kernel: weight packing, per-frame latent folding and
// the per-ray head (block combine + RGB_layer_2 + background merge).
#include "n3dt_device.h"
#include "n3dt_layout.h"

// ---------------------------------------------------------------------------------------------
// Weight packing.  Logical matrices and their sources: n3dt_layout.h.
// fp32 fragment order (v_mfma_f32_16x16x4_f32 A operand, 4 k-steps per 16-byte load):
//   e = ((ot*(K/16) + k4)*64 + lane)*4 + j  ->  W'[ot*16 + (lane&15)][16*k4 + 4*j + (lane>>4)]
// 16-bit fragment order (v_mfma_f32_32x32x16_{bf16,f16} A operand whose k order matches an
// accumulator tile reused as the B operand, cdna guide section 3):
//   e = ((ot*(K/16) + ks)*64 + lane)*8 + j  ->  W'[ot*32 + (lane&31)][32*(ks>>1) + 16*(ks&1) + 8*(j>>2) + 4*(lane>>5) + (j&3)]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float logical_weight(const N3dtMlpParams& p, int stage, int row, int col, int S, int A, int U) {
    const int in0 = N3DT_PE_DIM + S + U, in5 = N3DT_PE_DIM + S + N3DT_HID, inr = N3DT_HID + A;
    switch (stage) {
        case 0: return col < N3DT_PE_DIM ? p.weight[0][(size_t)row * in0 + col] : 0.0f;
        case 5:
            if (col < N3DT_PE_DIM) return p.weight[5][(size_t)row * in5 + col];
            if (col == N3DT_PE_DIM) return 0.0f;
            return p.weight[5][(size_t)row * in5 + N3DT_PE_DIM + S + (col - 64)];
        case 8: return row == 0 ? p.weight[8][col] : 0.0f;
        case 10: return p.weight[10][(size_t)row * inr + col];
        default: return p.weight[stage][(size_t)row * N3DT_HID + col];
    }
}

// C[m][n] (+)= sum_k A(m,k) B(n,k) for small products (operands given by element strides; fp32 FMA): the merged
// RGB matrix W_m = Wr1[:, 0:384] Wr0 of the 16-bit kernels and its un-merge in the training backward.
// 32 x 32 tile per 256-thread workgroup, 2 x 2 results per thread -- a few hundred workgroups where the generic
// 128 x 128 GEMM would launch six.
__global__ __launch_bounds__(256) void small_gemm_kernel(int M, int N, int K, const float* __restrict__ A, long sam, long sak,
                                                                 const float* __restrict__ Bm, long sbn, long sbk,
                                                                 float* __restrict__ C, long ldc, int accumulate) {
    __shared__ float As[32][33], Bs[32][33];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // the next k-slab's 4 + 4 elements per thread are fetched into registers while the current one is multiplied: the loop
    // used to be load -> barrier -> 32 FMAs -> barrier, twelve exposed global-load latencies in a row (43 us for 192 x 384 x 384)
    float ra[4], rb[4];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = t + 256 * u;
            // pick the index that is contiguous in memory as the fast one
            const int a_r = sak == 1 ? i >> 5 : i & 31, a_k = sak == 1 ? i & 31 : i >> 5;
            const int b_r = sbk == 1 ? i >> 5 : i & 31, b_k = sbk == 1 ? i & 31 : i >> 5;
            ra[u] = (m0 + a_r < M && k0 + a_k < K) ? A[(long)(m0 + a_r) * sam + (long)(k0 + a_k) * sak] : 0.0f;
            rb[u] = (n0 + b_r < N && k0 + b_k < K) ? Bm[(long)(n0 + b_r) * sbn + (long)(k0 + b_k) * sbk] : 0.0f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = t + 256 * u;
            const int a_r = sak == 1 ? i >> 5 : i & 31, a_k = sak == 1 ? i & 31 : i >> 5;
            const int b_r = sbk == 1 ? i >> 5 : i & 31, b_k = sbk == 1 ? i & 31 : i >> 5;
            As[a_k][a_r] = ra[u];
            Bs[b_k][b_r] = rb[u];
        }
        __syncthreads();
        if (k0 + 32 < K) fetch(k0 + 32);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float a0 = As[k][ty], a1 = As[k][ty + 16], b0 = Bs[k][tx], b1 = Bs[k][tx + 16];
            acc[0][0] = fmaf(a0, b0, acc[0][0]);
            acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]);
            acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < N) C[(long)m * ldc + n] = (accumulate ? C[(long)m * ldc + n] : 0.0f) + acc[i][j];
        }
}
extern "C" void n3dt_launch_small_gemm(int M, int N, int K, const float* A, long sam, long sak, const float* B, long sbn, long sbk,
                                       float* C, long ldc, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(small_gemm_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, s, M, N, K, A, sam, sak, B, sbn, sbk, C, ldc,
                       accumulate);
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}

// 16-bit precisions stream ONE merged matrix for RGB_layer_0 -> RGB_layer_1: there is no activation between the
// two layers (models.py:79-81), so W_m = Wr1[:, 0:384] . Wr0 (192 x 384) replaces 12 + 6 out tiles by 6.  It is
// stored at stage 9's offset (right behind the density stage, keeping the stream contiguous); stage 10's slot stays
// unused.  The exact-fp32 kernel keeps the two layers separate (the reference's operation order).
// `order16` (16-bit precisions): 0 = 32x32x16 fragment order above, 1 = 16x16x32 order (nerf_fwd_x16b.hip):
//   e = ((t*(K/32) + ks)*64 + lane)*8 + j  ->  W'[t*16 + (lane&15)][32*ks + 16*(j>>2) + 4*(lane>>4) + (j&3)]
__global__ void pack_mlp_kernel(N3dtMlpParams p, int precision, int order16, int S, int A, int U, const float* __restrict__ wm,
                                unsigned char* __restrict__ out) {
    const int stage = blockIdx.y;
    const bool merged = precision != N3DT_F32 && stage == 9;
    if (precision != N3DT_F32 && stage == 10) return;
    N3dtStage st = n3dt_stage(stage);
    if (merged) st = n3dt_stage(10);  // 192 x 384
    const size_t n = (size_t)st.N * st.K;
    const size_t base = n3dt_stage_offset(stage);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        int row, col;
        if (precision == N3DT_F32) {
            int j = e & 3, lane = (e >> 2) & 63;
            size_t t = e >> 8;
            int k4 = (int)(t % (st.K / 16)), ot = (int)(t / (st.K / 16));
            row = ot * 16 + (lane & 15);
            col = 16 * k4 + 4 * j + (lane >> 4);
        } else if (order16 == 0) {
            int j = e & 7, lane = (e >> 3) & 63;
            size_t t = e >> 9;
            int ks = (int)(t % (st.K / 16)), ot = (int)(t / (st.K / 16));
            row = ot * 32 + (lane & 31);
            col = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
        } else {
            int j = e & 7, lane = (e >> 3) & 63;
            size_t t = e >> 9;
            int ks = (int)(t % (st.K / 32)), ot = (int)(t / (st.K / 32));
            row = ot * 16 + (lane & 15);
            col = 32 * ks + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        }
        float v;
        if (merged) {
            v = wm[(size_t)row * N3DT_HID + col];
        } else {
            v = logical_weight(p, stage, row, col, S, A, U);
        }
        if (precision == N3DT_F32) {
            reinterpret_cast<float*>(out)[base + e] = v;
        } else if (precision == N3DT_BF16) {
            reinterpret_cast<unsigned short*>(out)[base + e] = f32_to_bf16_rne(v);
        } else if (precision == N3DT_BF16X3) {
            // piece (e >> 9) of the one-product order becomes pieces 2P (hi) and 2P + 1 (lo); 512 elements per piece
            const size_t P = (base + e) >> 9, in = (base + e) & 511;
            const __bf16 hi = (__bf16)v;
            const __bf16 lo = (__bf16)(v - (float)hi);
            reinterpret_cast<unsigned short*>(out)[(2 * P) * 512 + in] = __builtin_bit_cast(unsigned short, hi);
            reinterpret_cast<unsigned short*>(out)[(2 * P + 1) * 512 + in] = __builtin_bit_cast(unsigned short, lo);
        } else {
            _Float16 hv = (_Float16)v;
            reinterpret_cast<unsigned short*>(out)[base + e] = __builtin_bit_cast(unsigned short, hv);
        }
    }
}

// W2^T [192][256] and b2 [256], fp32, after the matrices
__global__ void pack_tail_kernel(N3dtMlpParams p, float* __restrict__ tail) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N3DT_G * N3DT_C) {
        int j = i / N3DT_C, c = i % N3DT_C;
        tail[i] = p.weight[11][(size_t)c * N3DT_G + j];
    } else if (i < N3DT_G * N3DT_C + N3DT_C) {
        tail[i] = p.bias[11][i - N3DT_G * N3DT_C];
    } else if (i < N3DT_G * N3DT_C + N3DT_C + 12 * 8 * 2 * 64) {
        // B fragments of W2^T for the MFMA head: lane (column r31 of tile, k group h) holds k = 16 ks + 8 h + j, j = 0..7
        const int t = i - (N3DT_G * N3DT_C + N3DT_C);
        const int lane = t & 63, piece = t >> 6, part = piece & 1, tile = (piece >> 1) & 7, ks = piece >> 4;
        const int col = 32 * tile + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
        unsigned short* dst = reinterpret_cast<unsigned short*>(tail + n3dt_tail_w2_frags_offset()) + (size_t)piece * 512 + lane * 8;
        for (int j = 0; j < 8; ++j) {
            const float v = p.weight[11][(size_t)col * N3DT_G + k0 + j];
            const __bf16 hi = (__bf16)v;
            const __bf16 out = part == 0 ? hi : (__bf16)(v - (float)hi);
            dst[j] = __builtin_bit_cast(unsigned short, out);
        }
    }
}

extern "C" void n3dt_launch_pack(const N3dtGeom* g, int precision, const N3dtMlpParams* p, void* packed, hipStream_t stream) {
    float* tail = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(packed) + n3dt_packed_tail_offset(precision));
    // the merged RGB matrix W_m = Wr1[:, 0:384] Wr0 (fp32, behind W2^T and b2) is formed once and packed from there
    float* wm = tail + N3DT_G * N3DT_C + N3DT_C;
    if (precision != N3DT_F32)
        n3dt_launch_small_gemm(192, 384, 384, p->weight[10], 384 + g->appea_dim, 1, p->weight[9], 1, 384, wm, 384, 0, stream);
    hipLaunchKernelGGL(pack_mlp_kernel, dim3(64, N3DT_NSTAGE), dim3(256), 0, stream, *p, precision, 0, g->shape_dim, g->appea_dim,
                       g->audio_dim, wm, reinterpret_cast<unsigned char*>(packed));
    if (precision == N3DT_BF16 || precision == N3DT_F16)
        hipLaunchKernelGGL(pack_mlp_kernel, dim3(64, N3DT_NSTAGE), dim3(256), 0, stream, *p, precision, 1, g->shape_dim, g->appea_dim,
                           g->audio_dim, wm, reinterpret_cast<unsigned char*>(packed) + n3dt_packed_region_b_offset(precision));
    const int n = N3DT_G * N3DT_C + N3DT_C + 12 * 8 * 2 * 64;
    hipLaunchKernelGGL(pack_tail_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *p, tail);
}

// ---------------------------------------------------------------------------------------------
// Per-frame bias table (layout: n3dt_layout.h).  The shape / audio / appearance codes are constant
// over a frame, so their weight columns collapse into a bias (replaces the expand+concat of
// NetWorks/HeadNeRFNet.py:84,149-152 and models.py:69,80):
//   b0'[o]  = b0[o]  + W0[o, 63:63+S].shape + W0[o, 63+S:].audio
//   b5'[o]  = b5[o]  + W5[o, 63:63+S].shape
//   br1'[o] = br1[o] + Wr1[o, 384:].appea
// grid (B, N3DT_NSTAGE), block 384
// ---------------------------------------------------------------------------------------------
// One wavefront per output row: the lanes stride over the row's latent columns (coalesced 256-byte reads) and
// reduce with a butterfly; the fp32 summation ORDER therefore differs from a serial loop (the products are the same) --
// a few ulp on a bias, inside every mode's budget (the exact-fp32 gate is 1e-3 on RGB, measured 8e-6).
// grid (B, N3DT_NSTAGE, FOLD_ROWGROUPS), block 256 (4 waves).  The first version ran one THREAD per row (a serial
// 243-term loop over row-strided, uncoalesced loads): 23 us per call even at B = 1, now a launch latency.
#define FOLD_ROWGROUPS 24
__global__ __launch_bounds__(256) void fold_latents_kernel(N3dtMlpParams p, int S, int A, int U, int merged, const float* __restrict__ shape,
                                                           const float* __restrict__ appea, const float* __restrict__ audio,
                                                           float* __restrict__ fold) {
    const int b = blockIdx.x, stage = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float code[512 + N3DT_HID];
    const int in0 = N3DT_PE_DIM + S + U, in5 = N3DT_PE_DIM + S + N3DT_HID, inr = N3DT_HID + A;
    const N3dtStage st = n3dt_stage(stage);
    float* out = fold + (size_t)b * N3DT_FOLD_STRIDE + n3dt_bias_offset(stage);
    // the vector every row of this stage is dotted with, and where the row's matching columns start
    int n_code = 0, col0 = 0, ld = 0;
    if (stage == 0) { n_code = S + U; col0 = N3DT_PE_DIM; ld = in0; }
    else if (stage == 5) { n_code = S; col0 = N3DT_PE_DIM; ld = in5; }
    else if (stage == 10) {
        // merged RGB_layer_0 -> RGB_layer_1 bias: br1' + Wr1[:, 0:384] . br0 -- one dot of the whole row with [br0 ; appea]
        n_code = merged ? N3DT_HID + A : A;
        col0 = merged ? 0 : N3DT_HID;
        ld = inr;
    }
    for (int i = threadIdx.x; i < n_code; i += blockDim.x) {
        float v;
        if (stage == 10) {
            const int j = merged ? i - N3DT_HID : i;
            v = j < 0 ? p.bias[9][i] : appea[(size_t)b * A + j];
        } else {
            v = i < S ? shape[(size_t)b * S + i] : audio[(size_t)b * U + (i - S)];
        }
        code[i] = v;
    }
    __syncthreads();
    if (stage == 8) {  // density: one real row
        for (int o = blockIdx.z * 4 + wave; o < st.N; o += 4 * FOLD_ROWGROUPS)
            if (lane == 0) out[o] = o == 0 ? p.bias[8][0] : 0.0f;
        return;
    }
    // FOLD_UNR rows of a wave at a time: the kernel is a chain of load round trips (one per row when the rows went one by one),
    // so a wave keeps the loads of several rows in flight together
    constexpr int FOLD_UNR = 4;
    constexpr int RSTEP = 4 * FOLD_ROWGROUPS;
    for (int o0 = blockIdx.z * 4 + wave; o0 < st.N; o0 += RSTEP * FOLD_UNR) {
        float acc[FOLD_UNR];
#pragma unroll
        for (int u = 0; u < FOLD_UNR; ++u) acc[u] = 0.0f;
        if (n_code > 0) {
            const float* w[FOLD_UNR];
#pragma unroll
            for (int u = 0; u < FOLD_UNR; ++u) w[u] = p.weight[st.layer] + (size_t)min(o0 + RSTEP * u, st.N - 1) * ld + col0;
            for (int i = lane; i < n_code; i += 64) {
                const float cv = code[i];
#pragma unroll
                for (int u = 0; u < FOLD_UNR; ++u) acc[u] = fmaf(w[u][i], cv, acc[u]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int u = 0; u < FOLD_UNR; ++u) acc[u] += __shfl_xor(acc[u], off, 64);
        }
#pragma unroll
        for (int u = 0; u < FOLD_UNR; ++u) {
            const int o = o0 + RSTEP * u;
            if (lane == 0 && o < st.N) out[o] = p.bias[st.layer][o] + acc[u];
        }
    }
}

extern "C" void n3dt_launch_fold(const N3dtGeom* g, const N3dtMlpParams* p, const float* shape, const float* appea,
                                 const float* audio, float* fold, int merged_rgb, hipStream_t stream) {
    hipLaunchKernelGGL(fold_latents_kernel, dim3(g->batch, N3DT_NSTAGE, FOLD_ROWGROUPS), dim3(256), 0, stream, *p, g->shape_dim,
                       g->appea_dim, g->audio_dim, merged_rgb, shape, appea, audio, fold);
}

// ---------------------------------------------------------------------------------------------
// include_vd (NetWorks/HeadNeRFNet.py:56-63,86,141-142): RGB_layer_1's view-direction columns as a per-ray bias.
//   rayfold[ray][o] = fold[frame(ray)][bias_offset(10) + o] + ray_bias[ray][o]      (what the fused kernels read, n3dt_layout.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rayfold_kernel(int n_rays, long total, const float* __restrict__ fold, const float* __restrict__ ray_bias,
                                                      float* __restrict__ rayfold) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over rays x 48 (four outputs per thread)
    if (i >= total * 48) return;
    const long ray = i / 48;
    const int o = (int)(i % 48) * 4;
    const f32x4 f = *reinterpret_cast<const f32x4*>(fold + (size_t)(ray / n_rays) * N3DT_FOLD_STRIDE + n3dt_bias_offset(10) + o);
    const f32x4 r = *reinterpret_cast<const f32x4*>(ray_bias + (size_t)ray * N3DT_RAYFOLD_STRIDE + o);
    *reinterpret_cast<f32x4*>(rayfold + (size_t)ray * N3DT_RAYFOLD_STRIDE + o) = f32x4{f[0] + r[0], f[1] + r[1], f[2] + r[2], f[3] + r[3]};
}
extern "C" void n3dt_launch_rayfold(const N3dtGeom* g, const float* fold, const float* ray_bias, hipStream_t s) {
    const long total = (long)g->batch * g->n_rays;
    float* rayfold = const_cast<float*>(fold) + n3dt_rayfold_offset(g->batch);
    hipLaunchKernelGGL(rayfold_kernel, dim3((unsigned)((total * 48 + 255) / 256)), dim3(256), 0, s, g->n_rays, total, fold, ray_bias, rayfold);
}

// ray_bias[ray][o] = sum_j w_vd[o][j] * Embedder_4(d(ray))[j]: direction as GenSamplePoints (utils.py:149-153), the encoder's
// channel order [d, sin(2^0 d), cos(2^0 d), ..., sin(2^3 d), cos(2^3 d)] (utils.py:20-51), accurate sinf / cosf, the 27 products
// of an output summed in ascending j (fp32 FMAs).  One wave per ray: lane o and o + 64, o + 128 own three outputs.
__global__ __launch_bounds__(256) void ray_vd_bias_kernel(N3dtGeom g, const float* __restrict__ w_vd, long ld_w, const float* __restrict__ xy,
                                                          const float* __restrict__ R, const float* __restrict__ Kinv,
                                                          float* __restrict__ ray_bias) {
    __shared__ float wl[192 * 27];
    for (int i = threadIdx.x; i < 192 * 27; i += blockDim.x) wl[i] = w_vd[(size_t)(i / 27) * ld_w + i % 27];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long total = (long)g.batch * g.n_rays;
    for (long rayg = (long)blockIdx.x * 4 + wave; rayg < total; rayg += (long)gridDim.x * 4) {
        const int b = (int)(rayg / g.n_rays), ray = (int)(rayg % g.n_rays);
        const float x = xy[(int64_t)b * g.xy_stride_b + 0 * g.xy_stride_c + (int64_t)ray * g.xy_stride_r];
        const float y = xy[(int64_t)b * g.xy_stride_b + 1 * g.xy_stride_c + (int64_t)ray * g.xy_stride_r];
        float d[3], l;
        n3dt_ray_setup(R + b * 9, Kinv + b * 9, x, y, d, l);
        float pe[27];
        pe[0] = d[0]; pe[1] = d[1]; pe[2] = d[2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float a = d[i] * (float)(1 << k);
                pe[3 + 6 * k + i] = sinf(a);
                pe[3 + 6 * k + 3 + i] = cosf(a);
            }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int o = lane + 64 * u;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 27; ++j) acc = fmaf(wl[o * 27 + j], pe[j], acc);
            ray_bias[(size_t)rayg * N3DT_RAYFOLD_STRIDE + o] = acc;
        }
    }
}
extern "C" void n3dt_launch_ray_vd_bias(const N3dtGeom* g, const float* w_vd, long ld_w, const float* xy, const float* R, const float* Kinv,
                                        float* ray_bias, hipStream_t s) {
    const long total = (long)g->batch * g->n_rays;
    long grid = (total + 3) / 4;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(ray_vd_bias_kernel, dim3((unsigned)grid), dim3(256), 0, s, *g, w_vd, ld_w, xy, R, Kinv, ray_bias);
}

// ---------------------------------------------------------------------------------------------
// Per-ray head.  Combines the per-block partials front to back (the transmittance of a later
// block is scaled by the product of the earlier blocks' (1 - alpha + 1e-10) factors: the same
// cumprod as NetWorks/utils.py:283-287, re-associated), then applies RGB_layer_2 once per ray:
//   fg[c] = sum_s w_s (W2 g_s + b2)[c] = W2[c,:] . (sum_s w_s g_s) + b2[c] * sum_s w_s
// (models.py:82 is linear for featmap_nc != 3), and merges the learned background
// (HeadNeRFNet.py:103-112): merge = fg + (1 - sum w) * bg_featmap.
// 8 rays per 256-thread workgroup, thread = output channel.
// ---------------------------------------------------------------------------------------------
#define HEAD_RAYS 8
#define HEAD_MAX_BPR 64  // n_samples <= 1024 (n3dt_api.hip) over 16-sample blocks
__global__ __launch_bounds__(256) void ray_head_kernel(N3dtGeom g, int bpr, int bs, const float* __restrict__ part,
                                                       const float* __restrict__ wlocal, const float* __restrict__ tail,
                                                       const float* __restrict__ bg_featmap, int bg_hwc, float* __restrict__ fg_feat,
                                                       float* __restrict__ bg_alpha, float* __restrict__ depth,
                                                       float* __restrict__ weight, float* __restrict__ merge_feat,
                                                       float* __restrict__ rayrec) {
    __shared__ float G[HEAD_RAYS][N3DT_G];
    __shared__ float pref[HEAD_RAYS][HEAD_MAX_BPR];
    __shared__ float wsum_s[HEAD_RAYS], dsum_s[HEAD_RAYS];
    const long nrays_total = (long)g.batch * g.n_rays;
    const long ray0 = (long)blockIdx.x * HEAD_RAYS;
    const int t = threadIdx.x;
    if (t < HEAD_RAYS) {
        long rg = ray0 + t;
        float Trun = 1.0f, ws = 0.0f, ds = 0.0f;
        if (rg < nrays_total) {
            for (int k = 0; k < bpr; ++k) {
                const float* po = part + ((size_t)rg * bpr + k) * N3DT_PART_STRIDE + N3DT_G;
                pref[t][k] = Trun;
                ws += Trun * po[0];
                ds += Trun * po[1];
                Trun *= po[2];
            }
        }
        wsum_s[t] = ws;
        dsum_s[t] = ds;
    }
    __syncthreads();
    for (int i = t; i < HEAD_RAYS * N3DT_G; i += 256) {
        int r = i / N3DT_G, j = i % N3DT_G;
        long rg = ray0 + r;
        float acc = 0.0f;
        if (rg < nrays_total)
            for (int k = 0; k < bpr; ++k) acc += pref[r][k] * part[((size_t)rg * bpr + k) * N3DT_PART_STRIDE + j];
        G[r][j] = acc;
        // training: the per-ray record the head backward reads (composited RGB_layer_1 sums, sum of weights, depth, T)
        if (rayrec && rg < nrays_total) rayrec[(size_t)rg * N3DT_PART_STRIDE + j] = acc;
    }
    __syncthreads();
    if (rayrec && t < HEAD_RAYS && ray0 + t < nrays_total) {
        float* rec = rayrec + (size_t)(ray0 + t) * N3DT_PART_STRIDE + N3DT_G;
        rec[0] = wsum_s[t];
        rec[1] = dsum_s[t];
        rec[2] = 0.0f;
        rec[3] = 0.0f;
    }
    const float* W2T = tail;
    const float b2 = tail[N3DT_G * N3DT_C + t];
    float acc[HEAD_RAYS];
#pragma unroll
    for (int r = 0; r < HEAD_RAYS; ++r) acc[r] = 0.0f;
    for (int j = 0; j < N3DT_G; ++j) {
        float wv = W2T[j * N3DT_C + t];
#pragma unroll
        for (int r = 0; r < HEAD_RAYS; ++r) acc[r] = fmaf(wv, G[r][j], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < HEAD_RAYS; ++r) {
        long rg = ray0 + r;
        if (rg >= nrays_total) break;
        float fg = acc[r] + b2 * wsum_s[r];
        float ba = 1.0f - wsum_s[r];
        if (fg_feat) fg_feat[(size_t)rg * N3DT_C + t] = fg;
        if (merge_feat) {
            int ray = (int)(rg % g.n_rays);
            // bg_hwc: the background map transposed to [N_r][C] (coalesced here); else the parameter's own [C][N_r]
            merge_feat[(size_t)rg * N3DT_C + t] = fg + ba * (bg_hwc ? bg_featmap[(size_t)ray * N3DT_C + t] : bg_featmap[(size_t)t * g.n_rays + ray]);
        }
        if (t == 0) {
            if (bg_alpha) bg_alpha[rg] = ba;
            if (depth) depth[rg] = dsum_s[r];
        }
    }
    if (weight) {
        for (int i = t; i < HEAD_RAYS * g.n_samples; i += 256) {
            int r = i / g.n_samples, s = i % g.n_samples;
            long rg = ray0 + r;
            if (rg < nrays_total) {
                int k = s / bs;
                weight[(size_t)rg * g.n_samples + s] = pref[r][k] * wlocal[((size_t)rg * bpr + k) * bs + (s % bs)];
            }
        }
    }
}

// The same head on the matrix pipe (16-bit render modes): 32 rays per workgroup; RGB_layer_2 as a 32 x 192 x 256 product on
// v_mfma_f32_32x32x16_bf16 with BOTH factors split into bf16 hi + lo (three products: hi*hi + hi*lo + lo*hi), i.e. ~16
// mantissa bits -- far inside the 16-bit modes' error budget; the fp32 parity mode keeps the FMA kernel above.
// The FMA kernel is VALU-bound (1 536 FMAs per thread): 100 us per 32 768 rays, 2 % of the render step.
typedef __bf16 rh_bf16x8 __attribute__((ext_vector_type(8)));
#define RH_RAYS 32
// __launch_bounds__(256, 4): four waves per SIMD = four workgroups per CU (128 registers, no spill; hipcc took 160 unasked, three
// workgroups per CU: 2 048 workgroups at 16 frames then ran in 2.7 rounds instead of 2): 64 -> 55 us per 65 536 rays.
__global__ __launch_bounds__(256, 4) void ray_head_mfma_kernel(N3dtGeom g, int bpr, int bs, const float* __restrict__ part,
                                                            const float* __restrict__ wlocal, const float* __restrict__ tail,
                                                            const float* __restrict__ bg_featmap, int bg_hwc,
                                                            float* __restrict__ fg_feat, float* __restrict__ bg_alpha,
                                                            float* __restrict__ depth, float* __restrict__ weight,
                                                            float* __restrict__ merge_feat, float* __restrict__ rayrec) {
    __shared__ __attribute__((aligned(16))) float G[RH_RAYS][N3DT_G + 4];  // +4: 16-byte aligned rows off the bank stride
    __shared__ float pref[RH_RAYS][HEAD_MAX_BPR];
    __shared__ float wsum_s[RH_RAYS], dsum_s[RH_RAYS];
    const long nrays_total = (long)g.batch * g.n_rays;
    const long ray0 = (long)blockIdx.x * RH_RAYS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < RH_RAYS) {
        const long rg = ray0 + t;
        float Trun = 1.0f, ws = 0.0f, ds = 0.0f;
        if (rg < nrays_total) {
            for (int k = 0; k < bpr; ++k) {
                const float* po = part + ((size_t)rg * bpr + k) * N3DT_PART_STRIDE + N3DT_G;
                pref[t][k] = Trun;
                ws += Trun * po[0];
                ds += Trun * po[1];
                Trun *= po[2];
            }
        }
        wsum_s[t] = ws;
        dsum_s[t] = ds;
    }
    __syncthreads();
    // G[ray][0:192] = sum_k pref[ray][k] * part[ray][k][0:192].  A workgroup's run time is a chain of memory round trips (38 us for
    // ONE workgroup at one head; diagnostic builds: this phase 13 us of them, the epilogue 10), so the phase issues everything it
    // needs at once: 16-byte loads (rows are 784 B apart: aligned), six quads per thread, all blocks of a ray, fully unrolled for
    // the usual one or two blocks per ray.
    static_assert(N3DT_G % 4 == 0 && (N3DT_PART_STRIDE * 4) % 16 == 0 && RH_RAYS * (N3DT_G / 4) % 256 == 0, "whole quads");
    constexpr int QPR = N3DT_G / 4, NQ = RH_RAYS * QPR / 256;  // quads per ray row, quads per thread
    if (bpr <= 2) {
        f32x4 v[NQ][2];
        float pf[NQ][2];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = t + 256 * u, r = i / QPR, q = i % QPR;
            const long rg = min(ray0 + r, nrays_total - 1);
            const float* p0 = part + (size_t)rg * bpr * N3DT_PART_STRIDE + 4 * q;
            v[u][0] = *reinterpret_cast<const f32x4*>(p0);
            v[u][1] = *reinterpret_cast<const f32x4*>(p0 + (bpr == 2 ? N3DT_PART_STRIDE : 0));
            pf[u][0] = pref[r][0];
            pf[u][1] = bpr == 2 ? pref[r][1] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = t + 256 * u, r = i / QPR, q = i % QPR;
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = pf[u][0] * v[u][0][e] + pf[u][1] * v[u][1][e];
            const bool in = ray0 + r < nrays_total;
            if (!in) a = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(&G[r][4 * q]) = a;
            if (rayrec && in) *reinterpret_cast<f32x4*>(rayrec + (size_t)(ray0 + r) * N3DT_PART_STRIDE + 4 * q) = a;
        }
    } else {
        for (int i = t; i < RH_RAYS * N3DT_G; i += 256) {
            const int r = i / N3DT_G, j = i % N3DT_G;
            const long rg = ray0 + r;
            float acc = 0.0f;
            if (rg < nrays_total)
                for (int k = 0; k < bpr; ++k) acc += pref[r][k] * part[((size_t)rg * bpr + k) * N3DT_PART_STRIDE + j];
            G[r][j] = acc;
            if (rayrec && rg < nrays_total) rayrec[(size_t)rg * N3DT_PART_STRIDE + j] = acc;
        }
    }
    __syncthreads();
    if (rayrec && t < RH_RAYS && ray0 + t < nrays_total) {
        float* rec = rayrec + (size_t)(ray0 + t) * N3DT_PART_STRIDE + N3DT_G;
        rec[0] = wsum_s[t];
        rec[1] = dsum_s[t];
        rec[2] = 0.0f;
        rec[3] = 0.0f;
    }
    // fg[ray][c] = sum_j G[ray][j] W2T[j][c]: wave w owns output channels 64 w .. 64 w + 63 (two 32-column tiles)
    const int r31 = lane & 31, h = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const rh_bf16x8* w2frags = reinterpret_cast<const rh_bf16x8*>(tail + n3dt_tail_w2_frags_offset());
#pragma unroll 2
    for (int ks = 0; ks < N3DT_G / 16; ++ks) {
        const int k0 = 16 * ks + 8 * h;
        rh_bf16x8 a_hi, a_lo;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&G[r31][k0]), g1 = *reinterpret_cast<const f32x4*>(&G[r31][k0 + 4]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? g0[j] : g1[j - 4];
            const __bf16 hi = (__bf16)v;
            a_hi[j] = hi;
            a_lo[j] = (__bf16)(v - (float)hi);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // W2^T as pre-split bf16 hi / lo B fragments (pack_tail_kernel): two 16-byte loads instead of 8 strided
            // dwords and their conversions per fragment
            const rh_bf16x8* fr = w2frags + ((size_t)(ks * 8 + 2 * wave + i) * 2) * 64 + lane;
            const rh_bf16x8 b_hi = fr[0], b_lo = fr[64];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[i], 0, 0, 0);
        }
    }
    // accumulator tile: column (output channel) on the lane, rays in the registers
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = 64 * wave + 32 * i + r31;
        const float b2 = tail[N3DT_G * N3DT_C + c];
        // (the sixteen background values first, from clamped ray indices, then the stores: no load waits behind a store)
        float bgv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long rg = min(ray0 + rr, nrays_total - 1);
            const int ray = (int)(rg % g.n_rays);
            bgv[r] = merge_feat ? (bg_hwc ? bg_featmap[(size_t)ray * N3DT_C + c] : bg_featmap[(size_t)c * g.n_rays + ray]) : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long rg = ray0 + rr;
            if (rg >= nrays_total) continue;
            const float fg = acc[i][r] + b2 * wsum_s[rr];
            if (fg_feat) fg_feat[(size_t)rg * N3DT_C + c] = fg;
            if (merge_feat) merge_feat[(size_t)rg * N3DT_C + c] = fg + (1.0f - wsum_s[rr]) * bgv[r];
        }
    }
    if (t < RH_RAYS && ray0 + t < nrays_total) {
        if (bg_alpha) bg_alpha[ray0 + t] = 1.0f - wsum_s[t];
        if (depth) depth[ray0 + t] = dsum_s[t];
    }
    if (weight) {
        for (int i = t; i < RH_RAYS * g.n_samples; i += 256) {
            const int r = i / g.n_samples, s = i % g.n_samples;
            const long rg = ray0 + r;
            if (rg < nrays_total) {
                const int k = s / bs;
                weight[(size_t)rg * g.n_samples + s] = pref[r][k] * wlocal[((size_t)rg * bpr + k) * bs + (s % bs)];
            }
        }
    }
}

extern "C" void n3dt_launch_ray_head_mfma(const N3dtGeom* g, int bpr, int bs, const float* part, const float* wlocal,
                                          const float* tail, const float* bg_featmap, int bg_hwc, float* fg_feat, float* bg_alpha,
                                          float* depth, float* weight, float* merge_feat, float* rayrec, hipStream_t stream) {
    const long nrays_total = (long)g->batch * g->n_rays;
    const int grid = (int)((nrays_total + RH_RAYS - 1) / RH_RAYS);
    hipLaunchKernelGGL(ray_head_mfma_kernel, dim3(grid), dim3(256), 0, stream, *g, bpr, bs, part, wlocal, tail,
                       merge_feat ? bg_featmap : nullptr, bg_hwc, fg_feat, bg_alpha, depth, weight, merge_feat, rayrec);
}

extern "C" void n3dt_launch_ray_head(const N3dtGeom* g, int bpr, int bs, const float* part, const float* wlocal,
                                     const float* tail, const float* bg_featmap, int bg_hwc, float* fg_feat, float* bg_alpha,
                                     float* depth, float* weight, float* merge_feat, hipStream_t stream) {
    const long nrays_total = (long)g->batch * g->n_rays;
    const int grid = (int)((nrays_total + HEAD_RAYS - 1) / HEAD_RAYS);
    hipLaunchKernelGGL(ray_head_kernel, dim3(grid), dim3(256), 0, stream, *g, bpr, bs, part, wlocal, tail,
                       merge_feat ? bg_featmap : nullptr, bg_hwc, fg_feat, bg_alpha, depth, weight, merge_feat, (float*)nullptr);
}

// training forward: also leaves the per-ray record [R][N3DT_PART_STRIDE] and the global sample weights behind
extern "C" void n3dt_launch_ray_head_rec(const N3dtGeom* g, int bpr, int bs, const float* part, const float* wlocal,
                                         const float* tail, const float* bg_featmap, float* fg_feat, float* bg_alpha, float* depth,
                                         float* weight, float* merge_feat, float* rayrec, hipStream_t stream) {
    // (the fused mixed-precision training forward: RGB_layer_2 on the matrix pipe with both factors split hi + lo, as in the 16-bit
    //  render modes -- the FMA form took 41 us for 8 192 rays, 1 536 FMAs per thread; N3DT_TRAIN_HEAD_FMA=1 keeps it)
    const long nrays_total = (long)g->batch * g->n_rays;
    static const bool fma = [] {
        const char* e = getenv("N3DT_TRAIN_HEAD_FMA");
        return e && atoi(e) != 0;
    }();
    if (!fma) {
        const int grid = (int)((nrays_total + RH_RAYS - 1) / RH_RAYS);
        hipLaunchKernelGGL(ray_head_mfma_kernel, dim3(grid), dim3(256), 0, stream, *g, bpr, bs, part, wlocal, tail,
                           merge_feat ? bg_featmap : nullptr, 0, fg_feat, bg_alpha, depth, weight, merge_feat, rayrec);
        return;
    }
    const int grid = (int)((nrays_total + HEAD_RAYS - 1) / HEAD_RAYS);
    hipLaunchKernelGGL(ray_head_kernel, dim3(grid), dim3(256), 0, stream, *g, bpr, bs, part, wlocal, tail,
                       merge_feat ? bg_featmap : nullptr, 0, fg_feat, bg_alpha, depth, weight, merge_feat, rayrec);
}

// [C][n] -> [n][C]
__global__ void chw_to_hwc_kernel(int C, int n, const float* __restrict__ src, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < n) ? src[(size_t)c * n + p] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int p = p0 + i, c = c0 + tx;
        if (c < C && p < n) dst[(size_t)p * C + c] = tile[tx][i];
    }
}

extern "C" void n3dt_launch_chw_to_hwc(int C, int n, const float* src, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(chw_to_hwc_kernel, dim3((n + 31) / 32, (C + 31) / 32), dim3(256), 0, stream, C, n, src, dst);
}

// ---------------------------------------------------------------------------------------------
// Hierarchical sample planes.  FineSample.forward, NetWorks/utils.py:211-254, one wave per ray:
//   pdf over the interior coarse weights w[1 .. Nc-2] (+1e-5 in the normaliser only), cdf = [0, cumsum(pdf)]
//   (Nc-1 entries), Nf+1 uniform samples u (linspace, or the caller's torch.rand in train mode),
//   inds = searchsorted(cdf, u, right=True), below = max(0, inds-1), above = min(Nc-2, inds), bins = midpoints of
//   the coarse planes, t = (u - cdf[below]) / (cdf[above] - cdf[below], 1 where < 1e-5),
//   z = bins[below] + t (bins[above] - bins[below]);  all planes = sort(coarse planes ++ z).
// The cumulative sum runs sequentially like torch.cumsum; the sort is a stable rank sort (<= 2 K elements per ray).
// ---------------------------------------------------------------------------------------------
#define FS_MAX_PLANES 2048
__global__ __launch_bounds__(64) void fine_sample_kernel(N3dtGeom g, int n_fine, const float* __restrict__ weight,
                                                         const float* __restrict__ T, const float* __restrict__ t_rand,
                                                         const float* __restrict__ u_in, float* __restrict__ z_out) {
    extern __shared__ float fs_lds[];
    const int Nc = g.n_samples, Nu = n_fine + 1, Nall = Nc + Nu;
    float* zc = fs_lds;           // [Nc] coarse planes (the first Nc of the Nc+1 edges, utils.py:83)
    float* cdf = zc + Nc;         // [Nc - 1]
    float* all = cdf + (Nc - 1);  // [Nall]
    const long rayg = blockIdx.x;
    const int lane = threadIdx.x;
    const int b = (int)(rayg / g.n_rays);
    const float rz1 = T[b * 3 + 2] - g.world_z1, rz2 = T[b * 3 + 2] - g.world_z2;
    const float* tr = t_rand ? t_rand + rayg * (Nc + 1) : nullptr;
    const float* w = weight + rayg * Nc;
    for (int j = lane; j < Nc; j += 64) {
        // z_planes_given: the caller holds the coarse planes (the fine_samp_func seam gets them in coarse_sample_dict["zvals"])
        const float z = g.z_planes_given ? tr[j] : n3dt_edge_z(rz1, rz2, j, Nc, tr);
        zc[j] = z;
        all[j] = z;
    }
    float part = 0.0f;
    for (int j = 1 + lane; j < Nc - 1; j += 64) part += w[j] + 1e-5f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    const float total = part;
    __syncthreads();
    if (lane == 0) {
        float run = 0.0f;
        cdf[0] = 0.0f;
        for (int j = 0; j < Nc - 2; ++j) {
            run += w[j + 1] / total;
            cdf[j + 1] = run;
        }
    }
    __syncthreads();
    for (int i = lane; i < Nu; i += 64) {
        const float u = u_in ? u_in[rayg * Nu + i] : n3dt_linspace01(i, Nu);
        int lo = 0, hi = Nc - 1;  // upper bound: number of cdf entries <= u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1;
            else hi = mid;
        }
        const int inds = lo;
        const int below = inds - 1 > 0 ? inds - 1 : 0;
        const int above = inds < Nc - 2 ? inds : Nc - 2;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = 0.5f * (zc[below + 1] + zc[below]), b1 = 0.5f * (zc[above + 1] + zc[above]);
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - c0) / denom;
        all[Nc + i] = b0 + t * (b1 - b0);
    }
    __syncthreads();
    float* out = z_out + rayg * Nall;
    for (int i = lane; i < Nall; i += 64) {
        const float v = all[i];
        int rank = 0;
        for (int j = 0; j < Nall; ++j) {
            const float o = all[j];
            rank += (o < v || (o == v && j < i)) ? 1 : 0;
        }
        out[rank] = v;
    }
}

extern "C" void n3dt_launch_fine_sample(const N3dtGeom* g, int n_fine, const float* weight, const float* T, const float* t_rand,
                                        const float* u, float* z_planes, hipStream_t stream) {
    const long rays = (long)g->batch * g->n_rays;
    const size_t lds = sizeof(float) * ((size_t)g->n_samples * 2 - 1 + g->n_samples + n_fine + 1);
    hipLaunchKernelGGL(fine_sample_kernel, dim3((unsigned)rays), dim3(64), lds, stream, *g, n_fine, weight, T, t_rand, u, z_planes);
}

// ---- n3dt_stage_inputs: up to N3DT_STAGE_MAX small copies in one launch (blockIdx.y = entry) ------------------------
__global__ void stage_kernel(N3dtStageCopy st) {
    const int e = blockIdx.y;
    if (e >= st.n) return;
    const float* __restrict__ src = st.src[e];
    float* __restrict__ dst = st.dst[e];
    const long n = st.count[e];
    const bool view = e == 0 && st.view_dims[0] > 0;
    const long d1 = st.view_dims[1], d2 = st.view_dims[2];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long j = i;
        if (view) {
            const long i2 = i % d2, i1 = (i / d2) % d1, i0 = i / (d2 * d1);
            j = i0 * st.view_strides[0] + i1 * st.view_strides[1] + i2 * st.view_strides[2];
        }
        dst[i] = src[j];
    }
}

extern "C" void n3dt_launch_stage(const N3dtStageCopy* st, hipStream_t stream) {
    long most = 1;
    for (int i = 0; i < st->n; ++i) most = st->count[i] > most ? st->count[i] : most;
    const int bx = (int)((most + 255) / 256 < 64 ? (most + 255) / 256 : 64);
    hipLaunchKernelGGL(stage_kernel, dim3(bx, st->n), dim3(256), 0, stream, *st);
}
