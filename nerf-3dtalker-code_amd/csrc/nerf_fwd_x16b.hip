// Fused volumetric-render kernel, 16-bit MFMA mode, 16x16x32 tiling (v_mfma_f32_16x16x32_{bf16,f16}).
//
// The same algorithm, weight stream and HBM traffic as nerf_fwd_x16.hip, re-tiled: a wavefront carries one block of
// 32 consecutive samples of a ray as TWO 16-sample column tiles; a weight piece (1 KiB, one A fragment) is now
// 16 output rows x 32 k and feeds two MFMAs, one per sample tile -- the same LDS bytes and MFMA cycles per FLOP as the
// 32x32x16 tiling.  Why it exists: under load the chip holds a higher clock on this MFMA shape (measured 1.12-1.15x
// the FLOP/s on random data at equal cycles, MI355X_MICROARCH.md "DVFS give-back" item 7), and this kernel is
// bounded by exactly that clock.
//
// Layouts (c = lane & 15, q = lane >> 4):
//   activations  B fragment of k-step s (channels 32s .. 32s+31), sample tile nb: element j =
//                H[channel 32s + 16(j>>2) + 4q + (j&3)][sample 16 nb + c]
//                = registers 0..3 of the accumulator tiles 2s (j < 4) and 2s+1 (j >= 4): a pair of finished 16-row
//                output tiles IS the next layer's B fragment, no lane movement
//   weights      piece (t, s): lane (r = lane & 15, q) element j = W'[16t + r][32s + 16(j>>2) + 4q + (j&3)]
//                (second matrix region of the packed buffer, nerf_aux.hip)
//   accumulator  tile t, sample tile nb: register r = H[16t + 4q + r][16 nb + c]
#include <stdlib.h>

#include "x16_core.h"

template <int N>
__device__ __forceinline__ float butterfly_n(float (&v)[N], const int c) {
    // sum over the N lanes of a group (lane index c within the group) of N per-lane values: lane c ends with the
    // value whose index is the bit reversal of c
    int m = N >> 1;
#pragma unroll
    for (int n = N; n > 1; n >>= 1) {
        const bool bit = (c & m) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float keep = bit ? v[2 * i + 1] : v[2 * i];
            const float send = bit ? v[2 * i] : v[2 * i + 1];
            v[i] = keep + __shfl_xor(send, m, 64);
        }
        m >>= 1;
    }
    return v[0];
}

enum { B_HIDDEN = 0, B_DENSITY = 2, B_COMPOSITE = 3 };

// One stage: out[16 NT x 32] = W'[16 NT x 32 KS] . in + bias (+ ReLU).  The KPE leading k-steps take their B operand
// from the positional-encoding fragments (registers, stage L0; the wave's LDS copy, skip stage L5).
template <int PREC, int WAVES, int KS, int KPE, int NT, int MODE>
__device__ __forceinline__ void x16b_stage(WeightStream<PREC, WAVES>& ws, const float* __restrict__ bias,
                                           const typename X16<PREC>::frag (&pe_reg)[2][2], const unsigned char* pe_lds,
                                           const typename X16<PREC>::frag (&hin)[2][12], typename X16<PREC>::frag (&hout)[2][12],
                                           float (&aux)[2], float* po, const bool live, const int lane) {
    typedef typename X16<PREC>::frag frag;
    const int c = lane & 15, q = lane >> 4;
    const frag ones = X16<PREC>::ones_frag();
    float bias_cur = bias[c];
    f32x4 prev[2];   // the even tile of a pair, waiting for its odd partner
    float red[16];   // composite: 4 tiles x 4 registers, reduced over the samples together
    static_for<0, NT>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        f32x4 acc[2];
        {
            // acc = bias, broadcast over the samples, by one extra MFMA (hi/lo split): lane r of the q == 0 group
            // holds bias[16t + r], fetched one tile ahead
            const frag bf = X16<PREC>::bias_frag(bias_cur, q == 0);
            if (t + 1 < NT) bias_cur = bias[(t + 1) * 16 + c];
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            const f32x4 binit = X16<PREC>::mfma16(bf, ones, zero);
            acc[0] = binit;
            acc[1] = binit;
        }
        static_for<0, KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            const frag a_cur = ws.template next<MODE == B_COMPOSITE, NT * KS, t * KS + ks>();
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                frag b;
                if (ks < KPE) {
                    if (pe_lds) b = *reinterpret_cast<const frag*>(pe_lds + (nb * 2 + ks) * X16_PIECE);
                    else b = pe_reg[nb][ks < 2 ? ks : 0];
                } else {
                    b = hin[nb][ks >= KPE ? ks - KPE : 0];
                }
                acc[nb] = X16<PREC>::mfma16(a_cur, b, acc[nb]);
            }
        });
        if (MODE == B_HIDDEN) {
            if ((t & 1) == 0) {
                prev[0] = acc[0];
                prev[1] = acc[1];
            } else {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = prev[nb][r];
                        v[4 + r] = acc[nb][r];
                    }
                    hout[nb][t >> 1] = X16<PREC>::relu(X16<PREC>::pack(v), 0);
                }
            }
        } else if (MODE == B_DENSITY) {
            if (t == 0) {
                aux[0] = acc[0][0];  // row 0 of the tile, valid on lanes with q == 0
                aux[1] = acc[1][0];
            }
        } else {
            // weighted RGB_layer_1 activations, summed over the 32 samples: both sample tiles per lane first, then a
            // butterfly over the 16 sample lanes, four tiles (16 values) at a time
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(t & 3) * 4 + r] = fmaxf(acc[0][r], 0.0f) * aux[0] + fmaxf(acc[1][r], 0.0f) * aux[1];
            if ((t & 3) == 3) {
                const float s = butterfly_n<16>(red, c);
                const int v = ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3);  // bit-reversed lane index
                if (live) po[16 * ((t - 3) + (v >> 2)) + 4 * q + (v & 3)] = s;
            }
        }
    });
}

template <int PREC, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void nerf_fwd_x16b_kernel(
    N3dtGeom g, const unsigned char* __restrict__ packed, const float* __restrict__ fold, const float* __restrict__ xy,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv, const float* __restrict__ t_rand,
    float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks) {
    typedef typename X16<PREC>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;

    WeightStream<PREC, WAVES> ws;
    ws.gsrc = packed + (size_t)wave * WeightStream<PREC, WAVES>::PPW * X16_PIECE + lane * 16;
    ws.ring = lds;
    ws.lds_addr0 = (unsigned)(size_t)(LDS_AS unsigned char*)lds + lane * 16;
    ws.wave = wave;
    ws.prologue_issue();  // the sampler / encoder below runs under these loads
    // per-wave LDS copy of the PE fragments for the skip stage: 4 lane-linear 1 KiB pieces
    unsigned char* pe_lds = lds + X16_NBUF * X16_CH * X16_PIECE + (size_t)wave * 4 * X16_PIECE + lane * 16;

    long blk = (long)blockIdx.x * WAVES + wave;
    const bool live = blk < total_blocks;
    if (!live) blk = total_blocks - 1;
    float* po = part + (size_t)blk * N3DT_PART_STRIDE;
    const int sb = (int)(blk % bpr);
    const long rayg = blk / bpr;
    const int ray = (int)(rayg % g.n_rays);
    const int frame = (int)(rayg / g.n_rays);
    float dist[2], zval[2];
    frag pe[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        float p[3];
        n3dt_sample_point(g, xy, R, T, Kinv, t_rand, frame, ray, sb * X16_BS + 16 * nb + c, p, dist[nb], zval[nb]);
        // phase in revolutions as hi + lo, so that the 2^k scaling of the encoder stays exact
        float rh[3], rl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float inv2pi_hi = 0.15915494f, inv2pi_lo = 6.2195e-09f;  // 1/(2 pi) split
            rh[i] = p[i] * inv2pi_hi;
            rl[i] = fmaf(p[i], inv2pi_hi, -rh[i]) + p[i] * inv2pi_lo;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = pe_fast(p[0], p[1], p[2], rh[0], rh[1], rh[2], rl[0], rl[1], rl[2], 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3));
            pe[nb][ks] = X16<PREC>::pack(v);
            *reinterpret_cast<frag*>(pe_lds + (nb * 2 + ks) * X16_PIECE) = pe[nb][ks];
        }
    }
    const float* fb = fold + (size_t)__builtin_amdgcn_readfirstlane(frame) * N3DT_FOLD_STRIDE;
    ws.prologue_wait();

    frag ha[2][12], hb[2][12];
    float aux[2];
    // FeaExt_module_0 (reference: NetWorks/models.py:69-71)
    x16b_stage<PREC, WAVES, 2, 2, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(0), pe, nullptr, ha, ha, aux, po, live, lane);
    // FeaExt_module_1..7 with the skip concat after layer 4 (models.py:72-76); fully unrolled (see nerf_fwd_x16.hip)
    x16b_stage<PREC, WAVES, 12, 0, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(1), pe, nullptr, ha, hb, aux, po, live, lane);
    x16b_stage<PREC, WAVES, 12, 0, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(2), pe, nullptr, hb, ha, aux, po, live, lane);
    x16b_stage<PREC, WAVES, 12, 0, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(3), pe, nullptr, ha, hb, aux, po, live, lane);
    x16b_stage<PREC, WAVES, 12, 0, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(4), pe, nullptr, hb, ha, aux, po, live, lane);
    x16b_stage<PREC, WAVES, 14, 2, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(5), pe, pe_lds, ha, hb, aux, po, live, lane);
    x16b_stage<PREC, WAVES, 12, 0, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(6), pe, nullptr, hb, ha, aux, po, live, lane);
    x16b_stage<PREC, WAVES, 12, 0, 24, B_HIDDEN>(ws, fb + n3dt_bias_offset(7), pe, nullptr, ha, hb, aux, po, live, lane);
    // density head on h7 (models.py:78,84): two 16-row tiles keep the stage a whole chunk; row 0 is the density
    x16b_stage<PREC, WAVES, 12, 0, 2, B_DENSITY>(ws, fb + n3dt_bias_offset(8), pe, nullptr, hb, ha, aux, po, live, lane);
    // alpha, in-block transmittance and weights (reference: NetWorks/utils.py:273-289); sample 16 nb + c of the block
    {
        float x[2], al[2], Tl[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float sp = __shfl(aux[nb], c, 64);  // row 0 lives on the q == 0 lanes
            const float sigma = fmaxf(sp, 0.0f);
            al[nb] = 1.0f - expf(-sigma * dist[nb]);
            x[nb] = 1.0f - al[nb] + 1e-10f;
            Tl[nb] = n3dt_exclusive_prod<16>(x[nb], c);
        }
        const float tot0 = __shfl(Tl[0] * x[0], 15, 16);  // product over the first 16 samples
        Tl[1] *= tot0;
        const float w0 = al[0] * Tl[0], w1 = al[1] * Tl[1];
        float s0 = w0 + w1, s1 = w0 * zval[0] + w1 * zval[1];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off, 16);
            s1 += __shfl_xor(s1, off, 16);
        }
        const float tprod = __shfl(Tl[1] * x[1], 15, 16);
        if (live && lane == 0) {
            po[N3DT_G + 0] = s0;
            po[N3DT_G + 1] = s1;
            po[N3DT_G + 2] = tprod;
            po[N3DT_G + 3] = 0.0f;
        }
        if (live && wlocal && q == 0) {
            wlocal[(size_t)blk * X16_BS + c] = w0;
            wlocal[(size_t)blk * X16_BS + 16 + c] = w1;
        }
        aux[0] = w0;
        aux[1] = w1;
    }
    // RGB_layer_0 -> RGB_layer_1 as ONE merged 192 x 384 layer on h7, relu, weighted by the sample weights and reduced
    const float* b10 = g.vd_dim > 0 ? fold + n3dt_rayfold_offset(g.batch) + (size_t)__builtin_amdgcn_readfirstlane((int)rayg) * N3DT_RAYFOLD_STRIDE
                                    : fb + n3dt_bias_offset(10);  // include_vd: per ray (n3dt_layout.h)
    x16b_stage<PREC, WAVES, 12, 0, 12, B_COMPOSITE>(ws, b10, pe, nullptr, hb, ha, aux, po, live, lane);
}

template <int PREC>
static void launch_x16b(const N3dtGeom* g, const void* packed, const float* fold, const float* xy, const float* R, const float* T,
                        const float* Kinv, const float* t_rand, float* part, float* wlocal, hipStream_t stream) {
    constexpr int WAVES = 8;
    const int bpr = (g->n_samples + X16_BS - 1) / X16_BS;
    const long total = (long)g->batch * g->n_rays * bpr;
    const int grid = (int)((total + WAVES - 1) / WAVES);
    const size_t lds_bytes = X16_NBUF * X16_CH * X16_PIECE + (size_t)WAVES * 4 * X16_PIECE;
    auto kern = nerf_fwd_x16b_kernel<PREC, WAVES>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, *g, reinterpret_cast<const unsigned char*>(packed), fold,
                       xy, R, T, Kinv, t_rand, part, wlocal, bpr, total);
}

// `packed_b` = the 16x16x32-ordered matrix region of the packed buffer
extern "C" void n3dt_launch_nerf_fwd_x16b(const N3dtGeom* g, int precision, const void* packed_b, const float* fold, const float* xy,
                                          const float* R, const float* T, const float* Kinv, const float* t_rand, float* part,
                                          float* wlocal, hipStream_t stream) {
    if (precision == N3DT_BF16) launch_x16b<N3DT_BF16>(g, packed_b, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
    else launch_x16b<N3DT_F16>(g, packed_b, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
}
