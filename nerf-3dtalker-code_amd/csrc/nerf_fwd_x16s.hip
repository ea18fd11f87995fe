// Fused render kernel, split-precision mode (N3DT_BF16X3): the parity-grade mode on the bf16 matrix pipe.
//
// Same mathematics, stream machinery and epilogue as nerf_fwd_x16.hip (one wavefront = one block of 32 consecutive samples of
// one ray; activations transposed, H^T[channel][sample], resident in registers as MFMA B operands; weights streamed
// L2 -> LDS -> fragments; reference: NetWorks/models.py:62-87, NetWorks/utils.py:43-51,268-309) -- but every operand of every
// product travels as TWO bf16 values, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together), and a
// product is three MFMAs:   W x  ~=  W_hi x_hi  +  (W_hi x_lo + W_lo x_hi)      (the lo*lo term is below 2^-16 relative).
//
// Why: with single bf16 operands the rounding of weights AND activations (2^-9 relative each, both matter equally -- measured
// by rounding them separately in the reference) reaches 2.4e-2 on RGB when the density head is sharp (fixture
// `contrast`), 24x the north-star gate of 1e-3; fp16 operands reach 3.7e-3.  Split operands bring 6e-5 -- the exact-fp32
// kernel's class -- at a third of the bf16 mode's rate instead of a twentieth.
//
// Shape: 4 wavefronts per workgroup, one per SIMD, 512 registers each (the hi and lo halves of two activation buffers are
// 384 of them).  Per 32-sample block the stream is 2 x 2 280 pieces (hi piece, lo piece, alternating), each pair feeding three
// MFMAs, so LDS bytes per MFMA drop to 2/3 of the one-product kernel's.  The correction products accumulate in their own
// accumulator (two independent MFMA chains per tile; the small terms are summed among themselves before they meet the large one).
#include <hip/hip_runtime.h>

#include "x16_core.h"

#define XS_WAVES 4
typedef X16<N3DT_BF16> XsT;
typedef XsT::frag xs_frag;
typedef WeightStream<N3DT_BF16, XS_WAVES, 2 * X16_NCHUNK> XsStream;

// v (fp32, 8 values) -> hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void xs_split(const float (&v)[8], xs_frag& hi, xs_frag& lo) {
    hi = XsT::pack(v);
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j] - (float)hi[j];
    lo = XsT::pack(r);
}

// One layer on one 32-sample block.  KS k-steps of 16 channels per out tile, the first KPE of them positional-encoding
// fragments (read from the wave's LDS copy: hi pieces 0..3, lo pieces 4..7), NT out tiles of 32 channels.
template <int KS, int KPE, int NT, int MODE>
__device__ __forceinline__ void xs_stage(XsStream& ws, const float* __restrict__ bias, const unsigned char* pe_lds, const xs_frag (&hin_hi)[24],
                                         const xs_frag (&hin_lo)[24], xs_frag (&hout_hi)[24], xs_frag (&hout_lo)[24], float& aux,
                                         float* const po, const bool live, const int lane) {
    const int h = lane >> 5, c = lane & 31;
    float red[32];
    const xs_frag ones = XsT::ones_frag();
    float bias_cur = bias[c];
    constexpr bool LAST = MODE == MODE_COMPOSITE;  // the stream ends with this stage: no prefetch past it
    static_for<0, NT>([&](auto ot_c) {
        constexpr int ot = decltype(ot_c)::value;
        f32x16 acc, cor;
        {
            // acc = bias, broadcast over the samples, by one MFMA (hi / lo split: x16_core.h); cor = 0
            const xs_frag bf = XsT::bias_frag(bias_cur, h == 0);
            if (ot + 1 < NT) bias_cur = bias[(ot + 1) * 32 + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) cor[r] = 0.0f;
            acc = XsT::mfma(bf, ones, cor);
        }
        static_for<0, KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int P = 2 * (ot * KS + ks);
            const xs_frag a_hi = ws.template next<LAST, 2 * NT * KS, P>();
            const xs_frag a_lo = ws.template next<LAST, 2 * NT * KS, P + 1>();
            xs_frag b_hi, b_lo;
            if constexpr (ks < KPE) {
                b_hi = *reinterpret_cast<const xs_frag*>(pe_lds + ks * X16_PIECE);
                b_lo = *reinterpret_cast<const xs_frag*>(pe_lds + (4 + ks) * X16_PIECE);
            } else {
                b_hi = hin_hi[ks - KPE];
                b_lo = hin_lo[ks - KPE];
            }
            acc = XsT::mfma(a_hi, b_hi, acc);
            cor = XsT::mfma(a_hi, b_lo, cor);
            cor = XsT::mfma(a_lo, b_hi, cor);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += cor[r];
        if constexpr (MODE == MODE_HIDDEN) {
            // ReLU in fp32, then the split: registers 8*half .. 8*half+7 of the tile are k-step 2*ot + half of the next layer
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(acc[8 * half + r], 0.0f);
                xs_split(v, hout_hi[2 * ot + half], hout_lo[2 * ot + half]);
            }
        } else if constexpr (MODE == MODE_DENSITY) {
            aux = acc[0];  // row 0 of the tile, valid on lanes with h == 0
        } else {
            // MODE_COMPOSITE: relu(RGB_layer_1) weighted by the sample weights; two tiles (32 values) feed one butterfly over the samples
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(ot & 1) * 16 + r] = fmaxf(acc[r], 0.0f) * aux;
            if constexpr (ot & 1) {
                const float s = butterfly32(red, c);
                // bit-reversed lane index = which of the 32 reduced values this lane ended up with
                const int v = ((c & 1) << 4) | ((c & 2) << 2) | (c & 4) | ((c & 8) >> 2) | ((c & 16) >> 4);
                const int reg = v & 15, tile = (ot - 1) + (v >> 4);
                if (live) po[tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h] = s;
            }
        }
    });
}

__global__ __launch_bounds__(XS_WAVES * 64, 1) void nerf_fwd_x16s_kernel(
    N3dtGeom g, const unsigned char* __restrict__ packed, const float* __restrict__ fold, const float* __restrict__ xy,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv, const float* __restrict__ t_rand,
    float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;

    XsStream ws;
    ws.gsrc = packed + (size_t)wave * XsStream::PPW * X16_PIECE + lane * 16;
    ws.ring = lds;
    ws.lds_addr0 = (unsigned)(size_t)(LDS_AS unsigned char*)lds + lane * 16;
    ws.wave = wave;
    ws.prologue_issue();  // the sampler / encoder below runs under these loads
    // per-wave LDS copy of the PE fragments: 4 hi pieces, then 4 lo pieces (lane-linear 1 KiB each)
    unsigned char* pe_lds = lds + X16_NBUF * X16_CH * X16_PIECE + (size_t)wave * 8 * X16_PIECE + lane * 16;

    long bidx = (long)blockIdx.x * XS_WAVES + wave;
    const bool live = bidx < total_blocks;
    if (!live) bidx = total_blocks - 1;
    float* po = part + (size_t)bidx * N3DT_PART_STRIDE;
    const int sb = (int)(bidx % bpr);
    const long rayg = bidx / bpr;
    const int ray = (int)(rayg % g.n_rays);
    const int frame = (int)(rayg / g.n_rays);
    float p[3], dist, zval;
    n3dt_sample_point(g, xy, R, T, Kinv, t_rand, frame, ray, sb * X16_BS + c, p, dist, zval);
    {
        // phase in revolutions as hi + lo, so that the 2^k scaling of the encoder stays exact (as nerf_fwd_x16.hip)
        float rh[3], rl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float inv2pi_hi = 0.15915494f, inv2pi_lo = 6.2195e-09f;  // 1/(2 pi) split
            rh[i] = p[i] * inv2pi_hi;
            rl[i] = fmaf(p[i], inv2pi_hi, -rh[i]) + p[i] * inv2pi_lo;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = pe_fast(p[0], p[1], p[2], rh[0], rh[1], rh[2], rl[0], rl[1], rl[2],
                               32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3));
            xs_frag hi, lo;
            xs_split(v, hi, lo);
            *reinterpret_cast<xs_frag*>(pe_lds + ks * X16_PIECE) = hi;
            *reinterpret_cast<xs_frag*>(pe_lds + (4 + ks) * X16_PIECE) = lo;
        }
    }
    const float* fb = fold + (size_t)__builtin_amdgcn_readfirstlane(frame) * N3DT_FOLD_STRIDE;
    x16_pin(fb);  // no scalar loads of kernel arguments once the fragment stream runs (x16_core.h)
    ws.prologue_wait();

    xs_frag ha_hi[24], ha_lo[24], hb_hi[24], hb_lo[24];
    float aux = 0.0f;
    // FeaExt_module_0 (reference: NetWorks/models.py:69-71)
    xs_stage<4, 4, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(0), pe_lds, ha_hi, ha_lo, ha_hi, ha_lo, aux, po, live, lane);
    // FeaExt_module_1..7 with the skip concat after layer 4 (models.py:72-76)
    xs_stage<24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(1), pe_lds, ha_hi, ha_lo, hb_hi, hb_lo, aux, po, live, lane);
    xs_stage<24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(2), pe_lds, hb_hi, hb_lo, ha_hi, ha_lo, aux, po, live, lane);
    xs_stage<24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(3), pe_lds, ha_hi, ha_lo, hb_hi, hb_lo, aux, po, live, lane);
    xs_stage<24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(4), pe_lds, hb_hi, hb_lo, ha_hi, ha_lo, aux, po, live, lane);
    xs_stage<28, 4, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(5), pe_lds, ha_hi, ha_lo, hb_hi, hb_lo, aux, po, live, lane);
    xs_stage<24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(6), pe_lds, hb_hi, hb_lo, ha_hi, ha_lo, aux, po, live, lane);
    xs_stage<24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(7), pe_lds, ha_hi, ha_lo, hb_hi, hb_lo, aux, po, live, lane);
    // density head on h7 (models.py:78,84); the bias rides in the accumulator
    xs_stage<24, 0, 1, MODE_DENSITY>(ws, fb + n3dt_bias_offset(8), pe_lds, hb_hi, hb_lo, ha_hi, ha_lo, aux, po, live, lane);
    // alpha, in-block transmittance and weights (reference: NetWorks/utils.py:273-289)
    {
        const float sp = __shfl(aux, c, 64);  // row 0 lives on the h == 0 half
        const float sigma = fmaxf(sp, 0.0f);
        const float alpha = 1.0f - expf(-sigma * dist);
        const float x = 1.0f - alpha + 1e-10f;
        const float Tl = n3dt_exclusive_prod<32>(x, c);
        const float w = alpha * Tl;
        float s0 = w, s1 = w * zval;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off, 32);
            s1 += __shfl_xor(s1, off, 32);
        }
        const float tprod = __shfl(Tl * x, 31, 32);
        if (live && lane == 0) {
            po[N3DT_G + 0] = s0;
            po[N3DT_G + 1] = s1;
            po[N3DT_G + 2] = tprod;
            po[N3DT_G + 3] = 0.0f;
        }
        if (live && wlocal && h == 0) wlocal[(size_t)bidx * X16_BS + c] = w;
        aux = w;
    }
    // RGB_layer_0 -> RGB_layer_1 as ONE merged 192 x 384 layer on h7 (no activation sits between them, models.py:79-81; merged
    // matrix and bias built by pack / fold), relu, weighted by the sample weights and reduced over the samples
    // include_vd: the merged RGB bias of this wave's RAY (n3dt_layout.h) instead of the frame's
    const float* b10 = g.vd_dim > 0 ? fold + n3dt_rayfold_offset(g.batch) + (size_t)__builtin_amdgcn_readfirstlane((int)rayg) * N3DT_RAYFOLD_STRIDE
                                    : fb + n3dt_bias_offset(10);
    xs_stage<24, 0, 6, MODE_COMPOSITE>(ws, b10, pe_lds, hb_hi, hb_lo, ha_hi, ha_lo, aux, po, live, lane);
}

extern "C" void n3dt_launch_nerf_fwd_x16s(const N3dtGeom* g, const void* packed, const float* fold, const float* xy, const float* R,
                                          const float* T, const float* Kinv, const float* t_rand, float* part, float* wlocal,
                                          hipStream_t stream) {
    const int bpr = (g->n_samples + X16_BS - 1) / X16_BS;
    const long total = (long)g->batch * g->n_rays * bpr;
    const int grid = (int)((total + XS_WAVES - 1) / XS_WAVES);
    const size_t lds_bytes = X16_NBUF * X16_CH * X16_PIECE + (size_t)XS_WAVES * 8 * X16_PIECE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_fwd_x16s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
    hipLaunchKernelGGL(nerf_fwd_x16s_kernel, dim3(grid), dim3(XS_WAVES * 64), lds_bytes, stream, *g,
                       reinterpret_cast<const unsigned char*>(packed), fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total);
}
