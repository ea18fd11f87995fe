// Fused volumetric-render kernel, exact-fp32 mode (the <=1e-3 RGB parity path).
//
// One wavefront = one block of 16 consecutive samples of one ray, carried through the whole
// latent-conditioned MLP with v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fmaf chain).
// Activations live as [channel][sample] rows in the wave's private LDS slab; a layer loads its
// K input rows into registers once (B operands), streams the packed weights (A operands,
// 16 B per lane per 4 MFMAs, coalesced 1 KiB per wave-instruction, L2-resident) and overwrites
// the slab in place.  Nothing per-sample ever reaches HBM: the wave composites its 16 samples
// (alpha, in-block transmittance scan, weighted feature sum) and emits one 196-float partial.
//
// Replaces (reference): NetWorks/utils.py:147-161,65-145 (sampler), :43-51 (embedder),
// NetWorks/models.py:62-87 (MLP), NetWorks/utils.py:268-309 (compositing, per-block part).
#include "n3dt_device.h"
#include "n3dt_layout.h"

#define F32_BS 16                 // samples per wave
#define F32_WAVES 4               // waves per workgroup
#define F32_ROWS (64 + 384 + 32)  // PE rows | hidden rows | density rows
#define F32_SLAB (F32_ROWS * F32_BS)

template <int K>
__device__ __forceinline__ void f32_layer(const float* __restrict__ Wp, const float* __restrict__ bias, const int N,
                                          const bool relu, const float* Hin, float* Hout, const int lane) {
    const int c = lane & 15, q = lane >> 4;
    float hin[K / 4];
#pragma unroll
    for (int ks = 0; ks < K / 4; ++ks) hin[ks] = Hin[(4 * ks + q) * F32_BS + c];
    __syncthreads();  // every lane holds its inputs before rows are overwritten in place
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + lane;
    for (int ot = 0; ot < N / 16; ot += 2) {
        f32x4 acc0 = *reinterpret_cast<const f32x4*>(bias + ot * 16 + 4 * q);
        f32x4 acc1 = *reinterpret_cast<const f32x4*>(bias + ot * 16 + 16 + 4 * q);
        const f32x4* w0 = wp + (size_t)ot * (K / 16) * 64;
        const f32x4* w1 = w0 + (K / 16) * 64;
#pragma unroll
        for (int k4 = 0; k4 < K / 16; ++k4) {
            f32x4 a0 = w0[k4 * 64];
            f32x4 a1 = w1[k4 * 64];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, hin[4 * k4 + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, hin[4 * k4 + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, hin[4 * k4 + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, hin[4 * k4 + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, hin[4 * k4 + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, hin[4 * k4 + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, hin[4 * k4 + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, hin[4 * k4 + 3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v0 = acc0[r], v1 = acc1[r];
            if (relu) {
                v0 = fmaxf(v0, 0.0f);
                v1 = fmaxf(v1, 0.0f);
            }
            Hout[(ot * 16 + 4 * q + r) * F32_BS + c] = v0;
            Hout[(ot * 16 + 16 + 4 * q + r) * F32_BS + c] = v1;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(F32_WAVES * 64, 1) void nerf_fwd_f32_kernel(
    N3dtGeom g, N3dtMlpParams prm, const float* __restrict__ packed, const float* __restrict__ fold,
    const float* __restrict__ xy, const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv,
    const float* __restrict__ t_rand, float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    float* slab = lds + wave * F32_SLAB;
    float* pe = slab;                        // rows 0..63
    float* h = slab + 64 * F32_BS;           // rows 64..447
    float* den = slab + (64 + 384) * F32_BS; // rows 448..479

    long blk = (long)blockIdx.x * F32_WAVES + wave;
    const bool live = blk < total_blocks;
    if (!live) blk = total_blocks - 1;  // keep the wave in the barriers; its stores are masked
    const int sb = (int)(blk % bpr);
    const long rayg = blk / bpr;
    const int ray = (int)(rayg % g.n_rays);
    const int b = (int)(rayg / g.n_rays);
    const int s = sb * F32_BS + c;

    float p[3], dist, zval;
    n3dt_sample_point(g, xy, R, T, Kinv, t_rand, b, ray, s, p, dist, zval);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = 4 * i + q;
        pe[row * F32_BS + c] = n3dt_pe_row_accurate(p, row);
    }
    __syncthreads();

    const float* fb = fold + (size_t)b * N3DT_FOLD_STRIDE;
    // FeaExt_module_0..7 with the skip concat after layer 4 (models.py:69-76)
    f32_layer<64>(packed + n3dt_stage_offset(0), fb + n3dt_bias_offset(0), 384, true, pe, h, lane);
#pragma unroll 1
    for (int l = 1; l < 5; ++l) f32_layer<384>(packed + n3dt_stage_offset(l), fb + n3dt_bias_offset(l), 384, true, h, h, lane);
    f32_layer<448>(packed + n3dt_stage_offset(5), fb + n3dt_bias_offset(5), 384, true, pe, h, lane);
#pragma unroll 1
    for (int l = 6; l < 8; ++l) f32_layer<384>(packed + n3dt_stage_offset(l), fb + n3dt_bias_offset(l), 384, true, h, h, lane);
    // density head (models.py:78; relu applied below), then the feature head (:79-82)
    f32_layer<384>(packed + n3dt_stage_offset(8), fb + n3dt_bias_offset(8), 32, false, h, den, lane);
    f32_layer<384>(packed + n3dt_stage_offset(9), fb + n3dt_bias_offset(9), 384, false, h, h, lane);
    // include_vd: RGB_layer_1's bias of this wave's RAY (frame entry + view-direction term, n3dt_layout.h) instead of the frame's
    const float* b10 = g.vd_dim > 0 ? fold + n3dt_rayfold_offset(g.batch) + (size_t)rayg * N3DT_RAYFOLD_STRIDE : fb + n3dt_bias_offset(10);
    f32_layer<384>(packed + n3dt_stage_offset(10), b10, 192, true, h, h, lane);

    // compositing of this 16-sample block (utils.py:273-309): local transmittance starts at 1
    float sigma = fmaxf(den[c], 0.0f);
    float alpha = 1.0f - expf(-sigma * dist);
    float x = 1.0f - alpha + 1e-10f;
    float Tl = n3dt_exclusive_prod<16>(x, c);
    float w = alpha * Tl;
    float wsum = w, dsum = w * zval;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        wsum += __shfl_xor(wsum, off, 16);
        dsum += __shfl_xor(dsum, off, 16);
    }
    float tprod = __shfl(Tl * x, 15, 16);
    if (q == 0) den[F32_BS + c] = w;  // density row 1 is free scratch
    __syncthreads();
    if (live) {
        float* po = part + (size_t)blk * N3DT_PART_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int j = lane + 64 * i;
            const float* grow = h + j * F32_BS;
            float acc = 0.0f;
#pragma unroll
            for (int pt = 0; pt < F32_BS; ++pt) acc += den[F32_BS + pt] * grow[pt];
            po[j] = acc;
        }
        if (lane == 0) {
            po[N3DT_G + 0] = wsum;
            po[N3DT_G + 1] = dsum;
            po[N3DT_G + 2] = tprod;
            po[N3DT_G + 3] = 0.0f;
        }
        if (wlocal && q == 0) wlocal[(size_t)blk * F32_BS + c] = w;
    }
}

extern "C" void n3dt_launch_nerf_fwd_f32(const N3dtGeom* g, const N3dtMlpParams* prm, const void* packed, const float* fold,
                                         const float* xy, const float* R, const float* T, const float* Kinv,
                                         const float* t_rand, float* part, float* wlocal, hipStream_t stream) {
    const int bpr = (g->n_samples + F32_BS - 1) / F32_BS;
    const long total = (long)g->batch * g->n_rays * bpr;
    const int grid = (int)((total + F32_WAVES - 1) / F32_WAVES);
    const size_t lds_bytes = (size_t)F32_WAVES * F32_SLAB * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_fwd_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds_bytes);
    hipLaunchKernelGGL(nerf_fwd_f32_kernel, dim3(grid), dim3(F32_WAVES * 64), lds_bytes, stream, *g, *prm,
                       reinterpret_cast<const float*>(packed), fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total);
}
