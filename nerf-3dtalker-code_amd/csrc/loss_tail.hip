// Fused loss tail: the three MSE data terms of the reference's loss in one pass over the images,
// and their gradient in one more -- no boolean-mask gathers (which cost PyTorch a host synchronisation per term).
//
// Reference (read as text; it needs torchvision/face_alignment to import): Utils/HeadNeRFLossUtils.py:125-146
//   bg_loss      = mean((bg_img - v)^2)
//   head_loss    = mse(nan_to_num(merge_img)[mask >= 0.5], gt[mask >= 0.5])
//   nonhead_loss = mean((nan_to_num(merge_img)[mask < 0.5] - v)^2)
// merge_img, gt: [B,3,P,P]; bg_img: [1,3,P,P]; mask: [B,1,P,P]; v = bg_value (1 for the white background).
// acc[8] = { sum_bg, sum_head, n_head, sum_nonhead, n_nonhead, n_bg, -, - } (element counts, i.e. 3 per pixel).
// terms[4] = { bg_loss, head_loss, nonhead_loss, total = (bg + head) + nonhead } (the reference sums in that order, :228-231).
#include "n3dt_device.h"

__device__ __forceinline__ float loss_clean(float r) {
    if (r != r) return 0.0f;  // nan_to_num(nan=0.0); +-inf map to +-FLT_MAX like torch
    if (r > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (r < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return r;
}

// One thread walks groups of four consecutive pixels of one image plane (16-byte loads; HW % 4 == 0 -- the host falls back
// to VEC = 1 otherwise).
template <int VEC>
__global__ __launch_bounds__(256) void loss_tail_fwd_kernel(int B, int HW, const float* __restrict__ merge, const float* __restrict__ bg,
                                                            const float* __restrict__ gt, const float* __restrict__ mask, float v,
                                                            float* __restrict__ acc) {
    float s_bg = 0.f, s_head = 0.f, n_head = 0.f, s_non = 0.f, n_non = 0.f;
    const size_t n_img = (size_t)B * 3 * HW / VEC, n_bg = (size_t)3 * HW / VEC, hwv = (size_t)HW / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_img; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / (3 * hwv), p = i % hwv;
        float r[VEC], t[VEC], m[VEC], q[VEC];
        if constexpr (VEC == 4) {
            const f32x4 rv = reinterpret_cast<const f32x4*>(merge)[i], tv = reinterpret_cast<const f32x4*>(gt)[i],
                        mv = reinterpret_cast<const f32x4*>(mask)[b * hwv + p];
            f32x4 qv = {v, v, v, v};
            if (i < n_bg) qv = reinterpret_cast<const f32x4*>(bg)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) { r[j] = rv[j]; t[j] = tv[j]; m[j] = mv[j]; q[j] = qv[j]; }
        } else {
            r[0] = merge[i]; t[0] = gt[i]; m[0] = mask[b * hwv + p]; q[0] = i < n_bg ? bg[i] : v;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float rc = loss_clean(r[j]);
            if (m[j] >= 0.5f) {
                const float d = rc - t[j];
                s_head += d * d;
                n_head += 1.0f;
            } else {
                const float d = rc - v;
                s_non += d * d;
                n_non += 1.0f;
            }
            const float d = q[j] - v;  // (0 outside the background image)
            s_bg += d * d;
        }
    }
    float vals[5] = {s_bg, s_head, n_head, s_non, n_non};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float x = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        vals[k] = x;
    }
    // one atomic per value and workgroup: thousands of same-address atomics serialise (0.5 ms at 512^2 before this)
    __shared__ float red[4][5];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[wave][k] = vals[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&acc[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// terms[4] = { bg_loss, head_loss, nonhead_loss, (bg + head) + nonhead }.  A separate one-thread launch: folding it into the
// kernel above ("last workgroup to arrive finishes") needs a release / acquire pair per workgroup, and at agent scope each of
// those writes the XCD's whole L2 back (buffer_wbl2) -- with the renderer's 9 MB of fresh output sitting dirty in it that made
// the pass 88 us instead of 20.
__global__ void loss_tail_finish_kernel(int HW, float* __restrict__ acc, float* __restrict__ terms) {
    acc[5] = 3.0f * (float)HW;
    const float t0 = acc[0] / acc[5];
    const float t1 = acc[1] / acc[2];  // 0/0 = nan for an empty mask, as torch's mean of an empty selection
    const float t2 = acc[3] / acc[4];
    terms[0] = t0;
    terms[1] = t1;
    terms[2] = t2;
    terms[3] = (t0 + t1) + t2;
}

// d_merge, d_bg for upstream gradients g[3] of the three terms and / or g_total of their sum (either may be NULL)
template <int VEC>
__global__ __launch_bounds__(256) void loss_tail_bwd_kernel(int B, int HW, const float* __restrict__ merge, const float* __restrict__ bg,
                                                            const float* __restrict__ gt, const float* __restrict__ mask, float v,
                                                            const float* __restrict__ acc, const float* __restrict__ g,
                                                            const float* __restrict__ g_total, float* __restrict__ d_merge,
                                                            float* __restrict__ d_bg) {
    const size_t n_img = (size_t)B * 3 * HW / VEC, n_bg = (size_t)3 * HW / VEC, hwv = (size_t)HW / VEC;
    const float gt_ = g_total ? g_total[0] : 0.0f;
    const float g0 = (g ? g[0] : 0.0f) + gt_, g1 = (g ? g[1] : 0.0f) + gt_, g2 = (g ? g[2] : 0.0f) + gt_;
    const float k_bg = 2.0f * g0 / acc[5], k_head = 2.0f * g1 / acc[2], k_non = 2.0f * g2 / acc[4];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_img; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / (3 * hwv), p = i % hwv;
        if constexpr (VEC == 4) {
            const f32x4 rv = reinterpret_cast<const f32x4*>(merge)[i], tv = reinterpret_cast<const f32x4*>(gt)[i],
                        mv = reinterpret_cast<const f32x4*>(mask)[b * hwv + p];
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = rv[j];
                const bool finite = (r == r) && r <= 3.4028234663852886e38f && r >= -3.4028234663852886e38f;
                const float d = mv[j] >= 0.5f ? k_head * (r - tv[j]) : k_non * (r - v);
                o[j] = finite ? d : 0.0f;  // nan_to_num has zero gradient where it replaced the value
            }
            reinterpret_cast<f32x4*>(d_merge)[i] = o;
            if (i < n_bg) {
                const f32x4 qv = reinterpret_cast<const f32x4*>(bg)[i];
                reinterpret_cast<f32x4*>(d_bg)[i] = f32x4{k_bg * (qv[0] - v), k_bg * (qv[1] - v), k_bg * (qv[2] - v), k_bg * (qv[3] - v)};
            }
        } else {
            const float r = merge[i];
            const bool finite = (r == r) && r <= 3.4028234663852886e38f && r >= -3.4028234663852886e38f;
            const float d = mask[b * hwv + p] >= 0.5f ? k_head * (r - gt[i]) : k_non * (r - v);
            d_merge[i] = finite ? d : 0.0f;
            if (i < n_bg) d_bg[i] = k_bg * (bg[i] - v);
        }
    }
}

// display conversion (talker_trainer.py:1205, Utils/RenderUtils.py:123-125): planar float [V,3,HW] in (0,1) ->
// interleaved uint8 [V,HW,3], (unsigned char)(x * 255) like numpy's astype(np.uint8) on in-range values
__global__ void img_to_uint8_kernel(int V, int HW, const float* __restrict__ img, unsigned char* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)V * HW) return;
    const size_t v = i / HW, p = i % HW;
    const float* src = img + v * 3 * (size_t)HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = fminf(fmaxf(src[(size_t)c * HW] * 255.0f, 0.0f), 255.0f);
        out[i * 3 + c] = (unsigned char)x;
    }
}
extern "C" void n3dt_launch_img_to_uint8(int V, int HW, const float* img, unsigned char* out, hipStream_t s) {
    hipLaunchKernelGGL(img_to_uint8_kernel, dim3((unsigned)(((size_t)V * HW + 255) / 256)), dim3(256), 0, s, V, HW, img, out);
}

static inline bool loss_vec4(int HW, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
    auto al = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
    return HW % 4 == 0 && al(a) && al(b) && al(c) && al(d) && al(e) && al(f);
}

extern "C" void n3dt_launch_loss_fwd(int B, int HW, const float* merge, const float* bg, const float* gt, const float* mask, float v,
                                     float* acc, float* terms, hipStream_t s) {
    (void)hipMemsetAsync(acc, 0, 8 * sizeof(float), s);
    const bool v4 = loss_vec4(HW, merge, bg, gt, mask, nullptr, nullptr);
    const size_t n = (size_t)B * 3 * HW / (v4 ? 4 : 1);
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    if (v4) hipLaunchKernelGGL(loss_tail_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, B, HW, merge, bg, gt, mask, v, acc);
    else hipLaunchKernelGGL(loss_tail_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, B, HW, merge, bg, gt, mask, v, acc);
    hipLaunchKernelGGL(loss_tail_finish_kernel, dim3(1), dim3(1), 0, s, HW, acc, terms);
}

extern "C" void n3dt_launch_loss_bwd(int B, int HW, const float* merge, const float* bg, const float* gt, const float* mask, float v,
                                     const float* acc, const float* g, const float* g_total, float* d_merge, float* d_bg, hipStream_t s) {
    const bool v4 = loss_vec4(HW, merge, bg, gt, mask, d_merge, d_bg);
    const size_t n = (size_t)B * 3 * HW / (v4 ? 4 : 1);
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (v4) hipLaunchKernelGGL(loss_tail_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, B, HW, merge, bg, gt, mask, v, acc, g, g_total, d_merge, d_bg);
    else hipLaunchKernelGGL(loss_tail_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, B, HW, merge, bg, gt, mask, v, acc, g, g_total, d_merge, d_bg);
}
