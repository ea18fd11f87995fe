// Fused loss tail: the three MSE data terms of the reference's loss in one pass over the images,
// and their gradient in one more -- no boolean-mask gathers (which cost PyTorch a host synchronisation per term).
//
// Reference (read as text; it needs torchvision/face_alignment to import): Utils/HeadNeRFLossUtils.py:125-146
//   bg_loss      = mean((bg_img - v)^2)
//   head_loss    = mse(nan_to_num(merge_img)[mask >= 0.5], gt[mask >= 0.5])
//   nonhead_loss = mean((nan_to_num(merge_img)[mask < 0.5] - v)^2)
// merge_img, gt: [B,3,P,P]; bg_img: [1,3,P,P]; mask: [B,1,P,P]; v = bg_value (1 for the white background).
// acc[6] = { sum_bg, sum_head, n_head, sum_nonhead, n_nonhead, n_bg } (element counts, i.e. 3 per pixel).
#include "n3dt_device.h"

__global__ void loss_tail_fwd_kernel(int B, int HW, const float* __restrict__ merge, const float* __restrict__ bg,
                                     const float* __restrict__ gt, const float* __restrict__ mask, float v, float* __restrict__ acc) {
    float s_bg = 0.f, s_head = 0.f, n_head = 0.f, s_non = 0.f, n_non = 0.f;
    const size_t n_img = (size_t)B * 3 * HW, n_bg = (size_t)3 * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_img; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / ((size_t)3 * HW), p = i % HW;
        float r = merge[i];
        if (r != r) r = 0.0f;  // nan_to_num(nan=0.0); +-inf map to +-FLT_MAX like torch
        else if (r > 3.4028234663852886e38f) r = 3.4028234663852886e38f;
        else if (r < -3.4028234663852886e38f) r = -3.4028234663852886e38f;
        if (mask[b * HW + p] >= 0.5f) {
            const float d = r - gt[i];
            s_head += d * d;
            n_head += 1.0f;
        } else {
            const float d = r - v;
            s_non += d * d;
            n_non += 1.0f;
        }
        if (i < n_bg) {
            const float d = bg[i] - v;
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

// terms[3] = { bg_loss, head_loss, nonhead_loss }
__global__ void loss_tail_finish_kernel(int HW, float* __restrict__ acc, float* __restrict__ terms) {
    acc[5] = 3.0f * (float)HW;
    terms[0] = acc[0] / acc[5];
    terms[1] = acc[1] / acc[2];  // 0/0 = nan for an empty mask, as torch's mean of an empty selection
    terms[2] = acc[3] / acc[4];
}

// d_merge, d_bg for upstream gradients g[3] of the three terms
__global__ void loss_tail_bwd_kernel(int B, int HW, const float* __restrict__ merge, const float* __restrict__ bg,
                                     const float* __restrict__ gt, const float* __restrict__ mask, float v,
                                     const float* __restrict__ acc, const float* __restrict__ g, float* __restrict__ d_merge,
                                     float* __restrict__ d_bg) {
    const size_t n_img = (size_t)B * 3 * HW, n_bg = (size_t)3 * HW;
    const float k_bg = 2.0f * g[0] / acc[5], k_head = 2.0f * g[1] / acc[2], k_non = 2.0f * g[2] / acc[4];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_img; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / ((size_t)3 * HW), p = i % HW;
        const float r = merge[i];
        const bool finite = (r == r) && r <= 3.4028234663852886e38f && r >= -3.4028234663852886e38f;
        float d;
        if (mask[b * HW + p] >= 0.5f) d = k_head * (r - gt[i]);
        else d = k_non * (r - v);
        d_merge[i] = finite ? d : 0.0f;  // nan_to_num has zero gradient where it replaced the value
        if (i < n_bg) d_bg[i] = k_bg * (bg[i] - v);
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

extern "C" void n3dt_launch_loss_fwd(int B, int HW, const float* merge, const float* bg, const float* gt, const float* mask, float v,
                                     float* acc, float* terms, hipStream_t s) {
    (void)hipMemsetAsync(acc, 0, 6 * sizeof(float), s);
    const size_t n = (size_t)B * 3 * HW;
    int grid = (int)((n + 255) / 256);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(grid), dim3(256), 0, s, B, HW, merge, bg, gt, mask, v, acc);
    hipLaunchKernelGGL(loss_tail_finish_kernel, dim3(1), dim3(1), 0, s, HW, acc, terms);
}

extern "C" void n3dt_launch_loss_bwd(int B, int HW, const float* merge, const float* bg, const float* gt, const float* mask, float v,
                                     const float* acc, const float* g, float* d_merge, float* d_bg, hipStream_t s) {
    const size_t n = (size_t)B * 3 * HW;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(grid), dim3(256), 0, s, B, HW, merge, bg, gt, mask, v, acc, g, d_merge, d_bg);
}
