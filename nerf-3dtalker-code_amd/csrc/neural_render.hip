// 2-D neural renderer (feature map -> RGB), fp32, ray-major ("NHWC") activations.
//
// Replaces (reference): NeuralRenderer.forward NetWorks/neural_renderer.py:72-91,
// PixelShuffleUpsample.forward NetWorks/PixelShuffleUpsample.py:36-45, Blur :15-18
// (kornia.filters.filter2d: depthwise [1,2,1]x[1,2,1]/16, reflect border) and
// nn.Upsample(scale_factor=2, bilinear, align_corners=False) :54-55.
//
// Every 1x1 convolution is a pixel-major GEMM  Y[pix][o] = X[pix][:] . W[o][:] + b[o]  on
// v_mfma_f32_32x32x2_f32 (exact fp32); pixel-shuffle is a store-address permutation in the
// second PSU GEMM's epilogue; the RGB skip pyramid stays planar [nb,3,h,w] like the output.
#include "n3dt_device.h"

#define EPI_LRELU 0   // Y = lrelu(acc + b)
#define EPI_PSU 1     // Y(pixel-shuffled) = lrelu(acc + b) + X.repeat(1,4,1,1)

struct GemmEpi {
    float* y;          // output
    const float* res;  // EPI_PSU: the block input X [M][C]
    int C;             // EPI_PSU: input channels (N == 4C)
    int H, W;          // EPI_PSU: input resolution (M == nb*H*W)
    float slope;       // leaky-relu slope, <0 = identity
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ X,
                                                       const float* __restrict__ Wt, const float* __restrict__ bias,
                                                       GemmEpi ea) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int lrow = tid >> 2, kq = tid & 3;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (m0 + lrow < M) a = *reinterpret_cast<const f32x4*>(X + (size_t)(m0 + lrow) * K + k0 + 4 * kq);
        if (n0 + lrow < N) b = *reinterpret_cast<const f32x4*>(Wt + (size_t)(n0 + lrow) * K + k0 + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[4 * kq + j][lrow] = a[j];
            Bs[4 * kq + j][lrow] = b[j];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            float av = As[kk + (lane >> 5)][wr * 32 + (lane & 31)];
            float bv = Bs[kk + (lane >> 5)][wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + wc * 32 + (lane & 31);
    if (n >= N) return;
    const float bn = bias[n];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        float v = acc[reg] + bn;
        if (ea.slope >= 0.0f) v = v > 0.0f ? v : v * ea.slope;
        if (EPI == EPI_LRELU) {
            ea.y[(size_t)m * N + n] = v;
        } else {
            // out + x.repeat(1,4,1,1) then pixel_shuffle(2): out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]
            v += ea.res[(size_t)m * ea.C + (n % ea.C)];
            const int c = n >> 2, di = (n >> 1) & 1, dj = n & 1;
            const int w = m % ea.W, h = (m / ea.W) % ea.H, img = m / (ea.W * ea.H);
            ea.y[(((size_t)img * 2 * ea.H + 2 * h + di) * 2 * ea.W + 2 * w + dj) * ea.C + c] = v;
        }
    }
}


__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Blur on [nb,H,W,C]: one thread = one pixel x 4 channels
__global__ void blur_nhwc_kernel(int nb, int H, int W, int C, const float* __restrict__ x, float* __restrict__ y) {
    const int c4 = C / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)nb * H * W * c4;
    if (i >= total) return;
    const N3dtDiv dc4 = n3dt_div(c4), dW = n3dt_div(W), dH = n3dt_div(H);
    int cq = n3dt_rem(i, dc4);
    size_t pix = n3dt_quot(i, dc4);
    const size_t prow = n3dt_quot(pix, dW);
    int w = n3dt_rem(pix, dW), h = n3dt_rem(prow, dH), img = (int)n3dt_quot(prow, dH);
    const float k[3] = {0.25f, 0.5f, 0.25f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int di = -1; di <= 1; ++di)
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            int hh = reflect1(h + di, H), ww = reflect1(w + dj, W);
            f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)img * H + hh) * W + ww) * C + 4 * cq);
            float kw = k[di + 1] * k[dj + 1];
            acc += kw * v;
        }
    *reinterpret_cast<f32x4*>(y + pix * C + 4 * cq) = acc;
}

// feat_2_rgb: net [nb*H*W][K] -> planar rgb [nb,3,H,W]; optional accumulate and final sigmoid
__global__ void to_rgb_kernel(int nb, int HW, int K, const float* __restrict__ net, const float* __restrict__ Wt,
                              const float* __restrict__ bias, const float* __restrict__ rgb_in, float* __restrict__ rgb_out,
                              int final_sigmoid) {
    extern __shared__ float wl[];  // [3][K]
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) wl[i] = Wt[i];
    __syncthreads();
    size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (size_t)nb * HW) return;
    const float* xr = net + pix * K;
    float a0 = bias[0], a1 = bias[1], a2 = bias[2];
    for (int k = 0; k < K; k += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0 = fmaf(wl[k + j], v[j], a0);
            a1 = fmaf(wl[K + k + j], v[j], a1);
            a2 = fmaf(wl[2 * K + k + j], v[j], a2);
        }
    }
    const N3dtDiv dHW = n3dt_div(HW);
    size_t img = n3dt_quot(pix, dHW), p = n3dt_rem(pix, dHW);
    size_t o = img * 3 * (size_t)HW + p;
    if (rgb_in) {
        a0 = rgb_in[o] + a0;
        a1 = rgb_in[o + HW] + a1;
        a2 = rgb_in[o + 2 * (size_t)HW] + a2;
    }
    if (final_sigmoid) {
        a0 = 1.0f / (1.0f + expf(-a0));
        a1 = 1.0f / (1.0f + expf(-a1));
        a2 = 1.0f / (1.0f + expf(-a2));
    }
    rgb_out[o] = a0;
    rgb_out[o + HW] = a1;
    rgb_out[o + 2 * (size_t)HW] = a2;
}

// The same projection with SIXTEEN lanes per pixel (K a multiple of 64): a lane takes K / 16 channels as 16-byte loads that a
// pixel's lanes issue side by side (256 contiguous bytes per instruction and pixel; the one-thread-per-pixel form above strides
// its lanes 4 K bytes apart and walks K / 4 dependent loads: 11 us for two 64 x 64 maps, a launch the whole chip cannot help
// with since it is 32 workgroups), then a 4-step butterfly.  First level of the 16-bit render path (K = 256).
__global__ __launch_bounds__(256) void to_rgb16_kernel(int nb, int HW, int K, const float* __restrict__ net, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, float* __restrict__ rgb_out) {
    extern __shared__ float wl[];  // [3][K]
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) wl[i] = Wt[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    size_t pix = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool valid = pix < (size_t)nb * HW;
    if (!valid) pix = (size_t)nb * HW - 1;  // keep the lane for the shuffles
    const float* xr = net + pix * K + 4 * l16;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int k = 0; k < K; k += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + k + 4 * l16), w1 = *reinterpret_cast<const f32x4*>(wl + K + k + 4 * l16),
                    w2 = *reinterpret_cast<const f32x4*>(wl + 2 * K + k + 4 * l16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0 = fmaf(w0[j], v[j], a0);
            a1 = fmaf(w1[j], v[j], a1);
            a2 = fmaf(w2[j], v[j], a2);
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off, 64);
        a1 += __shfl_xor(a1, off, 64);
        a2 += __shfl_xor(a2, off, 64);
    }
    if (valid && l16 == 0) {
        const N3dtDiv dHW = n3dt_div(HW);
        const size_t img = n3dt_quot(pix, dHW), p = n3dt_rem(pix, dHW), o = img * 3 * (size_t)HW + p;
        rgb_out[o] = a0 + bias[0];
        rgb_out[o + HW] = a1 + bias[1];
        rgb_out[o + 2 * (size_t)HW] = a2 + bias[2];
    }
}

__device__ __forceinline__ float bilinear_at(const float* __restrict__ x, int h, int w, int i, int j) {
    // value of the 2x bilinear upsample (align_corners=False) of x[h][w] at output pixel (i, j)
    float si = fmaxf(0.5f * ((float)i + 0.5f) - 0.5f, 0.0f);
    float sj = fmaxf(0.5f * ((float)j + 0.5f) - 0.5f, 0.0f);
    int i0 = (int)si, j0 = (int)sj;
    int i1 = i0 + (i0 < h - 1 ? 1 : 0), j1 = j0 + (j0 < w - 1 ? 1 : 0);
    float li = si - (float)i0, lj = sj - (float)j0;
    return (1.0f - li) * ((1.0f - lj) * x[i0 * w + j0] + lj * x[i0 * w + j1]) +
           li * ((1.0f - lj) * x[i1 * w + j0] + lj * x[i1 * w + j1]);
}

// rgb_upsample = bilinear x2 then Blur, planar [n_planes][h][w] -> [n_planes][2h][2w]
__global__ void rgb_up_kernel(int n_planes, int h, int w, const float* __restrict__ x, float* __restrict__ y) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int H2 = 2 * h, W2 = 2 * w;
    if (i >= (size_t)n_planes * H2 * W2) return;
    const N3dtDiv dW2 = n3dt_div(W2), dH2 = n3dt_div(H2);
    const size_t orow = n3dt_quot(i, dW2);
    int oj = n3dt_rem(i, dW2), oi = n3dt_rem(orow, dH2);
    size_t pl = n3dt_quot(orow, dH2);
    const float* xp = x + pl * (size_t)h * w;
    float wr[3], wc[3];
    n3dt_up_blur_w3(oi, h, wr);
    n3dt_up_blur_w3(oj, w, wc);
    const int bi = oi >> 1, bj = oj >> 1;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ii = min(max(bi - 1 + a, 0), h - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += (wr[a] * wc[b]) * xp[(size_t)ii * w + min(max(bj - 1 + b, 0), w - 1)];
    }
    y[i] = acc;
}

#include "neural_render_x16.inc"

static inline int nr_ch(int C, int i) {
    int v = C >> i;
    return v < 32 ? 32 : v;
}

static void launch_gemm(int epi, int M, int N, int K, const float* X, const float* Wt, const float* b, GemmEpi ea, hipStream_t s) {
    dim3 grid((M + 63) / 64, (N + 63) / 64);
    if (epi == EPI_LRELU) hipLaunchKernelGGL(gemm_f32_kernel<EPI_LRELU>, grid, dim3(256), 0, s, M, N, K, X, Wt, b, ea);
    else hipLaunchKernelGGL(gemm_f32_kernel<EPI_PSU>, grid, dim3(256), 0, s, M, N, K, X, Wt, b, ea);
}

// workspace carve (floats): t1 | ps | bl | netA | netB | rgbA | rgbB
struct NrCarve {
    size_t t1, ps, bl, net, rgb, total;
};

static NrCarve nr_carve(const N3dtGeom* g, int nb) {
    NrCarve c = {0, 0, 0, 0, 0, 0};
    const int C = g->feat_nc;
    for (int i = 0; i < g->n_blocks; ++i) {
        size_t h = (size_t)g->featmap_size << i, M = (size_t)nb * h * h;
        size_t ci = nr_ch(C, i), co = nr_ch(C, i + 1);
        if (M * 2 * ci > c.t1) c.t1 = M * 2 * ci;
        if (4 * M * ci > c.ps) c.ps = 4 * M * ci;
        if (4 * M * co > c.net) c.net = 4 * M * co;
    }
    c.bl = c.ps;
    size_t P = (size_t)g->featmap_size << g->n_blocks;
    c.rgb = (size_t)nb * 3 * P * P;
    c.total = c.t1 + c.ps + c.bl + 2 * c.net + 2 * c.rgb + nrf_pack_floats(g);
    return c;
}

extern "C" size_t n3dt_nr_workspace_floats(const N3dtGeom* g, int nb) { return nr_carve(g, nb).total; }

// pack_mode: 0 = pack the fused blocks' weights into the workspace tail and render; 1 = render with the stream a previous call
// left there; 2 = pack only.  The fp32 path reads the raw parameters (nothing to pack).
extern "C" void n3dt_launch_neural_render(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p,
                                          const float* featmap, float* img, float* ws, int pack_mode, hipStream_t s) {
    const NrCarve cv = nr_carve(g, nb);
    if (precision != N3DT_F32) {  // 16-bit activations, blur commuted behind feat_layers (neural_render_x16.inc)
        n3dt_launch_neural_render_x16(g, nb, precision, p, featmap, img, ws, cv.total, pack_mode, s);
        return;
    }
    if (pack_mode == 2) return;
    float* t1 = ws;
    float* ps = t1 + cv.t1;
    float* bl = ps + cv.ps;
    float* netA = bl + cv.bl;
    float* netB = netA + cv.net;
    float* rgbA = netB + cv.net;
    float* rgbB = rgbA + cv.rgb;
    const int C = g->feat_nc, fs = g->featmap_size, nblk = g->n_blocks;
    int h = fs;
    // rgb = rgb_upsample(feat_2_rgb_list[0](x))   (neural_renderer.py:75)
    {
        size_t npix = (size_t)nb * h * h;
        hipLaunchKernelGGL(to_rgb_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 3 * C * sizeof(float), s, nb, h * h, C,
                           featmap, p->to_rgb_w[0], p->to_rgb_b[0], (const float*)nullptr, rgbB, 0);
        size_t nout = (size_t)nb * 3 * 4 * h * h;
        hipLaunchKernelGGL(rgb_up_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, nb * 3, h, h, rgbB, rgbA);
    }
    const float* net = featmap;
    float* cur = netA;
    float* oth = netB;
    for (int i = 0; i < nblk; ++i) {
        const int ci = nr_ch(C, i), co = nr_ch(C, i + 1);
        const int M = nb * h * h;
        GemmEpi e1 = {t1, nullptr, 0, 0, 0, 0.2f};
        launch_gemm(EPI_LRELU, M, 2 * ci, ci, net, p->psu1_w[i], p->psu1_b[i], e1, s);
        GemmEpi e2 = {ps, net, ci, h, h, 0.2f};
        launch_gemm(EPI_PSU, M, 4 * ci, 2 * ci, t1, p->psu2_w[i], p->psu2_b[i], e2, s);
        h *= 2;
        {
            size_t n = (size_t)nb * h * h * (ci / 4);
            hipLaunchKernelGGL(blur_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nb, h, h, ci, ps, bl);
        }
        GemmEpi e3 = {cur, nullptr, 0, 0, 0, 0.2f};
        launch_gemm(EPI_LRELU, nb * h * h, co, ci, bl, p->feat_w[i], p->feat_b[i], e3, s);
        const bool last = (i == nblk - 1);
        size_t npix = (size_t)nb * h * h;
        // rgb = rgb + feat_2_rgb_list[i+1](net); sigmoid after the last block (neural_renderer.py:82-88)
        hipLaunchKernelGGL(to_rgb_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 3 * co * sizeof(float), s, nb, h * h, co,
                           cur, p->to_rgb_w[i + 1], p->to_rgb_b[i + 1], (const float*)rgbA, last ? img : rgbB, last ? 1 : 0);
        if (!last) {
            size_t nout = (size_t)nb * 3 * 4 * h * h;
            hipLaunchKernelGGL(rgb_up_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, nb * 3, h, h, rgbB, rgbA);
        }
        net = cur;
        float* t = cur;
        cur = oth;
        oth = t;
    }
}
