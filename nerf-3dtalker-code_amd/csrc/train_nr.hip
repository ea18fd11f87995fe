// Training path of the 2-D neural renderer: forward that keeps every stage's activations, and the
// backward (adjoints of the 1x1 convs, pixel-shuffle + residual, reflect-border blur, bilinear x2 and
// the RGB skip pyramid).  Written once, templated on the storage element of the maps:
//   float     exact fp32, every conv product a gemm32 launch (the path pinned against the reference's autograd);
//   nrt_bf16  mixed precision: bf16 maps and map gradients in HBM, bf16 MFMA products, fp32 parameters and gradients.
//
// Differentiates (reference): NetWorks/neural_renderer.py:72-91, NetWorks/PixelShuffleUpsample.py:36-45,
// Blur :15-18 (kornia filter2d semantics).
#include "gemm32.h"
#include "n3dt_device.h"

static inline int nr_ch(int C, int i) {
    int v = C >> i;
    return v < 32 ? 32 : v;
}
static inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

struct NrSaved {
    size_t t1[N3DT_MAX_BLOCKS], tv[N3DT_MAX_BLOCKS], bl[N3DT_MAX_BLOCKS], net[N3DT_MAX_BLOCKS], img, total;
};
// t1: lrelu(layer_1 x) [M][2C]; tv: lrelu(layer_2 t1) before the residual [M][4C]; bl: blurred shuffled map [4M][C];
// net: lrelu(feat conv) [4M][C/2] (= next stage's input); img: the sigmoid output (for sigmoid')
static NrSaved nr_saved_layout(const N3dtGeom* g, int nb) {
    NrSaved s;
    size_t o = 0;
    for (int i = 0; i < g->n_blocks; ++i) {
        const size_t h = (size_t)g->featmap_size << i, M = (size_t)nb * h * h;
        const size_t ci = nr_ch(g->feat_nc, i), co = nr_ch(g->feat_nc, i + 1);
        s.t1[i] = o; o += al64(M * 2 * ci);
        s.tv[i] = o; o += al64(M * 4 * ci);
        s.bl[i] = o; o += al64(4 * M * ci);
        s.net[i] = o; o += al64(4 * M * co);
    }
    const size_t P = (size_t)g->featmap_size << g->n_blocks;
    s.img = o; o += al64((size_t)nb * 3 * P * P);
    s.total = o;
    return s;
}

struct NrWs {
    size_t a, b, c, rgb0, rgb1, wt, total;  // three big ping-pong buffers, two planar rgb gradient buffers, a transposed weight
};
static NrWs nr_ws_layout(const N3dtGeom* g, int nb) {
    NrWs w;
    size_t big = 0;
    for (int i = 0; i < g->n_blocks; ++i) {
        const size_t h = (size_t)g->featmap_size << i, M = (size_t)nb * h * h;
        const size_t ci = nr_ch(g->feat_nc, i);
        if (4 * M * ci > big) big = 4 * M * ci;
    }
    const size_t P = (size_t)g->featmap_size << g->n_blocks;
    size_t o = 0;
    w.a = o; o += al64(big);
    w.b = o; o += al64(big);
    w.c = o; o += al64(big);
    w.rgb0 = o; o += al64((size_t)nb * 3 * P * P);
    w.rgb1 = o; o += al64((size_t)nb * 3 * P * P);
    w.wt = o; o += al64((size_t)8 * g->feat_nc * g->feat_nc);  // fp32 [2C][4C] at most (layer_2 of stage 0)
    w.total = o;
    return w;
}

// (the sizes cover both the layered layouts above and the fused mixed-precision path's, nr_train16.h: the size queries of the
// C ABI do not take the precision)
extern "C" size_t n3dt_nr_train16_saved_bytes(const N3dtGeom* g, int nb);
extern "C" size_t n3dt_nr_train16_ws_bytes(const N3dtGeom* g, int nb);
extern "C" size_t n3dt_nr_train_saved_floats(const N3dtGeom* g, int nb) {
    const size_t a = nr_saved_layout(g, nb).total, b = (n3dt_nr_train16_saved_bytes(g, nb) + 3) / 4;
    return a > b ? a : b;
}
extern "C" size_t n3dt_nr_train_ws_floats(const N3dtGeom* g, int nb) {
    const size_t a = nr_ws_layout(g, nb).total, b = (n3dt_nr_train16_ws_bytes(g, nb) + 3) / 4;
    return a > b ? a : b;
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Storage element of the feature maps and of their gradients: float (the exact path) or bf16 (the mixed-precision
// path: bf16 maps in HBM, fp32 arithmetic in registers, fp32 parameters and parameter gradients).  The kernels below are
// written once against these load / store helpers.
struct nrt_bf16 {
    unsigned short u;
};
typedef unsigned short nrt_u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float nrt_ld(const float* p) { return *p; }
__device__ __forceinline__ float nrt_ld(const nrt_bf16* p) { return __builtin_bit_cast(float, (unsigned)p->u << 16); }
__device__ __forceinline__ void nrt_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void nrt_st(nrt_bf16* p, float v) { p->u = __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ f32x4 nrt_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 nrt_ld4(const nrt_bf16* p) {
    const nrt_u16x4 u = *reinterpret_cast<const nrt_u16x4*>(p);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(float, (unsigned)u[j] << 16);
    return v;
}
__device__ __forceinline__ void nrt_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void nrt_st4(nrt_bf16* p, f32x4 v) {
    nrt_u16x4 u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = __builtin_bit_cast(unsigned short, (__bf16)v[j]);
    *reinterpret_cast<nrt_u16x4*>(p) = u;
}

// ---- forward pieces ---------------------------------------------------------------------------
// tv [M][4C] (+ residual x.repeat) -> pixel-shuffled ps [4M][C]   (PixelShuffleUpsample.py:36,41-42)
template <class T, class TX>
__global__ void nrt_shuffle_kernel(int nb, int H, int W, int C, const T* __restrict__ tv, const TX* __restrict__ x,
                                   T* __restrict__ ps) {
    // thread = 4 adjacent output channels of one output pixel (one 16-byte store)
    const int c4 = C >> 2;
    const size_t total = (size_t)nb * H * W * 4 * c4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const N3dtDiv dc4 = n3dt_div(c4), dC = n3dt_div(C), dW2 = n3dt_div(2 * W), dH2 = n3dt_div(2 * H);
    const int c = 4 * n3dt_rem(i, dc4);
    size_t opix = n3dt_quot(i, dc4);  // output pixel
    const size_t orow = n3dt_quot(opix, dW2);
    const int ow = n3dt_rem(opix, dW2), oh = n3dt_rem(orow, dH2), img = (int)n3dt_quot(orow, dH2);
    const int h = oh >> 1, di = oh & 1, w = ow >> 1, dj = ow & 1;
    const size_t m = ((size_t)img * H + h) * W + w;
    const T* tr = tv + m * 4 * C;
    const TX* xr = x + m * C;
    f32x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = 4 * (c + j) + 2 * di + dj;  // pixel_shuffle source channel; x.repeat(1,4,1,1) adds x[o % C]
        out[j] = nrt_ld(tr + o) + nrt_ld(xr + n3dt_rem(o, dC));
    }
    nrt_st4(ps + opix * C + c, out);
}

template <class T>
__global__ void nrt_blur_kernel(int nb, int H, int W, int C, const T* __restrict__ x, T* __restrict__ y) {
    const int c4 = C / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nb * H * W * c4) return;
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
            f32x4 v = nrt_ld4(x + (((size_t)img * H + hh) * W + ww) * C + 4 * cq);
            acc += (k[di + 1] * k[dj + 1]) * v;
        }
    nrt_st4(y + pix * C + 4 * cq, acc);
}

// planar rgb [nb,3,HW] (+)= W[3][K] net[pix][K] + b     (feat_2_rgb_list)
template <class T>
__global__ void nrt_to_rgb_kernel(int nb, int HW, int K, const T* __restrict__ net, const float* __restrict__ Wt,
                                  const float* __restrict__ bias, const float* __restrict__ rgb_in, float* __restrict__ rgb_out,
                                  int final_sigmoid) {
    extern __shared__ float wl[];
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) wl[i] = Wt[i];
    __syncthreads();
    size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (size_t)nb * HW) return;
    const T* xr = net + pix * K;
    float a0 = bias[0], a1 = bias[1], a2 = bias[2];
    for (int k = 0; k < K; ++k) {
        const float v = nrt_ld(xr + k);
        a0 = fmaf(wl[k], v, a0);
        a1 = fmaf(wl[K + k], v, a1);
        a2 = fmaf(wl[2 * K + k], v, a2);
    }
    const N3dtDiv dHW = n3dt_div(HW);
    size_t img = n3dt_quot(pix, dHW), p = n3dt_rem(pix, dHW), o = img * 3 * (size_t)HW + p;
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

__device__ __forceinline__ float nrt_bilinear_at(const float* __restrict__ x, int h, int w, int i, int j) {
    float si = fmaxf(0.5f * ((float)i + 0.5f) - 0.5f, 0.0f);
    float sj = fmaxf(0.5f * ((float)j + 0.5f) - 0.5f, 0.0f);
    int i0 = (int)si, j0 = (int)sj;
    int i1 = i0 + (i0 < h - 1 ? 1 : 0), j1 = j0 + (j0 < w - 1 ? 1 : 0);
    float li = si - (float)i0, lj = sj - (float)j0;
    return (1.0f - li) * ((1.0f - lj) * x[i0 * w + j0] + lj * x[i0 * w + j1]) + li * ((1.0f - lj) * x[i1 * w + j0] + lj * x[i1 * w + j1]);
}

__global__ void nrt_rgb_up_kernel(int n_planes, int h, int w, const float* __restrict__ x, float* __restrict__ y) {
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

// ---- backward pieces --------------------------------------------------------------------------
// 1-D adjoint of the reflect-border [1,2,1]/4 blur: which outputs read input j, with what weight.
// In closed form: input j is read by outputs j-1, j, j+1 only (the reflected reads of outputs 0 and
// n-1 land on j = 1 and j = n-2, i.e. on an existing neighbour), so three taps with border-dependent weights --
// no tap lists, no variable trip counts (the list form kept its index arrays in scratch).
__device__ __forceinline__ void blur_adj_w3(int j, int n, float (&w)[3]) {
    w[0] = j >= 1 ? (j == 1 ? 0.5f : 0.25f) : 0.0f;           // from output j-1 (+ output 0's reflected read when j == 1)
    w[1] = 0.5f;
    w[2] = j + 1 < n ? (j == n - 2 ? 0.5f : 0.25f) : 0.0f;    // from output j+1 (+ output n-1's reflected read when j == n-2)
}

template <class T>
__global__ void nrt_blur_adj_kernel(int nb, int H, int W, int C, const T* __restrict__ dy, T* __restrict__ dx) {
    const int c4 = C / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nb * H * W * c4) return;
    const N3dtDiv dc4 = n3dt_div(c4), dW = n3dt_div(W), dH = n3dt_div(H);
    int cq = n3dt_rem(i, dc4);
    size_t pix = n3dt_quot(i, dc4);
    const size_t prow = n3dt_quot(pix, dW);
    int w = n3dt_rem(pix, dW), h = n3dt_rem(prow, dH), img = (int)n3dt_quot(prow, dH);
    float wh[3], ww[3];
    blur_adj_w3(h, H, wh);
    blur_adj_w3(w, W, ww);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int hh = min(max(h + a - 1, 0), H - 1);  // clamped rows / columns carry weight 0
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int wc = min(max(w + b - 1, 0), W - 1);
            const f32x4 v = nrt_ld4(dy + (((size_t)img * H + hh) * W + wc) * C + 4 * cq);
            acc += (wh[a] * ww[b]) * v;
        }
    }
    nrt_st4(dx + pix * C + 4 * cq, acc);
}

// planar variant for the rgb pyramid (single channel planes)
__global__ void nrt_blur_adj_planar_kernel(int n_planes, int H, int W, const float* __restrict__ dy, float* __restrict__ dx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_planes * H * W) return;
    const N3dtDiv dW = n3dt_div(W), dH = n3dt_div(H);
    const size_t prow = n3dt_quot(i, dW);
    int w = n3dt_rem(i, dW), h = n3dt_rem(prow, dH);
    size_t pl = n3dt_quot(prow, dH);
    float wh[3], ww[3];
    blur_adj_w3(h, H, wh);
    blur_adj_w3(w, W, ww);
    const float* d = dy + pl * (size_t)H * W;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int hh = min(max(h + a - 1, 0), H - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += (wh[a] * ww[b]) * d[(size_t)hh * W + min(max(w + b - 1, 0), W - 1)];
    }
    dx[i] = acc;
}

// adjoint of the x2 bilinear upsample (align_corners=False): input m receives from outputs 2m-1..2m+2
__device__ __forceinline__ int bil_adj_taps(int m, int n, int idx[4], float wt[4]) {
    int c = 0;
    idx[c] = 2 * m; wt[c++] = m == 0 ? 1.0f : 0.75f;             // even output 2m: (m-1: .25, m: .75), clamped at 0
    idx[c] = 2 * m + 1; wt[c++] = m == n - 1 ? 1.0f : 0.75f;     // odd output 2m+1: (m: .75, m+1: .25), clamped at n-1
    if (m + 1 < n) { idx[c] = 2 * m + 2; wt[c++] = 0.25f; }      // even output 2(m+1) reads m with .25
    if (m >= 1) { idx[c] = 2 * m - 1; wt[c++] = 0.25f; }         // odd output 2(m-1)+1 reads m with .25
    return c;
}

__global__ void nrt_bilinear_adj_kernel(int n_planes, int h, int w, const float* __restrict__ dy /*[2h][2w]*/, float* __restrict__ dx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_planes * h * w) return;
    const N3dtDiv dw = n3dt_div(w), dh = n3dt_div(h);
    const size_t prow = n3dt_quot(i, dw);
    int jw = n3dt_rem(i, dw), jh = n3dt_rem(prow, dh);
    size_t pl = n3dt_quot(prow, dh);
    int ih[4], iw[4];
    float wh[4], ww[4];
    const int nh = bil_adj_taps(jh, h, ih, wh), nw = bil_adj_taps(jw, w, iw, ww);
    const float* d = dy + pl * (size_t)4 * h * w;
    float acc = 0.0f;
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) acc += (wh[a] * ww[b]) * d[(size_t)ih[a] * 2 * w + iw[b]];
    dx[i] = acc;
}

// d_pre = d_img * y (1 - y)
__global__ void nrt_sigmoid_bwd_kernel(size_t n, const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * y[i] * (1.0f - y[i]);
}

// dnet[pix][k] (+)= sum_c d_rgb[img][c][p] W[c][k], gated by lrelu'(net) when gate != nullptr
template <class T>
__global__ void nrt_to_rgb_bwd_kernel(int nb, int HW, int K, const float* __restrict__ d_rgb, const float* __restrict__ Wt,
                                      const T* __restrict__ gate, T* __restrict__ dnet, int accumulate) {
    extern __shared__ float wl[];
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) wl[i] = Wt[i];
    __syncthreads();
    const int k4 = K / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nb * HW * k4) return;
    const N3dtDiv dk4 = n3dt_div(k4), dHW = n3dt_div(HW);
    const int kq = n3dt_rem(i, dk4);
    const size_t pix = n3dt_quot(i, dk4), img = n3dt_quot(pix, dHW), p = n3dt_rem(pix, dHW), o = img * 3 * (size_t)HW + p;
    const float d0 = d_rgb[o], d1 = d_rgb[o + HW], d2 = d_rgb[o + 2 * (size_t)HW];
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = d0 * wl[4 * kq + j] + d1 * wl[K + 4 * kq + j] + d2 * wl[2 * K + 4 * kq + j];
    T* dst = dnet + pix * K + 4 * kq;
    if (gate) {
        f32x4 y = nrt_ld4(gate + pix * K + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = y[j] > 0.0f ? v[j] : 0.2f * v[j];
    }
    if (accumulate) v += nrt_ld4(dst);
    nrt_st4(dst, v);
}

// x *= lrelu'(gate)
template <class T>
__global__ void nrt_gate_kernel(size_t n, const T* __restrict__ gate, T* __restrict__ x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = nrt_ld(x + i);
        nrt_st(x + i, nrt_ld(gate + i) > 0.0f ? v : 0.2f * v);
    }
}

// pixel-shuffle + residual adjoint: dps [4M][C] -> dtv [M][4C] gated by lrelu'(tv), and dx_res[M][C] = sum over the 4 repeats
template <class T>
__global__ void nrt_unshuffle_kernel(int nb, int H, int W, int C, const T* __restrict__ dps, const T* __restrict__ tv,
                                     T* __restrict__ dtv, T* __restrict__ dxres) {
    const size_t total = (size_t)nb * H * W * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const N3dtDiv dC = n3dt_div(C), dW = n3dt_div(W), dH = n3dt_div(H);
    const int cp = n3dt_rem(i, dC);  // residual channel c' : receives every o with o % C == c'
    const size_t m = n3dt_quot(i, dC);
    const size_t prow = n3dt_quot(m, dW);
    const int w = n3dt_rem(m, dW), h = n3dt_rem(prow, dH), img = (int)n3dt_quot(prow, dH);
    float res = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = r * C + cp;
        const int c = o >> 2, di = (o >> 1) & 1, dj = o & 1;
        const float d = nrt_ld(dps + (((size_t)img * 2 * H + 2 * h + di) * 2 * W + 2 * w + dj) * C + c);
        res += d;
        const float y = nrt_ld(tv + m * 4 * C + o);
        nrt_st(dtv + m * 4 * C + o, y > 0.0f ? d : 0.2f * d);
    }
    nrt_st(dxres + i, res);
}

// out[n] += sum_m X[m][n]: one thread = 4 adjacent columns x a 256-row chunk, 8 independent float4 loads in flight
#define NRT_CS_ROWS 512
template <class T>
__global__ __launch_bounds__(256) void nrt_colsum_kernel(const T* __restrict__ X, long ldx, long rows, int N, float* __restrict__ out,
                                                         int chunk) {
    // thread = 4 adjacent columns x one row lane (256 / (N/4) row lanes; every N here is a multiple of 32, <= 1024)
    __shared__ f32x4 red[256];
    const int cg = N >> 2, lanes = 256 / cg;
    const int t = threadIdx.x, cq = t % cg, rl = t / cg;
    const long r0 = (long)blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
    const T* base = X + 4 * cq;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    long r = r0 + rl;
    if (rl < lanes) {
        for (; r + 7L * lanes < r1; r += 8L * lanes) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = nrt_ld4(base + (r + (long)u * lanes) * ldx);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] += v[u];
        }
        for (; r < r1; r += lanes) acc[0] += nrt_ld4(base + r * ldx);
    }
    red[t] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (t < cg) {
        f32x4 sum = red[t];
        for (int l = 1; l < lanes; ++l) sum += red[l * cg + t];
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&out[4 * t + j], sum[j]);
    }
}
template <class T>
static void launch_nrt_colsum(const T* X, long ldx, long rows, int N, float* out, hipStream_t s) {
    // about 384 workgroups: every workgroup ends in N atomics on the SAME N addresses, and thousands of same-address atomics
    // serialise (they, not the loads, set the time of the 1536-workgroup version)
    long chunk = (rows + 383) / 384;
    if (chunk < NRT_CS_ROWS) chunk = NRT_CS_ROWS;
    chunk = (chunk + 31) / 32 * 32;
    hipLaunchKernelGGL(nrt_colsum_kernel<T>, dim3((unsigned)((rows + chunk - 1) / chunk)), dim3(256), 0, s, X, ldx, rows, N, out, (int)chunk);
}

// db[c] += sum over images and pixels of planar d_rgb
__global__ void nrt_rgb_bias_kernel(int nb, int HW, const float* __restrict__ d_rgb, float* __restrict__ db) {
    __shared__ float red[256];
    const int c = blockIdx.y;
    float acc = 0.0f;
    for (int img = 0; img < nb; ++img) {
        const float* d = d_rgb + ((size_t)img * 3 + c) * HW;
        for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < (size_t)HW; p += (size_t)gridDim.x * blockDim.x) acc += d[p];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&db[c], red[0]);
}

// feat_2_rgb weight gradient: dW[k][c] += sum over images and pixels of d_rgb[img][k][pix] * net[img][pix][c]
// (a 3 x co result over up to 10^6 pixels: a reduction, not a GEMM).  Thread = 4 adjacent channels x one pixel lane;
// block = wg_pix consecutive pixels of one image (launch_to_rgb_wgrad picks it); co in {32, 64, 128, 256}.
// db (nullable): the bias gradient db[k] += sum over pixels of d_rgb[img][k][pix] rides on the channel-group-0 threads, which
// load d_rgb anyway (the stand-alone nrt_rgb_bias_kernel re-read it: one launch per level).
template <class T>
__global__ __launch_bounds__(256) void nrt_to_rgb_wgrad_kernel(int HW, int co, int wg_pix, const float* __restrict__ d_rgb,
                                                               const T* __restrict__ net, float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float red[256][13];
    float bsum[3] = {0.0f, 0.0f, 0.0f};
    const int cg = co >> 2, lanes = 256 / cg;
    const int t = threadIdx.x, c4 = (t % cg) * 4, pl = t / cg;
    const int chunks = (HW + wg_pix - 1) / wg_pix;
    const int img = blockIdx.x / chunks, p0 = (blockIdx.x % chunks) * wg_pix;
    const int p1 = min(HW, p0 + wg_pix);
    const float* d = d_rgb + (size_t)img * 3 * HW;
    const T* x = net + (size_t)img * HW * co + c4;
    float acc[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.0f;
    auto one = [&](const int p, const f32x4 v, const float g0, const float g1, const float g2) {
        const float gk[3] = {g0, g1, g2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bsum[k] += gk[k];
            acc[k][0] = fmaf(gk[k], v.x, acc[k][0]);
            acc[k][1] = fmaf(gk[k], v.y, acc[k][1]);
            acc[k][2] = fmaf(gk[k], v.z, acc[k][2]);
            acc[k][3] = fmaf(gk[k], v.w, acc[k][3]);
        }
        (void)p;
    };
    int p = p0 + pl;
    for (; p + 3 * lanes < p1; p += 4 * lanes) {  // four independent pixels in flight
        f32x4 v[4];
        float gg[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = nrt_ld4(x + (size_t)(p + u * lanes) * co);
#pragma unroll
            for (int k = 0; k < 3; ++k) gg[u][k] = d[(size_t)k * HW + p + u * lanes];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(p, v[u], gg[u][0], gg[u][1], gg[u][2]);
    }
    for (; p < p1; p += lanes)
        one(p, nrt_ld4(x + (size_t)p * co), d[p], d[(size_t)HW + p], d[(size_t)2 * HW + p]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[t][k * 4 + j] = acc[k][j];
    __syncthreads();
    // thread (k, channel) sums the pixel lanes
    for (int o = t; o < 3 * co; o += 256) {
        const int k = o / co, c = o % co;
        float sum = 0.0f;
        for (int l = 0; l < lanes; ++l) sum += red[l * cg + (c >> 2)][k * 4 + (c & 3)];
        atomicAdd(&dW[(size_t)k * co + c], sum);
    }
    if (db) {  // uniform over the workgroup: the channel-group-0 threads (one per pixel lane) hold the block's d_rgb sums
        __syncthreads();
        if (t % cg == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) red[pl][k] = bsum[k];
        }
        __syncthreads();
        if (t < 3) {
            float sum = 0.0f;
            for (int l = 0; l < lanes; ++l) sum += red[l][t];
            atomicAdd(&db[t], sum);
        }
    }
}
template <class T>
static void launch_to_rgb_wgrad(int nb, int HW, int co, const float* d_rgb, const T* net, float* dW, hipStream_t s, float* db = nullptr) {
    // pixels per workgroup: as many as leave ~512 workgroups (every workgroup ends in 3 co
    // same-address atomics: 1 536 workgroups x 96 atomics on 96 addresses were a third of the 44 us this took at 3 x 512^2)
    static const int target = [] {
        const char* e = getenv("N3DT_NR_RGBW_WGS");
        return e ? atoi(e) : 512;
    }();
    long pix = ((long)nb * HW + target - 1) / target;
    pix = (pix + 63) / 64 * 64;  // small maps (2 x 32^2 at 256 channels): 64-pixel workgroups instead of four of 512
    const int chunks = (int)((HW + pix - 1) / pix);
    hipLaunchKernelGGL(nrt_to_rgb_wgrad_kernel<T>, dim3(nb * chunks), dim3(256), 0, s, HW, co, (int)pix, d_rgb, net, dW, db);
}

#define GRID1(n) dim3((unsigned)(((size_t)(n) + 255) / 256)), dim3(256)

static Gemm32 mk(int M, int N, int K, const float* A, long lda, int ak, const float* B, long ldb, int bk, float* C, long ldc) {
    Gemm32 g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.a_kmajor = ak;
    g.B = B; g.ldb = ldb; g.b_kmajor = bk;
    g.C = C; g.ldc = ldc;
    g.bias = nullptr; g.bias_group_rows = 0; g.bias_ld = 0;
    g.act = G32_ACT_NONE;
    g.gate = nullptr; g.ldgate = 0; g.gate_act = G32_ACT_NONE;
    g.accumulate = 0; g.split_k = 1;
    g.a16 = g.b16 = g.c16 = g.gate16 = 0;
    return g;
}
// parameter-gradient products always ADD into their destination: atomics when K is split, += otherwise.
// The K (pixel) dimension is split so that the launch has about 1024 workgroups, at least 512 pixels each.
static void set_grad_split(Gemm32& q, long K) {
    const long tiles = (long)((q.M + 127) / 128) * ((q.N + 127) / 128);
    long s = 1024 / tiles, cap = K / 512;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    q.split_k = (int)s;
    q.accumulate = q.split_k <= 1 ? 1 : 0;
}

extern "C" void n3dt_launch_conv1x1_bf16(int, int, int, const void*, int, const float*, const float*, float, void*, hipStream_t);
extern "C" void n3dt_launch_conv1x1_bwd_bf16(int, int, int, const void*, const float*, int, const void*, void*, hipStream_t);
extern "C" void n3dt_launch_chw_to_hwc(int, int, const float*, float*, hipStream_t);

// input gradient of a 1x1 conv: dx[M][Nin] = dy[M][Kout] . W[Kout][Nin], then * lrelu'(gate) (mode 1) or += into dx (mode 2).
// fp32: the generic GEMM with its k-major B operand; bf16 maps: W is transposed into `wt` ([Nin][Kout], a few hundred KB) and
// the renderer's 16-bit GEMM runs with a gate / add epilogue.
template <class T>
static void conv_bwd_x(int M, int Nin, int Kout, const T* dy, const float* W, int mode, const T* res, T* dx, float* wt, hipStream_t s);
template <>
void conv_bwd_x<float>(int M, int Nin, int Kout, const float* dy, const float* W, int mode, const float* res, float* dx, float*,
                       hipStream_t s) {
    Gemm32 q = mk(M, Nin, Kout, dy, Kout, 0, W, Nin, 1, dx, Nin);
    if (mode == 1) { q.gate = res; q.ldgate = Nin; q.gate_act = G32_ACT_LRELU; }
    if (mode == 2) q.accumulate = 1;
    n3dt_gemm32(q, s);
}

// forward 1x1 conv + bias + LeakyReLU(0.2): the exact fp32 GEMM, or (bf16 maps) the renderer's own 16-bit GEMM with its
// LDS-staged 16-byte stores (neural_render_x16.inc)
template <class T>
static void conv_fwd(int M, int N, int K, const void* x, int x16, const float* W, const float* b, T* y, hipStream_t s);
template <>
void conv_fwd<float>(int M, int N, int K, const void* x, int, const float* W, const float* b, float* y, hipStream_t s) {
    Gemm32 q = mk(M, N, K, reinterpret_cast<const float*>(x), K, 0, W, K, 0, y, N);
    q.bias = b; q.act = G32_ACT_LRELU;
    n3dt_gemm32(q, s);
}

// bf16-storage GEMM operands travel through the descriptor's float pointers (gemm32.h)
template <class T>
static inline const float* as_f(const T* p) { return reinterpret_cast<const float*>(p); }
template <class T>
static inline float* as_f(T* p) { return reinterpret_cast<float*>(p); }
template <class T>
static constexpr int is16() { return sizeof(T) == 2 ? 1 : 0; }

// Weight gradient of a 1x1 conv on bf16 maps: out[co][ci] += sum_pix dY[pix][co] * X[pix][ci].
// The contraction runs over PIXELS, the row index of both (row-major) operands, with a few dozen to a few hundred output
// channels: a generic tiled GEMM spends its time transposing 128-wide tiles through LDS for a 32 x 64 result (0.55 TB/s).
// Here every wave owns TO x TI 32x32 output tiles and a private range of pixels and gathers its MFMA fragments straight from
// global memory -- lane (r, h) of an A fragment needs dY[pixel 8h+j][channel r], j = 0..7: eight 2-byte loads, each of which is
// one contiguous 64-byte row segment across the 32 lanes -- so there is no LDS, no barrier, and the operand bytes per MFMA are
// small enough (1 KiB) that the load pipe is nowhere near its limit.  fp32 atomics combine the pixel ranges.
typedef __bf16 nrt_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short nrt_u16x8 __attribute__((ext_vector_type(8)));
// The X operand is read through a loader XL (value of element [plane][pixel][channel] as float): a row-major matrix (DwRows),
// or the pixel-shuffled map ps = y + x.repeat rebuilt on the fly from the saved sub-pixel planes and the block input (DwPs).
// PM (plane mode):
//   0  one product; the four waves of a workgroup split its pixel chunk and hold partial sums of the SAME output tiles.
//   1  dY (and X) come as four sub-pixel planes of K pixels each that all add into the SAME output (d Wf): wave w walks the whole
//      chunk of plane w -- the plane is wave-uniform, and the four waves read the same rows of the shared operand (x) through L1.
// blockIdx.z = p (PM 0) walks independent products of one shape (d W2: plane q of d_tv against t1 gives rows 4c + q): dY, out, db
// advance by dy_plane / out_plane / db_plane per plane, db entries are db_stride apart.  (Measured and dropped: "wave = plane,
// own outputs" for d W2 -- t1 fetched once for the four planes, but four times the atomics per workgroup: 109 / 69 / 67 us
// became 158 / 115 / 95.  And PM 1 on the smallest block (32 x 64 outputs): 73 -> 150 us, so that one keeps PM 0.)
template <class TX>
struct DwRows {
    const TX* X;
    int ci;
    __device__ __forceinline__ float operator()(const int, const long pix, const int ch) const { return nrt_ld(X + pix * ci + ch); }
};
template <class TX>
struct DwPs {  // value = y_q[m][ch] + x[m][(4 ch + q) % C]
    const nrt_bf16* y;
    const TX* x;
    long M;
    int C;
    __device__ __forceinline__ float operator()(const int plane, const long pix, const int ch) const {
        // plane >= 0: wave-uniform sub-pixel, pix = m;  plane < 0 (PM 0): pix is the plane-major row q M + m
        const int q = plane >= 0 ? plane : (int)((pix >= M) + (pix >= 2 * M) + (pix >= 3 * M));
        const long m = plane >= 0 ? pix : pix - (long)q * M;
        return nrt_ld(y + ((long)q * M + m) * C + ch) + nrt_ld(x + m * C + ((4 * ch + q) & (C - 1)));
    }
};
struct DwPlanes {
    long dy_plane, out_plane;
    int db_plane, db_stride;
};
// Two waves per SIMD for the 8-tile instantiations that fit 256 registers without a spill (hipcc takes 272 unasked: ONE wave per
// SIMD); the pixel-split gather of the planes form (DwPs, PM = 0) spills 18 under that cap and is left alone.
template <class XL>
struct nrt_dw16_is_rows { static constexpr bool value = false; };
template <class T>
struct nrt_dw16_is_rows<DwRows<T>> { static constexpr bool value = true; };
template <int TO, int TI, class XL, int PM = 0>
__global__ __launch_bounds__(256, (TO * TI >= 8 && (PM == 1 || nrt_dw16_is_rows<XL>::value)) ? 2 : 1) void nrt_dw16_kernel(int co, int ci, long K, const nrt_bf16* __restrict__ dY, const XL X,
                                                       float* __restrict__ out, long ldo, long chunk, float* __restrict__ db, const DwPlanes pl) {
    // db (nullable): the bias gradient db[c] += sum over pixels of dY[pix][c] rides along on the workgroups of the first input
    // tile group -- the column sums of the fragments they load anyway (a separate column-sum pass re-read dY: 9 launches, 0.26 ms
    // of the 8 ms training step)
    __shared__ float rs_red[4][TO * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_i = (ci / 32 + TI - 1) / TI;
    const int o0 = (blockIdx.y / tiles_i) * TO * 32, i0 = (blockIdx.y % tiles_i) * TI * 32;
    // the workgroup's pixel chunk: split evenly over its 4 waves in multiples of 16 pixels (PM 0), or walked whole by every
    // wave on its own plane (PM 1, 2)
    const long c0 = (long)blockIdx.x * chunk, c1 = min(K, c0 + chunk);
    const long per = PM == 0 ? ((c1 - c0 + 3) / 4 + 15) / 16 * 16 : (c1 - c0);
    const long p0 = PM == 0 ? c0 + wave * per : c0, p1 = min(c1, p0 + per);
    if (PM != 0) dY += (long)wave * pl.dy_plane;
    if (PM == 0) {
        dY += (long)blockIdx.z * pl.dy_plane;
        out += (long)blockIdx.z * pl.out_plane;
        if (db) db += (long)blockIdx.z * pl.db_plane;
    }
    f32x16 acc[TO][TI];
    float rs[TO];
#pragma unroll
    for (int a = 0; a < TO; ++a) {
        rs[a] = 0.0f;
#pragma unroll
        for (int b = 0; b < TI; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
    }
    const bool do_rs = db != nullptr && (blockIdx.y % tiles_i) == 0;
    // channels past the layer's width (a 64 x 128 wave tile on a narrower layer) are clamped to a valid one: their tiles are
    // computed and dropped.  The main loop takes whole 16-pixel steps with no per-element guards, two steps in flight.
    // One 16-pixel step: ALL its loads first (2-byte gathers: lane r takes channel r of 8 consecutive pixels), then the TO x TI
    // products, then the bias row sums.  (With the row sums between the dY loads and the X loads, as first written, the X loads
    // waited for them: 80 / 54 / 57 / 34 us per launch of a config-3 step became 77 / 49 / 47 / 30; summing before the products
    // instead costs the 2 x 4-tile float-map instantiation a 16-byte spill.  Measured and dropped: the loads of step k + 1 issued
    // before the products of step k, raw values converted at use -- 78 us either way for the 1 x 2-tile instantiation: these
    // kernels are bound by the rate of the 2-byte gathers, not by a step's latency.)
    auto step = [&](const long p, const bool guard) {
        nrt_bf16x8 fa[TO], fb[TI];
        const long pb = p + 8 * h;
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            nrt_u16x8 u;
            const int ch = min(o0 + 32 * a + r, co - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = (!guard || pb + j < p1) ? dY[(pb + j) * co + ch].u : (unsigned short)0;
            fa[a] = __builtin_bit_cast(nrt_bf16x8, u);
        }
#pragma unroll
        for (int b = 0; b < TI; ++b) {
            nrt_u16x8 u;
            const int ch = min(i0 + 32 * b + r, ci - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (!guard || pb + j < p1) ? X(PM == 0 ? -1 : wave, pb + j, ch) : 0.0f;
                u[j] = __builtin_bit_cast(unsigned short, (__bf16)v);
            }
            fb[b] = __builtin_bit_cast(nrt_bf16x8, u);
        }
#pragma unroll
        for (int a = 0; a < TO; ++a)
#pragma unroll
            for (int b = 0; b < TI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        if (do_rs) {  // (after the products: the X fragments are dead by now)
#pragma unroll
            for (int a = 0; a < TO; ++a) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += (float)fa[a][j];
                rs[a] += t;
            }
        }
    };
    long p = p0;
    for (; p + 16 <= p1; p += 16) step(p, false);
    if (p < p1) step(p, true);
    if (do_rs) {  // uniform over the workgroup: one atomic per channel and workgroup
#pragma unroll
        for (int a = 0; a < TO; ++a) {
            const float t = rs[a] + __shfl_xor(rs[a], 32, 64);
            if (h == 0) rs_red[wave][32 * a + r] = t;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TO * 32; i += 256) {
            const int ch = o0 + i;
            if (ch < co) atomicAdd(db + (long)ch * pl.db_stride, (rs_red[0][i] + rs_red[1][i]) + (rs_red[2][i] + rs_red[3][i]));
        }
    }
    // The four waves of the workgroup hold four partial sums of the SAME output tiles (disjoint pixel ranges).  They are
    // combined through LDS first -- wave w sums tile t of all four for t % 4 == w -- and only then added to the result: a
    // quarter of the fp32 atomics (disabling them showed they were 40 % of the kernel: 83 -> 50 us per launch).
    extern __shared__ float wave_acc[];  // [4 waves][TO*TI tiles][16][64]
    constexpr int NT = TO * TI;
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) wave_acc[((wave * NT + a * TI + b) * 16 + q) * 64 + lane] = acc[a][b][q];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) {
            const int tl = a * TI + b;
            if ((tl & 3) != wave) continue;  // uniform per wave
            const int col = i0 + 32 * b + r;
            if (col >= ci) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = o0 + 32 * a + (q & 3) + 8 * (q >> 2) + 4 * h;
                float v = 0.0f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) v += wave_acc[((w4 * NT + tl) * 16 + q) * 64 + lane];
                if (row < co) atomicAdd(out + (long)row * ldo + col, v);
            }
        }
}
template <class XL, int PM = 0>
static void launch_dw16_x(int co, int ci, long K, const nrt_bf16* dY, const XL X, float* out, long ldo, float* db, hipStream_t s, int planes = 1,
                          const DwPlanes pl = DwPlanes{0, 0, 0, 1}) {
    // 64 x 128 outputs per wave (fewer when the layer is smaller); about 512 workgroups of 4 waves.  PM 0: >= 1024 pixels per
    // workgroup (256 per wave); PM 1: every wave walks the whole chunk of its plane, >= 256 pixels
    const int groups = ((co / 32 + 1) / 2) * ((ci / 32 + 3) / 4);
    const long min_chunk = PM == 0 ? 1024 : 256;
    long slices = 512 / ((long)groups * planes);
    if (slices > K / min_chunk) slices = K / min_chunk;
    if (slices < 1) slices = 1;
    const long chunk = ((K + slices - 1) / slices + 63) / 64 * 64;
    dim3 grid((unsigned)((K + chunk - 1) / chunk), groups, planes);
    if (co <= 32 && ci <= 64) {
        hipLaunchKernelGGL((nrt_dw16_kernel<1, 2, XL, PM>), dim3(grid.x, 1, planes), dim3(256), (size_t)4 * 2 * 16 * 64 * sizeof(float), s, co, ci, K,
                           dY, X, out, ldo, chunk, db, pl);
    } else {
        auto kern = nrt_dw16_kernel<2, 4, XL, PM>;
        const size_t lds = (size_t)4 * 8 * 16 * 64 * sizeof(float);  // 128 KiB
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, co, ci, K, dY, X, out, ldo, chunk, db, pl);
    }
}
template <class TX>
static void launch_dw16(int co, int ci, long K, const nrt_bf16* dY, const TX* X, float* out, long ldo, float* db, hipStream_t s) {
    launch_dw16_x<DwRows<TX>, 0>(co, ci, K, dY, DwRows<TX>{X, ci}, out, ldo, db, s);
}

// to fp32 at the boundary (d_featmap is fp32 in both modes)
template <class T>
__global__ void nrt_to_f32_kernel(size_t n, const T* __restrict__ x, float* __restrict__ y) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = nrt_ld(x + i);
}

template <>
void conv_bwd_x<nrt_bf16>(int M, int Nin, int Kout, const nrt_bf16* dy, const float* W, int mode, const nrt_bf16* res, nrt_bf16* dx,
                          float* wt, hipStream_t s) {
    n3dt_launch_chw_to_hwc(Kout, Nin, W, wt, s);  // W [Kout][Nin] -> [Nin][Kout]
    n3dt_launch_conv1x1_bwd_bf16(M, Nin, Kout, dy, wt, mode, res, dx, s);
}

// parameter gradients of a 1x1 conv: dW[co][ci] += dy^T x over M pixels, db[co] += column sums of dy
template <class T, class TX>
static void conv_bwd_w(int co, int ci, long M, const T* dy, const TX* x, float* dW, float* db, hipStream_t s);
template <>
void conv_bwd_w<float, float>(int co, int ci, long M, const float* dy, const float* x, float* dW, float* db, hipStream_t s) {
    Gemm32 w = mk(co, ci, (int)M, dy, co, 1, x, ci, 1, dW, ci);
    set_grad_split(w, M);
    n3dt_gemm32(w, s);
    launch_nrt_colsum<float>(dy, (long)co, M, co, db, s);
}
template <>
void conv_bwd_w<nrt_bf16, nrt_bf16>(int co, int ci, long M, const nrt_bf16* dy, const nrt_bf16* x, float* dW, float* db, hipStream_t s) {
    launch_dw16<nrt_bf16>(co, ci, M, dy, x, dW, ci, db, s);
}
template <>
void conv_bwd_w<nrt_bf16, float>(int co, int ci, long M, const nrt_bf16* dy, const float* x, float* dW, float* db, hipStream_t s) {
    launch_dw16<float>(co, ci, M, dy, x, dW, ci, db, s);
}

template <>
void conv_fwd<nrt_bf16>(int M, int N, int K, const void* x, int x16, const float* W, const float* b, nrt_bf16* y, hipStream_t s) {
    n3dt_launch_conv1x1_bf16(M, N, K, x, x16, W, b, 0.2f, y, s);
}

// featmap [nb][fs*fs][C] -> img [nb,3,P,P]; all intermediates kept in `saved`.
// T = storage of the maps (float: exact path, every product on the fp32 MFMA GEMM; nrt_bf16: mixed precision, bf16 maps and
// bf16 MFMA products).  The layouts count ELEMENTS, so the bf16 path addresses the same offsets on 2-byte elements and
// simply leaves the upper half of the float-sized buffers unused; the planar fp32 regions (img, rgb pyramid) keep their
// float offsets, which lie beyond every bf16 region.
template <class T>
static void nr_train_fwd(const N3dtGeom* g, int nb, const N3dtRenderParams* p, const float* featmap, float* img, float* saved_f,
                         float* ws_f, hipStream_t s) {
    constexpr int h16 = is16<T>();
    const NrSaved sv = nr_saved_layout(g, nb);
    const NrWs wl = nr_ws_layout(g, nb);
    const int C = g->feat_nc, nblk = g->n_blocks;
    T* saved = reinterpret_cast<T*>(saved_f);
    T* ws = reinterpret_cast<T*>(ws_f);
    float* rgbA = ws_f + wl.rgb0;
    float* rgbB = ws_f + wl.rgb1;
    float* img_saved = saved_f + sv.img;
    T* ps = ws + wl.a;
    int h = g->featmap_size;
    hipLaunchKernelGGL(nrt_to_rgb_kernel<float>, GRID1((size_t)nb * h * h), 3 * C * sizeof(float), s, nb, h * h, C, featmap, p->to_rgb_w[0],
                       p->to_rgb_b[0], (const float*)nullptr, rgbB, 0);
    hipLaunchKernelGGL(nrt_rgb_up_kernel, GRID1((size_t)nb * 3 * 4 * h * h), 0, s, nb * 3, h, h, rgbB, rgbA);
    const T* x = nullptr;  // level 0 reads the fp32 featmap
    for (int i = 0; i < nblk; ++i) {
        const int ci = nr_ch(C, i), co = nr_ch(C, i + 1), M = nb * h * h;
        conv_fwd<T>(M, 2 * ci, ci, i == 0 ? (const void*)featmap : (const void*)x, i == 0 ? 0 : h16, p->psu1_w[i], p->psu1_b[i],
                    saved + sv.t1[i], s);
        conv_fwd<T>(M, 4 * ci, 2 * ci, saved + sv.t1[i], h16, p->psu2_w[i], p->psu2_b[i], saved + sv.tv[i], s);
        if (i == 0)
            hipLaunchKernelGGL((nrt_shuffle_kernel<T, float>), GRID1((size_t)M * ci), 0, s, nb, h, h, ci, (const T*)(saved + sv.tv[i]), featmap, ps);
        else
            hipLaunchKernelGGL((nrt_shuffle_kernel<T, T>), GRID1((size_t)M * ci), 0, s, nb, h, h, ci, (const T*)(saved + sv.tv[i]), x, ps);
        h *= 2;
        hipLaunchKernelGGL(nrt_blur_kernel<T>, GRID1((size_t)nb * h * h * (ci / 4)), 0, s, nb, h, h, ci, (const T*)ps, saved + sv.bl[i]);
        conv_fwd<T>(nb * h * h, co, ci, saved + sv.bl[i], h16, p->feat_w[i], p->feat_b[i], saved + sv.net[i], s);
        const bool last = i == nblk - 1;
        hipLaunchKernelGGL(nrt_to_rgb_kernel<T>, GRID1((size_t)nb * h * h), 3 * co * sizeof(float), s, nb, h * h, co,
                           (const T*)(saved + sv.net[i]), p->to_rgb_w[i + 1], p->to_rgb_b[i + 1], (const float*)rgbA, last ? img_saved : rgbB,
                           last ? 1 : 0);
        if (!last) hipLaunchKernelGGL(nrt_rgb_up_kernel, GRID1((size_t)nb * 3 * 4 * h * h), 0, s, nb * 3, h, h, rgbB, rgbA);
        x = saved + sv.net[i];
    }
    const size_t P = (size_t)g->featmap_size << nblk;
    (void)hipMemcpyAsync(img, img_saved, sizeof(float) * nb * 3 * P * P, hipMemcpyDeviceToDevice, s);
}

// =====================================================================================================================
// Fused mixed-precision training path (bf16 maps), backward.  The forward (neural_render.hip: n3dt_launch_nr_train16_fwd)
// is the inference sequence with a save epilogue; what it keeps is listed in nr_train16.h.  Per block, from the last:
//   d_pre  = (d net from the next block + feat_2_rgb^T d rgb) * lrelu'(net)            nr16_dpre_kernel          raster [4M][CO]
//   d W_rgb, d b_rgb                                                                   (the exact path's reductions)
//   d hid  = Blur^T d_pre, written as the four sub-pixel planes                        nr16_blur_adj_q_kernel    [4][M][CO]
//   d Wf  += d hid^T ps,  ps = y + x.repeat rebuilt by the loader; d bf = column sums  nrt_dw16_kernel<DwPs>
//   d tv   = (d hid . Wf) * lrelu'(y)     -- the un-shuffle is the plane layout        gemm_h (gate epilogue)    [4][M][C]
//   d W2  += d tv_q^T t1 per plane (row 4c + q), d b2                                  nrt_dw16_kernel, grid.z = q
//   d t1   = (sum_q d tv_q . W2_q) * lrelu'(t1)                                        gemm_h, A = 4 regions     [M][2C]
//   d W1  += d t1^T x, d b1                                                            nrt_dw16_kernel
//   d x    = d t1 . W1 + sum_q d hid_q . R_q   (the residual's path, as in the forward) gemm_h, A = 5 regions    [M][C]
//   d rgb at the block's input resolution: Blur^T, bilinear^T                          (planar kernels)
// Against the layered version this drops, per block, the blur adjoint at full channel width (it runs on CO = C/2 channels
// here, the Blur being commuted behind feat_layers as in the forward), the un-shuffle pass, the separate gate pass and the
// three weight transposes (one pack launch per block instead).
#include "nr_train16.h"
#include "dw_rowmajor.h"
extern "C" void n3dt_launch_nr_train16_fwd(const N3dtGeom*, int, const N3dtRenderParams*, const float*, float*, unsigned char*, unsigned char*,
                                           hipStream_t);
extern "C" void n3dt_launch_gemm_regions_bf16(int, int, int, int, const void* const*, const int*, const int*, const float*, const void*, void*,
                                              hipStream_t);

// the three transposed / permuted fp32 matrices of every block (nr_train16.h: Nr16Wt); blockIdx.y = block
struct Nr16PackArgs {
    const float* W1[N3DT_MAX_BLOCKS];
    const float* W2[N3DT_MAX_BLOCKS];
    const float* Wf[N3DT_MAX_BLOCKS];
    float* out[N3DT_MAX_BLOCKS];
    int C[N3DT_MAX_BLOCKS], CO[N3DT_MAX_BLOCKS];
};
__global__ void nr16_pack_wt_kernel(const Nr16PackArgs a) {
    const int blk = blockIdx.y, C = a.C[blk], CO = a.CO[blk];
    const float* __restrict__ W1 = a.W1[blk];
    const float* __restrict__ W2 = a.W2[blk];
    const float* __restrict__ Wf = a.Wf[blk];
    float* __restrict__ out = a.out[blk];
    const Nr16Wt L = nr16_wt_layout(C, CO);
    const size_t n_g3 = (size_t)C * (2 * C + 4 * CO), total = L.g3 + n_g3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v;
        if (i < L.g2) {  // g1: Wt[c][o] = Wf[o][c]
            const int c = (int)(i / CO), o = (int)(i % CO);
            v = Wf[(size_t)o * C + c];
        } else if (i < L.g3) {  // g2: Wt[n][q C + c] = W2[4c + q][n]
            const size_t j = i - L.g2;
            const int n = (int)(j / (4 * C)), kk = (int)(j % (4 * C)), q = kk / C, c = kk % C;
            v = W2[(size_t)(4 * c + q) * (2 * C) + n];
        } else {  // g3: Wt[k][j] = W1[j][k] | R_q[o][k]
            const size_t j = i - L.g3;
            const int KK = 2 * C + 4 * CO, k = (int)(j / KK), kk = (int)(j % KK);
            if (kk < 2 * C) {
                v = W1[(size_t)kk * C + k];
            } else {
                const int r = kk - 2 * C, q = r / CO, o = r % CO;
                v = 0.0f;
                if ((k & 3) == q) {
                    const float* wr = Wf + (size_t)o * C;
                    const int c0 = (k - q) >> 2;
                    v = (wr[c0] + wr[c0 + C / 4]) + (wr[c0 + C / 2] + wr[c0 + 3 * (C / 4)]);  // the forward's summation order
                }
            }
        }
        out[i] = v;
    }
}

// d_pre[pix][k] = (d_in[pix][k] + sum_c d_rgb[img][c][p] Wrgb[c][k]) * lrelu'(net[pix][k]);  thread = pixel x 8 channels
__global__ __launch_bounds__(256) void nr16_dpre_kernel(int nb, int HW, int K, const float* __restrict__ d_rgb, const float* __restrict__ Wt,
                                                        const nrt_bf16* __restrict__ net, const nrt_bf16* __restrict__ d_in,
                                                        nrt_bf16* __restrict__ d_pre) {
    extern __shared__ float wl[];
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) wl[i] = Wt[i];
    __syncthreads();
    const int k8 = K / 8, k8s = 31 - __builtin_clz(k8);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nb * HW * k8) return;
    const int kq = (int)(i & (k8 - 1));
    const size_t pix = i >> k8s, img = pix / HW, p = pix % HW, o = img * 3 * (size_t)HW + p;
    const float d0 = d_rgb[o], d1 = d_rgb[o + HW], d2 = d_rgb[o + 2 * (size_t)HW];
    const nrt_u16x8 g = *reinterpret_cast<const nrt_u16x8*>(net + pix * K + 8 * kq);
    nrt_u16x8 din = (nrt_u16x8)(0);
    if (d_in) din = *reinterpret_cast<const nrt_u16x8*>(d_in + pix * K + 8 * kq);
    nrt_u16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = d0 * wl[8 * kq + j] + d1 * wl[K + 8 * kq + j] + d2 * wl[2 * K + 8 * kq + j];
        v += __builtin_bit_cast(float, (unsigned)din[j] << 16);
        const float y = __builtin_bit_cast(float, (unsigned)g[j] << 16);
        v = y > 0.0f ? v : 0.2f * v;
        out[j] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    *reinterpret_cast<nrt_u16x8*>(d_pre + pix * K + 8 * kq) = out;
}

// Adjoint of the reflect-border blur, raster [img][2H][2W][K] -> four sub-pixel planes [q][nb*H*W][K] (q = 2 di + dj is output
// pixel (2h + di, 2w + dj) of input pixel (h, w)).  One thread = one input pixel x 8 channels: the four 3 x 3 adjoint stencils
// share a 4 x 4 neighbourhood (16 loads of 16 bytes), weights from blur_adj_w3 (clamped taps carry weight 0).
__global__ __launch_bounds__(256) void nr16_blur_adj_q_kernel(int nb, int H, int W, int K, const nrt_bf16* __restrict__ d_pre,
                                                              nrt_bf16* __restrict__ d_hid) {
    const int k8 = K / 8, k8s = 31 - __builtin_clz(k8);
    const size_t Mq = (size_t)nb * H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mq * k8) return;
    const int kq = (int)(i & (k8 - 1));
    const size_t pix = i >> k8s;
    const int w = (int)(pix % W), h = (int)((pix / W) % H), img = (int)(pix / ((size_t)W * H));
    const int H2 = 2 * H, W2 = 2 * W;
    float wr[2][3], wc[2][3];
    blur_adj_w3(2 * h, H2, wr[0]);
    blur_adj_w3(2 * h + 1, H2, wr[1]);
    blur_adj_w3(2 * w, W2, wc[0]);
    blur_adj_w3(2 * w + 1, W2, wc[1]);
    float acc[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][b][j] = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {  // raster rows 2h-1 .. 2h+2
        const int r = min(max(2 * h - 1 + rr, 0), H2 - 1);
        float t4[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = min(max(2 * w - 1 + t, 0), W2 - 1);
            const nrt_u16x8 v = *reinterpret_cast<const nrt_u16x8*>(d_pre + (((size_t)img * H2 + r) * W2 + c) * K + 8 * kq);
#pragma unroll
            for (int j = 0; j < 8; ++j) t4[t][j] = __builtin_bit_cast(float, (unsigned)v[j] << 16);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // horizontal three taps for output columns 2w (taps t = 0..2) and 2w+1 (t = 1..3)
            const float h0 = wc[0][0] * t4[0][j] + wc[0][1] * t4[1][j] + wc[0][2] * t4[2][j];
            const float h1 = wc[1][0] * t4[1][j] + wc[1][1] * t4[2][j] + wc[1][2] * t4[3][j];
            if (rr < 3) {  // row 2h reads raster rows rr = 0..2
                acc[0][0][j] = fmaf(wr[0][rr < 3 ? rr : 0], h0, acc[0][0][j]);
                acc[0][1][j] = fmaf(wr[0][rr < 3 ? rr : 0], h1, acc[0][1][j]);
            }
            if (rr > 0) {  // row 2h+1 reads raster rows rr = 1..3
                acc[1][0][j] = fmaf(wr[1][rr > 0 ? rr - 1 : 0], h0, acc[1][0][j]);
                acc[1][1][j] = fmaf(wr[1][rr > 0 ? rr - 1 : 0], h1, acc[1][1][j]);
            }
        }
    }
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            nrt_u16x8 out;
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = __builtin_bit_cast(unsigned short, (__bf16)acc[di][dj][j]);
            *reinterpret_cast<nrt_u16x8*>(d_hid + ((size_t)(2 * di + dj) * Mq + pix) * K + 8 * kq) = out;
        }
}

// d_featmap[pix][k] = d_x[pix][k] (16-bit) + sum_c d_rgb0[img][c][p] Wrgb0[c][k]   (the stage-0 rgb branch), fp32 out
__global__ __launch_bounds__(256) void nr16_final_kernel(int nb, int HW, int K, const float* __restrict__ d_rgb, const float* __restrict__ Wt,
                                                         const nrt_bf16* __restrict__ dx, float* __restrict__ d_feat) {
    extern __shared__ float wl[];
    for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) wl[i] = Wt[i];
    __syncthreads();
    const int k4 = K / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nb * HW * k4) return;
    const int kq = (int)(i % k4);
    const size_t pix = i / k4, img = pix / HW, p = pix % HW, o = img * 3 * (size_t)HW + p;
    const float d0 = d_rgb[o], d1 = d_rgb[o + HW], d2 = d_rgb[o + 2 * (size_t)HW];
    const f32x4 x = nrt_ld4(dx + pix * K + 4 * kq);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = x[j] + (d0 * wl[4 * kq + j] + d1 * wl[K + 4 * kq + j] + d2 * wl[2 * K + 4 * kq + j]);
    *reinterpret_cast<f32x4*>(d_feat + pix * K + 4 * kq) = v;
}

static void nr_bwd16(const N3dtGeom* g, int nb, const N3dtRenderParams* p, const N3dtRenderGrads* gp, const float* featmap, const float* d_img,
                     const unsigned char* saved, float* d_featmap, unsigned char* ws, hipStream_t s) {
    const Nr16Saved sv = nr16_saved_layout(g, nb);
    const Nr16Ws wl = nr16_ws_layout(g, nb);
    const int C0 = g->feat_nc, nblk = g->n_blocks;
    const size_t P = (size_t)g->featmap_size << nblk;
    nrt_bf16* dpre = reinterpret_cast<nrt_bf16*>(ws + wl.dpre);
    nrt_bf16* dhid = reinterpret_cast<nrt_bf16*>(ws + wl.dhid);
    nrt_bf16* dtv = reinterpret_cast<nrt_bf16*>(ws + wl.dtv);
    nrt_bf16* dt1 = reinterpret_cast<nrt_bf16*>(ws + wl.dt1);
    nrt_bf16* dxa = reinterpret_cast<nrt_bf16*>(ws + wl.dxa);
    nrt_bf16* dxb = reinterpret_cast<nrt_bf16*>(ws + wl.dxb);
    float* drgb = reinterpret_cast<float*>(ws + wl.drgb);
    float* dtmp = reinterpret_cast<float*>(ws + wl.dtmp);
    // the transposed weights of every block (weights moved since the last step: packed per call)
    {
        Nr16PackArgs pa;
        for (int i = 0; i < nblk; ++i) {
            pa.C[i] = nr16_ch(C0, i); pa.CO[i] = nr16_ch(C0, i + 1);
            pa.W1[i] = p->psu1_w[i]; pa.W2[i] = p->psu2_w[i]; pa.Wf[i] = p->feat_w[i];
            pa.out[i] = reinterpret_cast<float*>(ws + wl.wt[i]);
        }
        hipLaunchKernelGGL(nr16_pack_wt_kernel, dim3(128, nblk), dim3(256), 0, s, pa);
    }
    // layer_2 / layer_1 weight gradients on the MLP's LDS-transposed kernel (row-major front end, dw_rowmajor.h) where the pixel
    // count allows it: entries 2 i (d W2: the four planes of d_tv as one wide operand against t1) and 2 i + 1 (d W1: d_t1 against x)
    N3dtDwRm dw[2 * N3DT_MAX_BLOCKS];
    bool dw_ok[2 * N3DT_MAX_BLOCKS];
    bool any_dw = false;
    for (int i = 0; i < nblk; ++i) {
        const int ci = nr16_ch(C0, i);
        const long hh = (long)g->featmap_size << i, M = (long)nb * hh * hh;
        // (gp == nullptr: frozen renderer -- single-image fitting -- no parameter gradient at all, only the dX chain)
        dw[2 * i] = N3dtDwRm{dtv, ci, 4, (size_t)M * ci, saved + sv.t1[i], 2 * ci, M, 4 * ci, 2 * ci, gp ? gp->psu2_w[i] : nullptr, 2 * ci, ci,
                             gp ? gp->psu2_b[i] : nullptr};
        dw[2 * i + 1] = N3dtDwRm{dt1, 2 * ci, 1, 0, i > 0 ? (const void*)(saved + sv.net[i - 1]) : nullptr, ci, M, 2 * ci, ci,
                                 gp ? gp->psu1_w[i] : nullptr, ci, 0, gp ? gp->psu1_b[i] : nullptr};
        const char* lds_env = getenv("N3DT_NR_DW_LDS");  // (read per call: a test flips it in-process)
        const bool off_env = lds_env && atoi(lds_env) == 0;
        const bool off = off_env || 2 * nblk > N3DT_DW_RM_MAX;  // (the reduction launch carries at most N3DT_DW_RM_MAX products)
        dw_ok[2 * i] = gp && !off && n3dt_dw_rowmajor_ok(&dw[2 * i]);
        dw_ok[2 * i + 1] = gp && !off && i > 0 && n3dt_dw_rowmajor_ok(&dw[2 * i + 1]);  // (block 0's x is the fp32 feature map)
        any_dw = any_dw || dw_ok[2 * i] || dw_ok[2 * i + 1];
    }
    float* dwpart = reinterpret_cast<float*>(ws + wl.dwpart);
    if (any_dw) n3dt_launch_dw_rowmajor_zero(dw, 2 * nblk, dwpart, s);
    int h = (int)P;
    hipLaunchKernelGGL(nrt_sigmoid_bwd_kernel, GRID1((size_t)nb * 3 * P * P), 0, s, (size_t)nb * 3 * P * P,
                       reinterpret_cast<const float*>(saved + sv.img), d_img, drgb);
    const nrt_bf16* dnet_in = nullptr;  // d net_i from block i + 1 (its d x), raster [4M][co]
    nrt_bf16* dx_out = dxa;
    for (int i = nblk - 1; i >= 0; --i) {
        const int ci = nr16_ch(C0, i), co = nr16_ch(C0, i + 1);
        const int hin = h / 2, M = nb * hin * hin, M4 = nb * h * h, HW = h * h;
        const nrt_bf16* net = reinterpret_cast<const nrt_bf16*>(saved + sv.net[i]);
        const nrt_bf16* y = reinterpret_cast<const nrt_bf16*>(saved + sv.y[i]);
        const nrt_bf16* t1 = reinterpret_cast<const nrt_bf16*>(saved + sv.t1[i]);
        const nrt_bf16* x16 = i > 0 ? reinterpret_cast<const nrt_bf16*>(saved + sv.net[i - 1]) : nullptr;
        const float* wt = reinterpret_cast<const float*>(ws + wl.wt[i]);
        const Nr16Wt L = nr16_wt_layout(ci, co);
        // feat_2_rgb[i + 1]: parameter gradients; d_pre
        if (gp) launch_to_rgb_wgrad<nrt_bf16>(nb, HW, co, drgb, net, gp->to_rgb_w[i + 1], s, gp->to_rgb_b[i + 1]);
        hipLaunchKernelGGL(nr16_dpre_kernel, GRID1((size_t)M4 * (co / 8)), 3 * co * sizeof(float), s, nb, HW, co, (const float*)drgb,
                           p->to_rgb_w[i + 1], net, dnet_in, dpre);
        hipLaunchKernelGGL(nr16_blur_adj_q_kernel, GRID1((size_t)M * (co / 8)), 0, s, nb, hin, hin, co, (const nrt_bf16*)dpre, dhid);
        // feat_layers: d Wf += d hid^T ps (+ d bf), over all four planes' pixels
        if (!gp) {
        } else if (i == 0)
            launch_dw16_x<DwPs<float>, 1>(co, ci, (long)M, dhid, DwPs<float>{y, featmap, (long)M, ci}, gp->feat_w[i], ci, gp->feat_b[i], s, 1,
                                          DwPlanes{(long)M * co, 0, 0, 1});
        else if (co > 32)
            launch_dw16_x<DwPs<nrt_bf16>, 1>(co, ci, (long)M, dhid, DwPs<nrt_bf16>{y, x16, (long)M, ci}, gp->feat_w[i], ci, gp->feat_b[i], s, 1,
                                             DwPlanes{(long)M * co, 0, 0, 1});
        else  // the 32 x 64 block: the four planes as 4 M rows, waves split the pixels
            launch_dw16_x<DwPs<nrt_bf16>, 0>(co, ci, (long)M4, dhid, DwPs<nrt_bf16>{y, x16, (long)M, ci}, gp->feat_w[i], ci, gp->feat_b[i], s);
        {   // d tv (planes) = (d hid . Wf) * lrelu'(y)
            const void* base[1] = {dhid};
            const int width[1] = {co}, ld[1] = {co};
            n3dt_launch_gemm_regions_bf16(M4, ci, co, 1, base, width, ld, wt + L.g1, y, dtv, s);
        }
        // layer_2: d W2[4c + q][:] += d tv_q^T t1 (+ d b2[4c + q]), one product per plane in one launch
        if (dw_ok[2 * i]) n3dt_launch_dw_rowmajor_one(dw, 2 * i, dwpart, s);
        else if (gp)
            launch_dw16_x<DwRows<nrt_bf16>, 0>(ci, 2 * ci, (long)M, dtv, DwRows<nrt_bf16>{t1, 2 * ci}, gp->psu2_w[i], (long)4 * 2 * ci, gp->psu2_b[i], s, 4,
                                               DwPlanes{(long)M * ci, (long)2 * ci, 1, 4});
        {   // d t1 = (sum_q d tv_q . W2_q) * lrelu'(t1)
            const void* base[4] = {dtv, dtv + (size_t)M * ci, dtv + (size_t)2 * M * ci, dtv + (size_t)3 * M * ci};
            const int width[4] = {ci, ci, ci, ci}, ld[4] = {ci, ci, ci, ci};
            n3dt_launch_gemm_regions_bf16(M, 2 * ci, 4 * ci, 4, base, width, ld, wt + L.g2, t1, dt1, s);
        }
        // layer_1: d W1 += d t1^T x (+ d b1)
        if (dw_ok[2 * i + 1]) n3dt_launch_dw_rowmajor_one(dw, 2 * i + 1, dwpart, s);
        else if (!gp) {
        } else if (i == 0) launch_dw16<float>(2 * ci, ci, (long)M, dt1, featmap, gp->psu1_w[i], ci, gp->psu1_b[i], s);
        else launch_dw16<nrt_bf16>(2 * ci, ci, (long)M, dt1, x16, gp->psu1_w[i], ci, gp->psu1_b[i], s);
        {   // d x = d t1 . W1 + sum_q d hid_q . R_q
            const void* base[5] = {dt1, dhid, dhid + (size_t)M * co, dhid + (size_t)2 * M * co, dhid + (size_t)3 * M * co};
            const int width[5] = {2 * ci, co, co, co, co}, ld[5] = {2 * ci, co, co, co, co};
            n3dt_launch_gemm_regions_bf16(M, ci, 2 * ci + 4 * co, 5, base, width, ld, wt + L.g3, nullptr, dx_out, s);
        }
        // rgb pyramid: the running rgb at this block's output came from rgb_upsample of the sum at its input resolution
        h = hin;
        hipLaunchKernelGGL(nrt_blur_adj_planar_kernel, GRID1((size_t)nb * 3 * 4 * h * h), 0, s, nb * 3, 2 * h, 2 * h, drgb, dtmp);
        hipLaunchKernelGGL(nrt_bilinear_adj_kernel, GRID1((size_t)nb * 3 * h * h), 0, s, nb * 3, h, h, dtmp, drgb);
        dnet_in = dx_out;
        dx_out = dx_out == dxa ? dxb : dxa;
    }
    // stage-0 rgb: feat_2_rgb_list[0](featmap); d featmap = d x_0 + its branch
    {
        const int fs = g->featmap_size, HW = fs * fs;
        if (gp) launch_to_rgb_wgrad<float>(nb, HW, C0, drgb, featmap, gp->to_rgb_w[0], s, gp->to_rgb_b[0]);
        hipLaunchKernelGGL(nr16_final_kernel, GRID1((size_t)nb * HW * (C0 / 4)), 3 * C0 * sizeof(float), s, nb, HW, C0, (const float*)drgb,
                           p->to_rgb_w[0], dnet_in, d_featmap);
    }
    if (any_dw) {  // products that did not take this route left their partial buffers zero: the reduction adds nothing for them
        n3dt_launch_dw_rowmajor_reduce(dw, 2 * nblk, dwpart, s);
    }
}

static bool nr16_enabled(const N3dtGeom* g) {
    // N3DT_NR_TRAIN_FUSED=0: the layered bf16 path (A/B).  LATCHED once per process: the two paths lay `saved` and the workspace
    // out differently (floats against byte offsets of bf16 planes), so a switch that moved between a forward and its backward
    // would have the backward read `saved` under the wrong layout.  (N3DT_NR_DW_LDS / N3DT_NR_BLUR_MFMA pick kernels inside one
    // layout and stay per-call switches: the tests compare both forms in one process.)
    static const bool on = [] {
        const char* e = getenv("N3DT_NR_TRAIN_FUSED");
        return !e || atoi(e) != 0;
    }();
    return on && nr16_supported(g);
}
extern "C" size_t n3dt_nr_train16_saved_bytes(const N3dtGeom* g, int nb) { return nr16_saved_layout(g, nb).total; }
extern "C" size_t n3dt_nr_train16_ws_bytes(const N3dtGeom* g, int nb) { return nr16_ws_layout(g, nb).total; }

extern "C" void n3dt_launch_nr_train_fwd(const N3dtGeom* g, int nb, const N3dtRenderParams* p, const float* featmap, float* img,
                                         float* saved, float* ws, int bf16, hipStream_t s) {
    if (bf16 && nr16_enabled(g))
        n3dt_launch_nr_train16_fwd(g, nb, p, featmap, img, reinterpret_cast<unsigned char*>(saved), reinterpret_cast<unsigned char*>(ws), s);
    else if (bf16) nr_train_fwd<nrt_bf16>(g, nb, p, featmap, img, saved, ws, s);
    else nr_train_fwd<float>(g, nb, p, featmap, img, saved, ws, s);
}

// gradients are ACCUMULATED into gp (same pointer layout as the parameters); d_featmap is overwritten
template <class T>
static void nr_bwd(const N3dtGeom* g, int nb, const N3dtRenderParams* p, const N3dtRenderGrads* gp, const float* featmap,
                   const float* d_img, const float* saved_f, float* d_featmap, float* ws_f, hipStream_t s) {
    const NrSaved sv = nr_saved_layout(g, nb);
    const NrWs wl = nr_ws_layout(g, nb);
    const int C = g->feat_nc, nblk = g->n_blocks;
    const size_t P = (size_t)g->featmap_size << nblk;
    const T* saved = reinterpret_cast<const T*>(saved_f);
    T* ws = reinterpret_cast<T*>(ws_f);
    T* bufA = ws + wl.a;
    T* bufB = ws + wl.b;
    T* bufC = ws + wl.c;
    float* drgb = ws_f + wl.rgb0;   // gradient w.r.t. the running rgb sum at the current resolution
    float* dtmp = ws_f + wl.rgb1;
    int h = (int)P;
    hipLaunchKernelGGL(nrt_sigmoid_bwd_kernel, GRID1((size_t)nb * 3 * P * P), 0, s, (size_t)nb * 3 * P * P, saved_f + sv.img, d_img, drgb);
    T* dnet = bufA;  // gradient w.r.t. net_i (stage output), [nb*h*h][co]
    for (int i = nblk - 1; i >= 0; --i) {
        const int ci = nr_ch(C, i), co = nr_ch(C, i + 1);
        const int hin = h / 2, M = nb * hin * hin, M4 = nb * h * h, HW = h * h;
        const T* net = saved + sv.net[i];
        // rgb = rgb_prev_up + feat_2_rgb[i+1](net): parameter grads, then d net (gated by lrelu'(net))
        if (gp) {  // (gp == nullptr: frozen renderer, only the input gradient is wanted)
            launch_to_rgb_wgrad<T>(nb, HW, co, drgb, net, gp->to_rgb_w[i + 1], s);
            hipLaunchKernelGGL(nrt_rgb_bias_kernel, dim3(64, 3), dim3(256), 0, s, nb, HW, drgb, gp->to_rgb_b[i + 1]);
        }
        // d net: from the rgb branch (+ from the next stage's input gradient, already in dnet when i < nblk-1)
        if (i == nblk - 1) {
            hipLaunchKernelGGL(nrt_to_rgb_bwd_kernel<T>, GRID1((size_t)M4 * (co / 4)), 3 * co * sizeof(float), s, nb, HW, co, drgb,
                               p->to_rgb_w[i + 1], net, dnet, 0);
        } else {
            // dnet currently holds dL/d(net) from stage i+1 (ungated); add the rgb branch, then gate once
            hipLaunchKernelGGL(nrt_to_rgb_bwd_kernel<T>, GRID1((size_t)M4 * (co / 4)), 3 * co * sizeof(float), s, nb, HW, co, drgb,
                               p->to_rgb_w[i + 1], (const T*)nullptr, dnet, 1);
            hipLaunchKernelGGL(nrt_gate_kernel<T>, GRID1((size_t)M4 * co), 0, s, (size_t)M4 * co, net, dnet);
        }
        // feat conv: net = lrelu(bl Wf^T + bf)
        {
            if (gp) conv_bwd_w<T, T>(co, ci, M4, (const T*)dnet, saved + sv.bl[i], gp->feat_w[i], gp->feat_b[i], s);
            conv_bwd_x<T>(M4, ci, co, dnet, p->feat_w[i], 0, nullptr, bufB, ws_f + wl.wt, s);  // d bl
        }
        // blur adjoint -> d ps (bufC), then un-shuffle into d tv (bufB, gated) and the residual gradient (bufA)
        hipLaunchKernelGGL(nrt_blur_adj_kernel<T>, GRID1((size_t)M4 * (ci / 4)), 0, s, nb, h, h, ci, (const T*)bufB, bufC);
        hipLaunchKernelGGL(nrt_unshuffle_kernel<T>, GRID1((size_t)M * ci), 0, s, nb, hin, hin, ci, (const T*)bufC, saved + sv.tv[i], bufB, bufA);
        T* dtv = bufB;    // [M][4ci]
        T* dxres = bufA;  // [M][ci]
        // layer_2: tv = lrelu(t1 W2^T + b2)
        {
            if (gp) conv_bwd_w<T, T>(4 * ci, 2 * ci, M, (const T*)dtv, saved + sv.t1[i], gp->psu2_w[i], gp->psu2_b[i], s);
            conv_bwd_x<T>(M, 2 * ci, 4 * ci, dtv, p->psu2_w[i], 1, saved + sv.t1[i], bufC, ws_f + wl.wt, s);  // d t1, gated by lrelu'(t1)
        }
        // layer_1: t1 = lrelu(x W1^T + b1);  dx = dt1 W1 + residual gradient
        {
            if (!gp) {
            } else if (i == 0) conv_bwd_w<T, float>(2 * ci, ci, M, (const T*)bufC, featmap, gp->psu1_w[i], gp->psu1_b[i], s);
            else conv_bwd_w<T, T>(2 * ci, ci, M, (const T*)bufC, saved + sv.net[i - 1], gp->psu1_w[i], gp->psu1_b[i], s);
            conv_bwd_x<T>(M, ci, 2 * ci, bufC, p->psu1_w[i], 2, dxres, dxres, ws_f + wl.wt, s);  // dx = dt1 W1 + residual gradient
        }
        // rgb pyramid: at stage i > 0 the running rgb came from rgb_upsample of the previous sum
        h = hin;
        if (i > 0) {
            hipLaunchKernelGGL(nrt_blur_adj_planar_kernel, GRID1((size_t)nb * 3 * 4 * h * h), 0, s, nb * 3, 2 * h, 2 * h, drgb, dtmp);
            hipLaunchKernelGGL(nrt_bilinear_adj_kernel, GRID1((size_t)nb * 3 * h * h), 0, s, nb * 3, h, h, dtmp, drgb);
        }
        dnet = dxres;  // = dL/d(x_i) = dL/d(net_{i-1}) (ungated) for the next iteration; lives in bufA
    }
    // stage-0 rgb: rgb_upsample(feat_2_rgb_list[0](featmap))
    {
        const int fs = g->featmap_size, HW = fs * fs;
        hipLaunchKernelGGL(nrt_blur_adj_planar_kernel, GRID1((size_t)nb * 3 * 4 * HW), 0, s, nb * 3, 2 * fs, 2 * fs, drgb, dtmp);
        hipLaunchKernelGGL(nrt_bilinear_adj_kernel, GRID1((size_t)nb * 3 * HW), 0, s, nb * 3, fs, fs, dtmp, drgb);
        if (gp) {
            launch_to_rgb_wgrad<float>(nb, HW, C, drgb, featmap, gp->to_rgb_w[0], s);
            hipLaunchKernelGGL(nrt_rgb_bias_kernel, dim3(64, 3), dim3(256), 0, s, nb, HW, drgb, gp->to_rgb_b[0]);
        }
        hipLaunchKernelGGL(nrt_to_rgb_bwd_kernel<T>, GRID1((size_t)nb * HW * (C / 4)), 3 * C * sizeof(float), s, nb, HW, C, drgb,
                           p->to_rgb_w[0], (const T*)nullptr, dnet, 1);
        const size_t n = (size_t)nb * HW * C;
        hipLaunchKernelGGL(nrt_to_f32_kernel<T>, GRID1(n), 0, s, n, (const T*)dnet, d_featmap);
    }
}

extern "C" void n3dt_launch_nr_bwd(const N3dtGeom* g, int nb, const N3dtRenderParams* p, const N3dtRenderGrads* gp,
                                   const float* featmap, const float* d_img, const float* saved, float* d_featmap, float* ws,
                                   int bf16, hipStream_t s) {
    if (bf16 && nr16_enabled(g))
        nr_bwd16(g, nb, p, gp, featmap, d_img, reinterpret_cast<const unsigned char*>(saved), d_featmap, reinterpret_cast<unsigned char*>(ws), s);
    else if (bf16) nr_bwd<nrt_bf16>(g, nb, p, gp, featmap, d_img, saved, d_featmap, ws, s);
    else nr_bwd<float>(g, nb, p, gp, featmap, d_img, saved, d_featmap, ws, s);
}
