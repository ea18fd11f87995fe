// Training path of the volumetric render: forward that keeps the per-layer
// activations, and the backward through compositing, the MLP and the latent folding.
//
// This file: the exact-fp32 path, layer by layer on point-major activations [P][C] in HBM, every product a
// launch of the generic fp32 MFMA GEMM (gemm32.h).  It trades the fused kernel's zero-traffic design for
// a backward that is simple to verify against the reference's autograd (tests/golden grad fixtures).
// The fused mixed-precision path (bf16 MFMA, fp32 gradients) lives in train_x16.inc, included at the end.
//
// Reference code differentiated here: NetWorks/models.py:62-87 (MLP), NetWorks/utils.py:268-309
// (compositing), NetWorks/HeadNeRFNet.py:84-112,149-152 (latent concat, merge).
#include "gemm32.h"
#include "n3dt_device.h"
#include "n3dt_layout.h"

// diagnostic: dst[r][c] = bf16(src[r][c]) as fp32, for n = rows * cols elements (cols per row, leading dimensions given)
__global__ void train_round_bf16_kernel(size_t n, int cols, long ld_src, long ld_dst, const float* __restrict__ src, float* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols, c = i % cols;
        dst[r * ld_dst + c] = (float)(__bf16)src[r * ld_src + c];
    }
}

#define XR_LD 388  // RGB_layer_0 output (384) | density pre-activation (col 384) | pad

struct TrainSaved {  // float offsets into the saved buffer
    size_t cat5, geo, h[8], xr, g, w, ray, fold, total;
};

static inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

static TrainSaved saved_layout(const N3dtGeom* g) {
    TrainSaved s;
    const size_t P = (size_t)g->batch * g->n_rays * g->n_samples, R = (size_t)g->batch * g->n_rays;
    size_t o = 0;
    s.cat5 = o; o += al64(P * 448);       // [P][448]: PE (64) | H4 (384)
    s.geo = o; o += al64(P * 2);          // dist, zval
    for (int l = 0; l < 8; ++l) {
        if (l == 4) { s.h[l] = s.cat5; continue; }  // H4 lives inside cat5 (column 64, ld 448)
        s.h[l] = o; o += al64(P * 384);
    }
    s.xr = o; o += al64(P * XR_LD);
    s.g = o; o += al64(P * 192);
    s.w = o; o += al64(P);
    s.ray = o; o += al64(R * N3DT_PART_STRIDE);  // per ray: G[192], wsum, dsum, tprod, pad
    s.fold = o; o += al64(n3dt_fold_region_floats(g->batch, g->n_rays, g->vd_dim));  // (+ the per-ray RGB_layer_1 bias table of include_vd)
    s.total = o;
    return s;
}

struct TrainWs {  // float offsets into the workspace
    size_t w5p, wc, bc, dha, dhb, dxr, dg, dgray, dwsum, dw5p, dwc, dfold, dpe, total;
};

static TrainWs ws_layout(const N3dtGeom* g) {
    TrainWs w;
    const size_t P = (size_t)g->batch * g->n_rays * g->n_samples, R = (size_t)g->batch * g->n_rays;
    size_t o = 0;
    w.w5p = o; o += al64(384 * 448);   // FeaExt_module_5 without its latent columns, PE padded to 64
    w.wc = o; o += al64(385 * 384);    // [RGB_layer_0 ; density_module]
    w.bc = o; o += al64(385);
    w.dha = o; o += al64(P * 384);
    w.dhb = o; o += al64(P * 384);
    w.dxr = o; o += al64(P * XR_LD);
    w.dg = o; o += al64(P * 192);
    w.dgray = o; o += al64(R * 192);
    w.dwsum = o; o += al64(R);
    w.dw5p = o; o += al64(384 * 448);
    w.dwc = o; o += al64(385 * 384);
    w.dfold = o; o += al64((size_t)g->batch * N3DT_FOLD_STRIDE);
    w.dpe = o; o += al64(P * 64);  // d PE, only touched when camera gradients are requested
    w.total = o;
    return w;
}

extern "C" size_t n3dt_train_saved_floats(const N3dtGeom* g) { return saved_layout(g).total; }
extern "C" size_t n3dt_train_ws_floats(const N3dtGeom* g) { return ws_layout(g).total; }

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
// per point: sample position, accurate PE into cat5[:, 0:64], (dist, zval) into geo
__global__ void train_sample_pe_kernel(N3dtGeom g, const float* __restrict__ xy, const float* __restrict__ R,
                                       const float* __restrict__ T, const float* __restrict__ Kinv,
                                       const float* __restrict__ t_rand, float* __restrict__ cat5, float* __restrict__ geo) {
    const size_t P = (size_t)g.batch * g.n_rays * g.n_samples;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread = (point, 16 PE rows)
    const size_t pt = i >> 2;
    const int q = (int)(i & 3);
    if (pt >= P) return;
    const int s = (int)(pt % g.n_samples);
    const size_t rg = pt / g.n_samples;
    const int ray = (int)(rg % g.n_rays), b = (int)(rg / g.n_rays);
    float p[3], dist, zval;
    n3dt_sample_point(g, xy, R, T, Kinv, t_rand, b, ray, s, p, dist, zval);
    float* row = cat5 + pt * 448;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = 4 * k + q;
        row[r] = n3dt_pe_row_accurate(p, r);
    }
    if (q == 0) {
        geo[2 * pt] = dist;
        geo[2 * pt + 1] = zval;
    }
}

// W5' [384][448] = W5[:, 0:63] | 0 | W5[:, 63+S : 63+S+384];  Wc [385][384] = [Wr0 ; wd], bc = [br0 ; bd]
__global__ void train_pack_kernel(N3dtMlpParams p, int S, float* __restrict__ w5p, float* __restrict__ wc, float* __restrict__ bc) {
    const int in5 = N3DT_PE_DIM + S + N3DT_HID;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 384 * 448) {
        const int r = i / 448, c = i % 448;
        float v = 0.0f;
        if (c < 63) v = p.weight[5][(size_t)r * in5 + c];
        else if (c >= 64) v = p.weight[5][(size_t)r * in5 + 63 + S + (c - 64)];
        w5p[i] = v;
    }
    if (i < 385 * 384) wc[i] = i < 384 * 384 ? p.weight[9][i] : p.weight[8][i - 384 * 384];
    if (i < 385) bc[i] = i < 384 ? p.bias[9][i] : p.bias[8][0];
}

// one wave per ray: weights, per-ray composite record (NetWorks/utils.py:273-309)
__global__ void train_composite_fwd_kernel(N3dtGeom g, const float* __restrict__ xr, const float* __restrict__ G,
                                           const float* __restrict__ geo, float* __restrict__ w_out, float* __restrict__ rayrec) {
    const long R = (long)g.batch * g.n_rays;
    const long ray = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= R) return;
    const int Ns = g.n_samples;
    const size_t p0 = (size_t)ray * Ns;
    float acc[3] = {0.f, 0.f, 0.f};
    float Trun = 1.0f, wsum = 0.0f, dsum = 0.0f;
    for (int s0 = 0; s0 < Ns; s0 += 64) {
        const int s = s0 + lane;
        float alpha = 0.0f, zv = 0.0f;
        if (s < Ns) {
            const float sigma = fmaxf(xr[(p0 + s) * XR_LD + 384], 0.0f);
            alpha = 1.0f - expf(-sigma * geo[2 * (p0 + s)]);
            zv = geo[2 * (p0 + s) + 1];
        }
        const float x = 1.0f - alpha + 1e-10f;
        // exclusive product over the 64 lanes
        float incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            float o = __shfl_up(incl, off, 64);
            if (lane >= off) incl *= o;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = alpha * Trun * excl;
        if (s < Ns) w_out[p0 + s] = w;
        float ws = w, ds = w * zv;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ws += __shfl_xor(ws, off, 64);
            ds += __shfl_xor(ds, off, 64);
        }
        wsum += ws;
        dsum += ds;
        // weighted feature sums: lane handles channels lane, lane+64, lane+128
        const int n = min(64, Ns - s0);
        for (int k = 0; k < n; ++k) {
            const float wk = __shfl(w, k, 64);
            const float* gr = G + (p0 + s0 + k) * 192;
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] += wk * gr[lane + 64 * i];
        }
        Trun *= __shfl(incl, 63, 64);
    }
    float* rec = rayrec + (size_t)ray * N3DT_PART_STRIDE;
#pragma unroll
    for (int i = 0; i < 3; ++i) rec[lane + 64 * i] = acc[i];
    if (lane == 0) {
        rec[N3DT_G + 0] = wsum;
        rec[N3DT_G + 1] = dsum;
        rec[N3DT_G + 2] = Trun;
        rec[N3DT_G + 3] = 0.0f;
    }
}

// per ray (256 threads = channels): head backward.
//   fg = W2 Gray + b2 wsum ; bg_alpha = 1 - wsum ; merge = fg + bg_alpha * bg
// d_fg_total = d_merge + d_fg ; d_ba_total = d_bg_alpha + sum_c d_merge[c] bg[c]
// outputs dGray[192] = W2^T d_fg_total ; d_wsum = b2 . d_fg_total - d_ba_total ; dfg_total saved for dW2
// HB_RAYS rays per workgroup: one workgroup per ray read all of W2 (196 KB) from L2 for 192 x 256 FMAs -- 1.6 GB of L2 reads per
// call at two heads (71 us); eight rays share each W2 element (20 us)
#define HB_RAYS 8
__global__ __launch_bounds__(256) void train_head_bwd_kernel(N3dtGeom g, const float* __restrict__ W2 /*[256][192]*/,
                                                             const float* __restrict__ b2, const float* __restrict__ bg,
                                                             const float* __restrict__ d_merge, const float* __restrict__ d_fg,
                                                             const float* __restrict__ d_ba, float* __restrict__ dfg_total,
                                                             float* __restrict__ dgray, float* __restrict__ dwsum, long n_rays_total) {
    __shared__ float sd[HB_RAYS][256];
    __shared__ float red[HB_RAYS][4];
    const long rg0 = (long)blockIdx.x * HB_RAYS;
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const float b2c = b2[c];
#pragma unroll
    for (int r = 0; r < HB_RAYS; ++r) {
        const long rg = rg0 + r;
        float df = 0.0f, part = 0.0f;
        if (rg < n_rays_total) {
            const int ray = (int)(rg % g.n_rays);
            const float dm = d_merge ? d_merge[rg * 256 + c] : 0.0f;
            df = dm + (d_fg ? d_fg[rg * 256 + c] : 0.0f);
            dfg_total[rg * 256 + c] = df;
            part = b2c * df - (d_merge ? dm * bg[(size_t)c * g.n_rays + ray] : 0.0f);
        }
        sd[r][c] = df;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) red[r][wave] = part;
    }
    __syncthreads();
    if (c < HB_RAYS && rg0 + c < n_rays_total)
        dwsum[rg0 + c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]) - (d_ba ? d_ba[rg0 + c] : 0.0f);
    if (c < 192) {
        float acc[HB_RAYS];
#pragma unroll
        for (int r = 0; r < HB_RAYS; ++r) acc[r] = 0.0f;
#pragma unroll 4
        for (int k = 0; k < 256; ++k) {
            const float w = W2[k * 192 + c];
#pragma unroll
            for (int r = 0; r < HB_RAYS; ++r) acc[r] = fmaf(w, sd[r][k], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < HB_RAYS; ++r)
            if (rg0 + r < n_rays_total) dgray[(rg0 + r) * 192 + c] = acc[r];
    }
}

// d_bg_featmap[c][ray] += sum_b bg_alpha[b][ray] * d_merge[b][ray][c];  db2[c] += sum_rays dfg_total * wsum
// block = BG_RAYS consecutive rays x 256 channels: one db2 atomic per channel and block (one block per ray made every
// block hit the same 256 addresses: 4096 same-address atomics each, 107 us)
#define BG_RAYS 16
__global__ void train_bg_b2_grad_kernel(N3dtGeom g, const float* __restrict__ rayrec, const float* __restrict__ d_merge,
                                        const float* __restrict__ dfg_total, float* __restrict__ d_bg, float* __restrict__ db2) {
    const int c = threadIdx.x;  // 256 threads
    float accb = 0.0f;
    for (int ray = blockIdx.x * BG_RAYS; ray < min(g.n_rays, (blockIdx.x + 1) * BG_RAYS); ++ray) {
        float acc = 0.0f;
        for (int b = 0; b < g.batch; ++b) {
            const long rg = (long)b * g.n_rays + ray;
            const float wsum = rayrec[rg * N3DT_PART_STRIDE + N3DT_G];
            if (d_merge) acc += (1.0f - wsum) * d_merge[rg * 256 + c];
            accb += wsum * dfg_total[rg * 256 + c];
        }
        if (d_bg && d_merge) d_bg[(size_t)c * g.n_rays + ray] += acc;
    }
    if (db2) atomicAdd(&db2[c], accb);  // (nullptr: frozen network)
}

// one wave per ray: compositing backward (see DESIGN.md for the derivation)
//   dw_s = dGray . G_s + d_wsum ;  dalpha_s = dw_s T_s - (sum_{t>s} dw_t w_t) / x_s
//   dsigma_s = dalpha_s * dist_s * (1 - alpha_s) * [sigma_pre > 0]   -> dxr[:, 384]
//   dG[p][j] = w_s * dGray[j] * [G > 0]
__global__ void train_composite_bwd_kernel(N3dtGeom g, const float* __restrict__ xr, const float* __restrict__ G,
                                           const float* __restrict__ geo, const float* __restrict__ w_in,
                                           const float* __restrict__ dgray, const float* __restrict__ dwsum,
                                           float* __restrict__ dG, float* __restrict__ dxr) {
    const long R = (long)g.batch * g.n_rays;
    const long ray = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= R) return;
    const int Ns = g.n_samples;
    const size_t p0 = (size_t)ray * Ns;
    float dgr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dgr[i] = dgray[ray * 192 + lane + 64 * i];
    const float dws = dwsum[ray];
    // walk the ray back to front carrying S = sum_{t>s} dw_t w_t; within a 64-sample chunk use a suffix scan
    float carry = 0.0f;
    const int nchunk = (Ns + 63) / 64;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int s0 = ch * 64, n = min(64, Ns - s0);
        // dw_s for this chunk: lane k computes the dot over its 3 channels for every sample, then reduce
        float dw = 0.0f;
        for (int k = 0; k < n; ++k) {
            const float* gr = G + (p0 + s0 + k) * 192;
            float part = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) part += dgr[i] * gr[lane + 64 * i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            if (lane == k) dw = part + dws;
        }
        const int s = s0 + lane;
        float w = 0.0f, sig = 0.0f, dist = 0.0f;
        if (s < Ns) {
            w = w_in[p0 + s];
            sig = xr[(p0 + s) * XR_LD + 384];
            dist = geo[2 * (p0 + s)];
        }
        const float sigma = fmaxf(sig, 0.0f);
        const float e = expf(-sigma * dist);  // 1 - alpha
        const float alpha = 1.0f - e;
        const float x = e + 1e-10f;           // == 1 - alpha + 1e-10 up to rounding
        float v = (s < Ns) ? dw * w : 0.0f;
        // exclusive suffix sum over lanes: suf[lane] = sum_{l > lane} v[l]
        float incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            float o = __shfl_down(incl, off, 64);
            if (lane + off < 64) incl += o;
        }
        const float suf = incl - v + carry;
        const float T = alpha > 0.0f ? w / alpha : 0.0f;  // only used multiplied by dw where alpha > 0 matters
        float dalpha = dw * T - suf / x;
        if (!(alpha > 0.0f)) {
            // alpha == 0 (sigma <= 0 or dist == 0): T is not recoverable from w; recompute it from the prefix
            // product lazily -- the relu gate below zeroes dsigma whenever sigma_pre <= 0, and dist == 0 zeroes it too
            dalpha = 0.0f;
        }
        float dsig = dalpha * dist * e;
        if (!(sig > 0.0f)) dsig = 0.0f;
        if (s < Ns) {
            dxr[(p0 + s) * XR_LD + 384] = dsig;
            dxr[(p0 + s) * XR_LD + 385] = dalpha * sigma * e;  // d dist (pad column; consumed by the camera backward only)
        }
        carry += __shfl(incl, 0, 64);
        // dG rows
        for (int k = 0; k < n; ++k) {
            const float wk = __shfl(w, k, 64);
            const float* gr = G + (p0 + s0 + k) * 192;
            float* dgo = dG + (p0 + s0 + k) * 192;
#pragma unroll
            for (int i = 0; i < 3; ++i) dgo[lane + 64 * i] = gr[lane + 64 * i] > 0.0f ? wk * dgr[i] : 0.0f;
        }
    }
}

#define CAM_RAYS 64
// Rays per workgroup of the camera backward kernels: CAM_RAYS where there are many rays (each block ends in 12 atomics on the
// same 12 addresses per frame: one block per 4 rays meant 2 048 atomics per address at config 3), fewer where there are few -- a
// single 32 x 32 frame (single-image fitting) was 16 workgroups on 256 CUs: 348 us for 65 536 points.
static inline int n3dt_cam_rays_per_block(int n_rays, int batch) {
    int r = CAM_RAYS;
    while (r > 4 && (long)((n_rays + r - 1) / r) * batch < 512) r >>= 1;
    return r;
}
// Camera backward (the single-image fitting use-case): one wave per ray.
//   p_s = T + (d l) z_s,  dist_s = (z_{s+1} - z_s) l,  d = w/|w|,  w = R c,  c = Kinv [x, y, 1],  l = -1/d_z
//   PE rows: [p, sin(2^k p), cos(2^k p)]  ->  dp = dPE_p + sum_k 2^k (cos * dPE_sin - sin * dPE_cos)
// every sample edge moves 1:1 with T_z (utils.py:125-126,142 and the convex jitter of :73-78), so dist does not
// depend on T and the edge term of the points is dp . (d l).
__global__ void train_camera_bwd_kernel(N3dtGeom g, const float* __restrict__ xy, const float* __restrict__ R,
                                        const float* __restrict__ T, const float* __restrict__ Kinv,
                                        const float* __restrict__ t_rand, const float* __restrict__ cat5,
                                        const float* __restrict__ dpe, const float* __restrict__ dxr, float* __restrict__ d_R,
                                        float* __restrict__ d_T, const int cam_rays) {
    // block = cam_rays (<= CAM_RAYS) consecutive rays of ONE frame (blockIdx.y), 4 waves taking rays in turn; the 12 results are summed
    // in registers and LDS and leave as 12 atomics per block (one block per 4 rays meant 2 048 atomics per address)
    __shared__ float cam_red[4][12];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y, Ns = g.n_samples;
    float sum_R[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sum_T[3] = {0.f, 0.f, 0.f};
    const int ray_end = min(g.n_rays, (int)(blockIdx.x + 1) * cam_rays);
    for (int ray = blockIdx.x * cam_rays + wave; ray < ray_end; ray += 4) {
    const long rayg = (long)b * g.n_rays + ray;
    const float* Rb = R + b * 9;
    const float* Kb = Kinv + b * 9;
    const float* Tb = T + b * 3;
    const float x = xy[(int64_t)b * g.xy_stride_b + (int64_t)ray * g.xy_stride_r];
    const float y = xy[(int64_t)b * g.xy_stride_b + g.xy_stride_c + (int64_t)ray * g.xy_stride_r];
    float c[3], w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] = Kb[i * 3 + 0] * x + Kb[i * 3 + 1] * y + Kb[i * 3 + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = Rb[i * 3 + 0] * c[0] + Rb[i * 3 + 1] * c[1] + Rb[i * 3 + 2] * c[2];
    const float n = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const float dh[3] = {w[0] / n, w[1] / n, w[2] / n};
    const float l = -1.0f / dh[2];
    const float dl[3] = {dh[0] * l, dh[1] * l, dh[2] * l};
    const float rz1 = Tb[2] - g.world_z1, rz2 = Tb[2] - g.world_z2;
    const float* tr = t_rand ? t_rand + ((int64_t)b * g.n_rays + ray) * (Ns + 1) : nullptr;
    float g_dl[3] = {0.f, 0.f, 0.f}, g_T[3] = {0.f, 0.f, 0.f}, g_l = 0.0f, g_tz = 0.0f;
    for (int s = lane; s < Ns; s += 64) {
        const size_t pt = (size_t)rayg * Ns + s;
        const float* pe = cat5 + pt * 448;
        const float* dq = dpe + pt * 64;
        float dp[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float acc = dq[d];
            float f = 1.0f;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float sn = pe[3 + 6 * k + d], cs = pe[3 + 6 * k + 3 + d];
                acc += f * (cs * dq[3 + 6 * k + d] - sn * dq[3 + 6 * k + 3 + d]);
                f *= 2.0f;
            }
            dp[d] = acc;
        }
        // hierarchical pass: the planes are given.  They are affine combinations (weights summing to 1, placed by the DETACHED
        // coarse weights, NetWorks/utils.py:219-252) of the coarse planes o_z + const, so d plane / d T_z = 1 as in the coarse pass
        const float z_lo = g.z_planes_given ? tr[s] : n3dt_edge_z(rz1, rz2, s, Ns, tr);
        const float z_hi = g.z_planes_given ? tr[s + 1] : n3dt_edge_z(rz1, rz2, s + 1, Ns, tr);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            g_dl[d] += dp[d] * z_lo;
            g_T[d] += dp[d];
        }
        g_tz += dp[0] * dl[0] + dp[1] * dl[1] + dp[2] * dl[2];
        g_l += dxr[pt * XR_LD + 385] * (z_hi - z_lo);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            g_dl[d] += __shfl_xor(g_dl[d], off, 64);
            g_T[d] += __shfl_xor(g_T[d], off, 64);
        }
        g_l += __shfl_xor(g_l, off, 64);
        g_tz += __shfl_xor(g_tz, off, 64);
    }
    {
        const float gl = g_l + g_dl[0] * dh[0] + g_dl[1] * dh[1] + g_dl[2] * dh[2];
        float g_dh[3] = {g_dl[0] * l, g_dl[1] * l, g_dl[2] * l + gl * l * l};
        const float dot = g_dh[0] * dh[0] + g_dh[1] * dh[1] + g_dh[2] * dh[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float gw = (g_dh[i] - dh[i] * dot) / n;
#pragma unroll
            for (int j = 0; j < 3; ++j) sum_R[i * 3 + j] += gw * c[j];
        }
        sum_T[0] += g_T[0];
        sum_T[1] += g_T[1];
        sum_T[2] += g_T[2] + g_tz;
    }
    }  // rays of this wave
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) cam_red[wave][i] = sum_R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) cam_red[wave][9 + i] = sum_T[i];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float v = (cam_red[0][threadIdx.x] + cam_red[1][threadIdx.x]) + (cam_red[2][threadIdx.x] + cam_red[3][threadIdx.x]);
        if (threadIdx.x < 9) {
            if (d_R) atomicAdd(&d_R[b * 9 + threadIdx.x], v);
        } else if (d_T) {
            atomicAdd(&d_T[b * 3 + threadIdx.x - 9], v);
        }
    }
}

// out[f][n] += sum over the rows of frame f of X[m][n]   (bias gradients; per frame for the folded biases).
// 256 threads = ceil(N/4) column groups x row lanes, 8 independent float4 loads in flight per thread, an LDS reduction over
// the row lanes, then N atomics per workgroup; grid (row chunks, frames) with about 384 chunks in total -- every workgroup's
// atomics hit the same N addresses, and thousands of those serialise.
__global__ __launch_bounds__(256) void train_colsum_kernel(const float* __restrict__ X, long ldx, int rows_per_frame, int N,
                                                           float* __restrict__ out, long ldo, int chunk) {
    __shared__ f32x4 red[256];
    const int cg = (N + 3) / 4, lanes = 256 / cg;
    const int t = threadIdx.x, cq = t % cg, rl = t / cg;
    const int f = blockIdx.y;
    const int r0 = blockIdx.x * chunk, r1 = min(rows_per_frame, r0 + chunk);
    const float* base = X + ((size_t)f * rows_per_frame) * ldx + 4 * cq;
    const bool vec = (4 * cq + 4 <= N) && ((ldx & 3) == 0) && ((((size_t)base) & 15) == 0);
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rl < lanes) {
        int r = r0 + rl;
        if (vec) {
            for (; r + 7 * lanes < r1; r += 8 * lanes) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (size_t)(r + u * lanes) * ldx);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u & 3] += v[u];
            }
            for (; r < r1; r += lanes) acc[0] += *reinterpret_cast<const f32x4*>(base + (size_t)r * ldx);
        } else {
            for (; r < r1; r += lanes)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * cq + j < N) acc[0][j] += base[(size_t)r * ldx + j];
        }
    }
    red[t] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (t < cg) {
        f32x4 s = red[t];
        for (int l = 1; l < lanes; ++l) s += red[l * cg + t];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * t + j < N) atomicAdd(&out[(size_t)f * ldo + 4 * t + j], s[j]);
    }
}

static void launch_colsum(const float* X, long ldx, int rows_per_frame, int frames, int N, float* out, long ldo, hipStream_t s) {
    int chunk = (int)(((long)rows_per_frame * frames + 383) / 384);
    if (chunk < 256) chunk = 256;
    hipLaunchKernelGGL(train_colsum_kernel, dim3((rows_per_frame + chunk - 1) / chunk, frames), dim3(256), 0, s, X, ldx, rows_per_frame, N,
                       out, ldo, chunk);
}

// latent folding backward.  dfold[f] holds d b0'[384] (offset 0), d b5' (offset 5*384), d brgb1' (offset of stage 10).
//   d_shape[f] = W0[:,63:63+S]^T db0' + W5[:,63:63+S]^T db5' ; d_audio[f] = W0[:,63+S:]^T db0' ; d_appea[f] = Wr1[:,384:]^T dbr1'
//   dW0[:,63:] += sum_f db0'[f] (x) [shape_f, audio_f] ; dW5[:,63:63+S] += sum_f db5'[f] (x) shape_f ; dWr1[:,384:] += ...
//   db0 += sum_f db0'[f] (same for b5, br1)
// grid (3, B, FOLDB_ROWGROUPS), block 256: a workgroup takes a slice of the layer's output rows (as one workgroup per (table,
// frame) the kernel was six workgroups walking 384 rows each, one atomic per row and thread: 60 us of latency at two heads);
// d_shape / d_appea / d_audio are accumulated with atomics over the row groups: the launcher zeroes them first
#define FOLDB_ROWGROUPS 16
__global__ void train_fold_bwd_kernel(N3dtMlpParams p, N3dtMlpGrads gp, int S, int A, int U, int B, const float* __restrict__ shape,
                                      const float* __restrict__ appea, const float* __restrict__ audio,
                                      const float* __restrict__ dfold, float* __restrict__ d_shape, float* __restrict__ d_appea,
                                      float* __restrict__ d_audio, const int frozen) {
    // frozen != 0: the network's parameters take no gradient (single-image fitting): only the code gradients are formed
    const int which = blockIdx.x, f = blockIdx.y, t = threadIdx.x;
    const int in0 = N3DT_PE_DIM + S + U, in5 = N3DT_PE_DIM + S + N3DT_HID, inr = N3DT_HID + A;
    const int layer = which == 0 ? 0 : (which == 1 ? 5 : 10);
    const int nout = which == 2 ? 192 : 384;
    const int ld = which == 0 ? in0 : (which == 1 ? in5 : inr);
    const int col0 = which == 2 ? N3DT_HID : N3DT_PE_DIM;
    const int ncode = which == 0 ? S + U : (which == 1 ? S : A);
    const float* db = dfold + (size_t)f * N3DT_FOLD_STRIDE + n3dt_bias_offset(which == 0 ? 0 : (which == 1 ? 5 : 10));
    __shared__ float sdb[384];
    for (int i = t; i < nout; i += blockDim.x) sdb[i] = db[i];
    __syncthreads();
    const int per = (nout + FOLDB_ROWGROUPS - 1) / FOLDB_ROWGROUPS;
    const int o0 = blockIdx.z * per, o1 = min(nout, o0 + per);
    // bias gradient (each frame adds its share)
    if (!frozen)
        for (int o = o0 + t; o < o1; o += blockDim.x) atomicAdd(&gp.bias[layer][o], sdb[o]);
    // code gradients and latent weight columns
    for (int i = t; i < ncode; i += blockDim.x) {
        float code;
        if (which == 2) code = appea[(size_t)f * A + i];
        else code = i < S ? shape[(size_t)f * S + i] : audio[(size_t)f * U + (i - S)];
        float acc = 0.0f;
#pragma unroll 4
        for (int o = o0; o < o1; ++o) {
            acc = fmaf(p.weight[layer][(size_t)o * ld + col0 + i], sdb[o], acc);
            if (!frozen) atomicAdd(&gp.weight[layer][(size_t)o * ld + col0 + i], sdb[o] * code);
        }
        if (which == 2) { if (d_appea) atomicAdd(&d_appea[(size_t)f * A + i], acc); }
        else if (i < S) { if (d_shape) atomicAdd(&d_shape[(size_t)f * S + i], acc); }
        else if (d_audio) atomicAdd(&d_audio[(size_t)f * U + (i - S)], acc);
    }
    (void)B;
}
// Zero up to three caller-owned buffers with as few memsets as their addresses allow: the host mirror hands out d_shape,
// d_appea, d_audio as consecutive slices of one allocation, which makes this ONE launch (each memset is a ~5 us launch in
// a 6 ms training step that had eleven of them).
static void zero_runs(float* const* ptr, const size_t* count, int n, hipStream_t s) {
    int i = 0;
    while (i < n) {
        if (!ptr[i] || count[i] == 0) { ++i; continue; }
        float* lo = ptr[i];
        size_t len = count[i];
        int j = i + 1;
        while (j < n && ptr[j] && count[j] > 0 && ptr[j] == lo + len) len += count[j++];
        (void)hipMemsetAsync(lo, 0, sizeof(float) * len, s);
        i = j;
    }
}

static void launch_fold_bwd(const N3dtMlpParams* p, const N3dtMlpGrads* gp, int S, int A, int U, int B, const float* shape, const float* appea,
                            const float* audio, const float* dfold, float* d_shape, float* d_appea, float* d_audio, hipStream_t s,
                            bool codes_zeroed = false) {
    // (d_shape is zeroed by the callers at the top of their backward: two tables add into it)
    if (!codes_zeroed) {
        if (d_appea) (void)hipMemsetAsync(d_appea, 0, sizeof(float) * (size_t)B * A, s);
        if (d_audio && U > 0) (void)hipMemsetAsync(d_audio, 0, sizeof(float) * (size_t)B * U, s);
    }
    static const N3dtMlpGrads no_grads = {};
    hipLaunchKernelGGL(train_fold_bwd_kernel, dim3(3, B, FOLDB_ROWGROUPS), dim3(256), 0, s, *p, gp ? *gp : no_grads, S, A, U, B, shape, appea, audio, dfold,
                       d_shape, d_appea, d_audio, gp ? 0 : 1);
}

// scatter the packed gradients back: dW5[:,0:63] += dW5'[:,0:63]; dW5[:,63+S:] += dW5'[:,64:]; dWr0 += dWc[0:384]; dwd += dWc[384]
__global__ void train_unpack_grads_kernel(N3dtMlpGrads gp, int S, const float* __restrict__ dw5p, const float* __restrict__ dwc,
                                          const float* __restrict__ dbc) {
    const int in5 = N3DT_PE_DIM + S + N3DT_HID;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 384 * 448) {
        const int r = i / 448, c = i % 448;
        if (c < 63) gp.weight[5][(size_t)r * in5 + c] += dw5p[i];
        else if (c >= 64) gp.weight[5][(size_t)r * in5 + 63 + S + (c - 64)] += dw5p[i];
    }
    if (i < 385 * 384) {
        if (i < 384 * 384) gp.weight[9][i] += dwc[i];
        else gp.weight[8][i - 384 * 384] += dwc[i];
    }
    if (i < 385) {
        if (i < 384) gp.bias[9][i] += dbc[i];
        else gp.bias[8][0] += dbc[i];
    }
}

// ---------------------------------------------------------------------------------------------
// orchestration
// ---------------------------------------------------------------------------------------------
extern "C" void n3dt_launch_fold(const N3dtGeom*, const N3dtMlpParams*, const float*, const float*, const float*, float*, int, hipStream_t);
extern "C" void n3dt_launch_rayfold(const N3dtGeom*, const float*, const float*, hipStream_t);
extern "C" void n3dt_launch_ray_head(const N3dtGeom*, int, int, const float*, const float*, const float*, const float*, int, float*,
                                     float*, float*, float*, float*, hipStream_t);

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

static int split_for(long K);
// parameter-gradient products always ADD into their destination: atomics when K is split, += otherwise
static void set_grad_split(Gemm32& q, long K) {
    q.split_k = split_for(K);
    q.accumulate = q.split_k <= 1 ? 1 : 0;
}
static int split_for(long K) {
    long s = K / 2048;
    if (s < 1) s = 1;
    if (s > 128) s = 128;
    return (int)s;
}

// `tail` = fp32 W2^T[192][256] | b2[256] from any packed buffer (n3dt_mlp_pack)
extern "C" void n3dt_launch_train_fwd(const N3dtGeom* g, const N3dtMlpParams* p, const float* tail, const float* xy, const float* R,
                                      const float* T, const float* Kinv, const float* shape, const float* appea, const float* audio,
                                      const float* t_rand, const float* bg_featmap, float* fg_feat, float* bg_alpha, float* depth,
                                      float* merge_feat, float* saved, float* ws, const float* ray_bias, hipStream_t s) {
    const TrainSaved sv = saved_layout(g);
    const TrainWs wl = ws_layout(g);
    const int P = g->batch * g->n_rays * g->n_samples, ppf = g->n_rays * g->n_samples;
    const int S = g->shape_dim, A = g->appea_dim, U = g->audio_dim;
    float* fold = saved + sv.fold;
    float* cat5 = saved + sv.cat5;
    n3dt_launch_fold(g, p, shape, appea, audio, fold, 0, s);
    if (ray_bias) n3dt_launch_rayfold(g, fold, ray_bias, s);  // include_vd: RGB_layer_1's bias per ray
    hipLaunchKernelGGL(train_pack_kernel, dim3((385 * 384 + 384 * 448 + 255) / 256), dim3(256), 0, s, *p, S, ws + wl.w5p, ws + wl.wc,
                       ws + wl.bc);
    hipLaunchKernelGGL(train_sample_pe_kernel, dim3((unsigned)(((size_t)P * 4 + 255) / 256)), dim3(256), 0, s, *g, xy, R, T, Kinv, t_rand,
                       cat5, saved + sv.geo);
    auto hptr = [&](int l) { return l == 4 ? cat5 + 64 : saved + sv.h[l]; };
    auto hld = [&](int l) { return l == 4 ? 448L : 384L; };
    // L0 (models.py:69-71): PE columns lead FeaExt_module_0's input, latents folded into the bias
    {
        Gemm32 q = mk(P, 384, 63, cat5, 448, 0, p->weight[0], 63 + S + U, 0, hptr(0), hld(0));
        q.bias = fold + n3dt_bias_offset(0); q.bias_group_rows = ppf; q.bias_ld = N3DT_FOLD_STRIDE; q.act = G32_ACT_RELU;
        n3dt_gemm32(q, s);
    }
    for (int l = 1; l < 8; ++l) {
        Gemm32 q = l == 5 ? mk(P, 384, 448, cat5, 448, 0, ws + wl.w5p, 448, 0, hptr(5), hld(5))
                          : mk(P, 384, 384, hptr(l - 1), hld(l - 1), 0, p->weight[l], 384, 0, hptr(l), hld(l));
        q.bias = fold + n3dt_bias_offset(l); q.bias_group_rows = ppf; q.bias_ld = N3DT_FOLD_STRIDE; q.act = G32_ACT_RELU;
        n3dt_gemm32(q, s);
    }
    {   // RGB_layer_0 | density_module (models.py:78-79)
        Gemm32 q = mk(P, 385, 384, hptr(7), 384, 0, ws + wl.wc, 384, 0, saved + sv.xr, XR_LD);
        q.bias = ws + wl.bc;
        n3dt_gemm32(q, s);
    }
    {   // RGB_layer_1 (+ folded appearance), relu (models.py:80-81)
        Gemm32 q = mk(P, 192, 384, saved + sv.xr, XR_LD, 0, p->weight[10], 384 + A, 0, saved + sv.g, 192);
        q.bias = fold + n3dt_bias_offset(10); q.bias_group_rows = ppf; q.bias_ld = N3DT_FOLD_STRIDE; q.act = G32_ACT_RELU;
        if (ray_bias) {  // include_vd: one bias row per ray (the samples of a ray are consecutive rows)
            q.bias = fold + n3dt_rayfold_offset(g->batch); q.bias_group_rows = g->n_samples; q.bias_ld = N3DT_RAYFOLD_STRIDE;
        }
        n3dt_gemm32(q, s);
    }
    const long Rr = (long)g->batch * g->n_rays;
    hipLaunchKernelGGL(train_composite_fwd_kernel, dim3((unsigned)((Rr + 3) / 4)), dim3(256), 0, s, *g, saved + sv.xr, saved + sv.g,
                       saved + sv.geo, saved + sv.w, saved + sv.ray);
    n3dt_launch_ray_head(g, 1, g->n_samples, saved + sv.ray, nullptr, tail, bg_featmap, 0, fg_feat, bg_alpha, depth, nullptr, merge_feat, s);
}

extern "C" void n3dt_launch_train_bwd(const N3dtGeom* g, const N3dtMlpParams* p, const N3dtMlpGrads* gp, const float* shape,
                                      const float* appea, const float* audio, const float* bg_featmap, const float* d_merge,
                                      const float* d_fg, const float* d_ba, const float* saved, float* d_bg_featmap, float* d_shape,
                                      float* d_appea, float* d_audio, const float* xy, const float* Rm, const float* Tv,
                                      const float* Kinv, const float* t_rand, float* d_R, float* d_T, float* ws, float* d_ray_bias,
                                      hipStream_t s) {
    const TrainSaved sv = saved_layout(g);
    const TrainWs wl = ws_layout(g);
    const int P = g->batch * g->n_rays * g->n_samples, ppf = g->n_rays * g->n_samples;
    const int S = g->shape_dim, A = g->appea_dim, U = g->audio_dim, B = g->batch;
    const long Rr = (long)B * g->n_rays;
    const float* cat5 = saved + sv.cat5;
    auto hptr = [&](int l) { return l == 4 ? cat5 + 64 : saved + sv.h[l]; };
    auto hld = [&](int l) { return l == 4 ? 448L : 384L; };
    float* dha = ws + wl.dha;
    float* dhb = ws + wl.dhb;
    float* dxr = ws + wl.dxr;
    float* dG = ws + wl.dg;
    float* dfold = ws + wl.dfold;
    // dha doubles as the [R][256] d_fg_total scratch before the MLP backward starts
    float* dfg_total = dha;
    (void)hipMemsetAsync(dfold, 0, sizeof(float) * (size_t)B * N3DT_FOLD_STRIDE, s);
    // the workspace is scratch: rebuild the packed W5' / [Wr0; wd] here rather than trusting the forward's copy
    hipLaunchKernelGGL(train_pack_kernel, dim3((385 * 384 + 384 * 448 + 255) / 256), dim3(256), 0, s, *p, S, ws + wl.w5p, ws + wl.wc,
                       ws + wl.bc);
    (void)hipMemsetAsync(ws + wl.dw5p, 0, sizeof(float) * 384 * 448, s);
    (void)hipMemsetAsync(ws + wl.dwc, 0, sizeof(float) * 385 * 384, s);
    (void)hipMemsetAsync(ws + wl.bc, 0, sizeof(float) * 385, s);  // re-used as the [d br0 | d bd] accumulator (Wc's bias is not needed in backward)
    if (d_shape) (void)hipMemsetAsync(d_shape, 0, sizeof(float) * (size_t)B * S, s);
    // ---- head: RGB_layer_2 once per ray + merge (models.py:82, HeadNeRFNet.py:103-112)
    hipLaunchKernelGGL(train_head_bwd_kernel, dim3((unsigned)((Rr + HB_RAYS - 1) / HB_RAYS)), dim3(256), 0, s, *g, p->weight[11], p->bias[11], bg_featmap, d_merge,
                       d_fg, d_ba, dfg_total, ws + wl.dgray, ws + wl.dwsum, (long)Rr);
    if (gp || d_bg_featmap)
        hipLaunchKernelGGL(train_bg_b2_grad_kernel, dim3((g->n_rays + BG_RAYS - 1) / BG_RAYS), dim3(256), 0, s, *g, saved + sv.ray, d_merge, dfg_total,
                           d_bg_featmap, gp ? gp->bias[11] : nullptr);
    if (gp) {   // dW2[256][192] += dfg_total^T Gray
        Gemm32 q = mk(256, 192, (int)Rr, dfg_total, 256, 1, saved + sv.ray, N3DT_PART_STRIDE, 1, gp->weight[11], 192);
        set_grad_split(q, Rr);
        n3dt_gemm32(q, s);
    }
    // ---- compositing
    hipLaunchKernelGGL(train_composite_bwd_kernel, dim3((unsigned)((Rr + 3) / 4)), dim3(256), 0, s, *g, saved + sv.xr, saved + sv.g,
                       saved + sv.geo, saved + sv.w, ws + wl.dgray, ws + wl.dwsum, dG, dxr);
    // ---- RGB_layer_1
    {
        Gemm32 q = mk(P, 384, 192, dG, 192, 0, p->weight[10], 384 + A, 1, dxr, XR_LD);  // dX = dG Wr1[:, 0:384]
        n3dt_gemm32(q, s);
        if (gp) {  // (gp == nullptr: frozen network, only input gradients are wanted)
            Gemm32 w = mk(192, 384, P, dG, 192, 1, saved + sv.xr, XR_LD, 1, gp->weight[10], 384 + A);  // dWr1[:, 0:384] += dG^T X
            set_grad_split(w, P);
            n3dt_gemm32(w, s);
        }
        launch_colsum(dG, 192L, ppf, B, 192, dfold + n3dt_bias_offset(10), (long)N3DT_FOLD_STRIDE, s);
        if (d_ray_bias) {  // include_vd: dL/d(ray_bias) = the same column sums, per ray
            (void)hipMemsetAsync(d_ray_bias, 0, sizeof(float) * (size_t)Rr * N3DT_RAYFOLD_STRIDE, s);
            launch_colsum(dG, 192L, g->n_samples, (int)Rr, 192, d_ray_bias, (long)N3DT_RAYFOLD_STRIDE, s);
        }
    }
    // ---- RGB_layer_0 | density: dH7 = dXR Wc, gated by relu(H7)
    {
        Gemm32 q = mk(P, 384, 385, dxr, XR_LD, 0, ws + wl.wc, 384, 1, dha, 384);
        q.gate = hptr(7); q.ldgate = hld(7); q.gate_act = G32_ACT_RELU;
        n3dt_gemm32(q, s);
        if (gp) {
            Gemm32 w = mk(385, 384, P, dxr, XR_LD, 1, hptr(7), hld(7), 1, ws + wl.dwc, 384);
            set_grad_split(w, P);
            n3dt_gemm32(w, s);
        }
        // bias grads of RGB_layer_0 (cols 0..383) and density (col 384): one frame group of all rows
        if (gp) launch_colsum(dxr, (long)XR_LD, P, 1, 385, ws + wl.bc, 0L, s);  // [d br0 (384) | d bd] into the packed scratch, copied out below
    }
    // ---- trunk, layers 7..0.  `dcur` = dL/dH_l (already gated by relu'(H_l))
    float* dcur = dha;
    float* dnext = dhb;
    for (int l = 7; l >= 0; --l) {
        // parameter gradients of layer l (none for a frozen network)
        if (!gp) {
        } else if (l == 5) {
            Gemm32 w = mk(384, 448, P, dcur, 384, 1, cat5, 448, 1, ws + wl.dw5p, 448);
            set_grad_split(w, P);
            n3dt_gemm32(w, s);
        } else if (l == 0) {
            Gemm32 w = mk(384, 63, P, dcur, 384, 1, cat5, 448, 1, gp->weight[0], 63 + S + U);
            set_grad_split(w, P);
            n3dt_gemm32(w, s);
        } else {
            Gemm32 w = mk(384, 384, P, dcur, 384, 1, hptr(l - 1), hld(l - 1), 1, gp->weight[l], 384);
            set_grad_split(w, P);
            n3dt_gemm32(w, s);
        }
        if (l == 0 || l == 5) {
            launch_colsum(dcur, 384L, ppf, B, 384, dfold + n3dt_bias_offset(l), (long)N3DT_FOLD_STRIDE, s);
        } else if (gp) {
            launch_colsum(dcur, 384L, P, 1, 384, gp->bias[l], 0L, s);
        }
        const bool want_cam = d_R || d_T;
        // DIAGNOSTIC (N3DT_DIAG_PE_BF16, bit 0: round dZ, bit 1: round the weights): the two d-PE products of THIS exact path with
        // operands rounded to bf16 -- what the fused bf16 path's last product does to the camera gradients, in isolation from
        // the rounding its dZ has already collected upstream (tools/cam_error_probe.py)
        static const int diag_pe = [] {
            const char* e = getenv("N3DT_DIAG_PE_BF16");
            return e ? atoi(e) : 0;
        }();
        const float* dz_pe = dcur;
        const float* w5_pe = ws + wl.w5p;
        const float* w0_pe = p->weight[0];
        long w5_ld = 448, w0_ld = 63 + S + U;
        if (want_cam && diag_pe && (l == 5 || l == 0)) {
            static float* wtmp = nullptr;  // diagnostic only: one lazily allocated scratch for the rounded weight columns
            if (!wtmp) (void)hipMalloc(&wtmp, sizeof(float) * 384 * 64);
            if (diag_pe & 1) {
                hipLaunchKernelGGL(train_round_bf16_kernel, dim3(1024), dim3(256), 0, s, (size_t)P * 384, 384, 384, 384, dcur, dnext);
                dz_pe = dnext;  // (dnext is written by this layer's input-gradient GEMM only after the products below)
            }
            if (diag_pe & 2) {
                if (l == 5) { hipLaunchKernelGGL(train_round_bf16_kernel, dim3(64), dim3(256), 0, s, (size_t)384 * 64, 64, 448, 64, ws + wl.w5p, wtmp); w5_pe = wtmp; w5_ld = 64; }
                else { hipLaunchKernelGGL(train_round_bf16_kernel, dim3(64), dim3(256), 0, s, (size_t)384 * 63, 63, 63 + S + U, 64, p->weight[0], wtmp); w0_pe = wtmp; w0_ld = 64; }
            }
        }
        if (want_cam && l == 5) {  // d PE from the skip layer: dH5 W5'[:, 0:64]
            Gemm32 q = mk(P, 64, 384, dz_pe, 384, 0, w5_pe, w5_ld, 1, ws + wl.dpe, 64);
            n3dt_gemm32(q, s);
        }
        if (want_cam && l == 0) {  // += dH0 W0[:, 0:63]
            Gemm32 q = mk(P, 63, 384, dz_pe, 384, 0, w0_pe, w0_ld, 1, ws + wl.dpe, 64);
            q.accumulate = 1;
            n3dt_gemm32(q, s);
            if (d_R) (void)hipMemsetAsync(d_R, 0, sizeof(float) * 9 * B, s);
            if (d_T) (void)hipMemsetAsync(d_T, 0, sizeof(float) * 3 * B, s);
            const int cr = n3dt_cam_rays_per_block(g->n_rays, B);
            hipLaunchKernelGGL(train_camera_bwd_kernel, dim3((g->n_rays + cr - 1) / cr, B), dim3(256), 0, s, *g, xy, Rm, Tv, Kinv, t_rand, cat5,
                               ws + wl.dpe, dxr, d_R, d_T, cr);
        }
        if (l == 0) break;
        // input gradient: dH_{l-1} = (dH_l W_l) * relu'(H_{l-1})
        Gemm32 q = l == 5 ? mk(P, 384, 384, dcur, 384, 0, ws + wl.w5p + 64, 448, 1, dnext, 384)
                          : mk(P, 384, 384, dcur, 384, 0, p->weight[l], 384, 1, dnext, 384);
        q.gate = hptr(l - 1); q.ldgate = hld(l - 1); q.gate_act = G32_ACT_RELU;
        n3dt_gemm32(q, s);
        float* t = dcur; dcur = dnext; dnext = t;
    }
    if (gp)
        hipLaunchKernelGGL(train_unpack_grads_kernel, dim3((384 * 448 + 255) / 256), dim3(256), 0, s, *gp, S, ws + wl.dw5p, ws + wl.dwc,
                           ws + wl.bc);
    launch_fold_bwd(p, gp, S, A, U, B, shape, appea, audio, dfold, d_shape, d_appea, d_audio, s);
}

#include "train_x16.inc"
