// The reference's inner seams as stand-alone operators, in the reference's own tensor layouts:
//   GenSamplePoints.forward   NetWorks/utils.py:147-161 (+ :65-145)   -> n3dt_launch_sample_points
//   Embedder.forward          NetWorks/utils.py:43-51                 -> n3dt_launch_embed
//   MLPforNeRF.forward        NetWorks/models.py:62-87                -> n3dt_launch_mlp_points
//   CalcRayColor.forward      NetWorks/utils.py:291-309               -> n3dt_launch_composite
// The fused render (n3dt_render_fwd) never materialises the tensors that cross these seams -- [B,63,N_r,N_s] encodings,
// [B,256,N_r,N_s] per-sample features -- which is where its speed comes from.  These operators exist for callers that
// address the sub-modules directly (SURVEY 8b "inner seam kept addressable"): exact fp32, unfused, inference only.
#include "gemm32.h"
#include "n3dt_device.h"

// ---- GenSamplePoints ---------------------------------------------------------------------------
// pts [B,3,Nr,Ns], zvals / z_dists [B,1,Nr,Ns], ray_d [B,3,Nr], ray_l [B,1,Nr]   (any output may be NULL)
__global__ void seam_sample_kernel(N3dtGeom g, const float* __restrict__ xy, const float* __restrict__ R, const float* __restrict__ T,
                                   const float* __restrict__ Kinv, const float* __restrict__ t_rand, float* __restrict__ pts,
                                   float* __restrict__ zvals, float* __restrict__ z_dists, float* __restrict__ ray_d,
                                   float* __restrict__ ray_l) {
    const size_t total = (size_t)g.batch * g.n_rays * g.n_samples;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = (int)(i % g.n_samples);
    const size_t rg = i / g.n_samples;
    const int ray = (int)(rg % g.n_rays), b = (int)(rg / g.n_rays);
    float p[3], dist, zval;
    n3dt_sample_point(g, xy, R, T, Kinv, t_rand, b, ray, s, p, dist, zval);
    const size_t M = (size_t)g.n_rays * g.n_samples, m = (size_t)ray * g.n_samples + s;
    if (pts)
#pragma unroll
        for (int d = 0; d < 3; ++d) pts[((size_t)b * 3 + d) * M + m] = p[d];
    if (zvals) zvals[(size_t)b * M + m] = zval;
    if (z_dists) z_dists[(size_t)b * M + m] = dist;
    if (s == 0 && (ray_d || ray_l)) {
        const float x = xy[(int64_t)b * g.xy_stride_b + (int64_t)ray * g.xy_stride_r];
        const float y = xy[(int64_t)b * g.xy_stride_b + g.xy_stride_c + (int64_t)ray * g.xy_stride_r];
        float d[3], l;
        n3dt_ray_setup(R + b * 9, Kinv + b * 9, x, y, d, l);
        if (ray_d)
#pragma unroll
            for (int k = 0; k < 3; ++k) ray_d[((size_t)b * 3 + k) * g.n_rays + ray] = d[k];
        if (ray_l) ray_l[(size_t)b * g.n_rays + ray] = l;
    }
}

extern "C" void n3dt_launch_sample_points(const N3dtGeom* g, const float* xy, const float* R, const float* T, const float* Kinv,
                                          const float* t_rand, float* pts, float* zvals, float* z_dists, float* ray_d, float* ray_l,
                                          hipStream_t s) {
    const size_t total = (size_t)g->batch * g->n_rays * g->n_samples;
    hipLaunchKernelGGL(seam_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *g, xy, R, T, Kinv, t_rand, pts, zvals,
                       z_dists, ray_d, ray_l);
}

// ---- Embedder ------------------------------------------------------------------------------------
// pts [B,3,M] -> pe [B,63,M]
__global__ void seam_embed_kernel(int B, size_t M, const float* __restrict__ pts, float* __restrict__ pe) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * M) return;
    const size_t b = i / M, m = i % M;
    const float p[3] = {pts[(b * 3 + 0) * M + m], pts[(b * 3 + 1) * M + m], pts[(b * 3 + 2) * M + m]};
    for (int r = 0; r < N3DT_PE_DIM; ++r) pe[(b * N3DT_PE_DIM + r) * M + m] = n3dt_pe_row_accurate(p, r);
}

extern "C" void n3dt_launch_embed(int B, size_t M, const float* pts, float* pe, hipStream_t s) {
    hipLaunchKernelGGL(seam_embed_kernel, dim3((unsigned)(((size_t)B * M + 255) / 256)), dim3(256), 0, s, B, M, pts, pe);
}

// the same encoder with n_freqs frequencies: pts [B,3,M] -> pe [B, 3 + 6 n_freqs, M] (n_freqs = 4: the reference's vd_encoder,
// NetWorks/HeadNeRFNet.py:30-31,61)
__global__ void seam_embed_freqs_kernel(int B, size_t M, int n_freqs, const float* __restrict__ pts, float* __restrict__ pe) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * M) return;
    const size_t b = i / M, m = i % M;
    const int rows = 3 + 6 * n_freqs;
    const float p[3] = {pts[(b * 3 + 0) * M + m], pts[(b * 3 + 1) * M + m], pts[(b * 3 + 2) * M + m]};
    for (int r = 0; r < 3; ++r) pe[(b * rows + r) * M + m] = p[r];
    for (int k = 0; k < n_freqs; ++k)
        for (int d = 0; d < 3; ++d) {
            const float a = p[d] * (float)(1 << k);
            pe[(b * rows + 3 + 6 * k + d) * M + m] = sinf(a);
            pe[(b * rows + 3 + 6 * k + 3 + d) * M + m] = cosf(a);
        }
}
extern "C" void n3dt_launch_embed_freqs(int B, size_t M, int n_freqs, const float* pts, float* pe, hipStream_t s) {
    hipLaunchKernelGGL(seam_embed_freqs_kernel, dim3((unsigned)(((size_t)B * M + 255) / 256)), dim3(256), 0, s, B, M, n_freqs, pts, pe);
}

// ---- MLPforNeRF ----------------------------------------------------------------------------------
// channel-major [B][C][M] -> point-major rows dst[(b*M + m) * ld + col0 + c]
__global__ void seam_to_rows_kernel(int B, int C, size_t M, const float* __restrict__ src, float* __restrict__ dst, long ld, int col0) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32;
    const size_t m0 = (size_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const size_t m = m0 + tx;
        tile[i][tx] = (c < C && m < M) ? src[((size_t)b * C + c) * M + m] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const size_t m = m0 + i;
        const int c = c0 + tx;
        if (c < C && m < M) dst[((size_t)b * M + m) * ld + col0 + c] = tile[tx][i];
    }
}
// point-major rows src[(b*M + m) * ld + c] -> channel-major [B][C][M], optional ReLU
__global__ void seam_to_planes_kernel(int B, int C, size_t M, const float* __restrict__ src, long ld, float* __restrict__ dst, int relu) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32;
    const size_t m0 = (size_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const size_t m = m0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (c < C && m < M) ? src[((size_t)b * M + m) * ld + c] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const size_t m = m0 + tx;
        if (c < C && m < M) {
            const float v = tile[tx][i];
            dst[((size_t)b * C + c) * M + m] = relu ? fmaxf(v, 0.0f) : v;
        }
    }
}

struct SeamMlpWs {  // float offsets; every row count is P = B*M
    size_t in0, cat5, h[2], rin, g, out, total;
    long ld0, ld5, ldr;
};
static SeamMlpWs seam_mlp_layout(const N3dtGeom* g, size_t P) {
    SeamMlpWs w;
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const int vp = N3DT_PE_DIM + g->shape_dim;
    w.ld0 = (vp + g->audio_dim + 3) & ~3;   // [embed_vps | audiostyle]
    w.ld5 = (vp + 384 + 3) & ~3;            // [embed_vps | h4]
    w.ldr = (384 + g->appea_dim + 3) & ~3;  // [RGB_layer_0 out | embed_vds]
    size_t o = 0;
    w.in0 = o; o += al(P * w.ld0);
    w.cat5 = o; o += al(P * w.ld5);
    w.h[0] = o; o += al(P * 384);
    w.h[1] = o; o += al(P * 384);
    w.rin = o; o += al(P * w.ldr);
    w.g = o; o += al(P * 192);
    w.out = o; o += al(P * 256);
    w.total = o;
    return w;
}
extern "C" size_t n3dt_seam_mlp_ws_floats(const N3dtGeom* g, size_t M) { return seam_mlp_layout(g, (size_t)g->batch * M).total; }

static Gemm32 smk(long M, int N, int K, const float* A, long lda, const float* Bw, long ldb, const float* bias, int relu, float* C, long ldc) {
    Gemm32 q;
    q.M = (int)M; q.N = N; q.K = K;
    q.A = A; q.lda = lda; q.a_kmajor = 0;
    q.B = Bw; q.ldb = ldb; q.b_kmajor = 0;
    q.C = C; q.ldc = ldc;
    q.bias = bias; q.bias_group_rows = 0; q.bias_ld = 0;
    q.act = relu ? G32_ACT_RELU : G32_ACT_NONE;
    q.gate = nullptr; q.ldgate = 0; q.gate_act = G32_ACT_NONE;
    q.accumulate = 0; q.split_k = 1;
    q.a16 = q.b16 = q.c16 = q.gate16 = 0;
    return q;
}

// audio [B,U,M] (NULL iff U == 0), embed_vps [B,63+S,M], embed_vds [B,A,M] -> rgb [B,256,M], density [B,1,M]
extern "C" void n3dt_launch_mlp_points(const N3dtGeom* g, size_t M, const N3dtMlpParams* p, const float* audio, const float* vps,
                                       const float* vds, float* rgb, float* density, float* ws, hipStream_t s) {
    const int B = g->batch, S = g->shape_dim, A = g->appea_dim, U = g->audio_dim, vp = N3DT_PE_DIM + S;
    const size_t P = (size_t)B * M;
    const SeamMlpWs wl = seam_mlp_layout(g, P);
    auto rows = [&](int C, const float* src, float* dst, long ld, int col0) {
        hipLaunchKernelGGL(seam_to_rows_kernel, dim3((unsigned)((M + 31) / 32), (C + 31) / 32, B), dim3(256), 0, s, B, C, M, src, dst, ld, col0);
    };
    float* in0 = ws + wl.in0;
    float* cat5 = ws + wl.cat5;
    float* rin = ws + wl.rin;
    rows(vp, vps, in0, wl.ld0, 0);
    if (U > 0) rows(U, audio, in0, wl.ld0, vp);
    rows(vp, vps, cat5, wl.ld5, 0);
    rows(A, vds, rin, wl.ldr, 384);
    // x = relu(FeaExt_i(x)); after i == 4: x = cat([embed_vps, x])   (models.py:69-76)
    const float* x = in0;
    long ldx = wl.ld0;
    int kx = vp + U;
    for (int i = 0; i < 8; ++i) {
        float* y = i == 4 ? cat5 + vp : ws + wl.h[i & 1];
        const long ldy = i == 4 ? wl.ld5 : 384;
        n3dt_gemm32(smk((long)P, 384, kx, x, ldx, p->weight[i], kx, p->bias[i], 1, y, ldy), s);
        if (i == 4) { x = cat5; ldx = wl.ld5; kx = vp + 384; }
        else { x = y; ldx = 384; kx = 384; }
    }
    // density = relu(density_module(x)); x = RGB_layer_0(x); x = relu(RGB_layer_1(cat([x, embed_vds]))); rgb = RGB_layer_2(x)  (:78-84)
    n3dt_gemm32(smk((long)P, 1, 384, x, 384, p->weight[8], 384, p->bias[8], 1, density, 1), s);
    n3dt_gemm32(smk((long)P, 384, 384, x, 384, p->weight[9], 384, p->bias[9], 0, rin, wl.ldr), s);
    n3dt_gemm32(smk((long)P, 192, 384 + A, rin, wl.ldr, p->weight[10], 384 + A, p->bias[10], 1, ws + wl.g, 192), s);
    n3dt_gemm32(smk((long)P, 256, 192, ws + wl.g, 192, p->weight[11], 192, p->bias[11], 0, ws + wl.out, 256), s);
    hipLaunchKernelGGL(seam_to_planes_kernel, dim3((unsigned)((M + 31) / 32), 8, B), dim3(256), 0, s, B, 256, M, ws + wl.out, 256L, rgb, 0);
}

// ---- CalcRayColor ---------------------------------------------------------------------------------
// rgb [B,C,Nr,Ns], density / z_dists / zvals [B,1,Nr,Ns] -> feat [B,C,Nr], bg_alpha / depth [B,1,Nr], weight [B,1,Nr,Ns]
// one wave per ray, 64 samples per pass
__global__ void seam_composite_kernel(int B, int Nr, int Ns, int C, const float* __restrict__ rgb, const float* __restrict__ density,
                                      const float* __restrict__ z_dists, const float* __restrict__ zvals, float* __restrict__ feat,
                                      float* __restrict__ bg_alpha, float* __restrict__ depth, float* __restrict__ weight) {
    extern __shared__ float w_lds[];  // [4 waves][Ns]
    const long rg = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rg >= (long)B * Nr) return;
    const int b = (int)(rg / Nr), ray = (int)(rg % Nr);
    float* wl = w_lds + (size_t)(threadIdx.x >> 6) * Ns;
    const size_t base = ((size_t)b * Nr + ray) * Ns;
    float Trun = 1.0f, wsum = 0.0f, dsum = 0.0f;
    for (int s0 = 0; s0 < Ns; s0 += 64) {
        const int s = s0 + lane;
        float alpha = 0.0f, z = 0.0f;
        if (s < Ns) {
            alpha = 1.0f - expf(-density[base + s] * z_dists[base + s]);  // utils.py:275
            z = zvals[base + s];
        }
        const float x = 1.0f - alpha + 1e-10f;                             // :283
        float incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_up(incl, off, 64);
            if (lane >= off) incl *= o;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = alpha * Trun * excl;                               // :287
        if (s < Ns) {
            wl[s] = w;
            if (weight) weight[base + s] = w;
        }
        float a = w, d = w * z;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            d += __shfl_xor(d, off, 64);
        }
        wsum += a;
        dsum += d;
        Trun *= __shfl(incl, 63, 64);
    }
    if (lane == 0) {
        if (bg_alpha) bg_alpha[(size_t)b * Nr + ray] = 1.0f - wsum;       // :306
        if (depth) depth[(size_t)b * Nr + ray] = dsum;
    }
    // features: lane = channel (c, c + 64, ...), samples in order (the weights of the ray sit in LDS)
    for (int c = lane; c < C; c += 64) {
        const float* col = rgb + (((size_t)b * C + c) * Nr + ray) * Ns;
        float acc = 0.0f;
        for (int s = 0; s < Ns; ++s) acc = fmaf(wl[s], col[s], acc);
        feat[((size_t)b * C + c) * Nr + ray] = acc;
    }
}

extern "C" void n3dt_launch_composite(int B, int Nr, int Ns, int C, const float* rgb, const float* density, const float* z_dists,
                                      const float* zvals, float* feat, float* bg_alpha, float* depth, float* weight, hipStream_t s) {
    const long rays = (long)B * Nr;
    hipLaunchKernelGGL(seam_composite_kernel, dim3((unsigned)((rays + 3) / 4)), dim3(256), sizeof(float) * 4 * Ns, s, B, Nr, Ns, C, rgb,
                       density, z_dists, zvals, feat, bg_alpha, depth, weight);
}
