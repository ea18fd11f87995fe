// extern "C" entry points of libn3dt.so (declared in include/n3dt.h).
// Host-side only: validates geometry, carves the caller's workspace and enqueues kernels on the
// caller's stream.  No allocation, no synchronisation; the only mutable global state is the opt-in
// measurement hook (n3dt_prof_*, mutex-guarded).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/n3dt.h"
#include "n3dt_layout.h"

extern "C" {
void n3dt_launch_pack(const N3dtGeom*, int, const N3dtMlpParams*, void*, hipStream_t);
void n3dt_launch_fold(const N3dtGeom*, const N3dtMlpParams*, const float*, const float*, const float*, float*, int, hipStream_t);
void n3dt_launch_ray_head(const N3dtGeom*, int, int, const float*, const float*, const float*, const float*, int, float*, float*,
                          float*, float*, float*, hipStream_t);
void n3dt_launch_ray_head_mfma(const N3dtGeom*, int, int, const float*, const float*, const float*, const float*, int, float*, float*,
                               float*, float*, float*, float*, hipStream_t);
void n3dt_launch_chw_to_hwc(int, int, const float*, float*, hipStream_t);
void n3dt_launch_rayfold(const N3dtGeom*, const float*, const float*, hipStream_t);
void n3dt_launch_ray_vd_bias(const N3dtGeom*, const float*, long, const float*, const float*, const float*, float*, hipStream_t);
void n3dt_launch_embed_freqs(int, size_t, int, const float*, float*, hipStream_t);
void n3dt_launch_sample_points(const N3dtGeom*, const float*, const float*, const float*, const float*, const float*, float*, float*,
                               float*, float*, float*, hipStream_t);
void n3dt_launch_embed(int, size_t, const float*, float*, hipStream_t);
size_t n3dt_seam_mlp_ws_floats(const N3dtGeom*, size_t);
void n3dt_launch_mlp_points(const N3dtGeom*, size_t, const N3dtMlpParams*, const float*, const float*, const float*, float*, float*, float*,
                            hipStream_t);
void n3dt_launch_composite(int, int, int, int, const float*, const float*, const float*, const float*, float*, float*, float*, float*,
                           hipStream_t);
void n3dt_launch_fine_sample(const N3dtGeom*, int, const float*, const float*, const float*, const float*, float*, hipStream_t);
void n3dt_launch_nerf_fwd_f32(const N3dtGeom*, const N3dtMlpParams*, const void*, const float*, const float*, const float*,
                              const float*, const float*, const float*, float*, float*, hipStream_t);
void n3dt_launch_nerf_fwd_x16(const N3dtGeom*, int, const void*, const float*, const float*, const float*, const float*,
                              const float*, const float*, float*, float*, hipStream_t);
void n3dt_launch_nerf_fwd_x16s(const N3dtGeom*, const void*, const float*, const float*, const float*, const float*, const float*,
                               const float*, float*, float*, hipStream_t);
void n3dt_launch_nerf_fwd_x16b(const N3dtGeom*, int, const void*, const float*, const float*, const float*, const float*,
                               const float*, const float*, float*, float*, hipStream_t);
size_t n3dt_nr_workspace_floats(const N3dtGeom*, int);
size_t n3dt_train_saved_floats(const N3dtGeom*);
size_t n3dt_train_ws_floats(const N3dtGeom*);
void n3dt_launch_train_fwd(const N3dtGeom*, const N3dtMlpParams*, const float*, const float*, const float*, const float*, const float*,
                           const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*,
                           float*, const float* /*ray_bias*/, hipStream_t);
void n3dt_launch_train_bwd(const N3dtGeom*, const N3dtMlpParams*, const N3dtMlpGrads*, const float*, const float*, const float*,
                           const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*,
                           const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float* /*d_ray_bias*/,
                           hipStream_t);
size_t n3dt_train16_saved_bytes(const N3dtGeom*);
size_t n3dt_train16_ws_bytes(const N3dtGeom*);
void n3dt_launch_train16_fwd(const N3dtGeom*, const N3dtMlpParams*, const void*, const float*, const float*, const float*, const float*,
                             const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*,
                             float*, void*, void*, const float* /*ray_bias*/, hipStream_t);
void n3dt_launch_train16_bwd(const N3dtGeom*, const N3dtMlpParams*, const N3dtMlpGrads*, const float*, const float*, const float*,
                             const float*, const float*, const float*, const float*, const void*, float*, float*, float*, float*,
                             const float*, const float*, const float*, const float*, const float*, float*, float*, void*, float* /*d_ray_bias*/,
                             hipStream_t);
void n3dt_launch_img_to_uint8(int, int, const float*, unsigned char*, hipStream_t);
void n3dt_launch_loss_fwd(int, int, const float*, const float*, const float*, const float*, float, float*, float*, hipStream_t);
void n3dt_launch_loss_bwd(int, int, const float*, const float*, const float*, const float*, float, const float*, const float*, const float*, float*,
                          float*, hipStream_t);
size_t n3dt_nr_train_saved_floats(const N3dtGeom*, int);
size_t n3dt_nr_train_ws_floats(const N3dtGeom*, int);
void n3dt_launch_nr_train_fwd(const N3dtGeom*, int, const N3dtRenderParams*, const float*, float*, float*, float*, int, hipStream_t);
void n3dt_launch_nr_bwd(const N3dtGeom*, int, const N3dtRenderParams*, const N3dtRenderGrads*, const float*, const float*, const float*,
                        float*, float*, int, hipStream_t);
void n3dt_launch_neural_render(const N3dtGeom*, int, int, const N3dtRenderParams*, const float*, float*, float*, int, hipStream_t);
void n3dt_launch_stage(const N3dtStageCopy*, hipStream_t);
}

static thread_local char g_err[256] = "";

static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

static int check_hip(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
        return N3DT_EHIP;
    }
    return N3DT_OK;
}

static int check_geom(const N3dtGeom* g, int precision) {
    if (!g) return fail(N3DT_EINVAL, "geometry is NULL");
    if (precision != N3DT_F32 && precision != N3DT_BF16 && precision != N3DT_F16 && precision != N3DT_BF16X3)
        return fail(N3DT_EINVAL, "unknown precision");
    if (g->batch < 1 || g->n_rays < 1 || g->n_samples < 1) return fail(N3DT_EINVAL, "batch, n_rays and n_samples must be >= 1");
    if (g->n_samples > 1024) return fail(N3DT_EINVAL, "n_samples > 1024 is not supported");
    if (g->hidden != 384) return fail(N3DT_EINVAL, "only mlp_hidden_nchannels == 384 is built");
    if (g->feat_nc != 256) return fail(N3DT_EINVAL, "only featmap_nc == 256 is built");
    if (g->shape_dim < 1 || g->appea_dim < 1 || g->audio_dim < 0) return fail(N3DT_EINVAL, "bad latent widths");
    if (g->shape_dim + g->audio_dim > 512 || g->appea_dim > 512) return fail(N3DT_EINVAL, "latent width > 512");
    if (g->vd_dim != 0 && g->vd_dim != 27) return fail(N3DT_EINVAL, "vd_dim must be 0 or 27 (include_vd: 3 + 6 * 4 channels)");
    return N3DT_OK;
}
// include_vd: the per-ray bias must come with the flag, and only with it
static int check_ray_bias(const N3dtGeom* g, const void* ray_bias, const char* who) {
    if ((g->vd_dim > 0) != (ray_bias != nullptr)) {
        snprintf(g_err, sizeof(g_err), "%s: the per-ray bias must be given iff vd_dim > 0", who);
        return N3DT_EINVAL;
    }
    return N3DT_OK;
}

// Tiling of the 16-bit fused render kernel: 1 = 32x32x16 MFMA, 8 waves x 32 samples; 2 = the same, 4 waves x 64 samples;
// 3 = 16x16x32 MFMA, 8 waves x 32 samples (nerf_fwd_x16b.hip).  N3DT_X16_TILING overrides the default at run time.
#ifndef N3DT_X16_DEFAULT_TILING
#define N3DT_X16_DEFAULT_TILING 1
#endif
static int x16_tiling() {
    static const int t = [] {
        const char* e = getenv("N3DT_X16_TILING");
        return e ? atoi(e) : N3DT_X16_DEFAULT_TILING;
    }();
    return t;
}

static inline int block_samples(int precision) { return precision == N3DT_F32 ? 16 : 32; }
static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct RenderCarve {
    size_t fold, part, wlocal, bghwc, total;  // byte offsets / total bytes
    int bpr, bs;
};

static RenderCarve render_carve(const N3dtGeom* g, int precision) {
    RenderCarve c;
    c.bs = block_samples(precision);
    c.bpr = (g->n_samples + c.bs - 1) / c.bs;
    const size_t blocks = (size_t)g->batch * g->n_rays * c.bpr;
    c.fold = 0;
    c.part = align256(n3dt_fold_region_floats(g->batch, g->n_rays, g->vd_dim) * sizeof(float));  // (+ the per-ray table of include_vd)
    c.wlocal = c.part + align256(blocks * (192 + 4) * sizeof(float));
    c.bghwc = c.wlocal + align256(blocks * c.bs * sizeof(float));  // the background map transposed to [N_r][C] for the merge
    c.total = c.bghwc + align256((size_t)g->n_rays * g->feat_nc * sizeof(float));
    return c;
}

// ---- opt-in measurement hook (see n3dt.h) ---------------------------------------------------
static std::mutex g_prof_mu;
static hipEvent_t* g_prof_ev = nullptr;  // 2 * g_prof_cap events
static int g_prof_cap = 0, g_prof_n = 0;

static void prof_free_locked(int created) {
    for (int i = 0; i < created; ++i) (void)hipEventDestroy(g_prof_ev[i]);
    delete[] g_prof_ev;
    g_prof_ev = nullptr;
    g_prof_cap = g_prof_n = 0;
}

extern "C" int n3dt_prof_enable(int max_records) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    prof_free_locked(2 * g_prof_cap);
    if (max_records <= 0) return N3DT_OK;
    g_prof_ev = new hipEvent_t[2 * (size_t)max_records];
    for (int i = 0; i < 2 * max_records; ++i)
        if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) {
            prof_free_locked(i);  // the events created so far do not leak
            return fail(N3DT_EHIP, "n3dt_prof_enable: hipEventCreate failed");
        }
    g_prof_cap = max_records;
    return N3DT_OK;
}

// A measured span: begin records an event on `s` and returns its slot (or -1: hook off / record list full), end records the
// closing event.  Used by n3dt_render_fwd (the fused MLP kernel) and by the fused training path (forward kernel, dX chain,
// weight-gradient stage: three spans per step, in that order) -- internal, not part of the ABI.
extern "C" int n3dt_prof_span_begin(hipStream_t s) {
    if (g_prof_cap <= 0) return -1;  // the unlocked read is the hook's documented contract: enabled from one thread, while idle
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof_n >= g_prof_cap) return -1;
    const int slot = g_prof_n++;
    (void)hipEventRecord(g_prof_ev[2 * slot], s);
    return slot;
}
extern "C" void n3dt_prof_span_end(int slot, hipStream_t s) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (slot < g_prof_cap) (void)hipEventRecord(g_prof_ev[2 * slot + 1], s);
}

extern "C" int n3dt_prof_collect(float* ms_out, int capacity, int* n_out) {
    if (!ms_out || !n_out) return fail(N3DT_EINVAL, "n3dt_prof_collect: NULL argument");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    int n = g_prof_n < capacity ? g_prof_n : capacity;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) return fail(N3DT_EHIP, "n3dt_prof_collect: sync failed");
        if (hipEventElapsedTime(&ms_out[i], g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess)
            return fail(N3DT_EHIP, "n3dt_prof_collect: elapsed failed");
    }
    *n_out = n;
    g_prof_n = 0;
    return N3DT_OK;
}

extern "C" int n3dt_abi_version(void) { return N3DT_ABI_VERSION; }
extern "C" const char* n3dt_last_error(void) { return g_err; }

extern "C" size_t n3dt_mlp_packed_bytes(const N3dtGeom* g, int precision) {
    if (check_geom(g, precision) != N3DT_OK) return 0;
    return n3dt_packed_tail_offset(precision) + n3dt_packed_tail_floats() * sizeof(float);
}

extern "C" int n3dt_mlp_pack(const N3dtGeom* g, int precision, const N3dtMlpParams* p, void* packed, void* stream) {
    int rc = check_geom(g, precision);
    if (rc) return rc;
    if (!p || !packed) return fail(N3DT_EINVAL, "n3dt_mlp_pack: NULL argument");
    for (int l = 0; l < N3DT_MLP_LAYERS; ++l)
        if (!p->weight[l] || !p->bias[l]) return fail(N3DT_EINVAL, "n3dt_mlp_pack: NULL parameter pointer");
    n3dt_launch_pack(g, precision, p, packed, (hipStream_t)stream);
    return check_hip("n3dt_mlp_pack");
}

extern "C" size_t n3dt_render_workspace_bytes(const N3dtGeom* g, int precision) {
    if (check_geom(g, precision) != N3DT_OK) return 0;
    return render_carve(g, precision).total;
}

extern "C" int n3dt_render_fwd(const N3dtGeom* g, int precision, const void* packed_mlp, const N3dtMlpParams* p, const float* xy,
                               const float* R, const float* T, const float* Kinv, const float* shape, const float* appea,
                               const float* audio, const float* t_rand, const float* bg_featmap, const float* ray_bias, float* fg_feat,
                               float* bg_alpha, float* depth, float* weight, float* merge_feat, void* workspace,
                               size_t workspace_bytes, void* stream) {
    int rc = check_geom(g, precision);
    if (rc) return rc;
    if ((rc = check_ray_bias(g, ray_bias, "n3dt_render_fwd")) != N3DT_OK) return rc;
    if (!packed_mlp || !p || !xy || !R || !T || !Kinv || !shape || !appea || !workspace)
        return fail(N3DT_EINVAL, "n3dt_render_fwd: NULL argument");
    if (!fg_feat && !merge_feat) return fail(N3DT_EINVAL, "n3dt_render_fwd: neither fg_feat nor merge_feat requested");
    if (g->audio_dim > 0 && !audio) return fail(N3DT_EINVAL, "n3dt_render_fwd: audio is NULL but audio_dim > 0");
    if (merge_feat && !bg_featmap) return fail(N3DT_EINVAL, "n3dt_render_fwd: merge_feat needs bg_featmap");
    if (g->z_planes_given && !t_rand) return fail(N3DT_EINVAL, "n3dt_render_fwd: z_planes_given but no planes passed as t_rand");
    const RenderCarve c = render_carve(g, precision);
    if (workspace_bytes < c.total) return fail(N3DT_EWORKSPACE, "n3dt_render_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    float* fold = (float*)(ws + c.fold);
    float* part = (float*)(ws + c.part);
    float* wlocal = (float*)(ws + c.wlocal);
    n3dt_launch_fold(g, p, shape, appea, audio, fold, precision != N3DT_F32, s);
    if (ray_bias) n3dt_launch_rayfold(g, fold, ray_bias, s);
    const int span = n3dt_prof_span_begin(s);
    if (precision == N3DT_F32)
        n3dt_launch_nerf_fwd_f32(g, p, packed_mlp, fold, xy, R, T, Kinv, t_rand, part, weight ? wlocal : nullptr, s);
    else if (precision == N3DT_BF16X3)
        n3dt_launch_nerf_fwd_x16s(g, packed_mlp, fold, xy, R, T, Kinv, t_rand, part, weight ? wlocal : nullptr, s);
    else if (x16_tiling() == 3 && g->vd_dim == 0)
        n3dt_launch_nerf_fwd_x16b(g, precision, (const unsigned char*)packed_mlp + n3dt_packed_region_b_offset(precision), fold, xy, R, T,
                                  Kinv, t_rand, part, weight ? wlocal : nullptr, s);
    else
        n3dt_launch_nerf_fwd_x16(g, precision, packed_mlp, fold, xy, R, T, Kinv, t_rand, part, weight ? wlocal : nullptr, s);
    n3dt_prof_span_end(span, s);
    const float* tail = (const float*)((const unsigned char*)packed_mlp + n3dt_packed_tail_offset(precision));
    const float* bghwc = bg_featmap;
    if (merge_feat && !g->bg_is_hwc) {  // [C][N_r] parameter -> [N_r][C]
        n3dt_launch_chw_to_hwc(g->feat_nc, g->n_rays, bg_featmap, (float*)(ws + c.bghwc), s);
        bghwc = (const float*)(ws + c.bghwc);
    }
    if (precision == N3DT_F32)
        n3dt_launch_ray_head(g, c.bpr, c.bs, part, wlocal, tail, bghwc, 1, fg_feat, bg_alpha, depth, weight, merge_feat, s);
    else
        n3dt_launch_ray_head_mfma(g, c.bpr, c.bs, part, wlocal, tail, bghwc, 1, fg_feat, bg_alpha, depth, weight, merge_feat, nullptr, s);
    return check_hip("n3dt_render_fwd");
}

// ---- stand-alone seams (csrc/seams.hip) ------------------------------------------------------------
extern "C" int n3dt_sample_points(const N3dtGeom* g, const float* xy, const float* R, const float* T, const float* Kinv,
                                  const float* t_rand, float* pts, float* zvals, float* z_dists, float* ray_d, float* ray_l, void* stream) {
    int rc = check_geom(g, N3DT_F32);
    if (rc) return rc;
    if (!xy || !R || !T || !Kinv) return fail(N3DT_EINVAL, "n3dt_sample_points: NULL argument");
    if (g->z_planes_given && !t_rand) return fail(N3DT_EINVAL, "n3dt_sample_points: z_planes_given but no planes passed as t_rand");
    n3dt_launch_sample_points(g, xy, R, T, Kinv, t_rand, pts, zvals, z_dists, ray_d, ray_l, (hipStream_t)stream);
    return check_hip("n3dt_sample_points");
}

extern "C" int n3dt_embed(int batch, size_t m, const float* pts, float* pe, void* stream) {
    if (batch < 1 || m < 1 || !pts || !pe) return fail(N3DT_EINVAL, "n3dt_embed: bad argument");
    n3dt_launch_embed(batch, m, pts, pe, (hipStream_t)stream);
    return check_hip("n3dt_embed");
}

extern "C" int n3dt_embed_freqs(int batch, size_t m, int n_freqs, const float* pts, float* pe, void* stream) {
    if (batch < 1 || m < 1 || n_freqs < 1 || n_freqs > 16 || !pts || !pe) return fail(N3DT_EINVAL, "n3dt_embed_freqs: bad argument");
    n3dt_launch_embed_freqs(batch, m, n_freqs, pts, pe, (hipStream_t)stream);
    return check_hip("n3dt_embed_freqs");
}

extern "C" int n3dt_ray_vd_bias(const N3dtGeom* g, const float* w_vd, int64_t ld_w, const float* xy, const float* R, const float* Kinv,
                                float* ray_bias, void* stream) {
    int rc = check_geom(g, N3DT_F32);
    if (rc) return rc;
    if (g->vd_dim != 27) return fail(N3DT_EINVAL, "n3dt_ray_vd_bias: vd_dim must be 27");
    if (!w_vd || ld_w < 27 || !xy || !R || !Kinv || !ray_bias) return fail(N3DT_EINVAL, "n3dt_ray_vd_bias: bad argument");
    n3dt_launch_ray_vd_bias(g, w_vd, (long)ld_w, xy, R, Kinv, ray_bias, (hipStream_t)stream);
    return check_hip("n3dt_ray_vd_bias");
}

extern "C" size_t n3dt_mlp_points_workspace_bytes(const N3dtGeom* g, size_t m) {
    if (check_geom(g, N3DT_F32) != N3DT_OK || m < 1) return 0;
    return n3dt_seam_mlp_ws_floats(g, m) * sizeof(float);
}

extern "C" int n3dt_mlp_points(const N3dtGeom* g, size_t m, const N3dtMlpParams* p, const float* audio, const float* embed_vps,
                               const float* embed_vds, float* rgb, float* density, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_geom(g, N3DT_F32);
    if (rc) return rc;
    if (m < 1 || !p || !embed_vps || !embed_vds || !rgb || !density || !workspace) return fail(N3DT_EINVAL, "n3dt_mlp_points: NULL argument");
    if (g->audio_dim > 0 && !audio) return fail(N3DT_EINVAL, "n3dt_mlp_points: audio is NULL but audio_dim > 0");
    if ((size_t)g->batch * m > (size_t)1 << 30) return fail(N3DT_EINVAL, "n3dt_mlp_points: more than 2^30 points");
    if (workspace_bytes < n3dt_mlp_points_workspace_bytes(g, m)) return fail(N3DT_EWORKSPACE, "n3dt_mlp_points: workspace too small");
    for (int l = 0; l < N3DT_MLP_LAYERS; ++l)
        if (!p->weight[l] || !p->bias[l]) return fail(N3DT_EINVAL, "n3dt_mlp_points: NULL parameter pointer");
    n3dt_launch_mlp_points(g, m, p, audio, embed_vps, embed_vds, rgb, density, (float*)workspace, (hipStream_t)stream);
    return check_hip("n3dt_mlp_points");
}

extern "C" int n3dt_composite(int batch, int n_rays, int n_samples, int channels, const float* rgb, const float* density,
                              const float* z_dists, const float* zvals, float* feat, float* bg_alpha, float* depth, float* weight,
                              void* stream) {
    if (batch < 1 || n_rays < 1 || n_samples < 1 || channels < 1 || !rgb || !density || !z_dists || !zvals || !feat)
        return fail(N3DT_EINVAL, "n3dt_composite: bad argument");
    if (n_samples > 4096) return fail(N3DT_EINVAL, "n3dt_composite: more than 4096 samples per ray");
    n3dt_launch_composite(batch, n_rays, n_samples, channels, rgb, density, z_dists, zvals, feat, bg_alpha, depth, weight,
                          (hipStream_t)stream);
    return check_hip("n3dt_composite");
}

extern "C" int n3dt_fine_sample(const N3dtGeom* g, int n_fine, const float* weight, const float* T, const float* t_rand, const float* u,
                                float* z_planes, void* stream) {
    int rc = check_geom(g, N3DT_F32);
    if (rc) return rc;
    if (!weight || !T || !z_planes) return fail(N3DT_EINVAL, "n3dt_fine_sample: NULL argument");
    if (g->z_planes_given && !t_rand) return fail(N3DT_EINVAL, "n3dt_fine_sample: z_planes_given but no coarse planes passed as t_rand");
    if (g->n_samples < 3) return fail(N3DT_EINVAL, "n3dt_fine_sample: needs at least 3 coarse samples");
    if (n_fine < 0 || g->n_samples + n_fine + 1 > 2048) return fail(N3DT_EINVAL, "n3dt_fine_sample: more than 2048 planes per ray");
    n3dt_launch_fine_sample(g, n_fine, weight, T, t_rand, u, z_planes, (hipStream_t)stream);
    return check_hip("n3dt_fine_sample");
}

extern "C" size_t n3dt_neural_render_workspace_bytes(const N3dtGeom* g, int nb) {
    if (!g || nb < 1 || g->n_blocks < 1 || g->n_blocks > N3DT_MAX_BLOCKS || g->featmap_size < 2) return 0;
    return n3dt_nr_workspace_floats(g, nb) * sizeof(float);
}

static int neural_render_common(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap, float* img,
                                void* workspace, size_t workspace_bytes, int pack_mode, void* stream, const char* who) {
    if (!g || !p || !workspace || (pack_mode != 2 && (!featmap || !img))) return fail(N3DT_EINVAL, "neural render: NULL argument");
    // the split-precision mode renders the 2-D stage on its fp16 path: 2.3e-4 on RGB where bf16 maps reach 1.6e-3 (fixture
    // `contrast`, features O(10)), inside the mode's 1e-3 budget without splitting the renderer's products as well
    if (precision == N3DT_BF16X3) precision = N3DT_F16;
    if (precision != N3DT_F32 && precision != N3DT_BF16 && precision != N3DT_F16) return fail(N3DT_EINVAL, "unknown precision");
    if (nb < 1) return fail(N3DT_EINVAL, "neural render: nb < 1");
    if (g->n_blocks < 1 || g->n_blocks > N3DT_MAX_BLOCKS) return fail(N3DT_EINVAL, "n_blocks must be in 1..8");
    if (g->feat_nc != 256) return fail(N3DT_EINVAL, "only featmap_nc == 256 is built");
    if (g->featmap_size < 2) return fail(N3DT_EINVAL, "featmap_size < 2 (reflect border needs 2 pixels)");
    {
        // the kernels index one map level with 32-bit in-plane offsets: pixels x channels of the largest level below 2^31
        const size_t side = (size_t)g->featmap_size << g->n_blocks, elems = (size_t)nb * side * side * 32;
        if (elems >= ((size_t)1 << 31)) return fail(N3DT_EINVAL, "neural render: nb x output pixels x 32 channels must stay below 2^31 (split the batch)");
    }
    if (workspace_bytes < n3dt_neural_render_workspace_bytes(g, nb)) return fail(N3DT_EWORKSPACE, "neural render workspace too small");
    for (int i = 0; i <= g->n_blocks; ++i)
        if (!p->to_rgb_w[i] || !p->to_rgb_b[i]) return fail(N3DT_EINVAL, "NULL feat_2_rgb parameter");
    for (int i = 0; i < g->n_blocks; ++i)
        if (!p->psu1_w[i] || !p->psu1_b[i] || !p->psu2_w[i] || !p->psu2_b[i] || !p->feat_w[i] || !p->feat_b[i])
            return fail(N3DT_EINVAL, "NULL neural-render block parameter");
    n3dt_launch_neural_render(g, nb, precision, p, featmap, img, (float*)workspace, pack_mode, (hipStream_t)stream);
    return check_hip(who);
}

extern "C" int n3dt_neural_render_fwd(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap,
                                      float* img, void* workspace, size_t workspace_bytes, void* stream) {
    return neural_render_common(g, nb, precision, p, featmap, img, workspace, workspace_bytes, 0, stream, "n3dt_neural_render_fwd");
}

extern "C" int n3dt_neural_render_pack(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return neural_render_common(g, nb, precision, p, nullptr, nullptr, workspace, workspace_bytes, 2, stream, "n3dt_neural_render_pack");
}

extern "C" int n3dt_neural_render_fwd_reuse(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap,
                                            float* img, void* workspace, size_t workspace_bytes, void* stream) {
    return neural_render_common(g, nb, precision, p, featmap, img, workspace, workspace_bytes, 1, stream, "n3dt_neural_render_fwd_reuse");
}

extern "C" int n3dt_chw_to_hwc(int C, int n, const float* src, float* dst, void* stream) {
    if (C < 1 || n < 1 || !src || !dst) return fail(N3DT_EINVAL, "n3dt_chw_to_hwc: bad argument");
    n3dt_launch_chw_to_hwc(C, n, src, dst, (hipStream_t)stream);
    return check_hip("n3dt_chw_to_hwc");
}

// ---- training path -----------------------------------------------------------------------------
static int check_train_geom(const N3dtGeom* g) {
    int rc = check_geom(g, N3DT_F32);
    if (rc) return rc;
    if ((size_t)g->batch * g->n_rays * g->n_samples > (size_t)1 << 30) return fail(N3DT_EINVAL, "training path: more than 2^30 sample points");
    return N3DT_OK;
}

// one size serves both training precisions (the fp32 layered path and the fused bf16 path lay the buffers out differently)
static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }
extern "C" size_t n3dt_render_train_saved_bytes(const N3dtGeom* g) {
    if (check_train_geom(g) != N3DT_OK) return 0;
    return max_sz(n3dt_train_saved_floats(g) * sizeof(float), n3dt_train16_saved_bytes(g));
}
extern "C" size_t n3dt_render_train_workspace_bytes(const N3dtGeom* g) {
    if (check_train_geom(g) != N3DT_OK) return 0;
    return max_sz(n3dt_train_ws_floats(g) * sizeof(float), n3dt_train16_ws_bytes(g));
}

extern "C" int n3dt_render_train_fwd(const N3dtGeom* g, int precision, const void* packed_mlp, const N3dtMlpParams* p, const float* xy, const float* R,
                                     const float* T, const float* Kinv, const float* shape, const float* appea, const float* audio,
                                     const float* t_rand, const float* bg_featmap, const float* ray_bias, float* fg_feat, float* bg_alpha,
                                     float* depth, float* merge_feat, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int rc = check_train_geom(g);
    if (rc) return rc;
    if ((rc = check_ray_bias(g, ray_bias, "n3dt_render_train_fwd")) != N3DT_OK) return rc;
    if (precision != N3DT_F32 && precision != N3DT_BF16) return fail(N3DT_EINVAL, "training precision must be N3DT_F32 or N3DT_BF16");
    if (!packed_mlp || !p || !xy || !R || !T || !Kinv || !shape || !appea || !fg_feat || !saved || !workspace)
        return fail(N3DT_EINVAL, "n3dt_render_train_fwd: NULL argument");
    if (g->audio_dim > 0 && !audio) return fail(N3DT_EINVAL, "n3dt_render_train_fwd: audio is NULL but audio_dim > 0");
    if (merge_feat && !bg_featmap) return fail(N3DT_EINVAL, "n3dt_render_train_fwd: merge_feat needs bg_featmap");
    if (saved_bytes < n3dt_render_train_saved_bytes(g)) return fail(N3DT_EWORKSPACE, "n3dt_render_train_fwd: saved buffer too small");
    if (workspace_bytes < n3dt_render_train_workspace_bytes(g)) return fail(N3DT_EWORKSPACE, "n3dt_render_train_fwd: workspace too small");
    // `packed_mlp` is the n3dt_mlp_pack buffer of the SAME precision: the fused bf16 path streams its bf16 matrices,
    // the fp32 path only reads the fp32 tail (W2^T, b2) behind them
    const float* tail = (const float*)((const unsigned char*)packed_mlp + n3dt_packed_tail_offset(precision));
    if (precision == N3DT_BF16) {
        n3dt_launch_train16_fwd(g, p, packed_mlp, tail, xy, R, T, Kinv, shape, appea, audio, t_rand, bg_featmap, fg_feat, bg_alpha, depth,
                                merge_feat, saved, workspace, ray_bias, (hipStream_t)stream);
        return check_hip("n3dt_render_train_fwd");
    }
    n3dt_launch_train_fwd(g, p, tail, xy, R, T, Kinv, shape, appea, audio, t_rand, bg_featmap, fg_feat, bg_alpha, depth, merge_feat,
                          (float*)saved, (float*)workspace, ray_bias, (hipStream_t)stream);
    return check_hip("n3dt_render_train_fwd");
}

extern "C" int n3dt_render_bwd(const N3dtGeom* g, int precision, const N3dtMlpParams* p, const N3dtMlpGrads* grads, const float* shape,
                               const float* appea, const float* audio, const float* bg_featmap, const float* d_merge_feat,
                               const float* d_fg_feat, const float* d_bg_alpha, const void* saved, size_t saved_bytes,
                               float* d_bg_featmap, float* d_shape, float* d_appea, float* d_audio, float* d_ray_bias, const float* xy,
                               const float* R, const float* T, const float* Kinv, const float* t_rand, float* d_R, float* d_T, void* workspace,
                               size_t workspace_bytes, void* stream) {
    int rc = check_train_geom(g);
    if (rc) return rc;
    if ((rc = check_ray_bias(g, d_ray_bias, "n3dt_render_bwd")) != N3DT_OK) return rc;
    if (precision != N3DT_F32 && precision != N3DT_BF16) return fail(N3DT_EINVAL, "training precision must be N3DT_F32 or N3DT_BF16");
    if (!p || !shape || !appea || !saved || !workspace) return fail(N3DT_EINVAL, "n3dt_render_bwd: NULL argument");
    if (!d_merge_feat && !d_fg_feat && !d_bg_alpha) return fail(N3DT_EINVAL, "n3dt_render_bwd: no incoming gradient");
    if (d_merge_feat && !bg_featmap) return fail(N3DT_EINVAL, "n3dt_render_bwd: d_merge_feat needs bg_featmap");
    if (g->audio_dim > 0 && !audio) return fail(N3DT_EINVAL, "n3dt_render_bwd: audio is NULL but audio_dim > 0");
    // grads == NULL: the network is frozen (single-image fitting) -- no parameter gradient is computed, only d codes / d cameras
    if (grads)
        for (int l = 0; l < N3DT_MLP_LAYERS; ++l)
            if (!grads->weight[l] || !grads->bias[l]) return fail(N3DT_EINVAL, "n3dt_render_bwd: NULL gradient pointer");
    if (!grads && d_bg_featmap) return fail(N3DT_EINVAL, "n3dt_render_bwd: grads == NULL (frozen network) but d_bg_featmap given");
    if (saved_bytes < n3dt_render_train_saved_bytes(g)) return fail(N3DT_EWORKSPACE, "n3dt_render_bwd: saved buffer too small");
    if (workspace_bytes < n3dt_render_train_workspace_bytes(g)) return fail(N3DT_EWORKSPACE, "n3dt_render_bwd: workspace too small");
    if ((d_R || d_T) && (!xy || !R || !T || !Kinv)) return fail(N3DT_EINVAL, "n3dt_render_bwd: camera gradients need xy, R, T, Kinv");
    if (precision == N3DT_BF16) {
        n3dt_launch_train16_bwd(g, p, grads, shape, appea, audio, bg_featmap, d_merge_feat, d_fg_feat, d_bg_alpha, saved, d_bg_featmap,
                                d_shape, d_appea, d_audio, xy, R, T, Kinv, t_rand, d_R, d_T, workspace, d_ray_bias, (hipStream_t)stream);
        return check_hip("n3dt_render_bwd");
    }
    n3dt_launch_train_bwd(g, p, grads, shape, appea, audio, bg_featmap, d_merge_feat, d_fg_feat, d_bg_alpha, (const float*)saved,
                          d_bg_featmap, d_shape, d_appea, d_audio, xy, R, T, Kinv, t_rand, d_R, d_T, (float*)workspace, d_ray_bias,
                          (hipStream_t)stream);
    return check_hip("n3dt_render_bwd");
}

static int check_nr(const N3dtGeom* g, int nb) {
    if (!g) return fail(N3DT_EINVAL, "geometry is NULL");
    if (nb < 1) return fail(N3DT_EINVAL, "nb < 1");
    if (g->n_blocks < 1 || g->n_blocks > N3DT_MAX_BLOCKS) return fail(N3DT_EINVAL, "n_blocks must be in 1..8");
    if (g->feat_nc != 256) return fail(N3DT_EINVAL, "only featmap_nc == 256 is built");
    if (g->featmap_size < 2) return fail(N3DT_EINVAL, "featmap_size < 2 (reflect border needs 2 pixels)");
    return N3DT_OK;
}

extern "C" size_t n3dt_neural_render_train_saved_bytes(const N3dtGeom* g, int nb) {
    if (check_nr(g, nb) != N3DT_OK) return 0;
    return n3dt_nr_train_saved_floats(g, nb) * sizeof(float);
}
extern "C" size_t n3dt_neural_render_train_workspace_bytes(const N3dtGeom* g, int nb) {
    if (check_nr(g, nb) != N3DT_OK) return 0;
    return n3dt_nr_train_ws_floats(g, nb) * sizeof(float);
}

extern "C" int n3dt_neural_render_train_fwd(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap, float* img,
                                            void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_nr(g, nb);
    if (rc) return rc;
    if (!p || !featmap || !img || !saved || !workspace) return fail(N3DT_EINVAL, "n3dt_neural_render_train_fwd: NULL argument");
    if (saved_bytes < n3dt_neural_render_train_saved_bytes(g, nb)) return fail(N3DT_EWORKSPACE, "neural render saved buffer too small");
    if (workspace_bytes < n3dt_neural_render_train_workspace_bytes(g, nb)) return fail(N3DT_EWORKSPACE, "neural render workspace too small");
    n3dt_launch_nr_train_fwd(g, nb, p, featmap, img, (float*)saved, (float*)workspace, precision == N3DT_BF16, (hipStream_t)stream);
    return check_hip("n3dt_neural_render_train_fwd");
}

extern "C" int n3dt_neural_render_bwd(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const N3dtRenderGrads* grads,
                                      const float* featmap, const float* d_img, const void* saved, size_t saved_bytes, float* d_featmap,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_nr(g, nb);
    if (rc) return rc;
    if (!p || !featmap || !d_img || !saved || !d_featmap || !workspace)  // (grads == NULL: frozen renderer, only d_featmap is wanted)
        return fail(N3DT_EINVAL, "n3dt_neural_render_bwd: NULL argument");
    if (saved_bytes < n3dt_neural_render_train_saved_bytes(g, nb)) return fail(N3DT_EWORKSPACE, "neural render saved buffer too small");
    if (workspace_bytes < n3dt_neural_render_train_workspace_bytes(g, nb)) return fail(N3DT_EWORKSPACE, "neural render workspace too small");
    n3dt_launch_nr_bwd(g, nb, p, grads, featmap, d_img, (const float*)saved, d_featmap, (float*)workspace, precision == N3DT_BF16, (hipStream_t)stream);
    return check_hip("n3dt_neural_render_bwd");
}

extern "C" int n3dt_img_to_uint8(int n_images, int pixels, const float* img, unsigned char* out, void* stream) {
    if (n_images < 1 || pixels < 1 || !img || !out) return fail(N3DT_EINVAL, "n3dt_img_to_uint8: bad argument");
    n3dt_launch_img_to_uint8(n_images, pixels, img, out, (hipStream_t)stream);
    return check_hip("n3dt_img_to_uint8");
}

extern "C" int n3dt_loss_fwd(int batch, int pixels, const float* merge_img, const float* bg_img, const float* gt, const float* mask,
                             float bg_value, float* acc, float* terms, void* stream) {
    if (batch < 1 || pixels < 1 || !merge_img || !bg_img || !gt || !mask || !acc || !terms) return fail(N3DT_EINVAL, "n3dt_loss_fwd: bad argument");
    n3dt_launch_loss_fwd(batch, pixels, merge_img, bg_img, gt, mask, bg_value, acc, terms, (hipStream_t)stream);
    return check_hip("n3dt_loss_fwd");
}

extern "C" int n3dt_loss_bwd(int batch, int pixels, const float* merge_img, const float* bg_img, const float* gt, const float* mask,
                             float bg_value, const float* acc, const float* g, const float* g_total, float* d_merge, float* d_bg, void* stream) {
    if (batch < 1 || pixels < 1 || !merge_img || !bg_img || !gt || !mask || !acc || (!g && !g_total) || !d_merge || !d_bg)
        return fail(N3DT_EINVAL, "n3dt_loss_bwd: bad argument");
    n3dt_launch_loss_bwd(batch, pixels, merge_img, bg_img, gt, mask, bg_value, acc, g, g_total, d_merge, d_bg, (hipStream_t)stream);
    return check_hip("n3dt_loss_bwd");
}

// ---- input staging + hipGraph replay (see n3dt.h) -------------------------------------------------------------
extern "C" int n3dt_stage_inputs(const N3dtStageCopy* st, void* stream) {
    if (!st || st->n < 1 || st->n > N3DT_STAGE_MAX) return fail(N3DT_EINVAL, "n3dt_stage_inputs: bad entry count");
    for (int i = 0; i < st->n; ++i)
        if (!st->src[i] || !st->dst[i] || st->count[i] < 1) return fail(N3DT_EINVAL, "n3dt_stage_inputs: NULL pointer or empty entry");
    if (st->view_dims[0] > 0) {
        if (st->view_dims[1] < 1 || st->view_dims[2] < 1 || st->view_dims[0] * st->view_dims[1] * st->view_dims[2] != st->count[0])
            return fail(N3DT_EINVAL, "n3dt_stage_inputs: view_dims do not multiply to count[0]");
        for (int d = 0; d < 3; ++d)
            if (st->view_strides[d] < 0) return fail(N3DT_EINVAL, "n3dt_stage_inputs: negative stride");
    }
    n3dt_launch_stage(st, (hipStream_t)stream);
    return check_hip("n3dt_stage_inputs");
}

struct N3dtGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int n3dt_graph_begin(void* stream) {
    (void)hipGetLastError();
    if (hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        return fail(N3DT_EHIP, "n3dt_graph_begin: hipStreamBeginCapture failed (the legacy NULL stream cannot be captured)");
    }
    return N3DT_OK;
}

extern "C" int n3dt_graph_end(void* stream, void** graph_out) {
    if (!graph_out) return fail(N3DT_EINVAL, "n3dt_graph_end: NULL argument");
    *graph_out = nullptr;
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture((hipStream_t)stream, &graph) != hipSuccess || !graph) {
        (void)hipGetLastError();
        return fail(N3DT_EHIP, "n3dt_graph_end: hipStreamEndCapture failed (a call inside the capture was not capturable)");
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(graph);
        return fail(N3DT_EHIP, "n3dt_graph_end: hipGraphInstantiate failed");
    }
    *graph_out = new N3dtGraph{graph, exec};
    return N3DT_OK;
}

extern "C" int n3dt_graph_launch(void* graph, void* stream) {
    if (!graph) return fail(N3DT_EINVAL, "n3dt_graph_launch: NULL graph");
    if (hipGraphLaunch(((N3dtGraph*)graph)->exec, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(N3DT_EHIP, "n3dt_graph_launch: hipGraphLaunch failed");
    }
    return N3DT_OK;
}

extern "C" int n3dt_graph_destroy(void* graph) {
    if (!graph) return N3DT_OK;
    N3dtGraph* g = (N3dtGraph*)graph;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return N3DT_OK;
}
