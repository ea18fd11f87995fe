/*
 * n3dt.h -- C ABI of libn3dt.so: the MI355X (gfx950) head-render hot path of NeRF-3DTalker.
 *
 * The reference has no FFI/operator seam for this path: the boundary is the Python module
 * HeadNeRFNet (reference: NetWorks/HeadNeRFNet.py:10-207).  These entry points are what a
 * binding for that module calls underneath; each one names the reference code it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch/ATen types.
 *   - every data pointer is a DEVICE pointer (hipMalloc'd or torch-owned); all memory is
 *     caller-owned, nothing is allocated or freed inside a call (hipGraph-capturable).
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls are
 *     stream-ordered, asynchronous and re-entrant (no global mutable state).
 *   - return 0 on success, a negative N3DT_E* code otherwise; never throws, never exits.
 *     n3dt_last_error() returns a thread-local message for the last failing call.
 *   - tensors are fp32 and contiguous unless a stride is given.  Feature maps cross this ABI
 *     ray-major ("NHWC"): [B, N_r, C]; images are planar [B, 3, P, P] like the reference.
 */
#ifndef N3DT_API_H_
#define N3DT_API_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N3DT_ABI_VERSION 5

/* arithmetic type of the MLP contraction */
#define N3DT_F32 0  /* v_mfma_f32_16x16x4_f32, exact fp32: the <=1e-3 RGB parity mode */
#define N3DT_BF16 1 /* v_mfma_f32_32x32x16_bf16, fp32 accumulate: the roofline mode  */
#define N3DT_F16 2  /* v_mfma_f32_32x32x16_f16,  fp32 accumulate                      */
/* bf16 MFMA with every operand split hi + lo (x = bf16(x) + bf16(x - bf16(x)), three products per product, ~16 mantissa bits):
 * the parity-grade mode on the matrix pipe -- holds the 1e-3 RGB gate on sharp networks where single bf16 / fp16 operands do
 * not, at a third of the bf16 rate.  Render calls only (n3dt_mlp_pack, n3dt_render_fwd,
 * n3dt_neural_render_fwd, where the 2-D renderer then runs its fp16 path); the training entry points take F32 or BF16. */
#define N3DT_BF16X3 3

#define N3DT_OK 0
#define N3DT_EINVAL (-1)    /* bad geometry / null pointer / unsupported size */
#define N3DT_EWORKSPACE (-2) /* workspace too small */
#define N3DT_EHIP (-3)      /* a HIP runtime call failed */

#define N3DT_PE_DIM 63      /* 3 + 6*10 (reference: NetWorks/HeadNeRFNet.py:27-28,50) */
#define N3DT_MLP_LAYERS 12  /* FeaExt_module_0..7, density_module, RGB_layer_0..2 */

/* Geometry of one call.  Mirrors what HeadNeRFNet reads from `opt` and from the input shapes
 * (reference: NetWorks/HeadNeRFNet.py:22-45,131; HeadNeRFOptions.py:5-34). */
typedef struct N3dtGeom {
    int32_t batch;        /* B frames */
    int32_t n_rays;       /* N_r rays per frame (featmap_size^2 for the reference reading) */
    int32_t n_samples;    /* N_s = opt.num_sample_coarse */
    int32_t hidden;       /* opt.mlp_hidden_nchannels; this build supports 384 */
    int32_t feat_nc;      /* opt.featmap_nc (C); this build supports 256 */
    int32_t shape_dim;    /* iden+expr(+gaze) code width, 179 (+eye_gaze_dim) */
    int32_t appea_dim;    /* text+illu code width, 127 */
    int32_t audio_dim;    /* 64, or 0 for the audio-less *_yuan variant */
    int32_t featmap_size; /* fs; n_rays == fs*fs whenever the neural renderer follows */
    int32_t n_blocks;     /* log2(pred_img_size / featmap_size) */
    float world_z1;       /* opt.world_z1 (2.5)  */
    float world_z2;       /* opt.world_z2 (-3.5) */
    /* element strides of batch_xy [B,2,N_r]; the trainer passes an expand()ed view
     * (reference: talker_trainer.py:768), so stride_b may be 0 */
    int64_t xy_stride_b, xy_stride_c, xy_stride_r;
    /* 0: the N_s+1 sample planes of a ray are the reference's linspace between world_z1 / world_z2 (jittered by
     * `t_rand` in train mode, NetWorks/utils.py:118-145);  1: the `t_rand` argument of the render calls carries the
     * planes themselves, [B,N_r,N_s+1] ascending camera-relative z (the hierarchical pass, utils.py:173-208) */
    int32_t z_planes_given;
    /* 0: `bg_featmap` arguments are the parameter as PyTorch holds it, [C][N_r] (neural_renderer.py:31-46) and the render call
     * transposes it; 1: the caller passes it already ray-major [N_r][C] (e.g. cached per parameter version): no transposition */
    int32_t bg_is_hwc;
    /* include_vd (ABI 5; reference: NetWorks/HeadNeRFNet.py:56-63,86,141-142).  0: RGB_layer_1's input is [RGB_layer_0 out |
     * appea_code], as every caller of the reference builds it.  27 (= 3 + 6 * vd_n_freqs): it is [RGB_layer_0 out | Embedder_4(ray
     * direction) | appea_code].  The direction is constant along a ray, so those 27 columns collapse into a PER-RAY bias of the
     * layer -- the render calls then take it as `ray_bias` [B, N_r, 192] (n3dt_ray_vd_bias computes it) and `weight[10]` is the
     * layer WITHOUT its 27 view-direction columns, [192, 384 + appea_dim] contiguous. */
    int32_t vd_dim;
} N3dtGeom;

/* The MLP's fp32 parameters as PyTorch owns them: weight[l] is [out_l, in_l] row-major
 * (Conv2d 1x1 weight viewed 2-D), bias[l] is [out_l]; order FeaExt_module_0..7, density_module,
 * RGB_layer_0, RGB_layer_1, RGB_layer_2 (reference: NetWorks/models.py:29-59). */
typedef struct N3dtMlpParams {
    const float* weight[N3DT_MLP_LAYERS];
    const float* bias[N3DT_MLP_LAYERS];
} N3dtMlpParams;

/* Neural-renderer parameters (reference: NetWorks/neural_renderer.py:49-69,
 * NetWorks/PixelShuffleUpsample.py:29-33).  Index i = block, 0 <= i < n_blocks (<= 8). */
#define N3DT_MAX_BLOCKS 8
typedef struct N3dtRenderParams {
    const float* to_rgb_w[N3DT_MAX_BLOCKS + 1]; /* feat_2_rgb_list.{0..n}: [3, c_i] */
    const float* to_rgb_b[N3DT_MAX_BLOCKS + 1];
    const float* psu1_w[N3DT_MAX_BLOCKS];       /* feat_upsample_list.i.layer_1: [2c, c]  */
    const float* psu1_b[N3DT_MAX_BLOCKS];
    const float* psu2_w[N3DT_MAX_BLOCKS];       /* feat_upsample_list.i.layer_2: [4c, 2c] */
    const float* psu2_b[N3DT_MAX_BLOCKS];
    const float* feat_w[N3DT_MAX_BLOCKS];       /* feat_layers.i: [c_{i+1}, c_i] */
    const float* feat_b[N3DT_MAX_BLOCKS];
} N3dtRenderParams;

/* Gradient buffers, same shapes and order as the parameters; every training entry point ACCUMULATES (+=)
 * into them, so the caller zeroes them (or passes PyTorch's .grad tensors to accumulate across calls). */
typedef struct N3dtMlpGrads {
    float* weight[N3DT_MLP_LAYERS];
    float* bias[N3DT_MLP_LAYERS];
} N3dtMlpGrads;

typedef struct N3dtRenderGrads {
    float* to_rgb_w[N3DT_MAX_BLOCKS + 1];
    float* to_rgb_b[N3DT_MAX_BLOCKS + 1];
    float* psu1_w[N3DT_MAX_BLOCKS];
    float* psu1_b[N3DT_MAX_BLOCKS];
    float* psu2_w[N3DT_MAX_BLOCKS];
    float* psu2_b[N3DT_MAX_BLOCKS];
    float* feat_w[N3DT_MAX_BLOCKS];
    float* feat_b[N3DT_MAX_BLOCKS];
} N3dtRenderGrads;

int n3dt_abi_version(void);
const char* n3dt_last_error(void);

/* ---- weights ------------------------------------------------------------------------------
 * Re-lay the MLP weights into the MFMA-fragment order (and dtype) the fused kernel streams.
 * PyTorch's optimizer mutates the fp32 parameters in place every step, so the binding
 * re-packs whenever a parameter's version counter changed (SURVEY 8b "Parameters / ownership").
 * Replaces: nothing in the reference (cuDNN consumes the Conv2d weights directly). */
size_t n3dt_mlp_packed_bytes(const N3dtGeom* g, int precision);
int n3dt_mlp_pack(const N3dtGeom* g, int precision, const N3dtMlpParams* p, void* packed, void* stream);

/* ---- volumetric render: a1..a7 of SURVEY 8a in one call --------------------------------------
 * Replaces, fused: GenSamplePoints.forward (NetWorks/utils.py:147-161), Embedder.forward
 * (utils.py:43-51), the latent expand+concat (HeadNeRFNet.py:84,149-152), MLPforNeRF.forward
 * (models.py:62-87), CalcRayColor.forward (utils.py:291-309) and the background merge
 * (HeadNeRFNet.py:103-112).
 *   xy      batch_xy, strides in g                R, Kinv [B,3,3]     T [B,3]
 *   shape [B,shape_dim]  appea [B,appea_dim]  audio [B,audio_dim] (NULL iff audio_dim==0)
 *   t_rand  [B,N_r,N_s+1] uniform noise for mode=="train" (utils.py:73-78), NULL for "test";
 *           with g->z_planes_given: the sample planes themselves (see N3dtGeom)
 *   bg_featmap  neural_render.bg_featmap [C, N_r] (NCHW parameter), NULL to skip the merge
 *   ray_bias    [B, N_r, 192] per-ray addend of RGB_layer_1's pre-activation: required iff g->vd_dim > 0 (include_vd), else NULL
 * outputs (any may be NULL, but one of fg_feat / merge_feat must be given):
 *   fg_feat [B,N_r,C]  bg_alpha [B,N_r]  depth [B,N_r]  weight [B,N_r,N_s]
 *   merge_feat [B,N_r,C] = fg_feat + bg_alpha * bg_featmap
 * `saved` (nullable, n3dt_render_saved_bytes) receives what n3dt_render_bwd needs. */
size_t n3dt_render_workspace_bytes(const N3dtGeom* g, int precision);
int n3dt_render_fwd(const N3dtGeom* g, int precision, const void* packed_mlp, const N3dtMlpParams* p,
                    const float* xy, const float* R, const float* T, const float* Kinv,
                    const float* shape, const float* appea, const float* audio, const float* t_rand,
                    const float* bg_featmap, const float* ray_bias,
                    float* fg_feat, float* bg_alpha, float* depth, float* weight, float* merge_feat,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- include_vd: the view-direction columns of RGB_layer_1 as a per-ray bias (ABI 5) -------------------------------
 * Replaces, for `include_vd=True`: GenSamplePoints' ray direction (NetWorks/utils.py:149-153), `vd_encoder(fg_dirs)`
 * (HeadNeRFNet.py:141-142: Embedder, 4 frequencies + input = 27 channels, utils.py:20-51), its expand over the samples and the
 * 27 matching columns of RGB_layer_1 (models.py:80; columns 384 .. 410 of its weight):
 *   ray_bias[b, r, o] = sum_j w_vd[o * ld_w + j] * Embedder_4(d(b, r))[j],   o < 192, j < 27
 * w_vd points at RGB_layer_1.weight[0][384] of the FULL parameter ([192, 384 + 27 + appea_dim], ld_w = its row length).
 * Inference helper; the differentiable path forms the same tensor with autograd on the host (a [B*N_r, 27] x [27, 192] product). */
int n3dt_ray_vd_bias(const N3dtGeom* g, const float* w_vd, int64_t ld_w, const float* xy, const float* R, const float* Kinv,
                     float* ray_bias, void* stream);

/* ---- hierarchical (fine) sample planes: SURVEY 8f row 4 ------------------------------------------
 * Replaces FineSample.forward (NetWorks/utils.py:211-263): inverse-CDF resampling of the coarse pass.
 *   g         geometry of the COARSE pass (n_samples = N_c >= 3).  z_planes_given = 0: the coarse planes are recomputed from
 *             T and t_rand; 1: `t_rand` carries them, [B,N_r,N_c+1] (the far edge is not read) -- the fine_samp_func seam,
 *             whose caller holds coarse_sample_dict["zvals"]
 *   n_fine    opt.num_sample_fine (N_f); n_fine + 1 samples are drawn
 *   weight    [B,N_r,N_c] compositing weights of the coarse pass (n3dt_render_fwd's `weight`)
 *   T, t_rand as given to the coarse pass (the coarse planes are recomputed from them)
 *   u         [B*N_r, N_f+1] uniform samples for mode=="train" (torch.rand at utils.py:227), NULL: linspace(0,1)
 *   z_planes  [B,N_r,N_c+N_f+1] out, ascending: the coarse planes merged with the new ones (utils.py:248).
 * The fine pass is then n3dt_render_fwd with n_samples = N_c + N_f, z_planes_given = 1, t_rand = z_planes and the
 * fine network's parameters (the reference's own call site, HeadNeRFNet.py:182-185, omits two arguments and cannot
 * run; this is that call with them supplied). */
int n3dt_fine_sample(const N3dtGeom* g, int n_fine, const float* weight, const float* T, const float* t_rand, const float* u,
                     float* z_planes, void* stream);

/* ---- the reference's inner seams as stand-alone operators ------------------------------------------
 * HeadNeRFNet keeps its sub-modules addressable (sample_func, vp_encoder, fg_CD_predictor, calc_color_func).
 * n3dt_render_fwd never materialises the tensors that cross those seams; these four calls do, in the reference's own
 * layouts, for callers that use the sub-modules directly.  Exact fp32, unfused, inference only.  M = N_r * N_s.
 *   n3dt_sample_points  GenSamplePoints.forward (NetWorks/utils.py:147-161): pts [B,3,N_r,N_s], zvals / z_dists
 *                       [B,1,N_r,N_s], ray_d [B,3,N_r], ray_l [B,1,N_r] (outputs may be NULL); t_rand as in n3dt_render_fwd
 *   n3dt_embed          Embedder.forward (utils.py:43-51): pts [B,3,M] -> pe [B,63,M]
 *   n3dt_mlp_points     MLPforNeRF.forward (NetWorks/models.py:62-87): audio [B,audio_dim,M] (NULL iff audio_dim == 0),
 *                       embed_vps [B,63+shape_dim,M], embed_vds [B,appea_dim,M] -> rgb [B,256,M], density [B,1,M]
 *                       (g supplies batch and the code widths; workspace from n3dt_mlp_points_workspace_bytes)
 *   n3dt_composite      CalcRayColor.forward (utils.py:291-309): rgb [B,C,N_r,N_s], density / z_dists / zvals
 *                       [B,1,N_r,N_s] -> feat [B,C,N_r], bg_alpha / depth [B,1,N_r], weight [B,1,N_r,N_s] (last three nullable) */
int n3dt_sample_points(const N3dtGeom* g, const float* xy, const float* R, const float* T, const float* Kinv, const float* t_rand,
                       float* pts, float* zvals, float* z_dists, float* ray_d, float* ray_l, void* stream);
int n3dt_embed(int batch, size_t m, const float* pts, float* pe, void* stream);
/* the same encoder with n_freqs frequencies (ABI 5): pts [B,3,M] -> pe [B, 3 + 6 n_freqs, M]; n_freqs = 4 is the reference's
 * vd_encoder (HeadNeRFNet.py:30-31,61), 10 equals n3dt_embed */
int n3dt_embed_freqs(int batch, size_t m, int n_freqs, const float* pts, float* pe, void* stream);
size_t n3dt_mlp_points_workspace_bytes(const N3dtGeom* g, size_t m);
int n3dt_mlp_points(const N3dtGeom* g, size_t m, const N3dtMlpParams* p, const float* audio, const float* embed_vps,
                    const float* embed_vds, float* rgb, float* density, void* workspace, size_t workspace_bytes, void* stream);
int n3dt_composite(int batch, int n_rays, int n_samples, int channels, const float* rgb, const float* density, const float* z_dists,
                   const float* zvals, float* feat, float* bg_alpha, float* depth, float* weight, void* stream);

/* ---- 2-D neural renderer: a8..a10 ----------------------------------------------------------------
 * Replaces NeuralRenderer.forward (NetWorks/neural_renderer.py:72-91) including
 * PixelShuffleUpsample.forward (PixelShuffleUpsample.py:36-45) and Blur (…:15-18, kornia filter2d).
 *   featmap [nb, fs, fs, C] ray-major   ->   img [nb, 3, P, P], P = fs << n_blocks
 * `precision` selects the arithmetic of the 1x1-conv GEMMs (N3DT_F32 exact; BF16/F16 inputs with
 * fp32 accumulate); blur, bilinear and the RGB pyramid are always fp32.
 * `nb` is the number of feature maps in this call (the binding renders the B merged maps and the
 * background map together, nb = B+1; reference calls the module twice, HeadNeRFNet.py:109,113).
 * Limit: nb * P * P * 32 < 2^31 (the kernels index one map level with 32-bit offsets: 255 maps at 512^2,
 * 63 at 1024^2); larger batches return N3DT_EINVAL before anything is launched -- split them. */
size_t n3dt_neural_render_workspace_bytes(const N3dtGeom* g, int nb);
int n3dt_neural_render_fwd(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap,
                           float* img, void* workspace, size_t workspace_bytes, void* stream);

/* The 16-bit modes re-lay the upsample blocks' fp32 weights into MFMA order at the start of every n3dt_neural_render_fwd
 * (the renderer takes raw parameters, so it cannot know whether an optimizer touched them): three 6 us launches, 2.5 % of
 * a one-head forward.  A caller that tracks its parameters (version counters) can split the two steps: the packed stream
 * lives in the tail of the caller-owned workspace.
 *   n3dt_neural_render_pack       re-packs the block weights into `workspace` (nothing else is touched);
 *   n3dt_neural_render_fwd_reuse  = n3dt_neural_render_fwd minus the packing: valid when the same workspace was last packed
 *                                   (by _pack or _fwd) for the same geometry, nb, precision and parameter VALUES.
 * fp32 precision reads the raw parameters: _pack is a no-op and _fwd_reuse equals _fwd. */
int n3dt_neural_render_pack(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, void* workspace,
                            size_t workspace_bytes, void* stream);
int n3dt_neural_render_fwd_reuse(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap,
                                 float* img, void* workspace, size_t workspace_bytes, void* stream);

/* ---- training path (fwd -> loss -> backward) -----------------------------------
 * `precision` = N3DT_F32: exact fp32 (the mode the gradient-parity tests pin); N3DT_BF16: every matrix product
 * on v_mfma_f32_32x32x16_bf16 with operands rounded to bf16 while staging (fp32 storage and accumulation), the
 * rest identical.  The forward keeps the per-layer activations in `saved` (caller-owned, sized by the
 * *_saved_bytes query) for the matching backward call, which must use the same precision; `workspace` is scratch.
 *
 * n3dt_render_train_fwd: same mathematics and outputs as n3dt_render_fwd (fg_feat, bg_alpha, depth?, merge_feat?).
 * n3dt_render_bwd: given dL/d(merge_feat) (and optionally dL/d(fg_feat), dL/d(bg_alpha), both nullable),
 *   accumulates parameter gradients into `grads`, adds dL/d(bg_featmap) [C,N_r] into d_bg_featmap (nullable)
 *   and writes dL/d(shape) [B,shape_dim], dL/d(appea) [B,appea_dim], dL/d(audio) [B,audio_dim] (each nullable).
 *   Differentiates NetWorks/models.py:62-87, NetWorks/utils.py:268-309, NetWorks/HeadNeRFNet.py:84-112,149-152.
 *   Camera gradients (the fitting use-case, FittingSingleImage_new.py:826-859): when d_R [B,3,3] and/or d_T [B,3]
 *   are non-NULL they receive dL/d(batch_Rmats), dL/d(batch_Tvecs); xy, R, T, Kinv (and t_rand if the forward
 *   used it) must then be the forward's inputs.  Pass NULL for all seven to skip that work.
 *   d_ray_bias [B, N_r, 192] (iff g->vd_dim > 0, else NULL) receives dL/d(ray_bias): the per-ray sums of RGB_layer_1's
 *   pre-activation gradient, from which the caller's autograd forms the gradients of the 27 view-direction columns (and, through
 *   the ray direction, of the camera rotation).
 *   `grads` == NULL (both n3dt_render_bwd and n3dt_neural_render_bwd): the network is FROZEN, as in single-image fitting
 *   (FittingSingleImage_new.py:826-859 optimises codes and cameras only) -- no parameter gradient is computed, only the
 *   gradients of the inputs (d_bg_featmap must then be NULL too). */
size_t n3dt_render_train_saved_bytes(const N3dtGeom* g);
size_t n3dt_render_train_workspace_bytes(const N3dtGeom* g);
int n3dt_render_train_fwd(const N3dtGeom* g, int precision, const void* packed_mlp, const N3dtMlpParams* p,
                          const float* xy, const float* R, const float* T, const float* Kinv,
                          const float* shape, const float* appea, const float* audio, const float* t_rand,
                          const float* bg_featmap, const float* ray_bias, float* fg_feat, float* bg_alpha, float* depth, float* merge_feat,
                          void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream);
int n3dt_render_bwd(const N3dtGeom* g, int precision, const N3dtMlpParams* p, const N3dtMlpGrads* grads,
                    const float* shape, const float* appea, const float* audio, const float* bg_featmap,
                    const float* d_merge_feat, const float* d_fg_feat, const float* d_bg_alpha,
                    const void* saved, size_t saved_bytes,
                    float* d_bg_featmap, float* d_shape, float* d_appea, float* d_audio, float* d_ray_bias,
                    const float* xy, const float* R, const float* T, const float* Kinv, const float* t_rand,
                    float* d_R, float* d_T,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Neural renderer with saved activations, and its backward (differentiates NetWorks/neural_renderer.py:72-91,
 * NetWorks/PixelShuffleUpsample.py:36-45).  d_featmap [nb,fs,fs,C] is overwritten; parameter gradients accumulate. */
size_t n3dt_neural_render_train_saved_bytes(const N3dtGeom* g, int nb);
size_t n3dt_neural_render_train_workspace_bytes(const N3dtGeom* g, int nb);
int n3dt_neural_render_train_fwd(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const float* featmap, float* img,
                                 void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, void* stream);
int n3dt_neural_render_bwd(const N3dtGeom* g, int nb, int precision, const N3dtRenderParams* p, const N3dtRenderGrads* grads,
                           const float* featmap, const float* d_img, const void* saved, size_t saved_bytes,
                           float* d_featmap, void* workspace, size_t workspace_bytes, void* stream);

/* ---- display conversion ----------------------------------------------------------------------------
 * Replaces `(img.permute(1,2,0).numpy() * 255).astype(np.uint8)` (talker_trainer.py:1205, Utils/RenderUtils.py:123-125):
 *   img [n_images,3,pixels] float in (0,1)  ->  out [n_images,pixels,3] uint8 */
int n3dt_img_to_uint8(int n_images, int pixels, const float* img, unsigned char* out, void* stream);

/* ---- fused loss tail -----------------------------------------------------------------
 * The three MSE data terms of the reference's loss (Utils/HeadNeRFLossUtils.py:125-146: bg_loss, head_loss,
 * nonhead_loss, including its nan_to_num) in one pass, and their gradient in one more; replaces three boolean-mask
 * gathers.  merge_img, gt [B,3,P,P]; bg_img [1,3,P,P]; mask [B,1,P,P] (head where >= 0.5); pixels = P*P.
 * n3dt_loss_fwd writes terms[4] = {bg, head, nonhead, (bg + head) + nonhead -- the reference's total, :228-231} and keeps
 * its sums/counts in acc[8] for n3dt_loss_bwd, which takes the upstream gradients g[3] of the three terms and / or g_total[1]
 * of their sum (each nullable, not both) and writes d_merge [B,3,P,P] and d_bg [1,3,P,P].  (ABI 4: acc grew from 6 to 8
 * floats, terms from 3 to 4, g_total is new.) */
int n3dt_loss_fwd(int batch, int pixels, const float* merge_img, const float* bg_img, const float* gt, const float* mask,
                  float bg_value, float* acc, float* terms, void* stream);
int n3dt_loss_bwd(int batch, int pixels, const float* merge_img, const float* bg_img, const float* gt, const float* mask,
                  float bg_value, const float* acc, const float* g, const float* g_total, float* d_merge, float* d_bg, void* stream);

/* [C, N_r] (NCHW parameter) -> [N_r, C]; used to feed bg_featmap to the renderer */
int n3dt_chw_to_hwc(int C, int n, const float* src, float* dst, void* stream);

/* ---- per-call host overhead: input staging + hipGraph replay ----------------------------------------
 * The reference's call shapes are small batches (validation B=1 talker_trainer.py:1119, fitting B=1 x 300 iterations
 * FittingSingleImage_new.py:887): a forward is ~17 short kernels and launch gaps are a quarter of the step.  Every entry
 * point above allocates nothing and only enqueues on `stream`, so a whole forward can be recorded once into a hipGraph
 * and replayed with ONE launch.  A replay reads the addresses recorded at capture time: the caller keeps static
 * buffers and refreshes them with n3dt_stage_inputs (one small kernel instead of one copy per tensor).
 *
 * n3dt_stage_inputs: n <= N3DT_STAGE_MAX copies src[i] -> dst[i] of count[i] floats in one launch.  Entry 0 may be a
 * strided 3-D view (batch_xy [B,2,N_r] with the strides the caller holds it in, e.g. an expand()ed grid): set
 * view_dims / view_strides (elements) and count[0] = product of view_dims; leave view_dims[0] = 0 for a flat copy. */
#define N3DT_STAGE_MAX 12
typedef struct N3dtStageCopy {
    const float* src[N3DT_STAGE_MAX];
    float* dst[N3DT_STAGE_MAX];
    int64_t count[N3DT_STAGE_MAX];
    int64_t view_dims[3], view_strides[3];
    int32_t n;
} N3dtStageCopy;
int n3dt_stage_inputs(const N3dtStageCopy* st, void* stream);

/* hipStreamBeginCapture (relaxed mode: other threads / streams are unaffected) ... hipStreamEndCapture +
 * hipGraphInstantiate.  Between begin and end, enqueue the n3dt_* calls of one forward on `stream`; nothing runs until
 * n3dt_graph_launch.  *graph_out is an opaque handle owned by the library until n3dt_graph_destroy. */
int n3dt_graph_begin(void* stream);
int n3dt_graph_end(void* stream, void** graph_out);
int n3dt_graph_launch(void* graph, void* stream);
int n3dt_graph_destroy(void* graph);

/* ---- measurement hook (bench.py only) ---------------------------------------------------------
 * While enabled, every n3dt_render_fwd brackets its fused MLP kernel launch (the roofline kernel,
 * not the fold / head kernels around it) with a hipEvent pair on the caller's stream.
 * n3dt_prof_collect synchronises those events and returns each launch's duration in ms.
 * Process-global and opt-in: the only mutable global state in the library; not for use while
 * other threads are rendering. */
int n3dt_prof_enable(int max_records);  /* 0 disables and frees the events */
int n3dt_prof_collect(float* ms_out, int capacity, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* N3DT_API_H_ */
