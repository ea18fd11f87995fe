"""Host-side launchers over the C ABI (include/n3dt.h).

torch is used here for device memory, the current HIP stream and nothing else: every tensor is
handed to libn3dt.so as a raw device pointer.  No CPU or eager-PyTorch fallback exists; calling
these with CPU tensors raises.
"""
import ctypes

import torch

from . import _lib
from ._lib import Geom, MlpParams, RenderParams, check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "libn3dt works on device memory only (no CPU fallback)"
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t):
    """fp32, contiguous, detached view of a (small) tensor"""
    return t.detach().to(dtype=torch.float32).contiguous()


class _Workspace:
    """Grow-only scratch buffers keyed by (role, device, stream): reuse is ordered by the stream the kernels are
    enqueued on, so two streams (or two modules driven from two streams) never share scratch.  A buffer that is
    outgrown goes back to the caching allocator, which is stream-aware for the stream it was allocated on -- the
    one that used it."""

    def __init__(self):
        self.buf = {}

    def get(self, key, nbytes, device):
        assert device.type == "cuda", "libn3dt works on device memory only (no CPU fallback)"
        k = (key, device.index, torch.cuda.current_stream(device).cuda_stream)
        b = self.buf.get(k)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self.buf[k] = b
        return b


WORKSPACE = _Workspace()


def make_geom(batch, n_rays, n_samples, hidden, feat_nc, shape_dim, appea_dim, audio_dim, featmap_size, n_blocks,
              world_z1, world_z2, xy_strides=(0, 0, 0), z_planes_given=0, bg_is_hwc=0, vd_dim=0):
    g = Geom()
    g.z_planes_given = int(z_planes_given)
    g.bg_is_hwc = int(bg_is_hwc)
    g.vd_dim = int(vd_dim)  # include_vd: 27, and the render calls take a per-ray bias of RGB_layer_1
    g.batch, g.n_rays, g.n_samples = int(batch), int(n_rays), int(n_samples)
    g.hidden, g.feat_nc = int(hidden), int(feat_nc)
    g.shape_dim, g.appea_dim, g.audio_dim = int(shape_dim), int(appea_dim), int(audio_dim)
    g.featmap_size, g.n_blocks = int(featmap_size), int(n_blocks)
    g.world_z1, g.world_z2 = float(world_z1), float(world_z2)
    g.xy_stride_b, g.xy_stride_c, g.xy_stride_r = [int(s) for s in xy_strides]
    return g


def mlp_params(weights, biases):
    """weights/biases: 12 fp32 contiguous device tensors in MLP_ORDER."""
    p = MlpParams()
    for i, (w, b) in enumerate(zip(weights, biases)):
        assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
        assert b.is_cuda and b.dtype == torch.float32 and b.is_contiguous()
        p.weight[i] = w.data_ptr()
        p.bias[i] = b.data_ptr()
    return p


def render_params(to_rgb, psu1, psu2, feat):
    """each argument: list of (weight, bias) fp32 contiguous device tensors"""
    p = RenderParams()
    for i, (w, b) in enumerate(to_rgb):
        p.to_rgb_w[i], p.to_rgb_b[i] = w.data_ptr(), b.data_ptr()
    for i, (w, b) in enumerate(psu1):
        p.psu1_w[i], p.psu1_b[i] = w.data_ptr(), b.data_ptr()
    for i, (w, b) in enumerate(psu2):
        p.psu2_w[i], p.psu2_b[i] = w.data_ptr(), b.data_ptr()
    for i, (w, b) in enumerate(feat):
        p.feat_w[i], p.feat_b[i] = w.data_ptr(), b.data_ptr()
    return p


def pack_mlp(geom, precision, params, device, out=None):
    nbytes = lib().n3dt_mlp_packed_bytes(ctypes.byref(geom), precision)
    if nbytes == 0:
        raise _lib.N3dtError("n3dt_mlp_packed_bytes: " + lib().n3dt_last_error().decode())
    packed = out if out is not None and out.numel() >= nbytes else torch.empty(nbytes, dtype=torch.uint8, device=device)
    check(lib().n3dt_mlp_pack(ctypes.byref(geom), precision, ctypes.byref(params), _ptr(packed), _stream()), "n3dt_mlp_pack")
    return packed


def render_workspace_bytes(geom, precision):
    return _bytes_or_raise(lib().n3dt_render_workspace_bytes(ctypes.byref(geom), precision), "n3dt_render_workspace_bytes")


def render_fwd(geom, precision, packed, params, xy, R, T, Kinv, shape, appea, audio, t_rand, bg_featmap,
               want_depth=False, want_weight=False, want_merge=True, merge_out=None, want_fg=True, weight_out=None, ws=None, ray_bias=None):
    """a1..a7 fused.  Returns dict(fg_feat [B,Nr,C]?, bg_alpha [B,Nr]?, depth?, weight?, merge_feat?).
    With merge_out / weight_out / ws given (and want_fg, want_depth off) the call allocates nothing: hipGraph-capturable."""
    dev = xy.device
    B, Nr, Ns, C = geom.batch, geom.n_rays, geom.n_samples, geom.feat_nc
    assert want_fg or want_merge
    out = {
        "fg_feat": torch.empty(B, Nr, C, dtype=torch.float32, device=dev) if want_fg else None,
        "bg_alpha": torch.empty(B, Nr, dtype=torch.float32, device=dev) if want_fg else None,
        "depth": torch.empty(B, Nr, dtype=torch.float32, device=dev) if want_depth else None,
        "weight": (weight_out if weight_out is not None else torch.empty(B, Nr, Ns, dtype=torch.float32, device=dev))
        if want_weight else None,
        "merge_feat": (merge_out if merge_out is not None else torch.empty(B, Nr, C, dtype=torch.float32, device=dev))
        if want_merge else None,
    }
    ws_bytes = render_workspace_bytes(geom, precision)
    if ws is None:
        ws = WORKSPACE.get("render", ws_bytes, dev)
    assert ws.numel() >= ws_bytes
    rc = lib().n3dt_render_fwd(
        ctypes.byref(geom), precision, _ptr(packed), ctypes.byref(params), _ptr(xy), _ptr(R), _ptr(T), _ptr(Kinv),
        _ptr(shape), _ptr(appea), _ptr(audio), _ptr(t_rand), _ptr(bg_featmap) if want_merge else None, _ptr(ray_bias),
        _ptr(out["fg_feat"]), _ptr(out["bg_alpha"]), _ptr(out["depth"]), _ptr(out["weight"]), _ptr(out["merge_feat"]),
        _ptr(ws), ws_bytes, _stream())
    check(rc, "n3dt_render_fwd")
    return out


def fine_sample(geom, n_fine, weight, T, t_rand=None, u=None, out=None):
    """Hierarchical sample planes (FineSample.forward): coarse weights [B,Nr,Nc] -> planes [B,Nr,Nc+n_fine+1]."""
    B, Nr, Nc = geom.batch, geom.n_rays, geom.n_samples
    z = out if out is not None else torch.empty(B, Nr, Nc + n_fine + 1, dtype=torch.float32, device=weight.device)
    check(lib().n3dt_fine_sample(ctypes.byref(geom), int(n_fine), _ptr(weight), _ptr(T), _ptr(t_rand), _ptr(u), _ptr(z), _stream()),
          "n3dt_fine_sample")
    return z


def img_to_uint8(img):
    """[V,3,P,P] float in (0,1) -> [V,P,P,3] uint8 on the device (the reference's display conversion)."""
    img = _f32c(img)
    V, _, P, Q = img.shape
    out = torch.empty(V, P, Q, 3, dtype=torch.uint8, device=img.device)
    check(lib().n3dt_img_to_uint8(V, P * Q, _ptr(img), _ptr(out), _stream()), "n3dt_img_to_uint8")
    return out


# ---- the reference's inner seams as stand-alone operators (csrc/seams.hip) -------------------------------------
def sample_points(geom, xy, R, T, Kinv, t_rand=None):
    """GenSamplePoints.forward: dict of pts [B,3,Nr,Ns], zvals / z_dists [B,1,Nr,Ns], ray_d [B,3,Nr], ray_l [B,1,Nr]."""
    B, Nr, Ns, dev = geom.batch, geom.n_rays, geom.n_samples, xy.device
    out = {"pts": torch.empty(B, 3, Nr, Ns, dtype=torch.float32, device=dev),
           "zvals": torch.empty(B, 1, Nr, Ns, dtype=torch.float32, device=dev),
           "z_dists": torch.empty(B, 1, Nr, Ns, dtype=torch.float32, device=dev),
           "ray_d": torch.empty(B, 3, Nr, dtype=torch.float32, device=dev),
           "ray_l": torch.empty(B, 1, Nr, dtype=torch.float32, device=dev)}
    check(lib().n3dt_sample_points(ctypes.byref(geom), _ptr(xy), _ptr(R), _ptr(T), _ptr(Kinv), _ptr(t_rand), _ptr(out["pts"]),
                                   _ptr(out["zvals"]), _ptr(out["z_dists"]), _ptr(out["ray_d"]), _ptr(out["ray_l"]), _stream()),
          "n3dt_sample_points")
    return out


def embed(pts):
    """Embedder.forward: pts [B,3,...] -> [B,63,...]."""
    pts = _f32c(pts)
    B = pts.shape[0]
    M = pts[0, 0].numel()
    pe = torch.empty((B, 63) + tuple(pts.shape[2:]), dtype=torch.float32, device=pts.device)
    check(lib().n3dt_embed(B, M, _ptr(pts), _ptr(pe), _stream()), "n3dt_embed")
    return pe


def embed_freqs(pts, n_freqs):
    """Embedder(N_freqs=n_freqs, include_input=True).forward: pts [B,3,...] -> [B, 3 + 6 n_freqs, ...]."""
    pts = _f32c(pts)
    B = pts.shape[0]
    M = pts[0, 0].numel()
    pe = torch.empty((B, 3 + 6 * n_freqs) + tuple(pts.shape[2:]), dtype=torch.float32, device=pts.device)
    check(lib().n3dt_embed_freqs(B, M, int(n_freqs), _ptr(pts), _ptr(pe), _stream()), "n3dt_embed_freqs")
    return pe


def ray_vd_bias(geom, w_rgb1, xy, R, Kinv, out=None):
    """include_vd: the per-ray bias of RGB_layer_1 from the ray direction (n3dt_ray_vd_bias).  w_rgb1: the layer's FULL 2-D
    weight [192, 384 + 27 + appea_dim] (contiguous); its columns 384 .. 410 are read in place."""
    assert w_rgb1.is_contiguous() and w_rgb1.dim() == 2 and w_rgb1.shape[0] == 192 and w_rgb1.shape[1] >= 384 + 27
    B, Nr = geom.batch, geom.n_rays
    rb = out if out is not None else torch.empty(B, Nr, 192, dtype=torch.float32, device=xy.device)
    w_vd = ctypes.c_void_p(w_rgb1.data_ptr() + 384 * 4)
    check(lib().n3dt_ray_vd_bias(ctypes.byref(geom), w_vd, int(w_rgb1.shape[1]), _ptr(xy), _ptr(R), _ptr(Kinv), _ptr(rb), _stream()),
          "n3dt_ray_vd_bias")
    return rb


def mlp_points(geom, params, audio, embed_vps, embed_vds):
    """MLPforNeRF.forward on materialised inputs [B,C,Nr,Ns]: returns (rgb [B,256,Nr,Ns], density [B,1,Nr,Ns])."""
    vps, vds = _f32c(embed_vps), _f32c(embed_vds)
    aud = _f32c(audio) if geom.audio_dim > 0 else None
    B = vps.shape[0]
    M = vps[0, 0].numel()
    tail = tuple(vps.shape[2:])
    rgb = torch.empty((B, geom.feat_nc) + tail, dtype=torch.float32, device=vps.device)
    dens = torch.empty((B, 1) + tail, dtype=torch.float32, device=vps.device)
    wb = _bytes_or_raise(lib().n3dt_mlp_points_workspace_bytes(ctypes.byref(geom), M), "n3dt_mlp_points_workspace_bytes")
    ws = WORKSPACE.get("seam_mlp", wb, vps.device)
    check(lib().n3dt_mlp_points(ctypes.byref(geom), M, ctypes.byref(params), _ptr(aud), _ptr(vps), _ptr(vds), _ptr(rgb), _ptr(dens),
                                _ptr(ws), wb, _stream()), "n3dt_mlp_points")
    return rgb, dens


def composite(rgb, density, z_dists, zvals):
    """CalcRayColor.forward: (feat [B,C,Nr], bg_alpha [B,1,Nr], depth [B,1,Nr], weight [B,1,Nr,Ns])."""
    rgb, density, z_dists, zvals = _f32c(rgb), _f32c(density), _f32c(z_dists), _f32c(zvals)
    B, C, Nr, Ns = rgb.shape
    dev = rgb.device
    feat = torch.empty(B, C, Nr, dtype=torch.float32, device=dev)
    ba = torch.empty(B, 1, Nr, dtype=torch.float32, device=dev)
    dp = torch.empty(B, 1, Nr, dtype=torch.float32, device=dev)
    w = torch.empty(B, 1, Nr, Ns, dtype=torch.float32, device=dev)
    check(lib().n3dt_composite(B, Nr, Ns, C, _ptr(rgb), _ptr(density), _ptr(z_dists), _ptr(zvals), _ptr(feat), _ptr(ba), _ptr(dp), _ptr(w),
                               _stream()), "n3dt_composite")
    return feat, ba, dp, w


def neural_render_workspace_bytes(geom, nb):
    ws_bytes = lib().n3dt_neural_render_workspace_bytes(ctypes.byref(geom), nb)
    if ws_bytes == 0:
        raise _lib.N3dtError("n3dt_neural_render_workspace_bytes: unsupported geometry")
    return ws_bytes


def neural_render_pack(geom, nb, rparams, precision, ws):
    """Re-pack the upsample blocks' weights into the tail of `ws` (16-bit modes; a no-op in fp32)."""
    check(lib().n3dt_neural_render_pack(ctypes.byref(geom), nb, precision, ctypes.byref(rparams), _ptr(ws), ws.numel(), _stream()),
          "n3dt_neural_render_pack")


def neural_render_fwd(geom, nb, rparams, featmap, precision=0, img=None, ws=None, reuse_packed=False):
    """featmap [nb, fs, fs, C] (ray-major) -> img [nb, 3, P, P].  With img / ws given the call allocates nothing.
    reuse_packed: `ws` was last packed (neural_render_pack / a previous call) for these parameter values: skip the packing."""
    dev = featmap.device
    P = geom.featmap_size << geom.n_blocks
    if img is None:
        img = torch.empty(nb, 3, P, P, dtype=torch.float32, device=dev)
    ws_bytes = neural_render_workspace_bytes(geom, nb)
    if ws is None:
        ws = WORKSPACE.get("nr", ws_bytes, dev)
    assert ws.numel() >= ws_bytes
    fn = lib().n3dt_neural_render_fwd_reuse if reuse_packed else lib().n3dt_neural_render_fwd
    check(fn(ctypes.byref(geom), nb, precision, ctypes.byref(rparams), _ptr(featmap), _ptr(img), _ptr(ws), ws_bytes, _stream()),
          "n3dt_neural_render_fwd")
    return img


def chw_to_hwc(src, C, n, dst=None):
    """[C, n] -> [n, C] on the device"""
    if dst is None:
        dst = torch.empty(n, C, dtype=torch.float32, device=src.device)
    check(lib().n3dt_chw_to_hwc(C, n, _ptr(src), _ptr(dst), _stream()), "n3dt_chw_to_hwc")
    return dst


# ---- input staging + hipGraph replay -----------------------------------------------------------------
def stage_inputs(pairs, view=None):
    """One launch copying every (src, dst) pair of fp32 device tensors (dst contiguous).  `view`: the first pair's source is
    read as a strided 3-D view -- pass the tensor itself (e.g. an expand()ed batch_xy), its sizes and strides are used."""
    st = _lib.Stage()
    assert 1 <= len(pairs) <= _lib.STAGE_MAX
    for i, (src, dst) in enumerate(pairs):
        assert src.is_cuda and dst.is_cuda and src.dtype == torch.float32 and dst.dtype == torch.float32 and dst.is_contiguous()
        assert src.numel() == dst.numel() and (i == 0 and view is not None or src.is_contiguous())
        st.src[i], st.dst[i], st.count[i] = src.data_ptr(), dst.data_ptr(), src.numel()
    if view is not None:
        assert view.dim() == 3
        for d in range(3):
            st.view_dims[d], st.view_strides[d] = view.shape[d], view.stride(d)
    st.n = len(pairs)
    check(lib().n3dt_stage_inputs(ctypes.byref(st), _stream()), "n3dt_stage_inputs")


def graph_begin(stream):
    check(lib().n3dt_graph_begin(ctypes.c_void_p(stream.cuda_stream)), "n3dt_graph_begin")


def graph_end(stream):
    h = ctypes.c_void_p()
    check(lib().n3dt_graph_end(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(h)), "n3dt_graph_end")
    return h


def graph_launch(handle):
    check(lib().n3dt_graph_launch(handle, _stream()), "n3dt_graph_launch")


def graph_destroy(handle):
    if handle:
        lib().n3dt_graph_destroy(handle)


# ---- training path -------------------------------------------------------------------------------
def _bytes_or_raise(n, what):
    if n == 0:
        raise _lib.N3dtError(what + ": " + lib().n3dt_last_error().decode())
    return n


def render_train_fwd(geom, packed, params, xy, R, T, Kinv, shape, appea, audio, t_rand, bg_featmap, precision=0, merge_out=None, ray_bias=None):
    """Forward with saved activations.  `packed` = pack_mlp(...) of the same precision.  Returns (out dict, saved buffer).
    merge_out: caller-owned [B, N_r, C] buffer for the merged map (a slice of the renderer's input batch)."""
    dev = xy.device
    B, Nr, C = geom.batch, geom.n_rays, geom.feat_nc
    if merge_out is not None:
        assert merge_out.is_contiguous() and tuple(merge_out.shape) == (B, Nr, C) and merge_out.dtype == torch.float32
    out = {
        "fg_feat": torch.empty(B, Nr, C, dtype=torch.float32, device=dev),
        "bg_alpha": torch.empty(B, Nr, dtype=torch.float32, device=dev),
        "merge_feat": merge_out if merge_out is not None else torch.empty(B, Nr, C, dtype=torch.float32, device=dev),
    }
    sbytes = _bytes_or_raise(lib().n3dt_render_train_saved_bytes(ctypes.byref(geom)), "n3dt_render_train_saved_bytes")
    wbytes = _bytes_or_raise(lib().n3dt_render_train_workspace_bytes(ctypes.byref(geom)), "n3dt_render_train_workspace_bytes")
    saved = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    ws = WORKSPACE.get("train", wbytes, dev)
    check(lib().n3dt_render_train_fwd(
        ctypes.byref(geom), precision, _ptr(packed), ctypes.byref(params), _ptr(xy), _ptr(R), _ptr(T), _ptr(Kinv), _ptr(shape), _ptr(appea),
        _ptr(audio), _ptr(t_rand), _ptr(bg_featmap), _ptr(ray_bias), _ptr(out["fg_feat"]), _ptr(out["bg_alpha"]), None, _ptr(out["merge_feat"]),
        _ptr(saved), sbytes, _ptr(ws), wbytes, _stream()), "n3dt_render_train_fwd")
    return out, saved


def render_bwd(geom, params, grads, shape, appea, audio, bg_featmap, d_merge, saved, cam=None, precision=0, d_bg=None, frozen=False):
    """Backward of render_train_fwd.  `grads` (MlpParams struct of zeroed tensors) is accumulated into.
    cam = (xy, R, T, Kinv, t_rand) requests camera gradients.
    Returns (d_bg_featmap [C,Nr], d_shape, d_appea, d_audio, d_R, d_T) -- and, with geom.vd_dim > 0 (include_vd), a seventh
    entry d_ray_bias [B, N_r, 192]."""
    dev = d_merge.device
    B, Nr, C = geom.batch, geom.n_rays, geom.feat_nc
    if frozen:  # grads is None: no parameter gradient, no d_bg_featmap
        assert grads is None
        d_bg = None
    elif d_bg is None:  # (a caller-provided buffer -- a slice of the gradient arena -- is already zeroed)
        d_bg = torch.zeros(C, Nr, dtype=torch.float32, device=dev)
    # the three code gradients side by side in one allocation: the library zeroes adjacent buffers with one launch
    S_, A_, U_ = geom.shape_dim, geom.appea_dim, geom.audio_dim
    codes = torch.empty(B * (S_ + A_ + U_), dtype=torch.float32, device=dev)
    d_shape = codes[:B * S_].view(B, S_)
    d_appea = codes[B * S_:B * (S_ + A_)].view(B, A_)
    d_audio = codes[B * (S_ + A_):].view(B, U_) if U_ > 0 else None
    d_R = d_T = None
    cam_ptrs = [None] * 5
    if cam is not None:
        d_R = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
        d_T = torch.empty(B, 3, dtype=torch.float32, device=dev)
        cam_ptrs = [_ptr(t) for t in cam]
    d_ray = torch.empty(B, Nr, 192, dtype=torch.float32, device=dev) if geom.vd_dim > 0 else None
    wbytes = lib().n3dt_render_train_workspace_bytes(ctypes.byref(geom))
    ws = WORKSPACE.get("train", wbytes, dev)
    check(lib().n3dt_render_bwd(
        ctypes.byref(geom), precision, ctypes.byref(params), None if grads is None else ctypes.byref(grads), _ptr(shape), _ptr(appea), _ptr(audio),
        _ptr(bg_featmap),
        _ptr(d_merge), None, None, _ptr(saved), saved.numel(), _ptr(d_bg), _ptr(d_shape), _ptr(d_appea), _ptr(d_audio), _ptr(d_ray),
        *cam_ptrs, _ptr(d_R), _ptr(d_T), _ptr(ws), wbytes, _stream()), "n3dt_render_bwd")
    if d_ray is not None:
        return d_bg, d_shape, d_appea, d_audio, d_R, d_T, d_ray
    return d_bg, d_shape, d_appea, d_audio, d_R, d_T


def neural_render_train_fwd(geom, nb, rparams, featmap, precision=0):
    dev = featmap.device
    P = geom.featmap_size << geom.n_blocks
    img = torch.empty(nb, 3, P, P, dtype=torch.float32, device=dev)
    sbytes = _bytes_or_raise(lib().n3dt_neural_render_train_saved_bytes(ctypes.byref(geom), nb), "n3dt_neural_render_train_saved_bytes")
    wbytes = _bytes_or_raise(lib().n3dt_neural_render_train_workspace_bytes(ctypes.byref(geom), nb),
                             "n3dt_neural_render_train_workspace_bytes")
    saved = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    ws = WORKSPACE.get("nr_train", wbytes, dev)
    check(lib().n3dt_neural_render_train_fwd(ctypes.byref(geom), nb, precision, ctypes.byref(rparams), _ptr(featmap), _ptr(img), _ptr(saved), sbytes,
                                             _ptr(ws), wbytes, _stream()), "n3dt_neural_render_train_fwd")
    return img, saved


def neural_render_bwd(geom, nb, rparams, rgrads, featmap, d_img, saved, precision=0):
    dev = featmap.device
    d_feat = torch.empty_like(featmap)
    wbytes = lib().n3dt_neural_render_train_workspace_bytes(ctypes.byref(geom), nb)
    ws = WORKSPACE.get("nr_train", wbytes, dev)
    check(lib().n3dt_neural_render_bwd(ctypes.byref(geom), nb, precision, ctypes.byref(rparams), None if rgrads is None else ctypes.byref(rgrads),
                                       _ptr(featmap), _ptr(d_img),
                                       _ptr(saved), saved.numel(), _ptr(d_feat), _ptr(ws), wbytes, _stream()), "n3dt_neural_render_bwd")
    return d_feat
