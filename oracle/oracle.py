"""ctypes binding of the CPU restatement (oracle/n3dt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MLP_ORDER = ["FeaExt_module_%d" % i for i in range(8)] + ["density_module", "RGB_layer_0", "RGB_layer_1", "RGB_layer_2"]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _ptr_array(arrs):
    keep = [_f32(a) for a in arrs]
    arr = (ctypes.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
    return arr, keep


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def mlp_weight_list(sd, prefix="fg_CD_predictor."):
    out = []
    for name in MLP_ORDER:
        w = _np(sd[prefix + name + ".weight"])
        out.append(w.reshape(w.shape[0], w.shape[1]))
        out.append(_np(sd[prefix + name + ".bias"]))
    return out


def nr_weight_list(sd, n_blocks, prefix="neural_render."):
    def wb(name):
        w = _np(sd[prefix + name + ".weight"])
        return [w.reshape(w.shape[0], w.shape[1]), _np(sd[prefix + name + ".bias"])]
    out = wb("feat_2_rgb_list.0")
    for i in range(n_blocks):
        out += wb("feat_upsample_list.%d.layer_1" % i) + wb("feat_upsample_list.%d.layer_2" % i)
        out += wb("feat_layers.%d" % i) + wb("feat_2_rgb_list.%d" % (i + 1))
    return out


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


def num_threads():
    return int(lib().orc_num_threads())


def sample(xy, R, T, Kinv, n_samples, world_z1=2.5, world_z2=-3.5, t_rand=None):
    xy, R, T, Kinv = _f32(xy), _f32(R), _f32(T).reshape(-1, 3), _f32(Kinv)
    B, _, Nr = xy.shape
    Ns = n_samples
    tr = _f32(t_rand) if t_rand is not None else None
    ray_d = np.empty((B, 3, Nr), np.float32)
    ray_l = np.empty((B, 1, Nr), np.float32)
    pts = np.empty((B, 3, Nr, Ns), np.float32)
    zvals = np.empty((B, 1, Nr, Ns), np.float32)
    z_dists = np.empty((B, 1, Nr, Ns), np.float32)
    lib().orc_sample(B, Nr, Ns, _p(xy), _p(R), _p(T), _p(Kinv), ctypes.c_float(world_z1), ctypes.c_float(world_z2),
                     _p(tr), _p(ray_d), _p(ray_l), _p(pts), _p(zvals), _p(z_dists))
    return {"ray_d": ray_d, "ray_l": ray_l, "pts": pts, "zvals": zvals, "z_dists": z_dists}


def fine_sample(weight, zvals, n_fine, u=None):
    """FineSample.forward: weight / zvals [B,1,Nr,Nc] (coarse) -> planes [B,Nr,Nc+n_fine+1]."""
    weight, zvals = _f32(weight), _f32(zvals)
    B, _, Nr, Nc = weight.shape
    uu = _f32(u) if u is not None else None
    out = np.empty((B, Nr, Nc + n_fine + 1), np.float32)
    lib().orc_fine_sample(ctypes.c_long(B * Nr), Nc, int(n_fine), _p(weight), _p(zvals), _p(uu), _p(out))
    return out


def sample_planes(xy, R, T, Kinv, planes):
    """FineSample._calc_sample_points_by_zvals: planes [B,Nr,N+1] -> pts / zvals / z_dists over N samples."""
    xy, R, T, Kinv, planes = _f32(xy), _f32(R), _f32(T).reshape(-1, 3), _f32(Kinv), _f32(planes)
    B, _, Nr = xy.shape
    N = planes.shape[-1] - 1
    pts = np.empty((B, 3, Nr, N), np.float32)
    zvals = np.empty((B, 1, Nr, N), np.float32)
    z_dists = np.empty((B, 1, Nr, N), np.float32)
    lib().orc_sample_planes(B, Nr, N, _p(xy), _p(R), _p(T), _p(Kinv), _p(planes), _p(pts), _p(zvals), _p(z_dists))
    return {"pts": pts, "zvals": zvals, "z_dists": z_dists}


def embed(pts):
    pts = _f32(pts)
    B = pts.shape[0]
    M = int(np.prod(pts.shape[2:]))
    pe = np.empty((B, 63) + pts.shape[2:], np.float32)
    lib().orc_embed(B, ctypes.c_long(M), _p(pts), _p(pe))
    return pe


def mlp(sd, pe, shape, appea, audio, C=256, H=384, prefix="fg_CD_predictor.", vd=None):
    """vd: [B,27,...] the encoded view directions (include_vd=True), else None."""
    pe, shape, appea = _f32(pe), _f32(shape), _f32(appea)
    audio = _f32(audio) if audio is not None and np.asarray(audio).size else None
    B = pe.shape[0]
    M = int(np.prod(pe.shape[2:]))
    wl, keep = _ptr_array(mlp_weight_list(sd, prefix))
    rgb = np.empty((B, C) + pe.shape[2:], np.float32)
    dens = np.empty((B, 1) + pe.shape[2:], np.float32)
    vd = _f32(vd) if vd is not None else None
    lib().orc_mlp_vd(B, ctypes.c_long(M), H, C, shape.shape[1], appea.shape[1], 0 if audio is None else audio.shape[1],
                     0 if vd is None else vd.shape[1], wl, _p(pe), _p(shape), _p(appea), _p(audio), _p(vd), _p(rgb), _p(dens))
    return rgb, dens


def composite(rgb, density, z_dists, zvals):
    rgb, density, z_dists, zvals = _f32(rgb), _f32(density), _f32(z_dists), _f32(zvals)
    B, C, Nr, Ns = rgb.shape
    fg = np.empty((B, C, Nr), np.float32)
    ba = np.empty((B, 1, Nr), np.float32)
    dp = np.empty((B, 1, Nr), np.float32)
    w = np.empty((B, 1, Nr, Ns), np.float32)
    lib().orc_composite(B, Nr, Ns, C, _p(rgb), _p(density), _p(z_dists), _p(zvals), _p(fg), _p(ba), _p(dp), _p(w))
    return fg, ba, dp, w


def neural_render(sd, x, n_blocks, debug=False):
    x = _f32(x)
    B, C, fs, _ = x.shape
    P = fs << n_blocks
    wl, keep = _ptr_array(nr_weight_list(sd, n_blocks))
    img = np.empty((B, 3, P, P), np.float32)
    dbg = [None, None, None]
    if debug:
        c1 = max(C // 2, 32)
        dbg = [np.empty((B, 3, 2 * fs, 2 * fs), np.float32), np.empty((B, C, 2 * fs, 2 * fs), np.float32),
               np.empty((B, c1, 2 * fs, 2 * fs), np.float32)]
    lib().orc_neural_render(B, C, fs, n_blocks, wl, _p(x), _p(img), _p(dbg[0]), _p(dbg[1]), _p(dbg[2]))
    return (img, dbg) if debug else img


def blur(x):
    x = _f32(x)
    B, C, h, w = x.shape
    y = np.empty_like(x)
    lib().orc_blur(B * C, h, w, _p(x), _p(y))
    return y


def forward(sd, opt, inp, t_rand=None, skip_neural_render=False, include_vd=False):
    """Whole path on the CPU.  inp: dict with the reference's kwarg names (numpy or torch).  include_vd: the module built with
    include_vd=True (RGB_layer_1 takes the 27-channel encoding of the ray direction, NetWorks/HeadNeRFNet.py:56-63)."""
    import math
    xy = _f32(_np(inp["batch_xy"]))
    B, _, Nr = xy.shape
    fs, C, H = opt.featmap_size, opt.featmap_nc, opt.mlp_hidden_nchannels
    nb = int(math.log2(opt.pred_img_size) - math.log2(fs))
    P = opt.pred_img_size
    shape, appea = _f32(_np(inp["shape_code"])), _f32(_np(inp["appea_code"]))
    audio = inp.get("audiostyle")
    audio = _f32(_np(audio)) if audio is not None and _np(audio).size else None
    mw, k1 = _ptr_array(mlp_weight_list(sd))
    nw, k2 = _ptr_array(nr_weight_list(sd, nb))
    bgf = _f32(_np(sd["neural_render.bg_featmap"]))
    R, T, K = _f32(_np(inp["batch_Rmats"])), _f32(_np(inp["batch_Tvecs"])).reshape(-1, 3), _f32(_np(inp["batch_inv_inmats"]))
    tr = _f32(_np(t_rand)) if t_rand is not None else None
    fg = np.empty((B, C, Nr), np.float32)
    ba = np.empty((B, 1, Nr), np.float32)
    merge_img = np.empty((B, 3, P, P), np.float32)
    bg_img = np.empty((1, 3, P, P), np.float32)
    lib().orc_forward_vd(B, Nr, opt.num_sample_coarse, fs, nb, H, C, shape.shape[1], appea.shape[1],
                      0 if audio is None else audio.shape[1], 27 if include_vd else 0, mw, nw, _p(bgf), _p(xy), _p(R), _p(T), _p(K),
                      ctypes.c_float(opt.world_z1), ctypes.c_float(opt.world_z2), _p(tr), _p(shape), _p(appea), _p(audio),
                      _p(fg), _p(ba), _p(merge_img), _p(bg_img), int(skip_neural_render))
    return {"fg_feat": fg, "bg_alpha": ba, "merge_img": merge_img, "bg_img": bg_img}


def embed_dirs(ray_d, n_samples, n_freqs=4):
    """vd_encoder(fg_dirs) (HeadNeRFNet.py:141-142): ray_d [B,3,Nr] -> [B, 3 + 6 n_freqs, Nr, n_samples] (numpy fp32, the encoder's
    channel order of NetWorks/utils.py:20-51; the direction is the same at every sample of a ray)."""
    d = _f32(ray_d)
    feats = [d]
    for k in range(n_freqs):
        a = (d * np.float32(2.0 ** k)).astype(np.float32)
        feats += [np.sin(a, dtype=np.float32), np.cos(a, dtype=np.float32)]
    pe = np.concatenate(feats, axis=1)
    return np.ascontiguousarray(np.repeat(pe[:, :, :, None], n_samples, axis=3))


def forward_hier(sd, opt, inp, t_rand=None, fine_u=None, include_vd=False):
    """Coarse pass + hierarchical pass (HeadNeRFNet._forward with hier_sampling=True, the call at HeadNeRFNet.py:182-185
    completed with its two missing arguments).  Returns the fine planes and both passes' composited features / images."""
    import math
    xy = _f32(_np(inp["batch_xy"]))
    B, _, Nr = xy.shape
    fs, C = opt.featmap_size, opt.featmap_nc
    nb = int(math.log2(opt.pred_img_size) - math.log2(fs))
    shape, appea = _np(inp["shape_code"]), _np(inp["appea_code"])
    audio = _np(inp["audiostyle"]) if inp.get("audiostyle") is not None else None
    R, T, K = _np(inp["batch_Rmats"]), _np(inp["batch_Tvecs"]), _np(inp["batch_inv_inmats"])
    s = sample(xy, R, T, K, opt.num_sample_coarse, opt.world_z1, opt.world_z2, None if t_rand is None else _np(t_rand))
    vd_c = embed_dirs(s["ray_d"], s["pts"].shape[-1]) if include_vd else None
    rgb, dens = mlp(sd, embed(s["pts"]), shape, appea, audio, C, opt.mlp_hidden_nchannels, vd=vd_c)
    cfg, cba, _, cw = composite(rgb, dens, s["z_dists"], s["zvals"])
    planes = fine_sample(cw, s["zvals"], opt.num_sample_fine, None if fine_u is None else _np(fine_u))
    f = sample_planes(xy, R, T, K, planes)
    vd_f = embed_dirs(s["ray_d"], f["pts"].shape[-1]) if include_vd else None
    rgb, dens = mlp(sd, embed(f["pts"]), shape, appea, audio, C, opt.mlp_hidden_nchannels, prefix="fine_fg_CD_predictor.", vd=vd_f)
    ffg, fba, _, fw = composite(rgb, dens, f["z_dists"], f["zvals"])
    bg = _f32(_np(sd["neural_render.bg_featmap"]))
    out = {"planes": planes, "coarse_weight": cw, "coarse_fg": cfg, "fine_fg": ffg, "fine_bg_alpha": fba, "fine_weight": fw}
    for name, fg, ba in (("coarse", cfg, cba), ("fine", ffg, fba)):
        merge = fg.reshape(B, C, fs, fs) + ba.reshape(B, 1, fs, fs) * bg
        out[name + "_merge_img"] = neural_render(sd, merge, nb)
    return out
