"""ctypes binding of libn3dt.so (C ABI: include/n3dt.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing
this module's `lib()` raises.  Build it with `make -C nerf-3dtalker-code_amd` (or
`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("N3DT_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libn3dt.so")

F32, BF16, F16, BF16X3 = 0, 1, 2, 3
PRECISIONS = {"fp32": F32, "f32": F32, "bf16": BF16, "fp16": F16, "f16": F16, "bf16x3": BF16X3}
MLP_LAYERS = 12
MAX_BLOCKS = 8

MLP_ORDER = ["FeaExt_module_%d" % i for i in range(8)] + ["density_module", "RGB_layer_0", "RGB_layer_1", "RGB_layer_2"]


class Geom(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32), ("n_rays", ctypes.c_int32), ("n_samples", ctypes.c_int32),
        ("hidden", ctypes.c_int32), ("feat_nc", ctypes.c_int32), ("shape_dim", ctypes.c_int32),
        ("appea_dim", ctypes.c_int32), ("audio_dim", ctypes.c_int32), ("featmap_size", ctypes.c_int32),
        ("n_blocks", ctypes.c_int32), ("world_z1", ctypes.c_float), ("world_z2", ctypes.c_float),
        ("xy_stride_b", ctypes.c_int64), ("xy_stride_c", ctypes.c_int64), ("xy_stride_r", ctypes.c_int64),
        ("z_planes_given", ctypes.c_int32), ("bg_is_hwc", ctypes.c_int32), ("vd_dim", ctypes.c_int32),
    ]


class MlpParams(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p * MLP_LAYERS), ("bias", ctypes.c_void_p * MLP_LAYERS)]


class RenderParams(ctypes.Structure):
    _fields_ = [
        ("to_rgb_w", ctypes.c_void_p * (MAX_BLOCKS + 1)), ("to_rgb_b", ctypes.c_void_p * (MAX_BLOCKS + 1)),
        ("psu1_w", ctypes.c_void_p * MAX_BLOCKS), ("psu1_b", ctypes.c_void_p * MAX_BLOCKS),
        ("psu2_w", ctypes.c_void_p * MAX_BLOCKS), ("psu2_b", ctypes.c_void_p * MAX_BLOCKS),
        ("feat_w", ctypes.c_void_p * MAX_BLOCKS), ("feat_b", ctypes.c_void_p * MAX_BLOCKS),
    ]


EXPORTS = [
    "n3dt_abi_version", "n3dt_last_error", "n3dt_mlp_packed_bytes", "n3dt_mlp_pack",
    "n3dt_render_workspace_bytes", "n3dt_render_fwd", "n3dt_neural_render_workspace_bytes",
    "n3dt_neural_render_fwd", "n3dt_chw_to_hwc", "n3dt_prof_enable", "n3dt_prof_collect",
    "n3dt_render_train_saved_bytes", "n3dt_render_train_workspace_bytes", "n3dt_render_train_fwd", "n3dt_render_bwd",
    "n3dt_neural_render_train_saved_bytes", "n3dt_neural_render_train_workspace_bytes",
    "n3dt_neural_render_train_fwd", "n3dt_neural_render_bwd", "n3dt_loss_fwd", "n3dt_loss_bwd", "n3dt_fine_sample",
    "n3dt_img_to_uint8", "n3dt_sample_points", "n3dt_embed", "n3dt_mlp_points_workspace_bytes", "n3dt_mlp_points", "n3dt_composite",
    "n3dt_ray_vd_bias", "n3dt_embed_freqs",
    "n3dt_neural_render_pack", "n3dt_neural_render_fwd_reuse", "n3dt_stage_inputs", "n3dt_graph_begin", "n3dt_graph_end", "n3dt_graph_launch", "n3dt_graph_destroy",
]

STAGE_MAX = 12


class Stage(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p * STAGE_MAX), ("dst", ctypes.c_void_p * STAGE_MAX), ("count", ctypes.c_int64 * STAGE_MAX),
                ("view_dims", ctypes.c_int64 * 3), ("view_strides", ctypes.c_int64 * 3), ("n", ctypes.c_int32)]


_LIB = None


class N3dtError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise N3dtError("libn3dt.so not found at %s -- the HIP extension is required (no CPU fallback); "
                        "run `make -C nerf-3dtalker-code_amd`" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.n3dt_abi_version.restype = ci
    L.n3dt_last_error.restype = ctypes.c_char_p
    L.n3dt_mlp_packed_bytes.restype = sz
    L.n3dt_mlp_packed_bytes.argtypes = [ctypes.POINTER(Geom), ci]
    L.n3dt_mlp_pack.restype = ci
    L.n3dt_mlp_pack.argtypes = [ctypes.POINTER(Geom), ci, ctypes.POINTER(MlpParams), vp, vp]
    L.n3dt_render_workspace_bytes.restype = sz
    L.n3dt_render_workspace_bytes.argtypes = [ctypes.POINTER(Geom), ci]
    L.n3dt_render_fwd.restype = ci
    L.n3dt_render_fwd.argtypes = [ctypes.POINTER(Geom), ci, vp, ctypes.POINTER(MlpParams)] + [vp] * 15 + [vp, sz, vp]
    L.n3dt_neural_render_workspace_bytes.restype = sz
    L.n3dt_neural_render_workspace_bytes.argtypes = [ctypes.POINTER(Geom), ci]
    L.n3dt_neural_render_fwd.restype = ci
    L.n3dt_neural_render_fwd.argtypes = [ctypes.POINTER(Geom), ci, ci, ctypes.POINTER(RenderParams), vp, vp, vp, sz, vp]
    L.n3dt_neural_render_fwd_reuse.restype = ci
    L.n3dt_neural_render_fwd_reuse.argtypes = [ctypes.POINTER(Geom), ci, ci, ctypes.POINTER(RenderParams), vp, vp, vp, sz, vp]
    L.n3dt_neural_render_pack.restype = ci
    L.n3dt_neural_render_pack.argtypes = [ctypes.POINTER(Geom), ci, ci, ctypes.POINTER(RenderParams), vp, sz, vp]
    L.n3dt_chw_to_hwc.restype = ci
    L.n3dt_chw_to_hwc.argtypes = [ci, ci, vp, vp, vp]
    gp, mp, rp = ctypes.POINTER(Geom), ctypes.POINTER(MlpParams), ctypes.POINTER(RenderParams)
    L.n3dt_render_train_saved_bytes.restype = sz
    L.n3dt_render_train_saved_bytes.argtypes = [gp]
    L.n3dt_render_train_workspace_bytes.restype = sz
    L.n3dt_render_train_workspace_bytes.argtypes = [gp]
    L.n3dt_render_train_fwd.restype = ci
    L.n3dt_render_train_fwd.argtypes = [gp, ci, vp, mp] + [vp] * 14 + [vp, sz, vp, sz, vp]
    L.n3dt_render_bwd.restype = ci
    L.n3dt_render_bwd.argtypes = [gp, ci, mp, mp] + [vp] * 7 + [vp, sz] + [vp] * 12 + [vp, sz, vp]
    L.n3dt_neural_render_train_saved_bytes.restype = sz
    L.n3dt_neural_render_train_saved_bytes.argtypes = [gp, ci]
    L.n3dt_neural_render_train_workspace_bytes.restype = sz
    L.n3dt_neural_render_train_workspace_bytes.argtypes = [gp, ci]
    L.n3dt_neural_render_train_fwd.restype = ci
    L.n3dt_neural_render_train_fwd.argtypes = [gp, ci, ci, rp, vp, vp, vp, sz, vp, sz, vp]
    L.n3dt_neural_render_bwd.restype = ci
    L.n3dt_neural_render_bwd.argtypes = [gp, ci, ci, rp, rp, vp, vp, vp, sz, vp, vp, sz, vp]
    L.n3dt_loss_fwd.restype = ci
    L.n3dt_loss_fwd.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp]
    L.n3dt_loss_bwd.restype = ci
    L.n3dt_loss_bwd.argtypes = [ci, ci, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, vp, vp]
    L.n3dt_prof_enable.restype = ci
    L.n3dt_prof_enable.argtypes = [ci]
    L.n3dt_prof_collect.restype = ci
    L.n3dt_prof_collect.argtypes = [ctypes.POINTER(ctypes.c_float), ci, ctypes.POINTER(ci)]
    L.n3dt_fine_sample.restype = ci
    L.n3dt_fine_sample.argtypes = [gp, ci, vp, vp, vp, vp, vp, vp]
    L.n3dt_img_to_uint8.restype = ci
    L.n3dt_img_to_uint8.argtypes = [ci, ci, vp, vp, vp]
    L.n3dt_sample_points.restype = ci
    L.n3dt_sample_points.argtypes = [gp] + [vp] * 10 + [vp]
    L.n3dt_embed.restype = ci
    L.n3dt_embed.argtypes = [ci, sz, vp, vp, vp]
    L.n3dt_embed_freqs.restype = ci
    L.n3dt_embed_freqs.argtypes = [ci, sz, ci, vp, vp, vp]
    L.n3dt_ray_vd_bias.restype = ci
    L.n3dt_ray_vd_bias.argtypes = [gp, vp, ctypes.c_int64, vp, vp, vp, vp, vp]
    L.n3dt_mlp_points_workspace_bytes.restype = sz
    L.n3dt_mlp_points_workspace_bytes.argtypes = [gp, sz]
    L.n3dt_mlp_points.restype = ci
    L.n3dt_mlp_points.argtypes = [gp, sz, mp] + [vp] * 5 + [vp, sz, vp]
    L.n3dt_composite.restype = ci
    L.n3dt_composite.argtypes = [ci, ci, ci, ci] + [vp] * 8 + [vp]
    L.n3dt_stage_inputs.restype = ci
    L.n3dt_stage_inputs.argtypes = [ctypes.POINTER(Stage), vp]
    L.n3dt_graph_begin.restype = ci
    L.n3dt_graph_begin.argtypes = [vp]
    L.n3dt_graph_end.restype = ci
    L.n3dt_graph_end.argtypes = [vp, ctypes.POINTER(vp)]
    L.n3dt_graph_launch.restype = ci
    L.n3dt_graph_launch.argtypes = [vp, vp]
    L.n3dt_graph_destroy.restype = ci
    L.n3dt_graph_destroy.argtypes = [vp]
    if L.n3dt_abi_version() != 5:
        raise N3dtError("libn3dt.so ABI version mismatch")
    _LIB = L
    return L


PROF_ACTIVE = False


def prof_enable(max_records):
    """The measurement hook of bench.py (n3dt_prof_enable).  While it is active forward() does not replay hipGraphs:
    the hook's events are recorded by the launching call, which a replay never executes."""
    global PROF_ACTIVE
    check(lib().n3dt_prof_enable(int(max_records)), "n3dt_prof_enable")
    PROF_ACTIVE = max_records > 0


def check(rc, what):
    if rc != 0:
        raise N3dtError("%s failed (%d): %s" % (what, rc, lib().n3dt_last_error().decode()))
