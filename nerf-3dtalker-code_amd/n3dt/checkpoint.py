"""Checkpoint format of the reference trainer (talker_trainer.py:557-567,736-749,913-937,1154-1165).

A checkpoint is a dict {"para": {featmap_size, featmap_nc, pred_img_size}, "net": state_dict, ...optimizer and
epoch entries...}; `net` uses exactly the keys/shapes of n3dt.HeadNeRFNet (buffers `...blur_layer.f`,
`neural_render.rgb_upsample.1.f` included), so real `model_Reso*.pth` files load here unchanged.
"""
import torch

from .headnerf import HeadNeRFNet
from .options import BaseOptions


def load_ckpt(model, state_dict):
    """Size-tolerant copy, as the trainer's load_ckpt: copy every entry whose name and shape match, report the rest."""
    own = model.state_dict()
    skipped = []
    for k, v in state_dict.items():
        if k in own and tuple(own[k].shape) == tuple(v.shape):
            own[k].copy_(v)
        else:
            skipped.append(k)
    for m in model.modules():  # packed weight copies follow version counters; be explicit anyway
        if hasattr(m, "invalidate_packed"):
            m.invalidate_packed()
    return skipped


def extend_for_gaze(state_dict, eye_gaze_dim):
    """Zero-pad the input columns of FeaExt_module_0/5 when eye-gaze features are appended to the shape code
    (talker_trainer.py:736-746)."""
    out = dict(state_dict)
    for key in ("fg_CD_predictor.FeaExt_module_5.weight", "fg_CD_predictor.FeaExt_module_0.weight"):
        w = out[key]
        r = w.shape[0]
        out[key] = torch.cat((w, torch.zeros((r, eye_gaze_dim, 1, 1), dtype=w.dtype)), 1)
    return out


def save_checkpoint(path, net, opt, epoch=0, optimizer=None, scheduler=None, extra=None):
    state = {"epoch": epoch, "net": net.state_dict(), "para": opt.para() if hasattr(opt, "para") else {
        "featmap_size": opt.featmap_size, "featmap_nc": opt.featmap_nc, "pred_img_size": opt.pred_img_size}}
    if optimizer is not None:
        state["optim_state"] = optimizer.state_dict()
    if scheduler is not None:
        state["scheule_state"] = scheduler.state_dict()  # (sic) the reference's key
    if extra:
        state.update(extra)
    torch.save(state, path)
    return state


def build_from_checkpoint(path_or_dict, include_gaze=False, eye_gaze_dim=2, strict=True, **net_kwargs):
    """`para` -> BaseOptions -> HeadNeRFNet -> weights, as talker_trainer.py:687-699 / FittingSingleImage_new.py:647-652."""
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, str) else path_or_dict
    opt = BaseOptions(ck["para"])
    net = HeadNeRFNet(opt, include_vd=False, hier_sampling=False, include_gaze=include_gaze, eye_gaze_dim=eye_gaze_dim,
                      **net_kwargs)
    sd = ck["net"]
    k0 = "fg_CD_predictor.FeaExt_module_0.weight"
    if include_gaze and sd[k0].shape[1] != net.state_dict()[k0].shape[1]:
        sd = extend_for_gaze(sd, eye_gaze_dim)
    if strict:
        net.load_state_dict(sd, strict=True)
    else:
        load_ckpt(net, sd)
    return net, opt
