"""HeadNeRFNet: the host-side mirror of the reference module, running on libn3dt.so.

Same constructor, same forward() signature / keyword names, same result dict and the same
state-dict keys and shapes as the reference (NetWorks/HeadNeRFNet.py:10-207), so
`talker_trainer.py` needs only its import line changed.  Everything between the
inputs and the two images is computed by hand-written gfx950 kernels behind the C ABI of
include/n3dt.h; there is no eager-PyTorch or CPU fallback -- a missing library or a CPU tensor
raises.

Sub-modules keep the reference's attribute names (sample_func, vp_encoder, fg_CD_predictor,
calc_color_func, neural_render) so the inner seams stay addressable.
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .parallel import _graph_task_id as parallel_graph_task_id


class _Conv1x1(nn.Module):
    """Parameter holder with nn.Conv2d(k=1) state-dict keys/shapes ([out,in,1,1], [out])."""

    def __init__(self, cin, cout, w_init, b_init="default_b"):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))
        self.bias = nn.Parameter(torch.empty(cout))
        self.reset(w_init, b_init)

    def reset(self, w_init, b_init):
        cout, cin = self.weight.shape[0], self.weight.shape[1]
        with torch.no_grad():
            if w_init == "xavier":  # reference: models.py:8-11 (_xavier_init)
                a = math.sqrt(6.0 / (cin + cout))
            else:  # nn.Conv2d default, kaiming_uniform(a=sqrt(5))
                a = 1.0 / math.sqrt(cin)
            self.weight.uniform_(-a, a)
            if b_init == "zero":
                self.bias.zero_()
            else:
                self.bias.uniform_(-1.0 / math.sqrt(cin), 1.0 / math.sqrt(cin))

    def w2d(self):
        return self.weight.view(self.weight.shape[0], self.weight.shape[1])


class MLPforNeRF(nn.Module):
    """Parameters of the latent-conditioned MLP (reference: NetWorks/models.py:13-59)."""

    def __init__(self, vp_channels, vd_channels, n_layers=8, h_channel=256, res_nfeat=3, audio_dim=64):
        super().__init__()
        assert n_layers == 8, "the fused kernels are built for the 8-layer trunk"
        self.vp_channels, self.vd_channels = vp_channels, vd_channels
        self.h_channel, self.res_nfeat, self.audio_dim = h_channel, res_nfeat, audio_dim
        self.n_layers = n_layers
        self.skips = [n_layers // 2]
        self.add_module("FeaExt_module_0", _Conv1x1(vp_channels + audio_dim, h_channel, "default_w"))
        for i in range(n_layers - 1):
            cin = h_channel + vp_channels if i in self.skips else h_channel
            self.add_module("FeaExt_module_%d" % (i + 1), _Conv1x1(cin, h_channel, "xavier"))
        self.add_module("density_module", _Conv1x1(h_channel, 1, "xavier", "zero"))
        self.add_module("RGB_layer_0", _Conv1x1(h_channel, h_channel, "xavier"))
        self.add_module("RGB_layer_1", _Conv1x1(h_channel + vd_channels, h_channel // 2, "default_w"))
        self.add_module("RGB_layer_2", _Conv1x1(h_channel // 2, res_nfeat, "default_w"))

    def layers(self):
        return [self._modules[n] for n in _lib.MLP_ORDER]

    @torch.no_grad()
    def forward(self, audiostyle, batch_embed_vps, batch_embed_vds):
        """The reference's stand-alone seam (NetWorks/models.py:62-87): inputs [B,C,N_r,N_s] -> (rgb, density).
        Unfused, exact fp32, inference only -- HeadNeRFNet.forward() never materialises these tensors."""
        B = batch_embed_vps.shape[0]
        ws = [m.w2d().detach() for m in self.layers()]
        bs = [m.bias.detach() for m in self.layers()]
        geom = ops.make_geom(B, 1, 1, self.h_channel, self.res_nfeat, self.vp_channels - 63, self.vd_channels, self.audio_dim, 2, 1,
                             0.0, 0.0)
        return ops.mlp_points(geom, ops.mlp_params(ws, bs), audiostyle if self.audio_dim > 0 else None, batch_embed_vps,
                              batch_embed_vds)


class Blur(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("f", torch.tensor([1.0, 2.0, 1.0]))


class PixelShuffleUpsample(nn.Module):
    """Parameters of one upsample block (reference: NetWorks/PixelShuffleUpsample.py:21-33)."""

    def __init__(self, in_feature):
        super().__init__()
        self.in_feature = in_feature
        self.layer_1 = _Conv1x1(in_feature, in_feature * 2, "default_w")
        self.layer_2 = _Conv1x1(in_feature * 2, in_feature * 4, "default_w")
        self.blur_layer = Blur()


class NeuralRenderer(nn.Module):
    """2-D renderer, feature map -> RGB (reference: NetWorks/neural_renderer.py:11-91)."""

    def __init__(self, bg_type="white", feat_nc=256, out_dim=3, final_actvn=True, min_feat=32, featmap_size=32,
                 img_size=256, **kwargs):
        super().__init__()
        assert out_dim == 3 and final_actvn and min_feat == 32
        self.bg_type = bg_type
        self._packed_sig = {}  # workspace address -> (parameter versions, nb, precision) its packed block weights were made from
        self._own_ws = {}
        self.train_precision = "fp32"  # "bf16": matrix products of the differentiable path on bf16 MFMA
        self._arena_owner = None        # the HeadNeRFNet whose gradient arena this renderer's backward writes into
        self.featmap_size = featmap_size
        self.n_feat = feat_nc
        self.out_dim = out_dim
        self.n_blocks = int(math.log2(img_size) - math.log2(featmap_size))
        self.min_feat = min_feat
        nf, nb = feat_nc, self.n_blocks
        if bg_type == "white":
            bg = torch.ones((1, nf, featmap_size, featmap_size), dtype=torch.float32)
        elif bg_type == "black":
            bg = torch.zeros((1, nf, featmap_size, featmap_size), dtype=torch.float32)
        else:
            raise ValueError("Error bg_type")  # the reference prints and exit(0)s here (neural_renderer.py:37-40)
        self.register_parameter("bg_featmap", nn.Parameter(bg))
        self.feat_upsample_list = nn.ModuleList([PixelShuffleUpsample(max(nf // (2 ** i), min_feat)) for i in range(nb)])
        self.rgb_upsample = nn.ModuleList([nn.Identity(), Blur()])  # keys: rgb_upsample.1.f
        self.feat_2_rgb_list = nn.ModuleList(
            [_Conv1x1(nf, out_dim, "default_w")] +
            [_Conv1x1(max(nf // (2 ** (i + 1)), min_feat), out_dim, "default_w") for i in range(nb)])
        self.feat_layers = nn.ModuleList(
            [_Conv1x1(max(nf // (2 ** i), min_feat), max(nf // (2 ** (i + 1)), min_feat), "default_w") for i in range(nb)])

    def get_bg_featmap(self):
        return self.bg_featmap

    def _geom(self, nb):
        return ops.make_geom(nb, self.featmap_size ** 2, 1, 384, self.n_feat, 1, 1, 0, self.featmap_size, self.n_blocks, 0, 0)

    def _rparams(self):
        def wb(m):
            return (m.w2d().detach().contiguous(), m.bias.detach().contiguous())
        return ops.render_params([wb(m) for m in self.feat_2_rgb_list], [wb(m.layer_1) for m in self.feat_upsample_list],
                                 [wb(m.layer_2) for m in self.feat_upsample_list], [wb(m) for m in self.feat_layers])

    def _flat_modules(self):
        """(weight, bias) holders in the order of _rparams_from: to_rgb[0..n], psu1[0..n-1], psu2[0..n-1], feat[0..n-1]"""
        return (list(self.feat_2_rgb_list) + [m.layer_1 for m in self.feat_upsample_list] +
                [m.layer_2 for m in self.feat_upsample_list] + list(self.feat_layers))

    def _rparams_from(self, tensors):
        """tensors: [w, b] pairs flattened in _flat_modules() order (2-D weights)"""
        n = self.n_blocks
        pairs = [(tensors[2 * i], tensors[2 * i + 1]) for i in range(len(tensors) // 2)]
        return ops.render_params(pairs[:n + 1], pairs[n + 1:2 * n + 1], pairs[2 * n + 1:3 * n + 1], pairs[3 * n + 1:])

    def render_hwc_train(self, featmap_hwc):
        """Differentiable [nb, fs, fs, C] -> [nb, 3, P, P] (exact fp32)."""
        flat = []
        for m in self._flat_modules():
            flat += [m.weight, m.bias]
        return _NeuralRenderFn.apply(self, None, 0, featmap_hwc, None, *flat)

    def _param_sig(self):
        return tuple((p.data_ptr(), p._version) for m in self._flat_modules() for p in (m.weight, m.bias))

    def ensure_packed(self, nb, precision, ws):
        """16-bit modes: the blocks' weights in MFMA order live in the tail of the workspace `ws`; (re-)pack them there
        when a parameter's version counter moved since this workspace was last packed.  Returns the parameter struct."""
        rp = self._rparams()
        prec = _lib.PRECISIONS[precision]
        if prec != _lib.F32:
            sig = (self._param_sig(), nb, prec)
            if self._packed_sig.get(ws.data_ptr()) != sig:
                ops.neural_render_pack(self._geom(nb), nb, rp, prec, ws)
                self._packed_sig[ws.data_ptr()] = sig
        return rp

    def invalidate_packed(self):
        self._packed_sig.clear()

    def render_hwc(self, featmap_hwc, precision="fp32", img=None, ws=None):
        """[nb, fs, fs, C] ray-major feature maps -> [nb, 3, P, P].  img / ws: caller-owned buffers (nothing is allocated
        then); without `ws` the module keeps one grow-only workspace per (device, stream), which also holds its packed weights."""
        nb = featmap_hwc.shape[0]
        assert featmap_hwc.is_contiguous()
        # the kernels index one map level with 32-bit offsets (nb x output pixels x 32 channels < 2^31: 255 maps at 512^2,
        # 63 at 1024^2); a larger batch -- a long novel-view sweep -- goes through in slices
        side = self.featmap_size << self.n_blocks
        cap = getattr(self, "_max_maps_per_call", None) or max(1, ((1 << 31) - 1) // (side * side * 32))
        if nb > cap:
            assert ws is None, "a caller-owned workspace is sized for one call"
            out = img if img is not None else torch.empty(nb, 3, side, side, dtype=torch.float32, device=featmap_hwc.device)
            for i in range(0, nb, cap):
                self.render_hwc(featmap_hwc[i:i + cap], precision, img=out[i:i + cap])
            return out
        geom = self._geom(nb)
        if ws is None:
            # ONE grow-only workspace per (device, stream), sized for the largest batch seen (a caller whose batch varies -- a
            # last partial batch, validation at 1 next to sweeps at 16 -- must not keep a workspace per size alive).  The
            # packed block weights sit at an offset that depends on nb; the pack signature carries nb, so a different nb re-packs.
            dev = featmap_hwc.device
            key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
            need = ops.neural_render_workspace_bytes(geom, nb)
            ws = self._own_ws.get(key)
            if ws is None or ws.numel() < need:
                if ws is not None:
                    self._packed_sig.pop(ws.data_ptr(), None)
                ws = self._own_ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
                self._packed_sig.pop(ws.data_ptr(), None)  # (a fresh allocation may sit where a released one sat)
        rp = self.ensure_packed(nb, precision, ws)
        return ops.neural_render_fwd(geom, nb, rp, featmap_hwc, _lib.PRECISIONS[precision], img=img, ws=ws, reuse_packed=True)

    def forward(self, x):
        """x: [nb, C, fs, fs] like the reference module."""
        nb, C, fs, _ = x.shape
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self.render_hwc_train(x.permute(0, 2, 3, 1).contiguous())
        hwc = torch.stack([ops.chw_to_hwc(x[i].contiguous(), C, fs * fs) for i in range(nb)]).view(nb, fs, fs, C)
        return self.render_hwc(hwc.contiguous())



def _zeros_like_many(tensors):
    """Zeroed gradient buffers for a list of parameter tensors as views of ONE allocation (one fill kernel instead of one
    per tensor: the training step launched 70 of them); every view starts on a 256-byte boundary."""
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += (t.numel() + 63) // 64 * 64
    flat = torch.zeros(total, dtype=tensors[0].dtype, device=tensors[0].device)
    return [flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, tensors)]


class _RenderFn(torch.autograd.Function):
    """a1..a7 with saved activations (n3dt_render_train_fwd) and its backward (n3dt_render_bwd).
    Inputs after `net`/`geom`: xy, Kinv, t_rand (no grad), then R, T, shape, appea, audio, bg_featmap, ray_bias and the
    24 MLP parameter tensors (all differentiable; R/T gradients are computed only when they require grad).
    ray_bias [B, N_r, 192] (include_vd, else None): the per-ray addend of RGB_layer_1's pre-activation; the caller forms it from
    the ray directions with autograd, so its gradient carries on to the 27 view-direction columns and to the cameras."""

    @staticmethod
    def forward(ctx, net, geom, merge_out, xy, Kinv, t_rand, R, T, shape, appea, audio, bg_featmap, ray_bias, *mlp):
        """merge_out: None, or a _Slot whose `.t` [B, N_r, C] (a slice of the renderer's input batch) receives the merged
        map and becomes the output.  (Handed over inside a plain object, not as a tensor argument: autograd then sees a fresh
        output, not a modified input.)"""
        ctx.want_cam = R.requires_grad or T.requires_grad
        ctx.T_shape = T.shape
        R = ops._f32c(R)
        T = ops._f32c(T).view(-1, 3)
        ws = [t.detach().view(t.shape[0], -1).contiguous() for t in mlp[:12]]
        bs = [t.detach().contiguous() for t in mlp[12:]]
        params = ops.mlp_params(ws, bs)
        # "bf16" = the fused mixed-precision path (bf16 MFMA products, fp32 parameters and gradients)
        ctx.prec = _lib.PRECISIONS[net.train_precision]
        # Weights under training change every step, and not every optimizer moves the version counters the packed-weight cache
        # follows: torch.optim.Adam(fused=True) updates the parameters WITHOUT bumping `_version` (seen: one pack in 13 steps).
        # So the differentiable path re-packs whenever a weight requires grad (two small launches, in place).
        packed = net._packed(geom, ctx.prec, params, ws, bs, force=any(t.requires_grad for t in mlp))
        shape_c, appea_c = ops._f32c(shape), ops._f32c(appea)
        audio_c = ops._f32c(audio) if geom.audio_dim > 0 else None
        bg = bg_featmap.detach().reshape(geom.feat_nc, -1).contiguous()
        out, saved = ops.render_train_fwd(geom, packed, params, xy, R, T, Kinv, shape_c, appea_c, audio_c, t_rand, bg, ctx.prec,
                                          merge_out=None if merge_out is None else merge_out.t,
                                          ray_bias=None if ray_bias is None else ops._f32c(ray_bias))
        ctx.geom, ctx.saved, ctx.keep = geom, saved, (ws, bs, shape_c, appea_c, audio_c, bg)
        ctx.cam = (xy, R, T, Kinv, t_rand) if ctx.want_cam else None
        ctx.bg_shape = bg_featmap.shape
        ctx.mlp_shapes = [t.shape for t in mlp]
        ctx.net, ctx.param_objs = net, list(mlp) + [bg_featmap]
        return out["merge_feat"]

    @staticmethod
    def backward(ctx, d_merge):
        ws, bs, shape_c, appea_c, audio_c, bg = ctx.keep
        geom = ctx.geom
        # gradient buffers: slices of the module's persistent arena (zeroed by one fill per backward pass, and what a
        # multi-GPU step all-reduces in place) -- or fresh zeroed buffers when the arena cannot be used (see FlatGrads.hand_out)
        vd = geom.vd_dim > 0
        if not (any(ctx.needs_input_grad[13:]) or ctx.needs_input_grad[11]):
            # FROZEN network (single-image fitting optimises codes and cameras only, FittingSingleImage_new.py:826-859): no
            # parameter gradient is wanted, so none is computed -- the weight-gradient stage is a quarter of a fitting iteration
            res = ops.render_bwd(geom, ops.mlp_params(ws, bs), None, shape_c, appea_c, audio_c, bg,
                                 d_merge.contiguous(), ctx.saved, ctx.cam, ctx.prec, frozen=True)
            _, d_shape, d_appea, d_audio, d_R, d_T = res[:6]
            ctx.saved = None
            if d_T is not None:
                d_T = d_T.view(ctx.T_shape)
            return (None, None, None, None, None, None, d_R, d_T, d_shape, d_appea, d_audio, None, res[6] if vd else None,
                    *([None] * len(ctx.mlp_shapes)))
        views = ctx.net._hand_out_grads(ctx.param_objs)
        if views is None:
            gws, gbs, d_bg_out = _zeros_like_many(ws), _zeros_like_many(bs), None
        else:
            gws = [v.view(w.shape) for v, w in zip(views[:12], ws)]
            gbs, d_bg_out = views[12:24], views[24].view(geom.feat_nc, geom.n_rays)
        res = ops.render_bwd(geom, ops.mlp_params(ws, bs), ops.mlp_params(gws, gbs), shape_c, appea_c, audio_c, bg, d_merge.contiguous(),
                             ctx.saved, ctx.cam, ctx.prec, d_bg=d_bg_out)
        d_bg, d_shape, d_appea, d_audio, d_R, d_T = res[:6]
        d_ray = res[6] if vd else None
        ctx.saved = None
        grads = [g.view(s) for g, s in zip(gws + gbs, ctx.mlp_shapes)]
        del gws, gbs, views, d_bg_out  # the returned views must be the only references (autograd then adopts them as .grad)
        if any(ctx.needs_input_grad[13:]) or ctx.needs_input_grad[11]:
            # parameter gradients went out: an optimizer step follows, possibly one that leaves the version counters alone
            # (fused Adam) -- void every packed / transposed copy so the next forward of any kind rebuilds them
            ctx.net.invalidate_packed()
        if d_T is not None:
            d_T = d_T.view(ctx.T_shape)
        return (None, None, None, None, None, None, d_R, d_T, d_shape, d_appea, d_audio, d_bg.view(ctx.bg_shape), d_ray, *grads)


class _Slot:
    """A tensor handed to an autograd Function outside its tensor arguments (see _RenderFn.forward)."""

    def __init__(self, t):
        self.t = t


def _adjacent(parts):
    """parts: contiguous tensors (or None).  The one tensor [sum of leading dims, ...] they form when they are consecutive
    slices of one allocation, else None."""
    if any(p is None or not p.is_contiguous() for p in parts):
        return None
    p0 = parts[0]
    off = p0.storage_offset()
    for p in parts:
        if p.untyped_storage().data_ptr() != p0.untyped_storage().data_ptr() or p.storage_offset() != off or \
                p.shape[1:] != p0.shape[1:] or p.dtype != p0.dtype:
            return None
        off += p.numel()
    n = sum(p.shape[0] for p in parts)
    return torch.as_strided(p0, (n,) + tuple(p0.shape[1:]), p0.stride(), p0.storage_offset())


class _NeuralRenderFn(torch.autograd.Function):
    """a8..a10 with saved activations and the hand-written backward (n3dt_neural_render_bwd).

    Two call shapes.  (1) `apply(nr, None, 0, featmap, None, *flat)`: featmap [nb, fs, fs, C] -> img [nb, 3, P, P] (the
    module's own forward()).  (2) HeadNeRFNet's training step, `apply(nr, maps, n_split, merged, bg_featmap, *flat)`: `maps`
    (a _Slot) is the renderer's whole input batch [nb + 1, fs, fs, C]; its first nb maps were written in place by the
    _RenderFn calls whose outputs `merged` (a tuple of [B, N_r, C] slices of it, concatenated by construction) are; the last
    slot is filled here from bg_featmap [1, C, fs, fs] (one transposing launch).  Outputs: the merged images and the
    background image as separate tensors (slices of one buffer) -- no torch.cat on the way in, no slice-backward
    zeros + copy + add on the way out."""

    @staticmethod
    def forward(ctx, nr, maps, n_merged, featmap, bg_featmap, *rest):
        merged, flat = rest[:n_merged], rest[n_merged:]
        if maps is None:
            fm = featmap.detach().contiguous()
            nb = fm.shape[0]
        else:
            fm = maps.t
            nb = fm.shape[0]
            fs, C = nr.featmap_size, nr.n_feat
            off = fm.storage_offset()
            for m in merged:  # the slices _RenderFn wrote
                assert m.untyped_storage().data_ptr() == fm.untyped_storage().data_ptr() and m.storage_offset() == off
                off += m.numel()
            assert off == fm.storage_offset() + (nb - 1) * fs * fs * C
            ops.chw_to_hwc(bg_featmap.detach().reshape(C, fs * fs), C, fs * fs, fm[nb - 1].view(fs * fs, C))
        geom = nr._geom(nb)
        tensors = [t.detach().view(t.shape[0], -1).contiguous() if t.dim() == 4 else t.detach().contiguous() for t in flat]
        rp = nr._rparams_from(tensors)
        ctx.prec = _lib.PRECISIONS[nr.train_precision]
        img, saved = ops.neural_render_train_fwd(geom, nb, rp, fm, ctx.prec)
        ctx.nr, ctx.geom, ctx.nb, ctx.saved, ctx.keep = nr, geom, nb, saved, (tensors, fm)
        ctx.shapes = [t.shape for t in flat]
        ctx.param_objs = list(flat)
        ctx.split = None if maps is None else [m.shape[0] for m in merged]
        ctx.merged_shapes = [m.shape for m in merged]
        ctx.set_materialize_grads(False)
        if maps is None:
            return img
        outs, lo = [], 0
        for n in ctx.split:
            outs.append(img[lo:lo + n])
            lo += n
        return (*outs, img[lo:])

    @staticmethod
    def backward(ctx, *d_imgs):
        tensors, fm = ctx.keep
        nb = ctx.nb
        P = d_imgs[0].shape[-1] if d_imgs[0] is not None else (ctx.nr.featmap_size << ctx.nr.n_blocks)
        if ctx.split is None:
            d_img = d_imgs[0].contiguous()
        else:
            d_img = _adjacent(d_imgs)  # the fused loss tail returns them as consecutive slices of one buffer
            if d_img is None:
                sizes = ctx.split + [1]
                d_img = torch.cat([torch.zeros(n, 3, P, P, dtype=torch.float32, device=fm.device) if d is None else d.float()
                                   for d, n in zip(d_imgs, sizes)])
        n_lead = 5 + (0 if ctx.split is None else len(ctx.split))
        if not any(ctx.needs_input_grad[n_lead:]):  # frozen renderer (fitting): the input gradient only
            d_feat = ops.neural_render_bwd(ctx.geom, nb, ctx.nr._rparams_from(tensors), None, fm, d_img, ctx.saved, ctx.prec)
            grads = [None] * len(ctx.shapes)
        else:
            owner = ctx.nr._arena_owner
            views = owner._hand_out_grads(ctx.param_objs) if owner is not None else None
            gt = _zeros_like_many(list(tensors)) if views is None else [v.view(t.shape) for v, t in zip(views, tensors)]
            d_feat = ops.neural_render_bwd(ctx.geom, nb, ctx.nr._rparams_from(tensors), ctx.nr._rparams_from(gt), fm,
                                           d_img, ctx.saved, ctx.prec)
            grads = [g.view(s) for g, s in zip(gt, ctx.shapes)]
            del gt, views
        ctx.saved = None
        if ctx.split is None:
            return (None, None, None, d_feat, None, *grads)
        fs, C = ctx.nr.featmap_size, ctx.nr.n_feat
        d_merged, lo = [], 0
        for n, shp in zip(ctx.split, ctx.merged_shapes):
            d_merged.append(d_feat[lo:lo + n].view(shp))
            lo += n
        d_bg = None
        if ctx.needs_input_grad[4]:
            d_bg = ops.chw_to_hwc(d_feat[nb - 1].view(fs * fs, C), fs * fs, C).view(1, C, fs, fs)  # [N_r, C] -> [C, N_r]
        return (None, None, None, None, d_bg, *d_merged, *grads)


class FineSample(nn.Module):
    """fine_samp_func seam (NetWorks/utils.py:164-265) as a stand-alone operator with the reference's call:
    `fine_samp_func(batch_weight [B,1,N_r,N_c], coarse_sample_dict, disturb)` -> {"pts", "dirs", "zvals", "z_dists"} on
    the N_c + N_f sorted samples.  The planes come from n3dt_fine_sample (inverse CDF of the DETACHED interior weights,
    stable merge with the coarse planes), the points from n3dt_sample_points on those planes.  forward() of the module
    itself never materialises these tensors (HeadNeRFNet.fine_planes feeds the fused kernels directly)."""

    def __init__(self, opt):
        super().__init__()
        self.n_sample = opt.num_sample_fine + 1
        self.world_z1, self.world_z2 = opt.world_z1, opt.world_z2

    @torch.no_grad()
    def forward(self, batch_weight, coarse_sample_dict, disturb, fine_u=None):
        cz = coarse_sample_dict["zvals"]
        B, _, n_r, n_c = cz.shape
        dev = cz.device
        n_f = self.n_sample - 1
        # the operators work from the camera, as the fused path does: recover it from the dict's ray tensors.  The coarse
        # planes themselves are passed through (they are the dict's zvals plus the far edge the reference dropped).
        ray_o = coarse_sample_dict["batch_ray_o"]                    # [B,3,N_r,1] (= T broadcast)
        T = ops._f32c(ray_o[:, :, 0, 0])
        if disturb and fine_u is None:                               # the reference's torch.rand(num_temp, NFsample)
            fine_u = torch.rand(B * n_r, n_f + 1, device=dev, dtype=torch.float32)
        geom = ops.make_geom(B, n_r, n_c, 384, 256, 179, 127, 64, 2, 1, self.world_z1, self.world_z2, z_planes_given=1)
        coarse_planes = torch.nn.functional.pad(ops._f32c(cz).view(B, n_r, n_c), (0, 1))  # [B,N_r,N_c+1]; the far edge is not read
        planes = ops.fine_sample(geom, n_f, ops._f32c(batch_weight).view(B, n_r, n_c), T, t_rand=coarse_planes,
                                 u=None if fine_u is None else ops._f32c(fine_u))
        zv = planes[:, :, :-1].unsqueeze(1)
        ray_d, ray_l = coarse_sample_dict["batch_ray_d"], coarse_sample_dict["batch_ray_l"]
        z_dists = (planes[:, :, 1:] - planes[:, :, :-1]).unsqueeze(1) * ray_l
        pts = ray_o + ray_d * ray_l * zv
        return {"pts": pts, "dirs": ray_d.expand(-1, -1, -1, zv.size(-1)), "zvals": zv, "z_dists": z_dists}


class GenSamplePoints(nn.Module):
    """sample_func seam (NetWorks/utils.py:55-161) as a stand-alone operator: same call, same result dict."""

    def __init__(self, opt):
        super().__init__()
        self.world_z1, self.world_z2, self.n_sample_fg = opt.world_z1, opt.world_z2, opt.num_sample_coarse

    @torch.no_grad()
    def forward(self, batch_xy, batch_Rmat, batch_Tvec, batch_inv_inmat, disturb, t_rand=None):
        B, _, n_r = batch_xy.shape
        xy = batch_xy if batch_xy.dtype == torch.float32 else batch_xy.float()
        if disturb and t_rand is None:  # the reference's torch.rand_like(zvals) (utils.py:77)
            t_rand = torch.rand(B, n_r, self.n_sample_fg + 1, device=xy.device, dtype=torch.float32)
        geom = ops.make_geom(B, n_r, self.n_sample_fg, 384, 256, 179, 127, 64, 2, 1, self.world_z1, self.world_z2, xy.stride())
        T = ops._f32c(batch_Tvec).view(B, 3)
        o = ops.sample_points(geom, xy, ops._f32c(batch_Rmat), T, ops._f32c(batch_inv_inmat), None if t_rand is None else ops._f32c(t_rand))
        ray_d = o["ray_d"].unsqueeze(-1)
        return {"pts": o["pts"], "dirs": ray_d.expand(-1, -1, -1, self.n_sample_fg), "zvals": o["zvals"], "z_dists": o["z_dists"],
                "batch_ray_o": T.view(B, 3, 1, 1).expand(B, 3, n_r, 1), "batch_ray_d": ray_d, "batch_ray_l": o["ray_l"].unsqueeze(-1)}


class Embedder(nn.Module):
    """vp_encoder / vd_encoder seam (NetWorks/utils.py:6-51): [B,3,...] -> [B, 3 + 6 N_freqs, ...]."""

    def __init__(self, N_freqs=10, include_input=True):
        super().__init__()
        assert include_input, "the kernels are built with include_input=True (the reference's only setting)"
        self.N_freqs, self.include_input = N_freqs, include_input

    @torch.no_grad()
    def forward(self, x):
        # 10 frequencies: vp_encoder; 4: vd_encoder (HeadNeRFNet.py:27-31,54,61)
        return ops.embed(x) if self.N_freqs == 10 else ops.embed_freqs(x, self.N_freqs)


class CalcRayColor(nn.Module):
    """calc_color_func seam (NetWorks/utils.py:268-309): same call, same 4-tuple."""

    @torch.no_grad()
    def forward(self, fg_vps, batch_rgb, batch_density, batch_dists, batch_z_vals):
        return ops.composite(batch_rgb, batch_density, batch_dists, batch_z_vals)


class HeadNeRFNet(nn.Module):
    def __init__(self, opt, include_vd, hier_sampling, include_gaze=False, eye_gaze_dim=2, audio_dim=64, precision="fp32",
                 train_precision="fp32", use_graph=None, graph_static_outputs=False):
        super().__init__()
        # hier_sampling=True: the reference builds FineSample + a second MLP (HeadNeRFNet.py:67-74) but its call site omits
        # two arguments (:182-185) and raises TypeError; here the fine pass runs, with those arguments supplied
        # (inference and training, camera gradients included: test_gradients_through_the_hierarchical_pass_including_the_cameras)
        # include_vd=True (HeadNeRFNet.py:56-63,86,141-142; no caller of the reference sets it): RGB_layer_1 takes 27 more input
        # channels, the 4-frequency encoding of the ray direction.  The direction is constant along a ray, so those columns are a
        # per-ray bias of the layer (one 27-wide product per ray instead of per point): _vd_ray_bias / ops.ray_vd_bias.
        self.hier_sampling = hier_sampling
        self.include_vd = include_vd
        self.include_gaze = include_gaze
        self.eye_gaze_dim = eye_gaze_dim
        self.audio_dim = audio_dim
        self.precision = precision              # inference path: "fp32" (parity) | "bf16" | "fp16"
        assert train_precision in ("fp32", "bf16")
        self.train_precision = train_precision  # differentiable path: "fp32" (exact) | "bf16" (bf16-MFMA products)
        self._build_info(opt)
        self._build_tool_funcs()
        self.neural_render.train_precision = train_precision
        object.__setattr__(self.neural_render, "_arena_owner", self)  # (not a sub-module registration: it is the parent)
        self._pack_cache = {}
        # use_grad_arena: the backward kernels accumulate into slices of ONE persistent flat buffer (grad_arena()) and those
        # slices become the parameters' .grad -- one fill per step instead of an allocation + fill per tensor list, and the
        # buffer a data-parallel step all-reduces in place (n3dt/parallel.py).  A `.grad` tensor is therefore overwritten by
        # the next backward that follows a zero_grad(set_to_none=True): keep a clone, not a reference, to look at it later.
        self.use_grad_arena = True
        self._grad_arena = None
        # use_graph=True (or N3DT_GRAPH=1): mode="test" forwards are recorded once per call shape into a hipGraph and replayed
        # with one launch (see _forward_graph).  OFF by default: measured on MI355X / ROCm 7.2 a replay is 3-7 % SLOWER than the
        # stream-ordered launches it replaces (one head: 0.72-0.74 ms against 0.68-0.70; config 4: 5 570 against 5 980 frames/s)
        # -- the forward is GPU-bound even at one head, its launches were already hidden behind the kernels, and the replay adds
        # per-node scheduling.  It remains for hosts whose CPU is the bottleneck (the per-forward host cost drops to three calls).
        # graph_static_outputs=True returns views of the graph's own output buffer (valid until the next forward of the same
        # shape) instead of a copy.
        self.use_graph = (os.environ.get("N3DT_GRAPH", "0") == "1") if use_graph is None else bool(use_graph)
        self.graph_static_outputs = graph_static_outputs
        self._graphs = {}
        self._bg_cache = None
        self._maps_cache = {}
        self._w10c_cache = {}
        # a (strict or not) load_state_dict replaces every weight: drop the packed copies
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.invalidate_packed())

    def grad_arena(self):
        """The persistent flat gradient buffer over this module's trainable parameters (n3dt.parallel.FlatGrads), built on
        first use and rebuilt when the parameter list, a requires_grad flag or the device changed."""
        from . import parallel
        params = [p for p in self.parameters() if p.requires_grad]
        a = self._grad_arena
        if a is None or not a.matches(params):
            a = self._grad_arena = parallel.FlatGrads(params)
            for p in self.parameters():
                p._n3dt_arena = a if p.requires_grad else None
        return a

    def _hand_out_grads(self, params):
        if not self.use_grad_arena or not all(p.requires_grad for p in params):
            return None
        return self.grad_arena().hand_out(params)

    def _bg_hwc(self):
        """neural_render.bg_featmap [1,C,fs,fs] as the kernels read it, ray-major [fs*fs, C]: transposed once per parameter
        version into a buffer whose address never changes (recorded graphs read it)."""
        bg = self.neural_render.bg_featmap
        ver = (bg.data_ptr(), bg._version)
        c = self._bg_cache
        if c is None or c[0] != ver:
            fs, C = self.featmap_size, self.featmap_nc
            dst = c[1] if c is not None and c[1].device == bg.device else torch.empty(fs * fs, C, dtype=torch.float32, device=bg.device)
            ops.chw_to_hwc(bg.detach().view(C, fs * fs), C, fs * fs, dst)
            self._bg_cache = c = (ver, dst)
            self._bg_serial = getattr(self, "_bg_serial", 0) + 1  # moves with every re-transpose (what the batch buffers compare)
        return c[1]

    def invalidate_packed(self):
        """Forget the packed (MFMA-ordered, 16-bit) copies of the MLP weights.  They are rebuilt on the next call.
        The cache follows the parameters' version counters, which optimizers, `copy_` under no_grad and
        `load_state_dict` all move; writes through `.data` (the reference's own `load_ckpt` does
        `model.state_dict()[k].data.copy_(v)`, talker_trainer.py:557-567) do NOT move them -- call this after such a
        write, or use n3dt.checkpoint.load_ckpt, which does."""
        # the BUFFERS stay (recorded hipGraphs hold their addresses; the next call re-packs / re-transposes into them in
        # place, stream-ordered); only the versions they were made from are forgotten
        for k, (ver, buf) in list(self._pack_cache.items()):
            self._pack_cache[k] = (None, buf)
        if self._bg_cache is not None:
            self._bg_cache = (None, self._bg_cache[1])
        for k, (ver, buf) in list(self._w10c_cache.items()):
            self._w10c_cache[k] = (None, buf)
        for k, ent in list(self._maps_cache.items()):
            self._maps_cache[k] = (None,) + tuple(ent[1:])
        self.neural_render.invalidate_packed()
        for e in self._graphs.values():
            e["bg_ver"] = None

    def _build_info(self, opt):
        self.num_sample_coarse = opt.num_sample_coarse
        self.num_sample_fine = opt.num_sample_fine
        self.vp_n_freqs = 10
        self.include_input_for_vp_embeder = True
        self.vd_n_freqs = 4
        self.include_input_for_vd_embeder = True
        self.mlp_h_channel = opt.mlp_hidden_nchannels
        self.base_shape_code_dims = opt.iden_code_dims + opt.expr_code_dims
        self.base_appea_code_dims = opt.text_code_dims + opt.illu_code_dims
        self.featmap_size = opt.featmap_size
        self.featmap_nc = opt.featmap_nc
        self.pred_img_size = opt.pred_img_size
        self.opt = opt

    def _build_tool_funcs(self):
        vp_channels = self.base_shape_code_dims + self.vp_n_freqs * 6 + 3
        if self.include_gaze:
            vp_channels += self.eye_gaze_dim
        vd_channels = self.base_appea_code_dims
        if self.include_vd:
            vd_channels += self.vd_n_freqs * 6 + 3
            self.vd_encoder = Embedder(N_freqs=self.vd_n_freqs, include_input=self.include_input_for_vd_embeder)
        self.vp_encoder = Embedder(N_freqs=self.vp_n_freqs, include_input=self.include_input_for_vp_embeder)
        self.sample_func = GenSamplePoints(self.opt)
        self.fg_CD_predictor = MLPforNeRF(vp_channels=vp_channels, vd_channels=vd_channels, h_channel=self.mlp_h_channel,
                                          res_nfeat=self.featmap_nc, audio_dim=self.audio_dim)
        if self.hier_sampling:
            self.fine_samp_func = FineSample(self.opt)
            self.fine_fg_CD_predictor = MLPforNeRF(vp_channels=vp_channels, vd_channels=vd_channels, h_channel=self.mlp_h_channel,
                                                   res_nfeat=self.featmap_nc, audio_dim=self.audio_dim)
        self.calc_color_func = CalcRayColor()
        self.neural_render = NeuralRenderer(bg_type=self.opt.bg_type, feat_nc=self.featmap_nc, out_dim=3, final_actvn=True,
                                            min_feat=32, featmap_size=self.featmap_size, img_size=self.pred_img_size)

    # ------------------------------------------------------------------------------------------
    def _shape_dim(self):
        return self.base_shape_code_dims + (self.eye_gaze_dim if self.include_gaze else 0)

    def _geom(self, batch, n_rays, xy, n_samples=None, z_planes_given=0, bg_is_hwc=0):
        return ops.make_geom(batch, n_rays, n_samples or self.num_sample_coarse, self.mlp_h_channel, self.featmap_nc,
                             self._shape_dim(), self.base_appea_code_dims, self.audio_dim, self.featmap_size,
                             self.neural_render.n_blocks, self.opt.world_z1, self.opt.world_z2, xy.stride(), z_planes_given, bg_is_hwc,
                             vd_dim=self._vd_dim())

    def _vd_dim(self):
        return self.vd_n_freqs * 6 + 3 if self.include_vd else 0

    def _w10_compact(self, fine=False):
        """include_vd: RGB_layer_1's weight WITHOUT its 27 view-direction columns, [192, 384 + appea] contiguous -- what the
        library packs and folds; kept per parameter version in a buffer whose address does not change."""
        layer = (self.fine_fg_CD_predictor if fine else self.fg_CD_predictor).RGB_layer_1
        w = layer.w2d().detach()
        ver = (w.data_ptr(), layer.weight._version)
        hit = self._w10c_cache.get(fine)
        if hit is None or hit[0] != ver or hit[1].device != w.device:
            h, vd = self.mlp_h_channel, self._vd_dim()
            buf = hit[1] if hit is not None and hit[1].device == w.device else torch.empty(w.shape[0], w.shape[1] - vd, dtype=torch.float32, device=w.device)
            buf[:, :h].copy_(w[:, :h])
            buf[:, h:].copy_(w[:, h + vd:])
            hit = self._w10c_cache[fine] = (ver, buf)
        return hit[1]

    def _mlp_params(self, fine=False):
        layers = (self.fine_fg_CD_predictor if fine else self.fg_CD_predictor).layers()
        ws = [m.w2d().detach() for m in layers]
        bs = [m.bias.detach() for m in layers]
        if self.include_vd:
            ws[10] = self._w10_compact(fine)
        return ops.mlp_params(ws, bs), ws, bs

    def _vd_ray_bias(self, xy, batch_Rmats, batch_inv_inmats, fine=False):
        """include_vd, differentiable: ray_bias [B, N_r, 192] = Embedder_4(ray direction) . RGB_layer_1.weight[:, 384:411]^T with
        autograd (gradients reach the 27 weight columns and batch_Rmats).  The direction as GenSamplePoints builds it
        (NetWorks/utils.py:149-153), the encoder's channel order as Embedder (:20-51).  A [B*N_r, 27] x [27, 192] product: plumbing
        next to the sample-point work the kernels do; the inference path uses the HIP kernel (ops.ray_vd_bias)."""
        B, _, n_r = xy.shape
        ones = torch.ones(B, 1, n_r, dtype=torch.float32, device=xy.device)
        d = batch_Rmats.float().bmm(batch_inv_inmats.float().bmm(torch.cat([xy, ones], dim=1)))  # [B,3,N_r]
        d = d / torch.norm(d, dim=1, keepdim=True)
        feats = [d]
        for k in range(self.vd_n_freqs):
            feats += [torch.sin(d * float(2 ** k)), torch.cos(d * float(2 ** k))]
        pe = torch.cat(feats, dim=1)  # [B,27,N_r]
        h = self.mlp_h_channel
        w = (self.fine_fg_CD_predictor if fine else self.fg_CD_predictor).RGB_layer_1.weight
        w_vd = w[:, h:h + self._vd_dim(), 0, 0]  # [192,27]
        return torch.einsum("bjr,oj->bro", pe, w_vd).contiguous()

    def _packed(self, geom, precision, params, ws, bs, force=False):
        """Packed weights, re-packed whenever the optimizer (or a load) touched a parameter."""
        key = (precision, ws[0].device.index, ws[0].data_ptr())
        ver = tuple((t.data_ptr(), t._version) for t in ws + bs)
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != ver or force:
            # re-pack INTO the existing buffer (stream-ordered): recorded hipGraphs keep reading a valid address
            hit = (ver, ops.pack_mlp(geom, precision, params, ws[0].device, out=None if hit is None else hit[1]))
            self._pack_cache[key] = hit
            if len(self._pack_cache) > 16:  # stale keys (parameters re-allocated by .to() / a new dtype) do not pile up
                for k in [k for k in self._pack_cache if k != key][:len(self._pack_cache) - 16]:
                    del self._pack_cache[k]
        return hit[1]

    def render_features(self, batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats,
                        t_rand=None, want_depth=False, want_weight=False, want_merge=True, precision=None, merge_out=None,
                        z_planes=None, want_fg=True, weight_out=None, workspace=None):
        """Rays -> composited feature map (seams a1..a7).  Outputs are ray-major [B, N_r, C].
        z_planes [B, N_r, N+1] (from fine_planes()): the hierarchical pass -- those planes, the fine network."""
        prec = _lib.PRECISIONS[precision or self.precision]
        B, tv, n_r = batch_xy.size()
        assert tv == 2
        xy = batch_xy if batch_xy.dtype == torch.float32 else batch_xy.float()
        if z_planes is not None:
            assert self.hier_sampling and t_rand is None
            t_rand = z_planes
            geom = self._geom(B, n_r, xy, n_samples=z_planes.shape[-1] - 1, z_planes_given=1, bg_is_hwc=1)
        else:
            geom = self._geom(B, n_r, xy, bg_is_hwc=1)
        params, ws, bs = self._mlp_params(fine=z_planes is not None)
        packed = self._packed(geom, prec, params, ws, bs)
        audio = ops._f32c(audiostyle) if self.audio_dim > 0 else None
        ray_bias = None
        if self.include_vd:  # the view-direction columns of RGB_layer_1 as a per-ray bias (one small launch)
            w10 = (self.fine_fg_CD_predictor if z_planes is not None else self.fg_CD_predictor).RGB_layer_1.w2d().detach()
            ray_bias = ops.ray_vd_bias(geom, w10, xy, ops._f32c(batch_Rmats), ops._f32c(batch_inv_inmats))
        out = ops.render_fwd(geom, prec, packed, params, xy, ops._f32c(batch_Rmats), ops._f32c(batch_Tvecs).view(B, 3),
                             ops._f32c(batch_inv_inmats), ops._f32c(shape_code), ops._f32c(appea_code), audio,
                             None if t_rand is None else ops._f32c(t_rand),
                             self._bg_hwc() if want_merge else None,  # ray-major, cached per parameter version
                             want_depth=want_depth, want_weight=want_weight, want_merge=want_merge, merge_out=merge_out,
                             want_fg=want_fg, weight_out=weight_out, ws=workspace, ray_bias=ray_bias)
        return out

    def fine_planes(self, batch_xy, coarse_weight, batch_Tvecs, t_rand=None, fine_u=None, out=None):
        """FineSample.forward (NetWorks/utils.py:211-263): coarse compositing weights [B, N_r, N_c] -> the
        N_c + N_f + 1 ascending sample planes of the fine pass.  fine_u [B*N_r, N_f+1]: the uniform samples of train mode."""
        B, _, n_r = batch_xy.size()
        xy = batch_xy if batch_xy.dtype == torch.float32 else batch_xy.float()
        return ops.fine_sample(self._geom(B, n_r, xy), self.num_sample_fine, ops._f32c(coarse_weight), ops._f32c(batch_Tvecs).view(B, 3),
                               None if t_rand is None else ops._f32c(t_rand), None if fine_u is None else ops._f32c(fine_u), out=out)

    def _forward(self, for_train, batch_xy, batch_uv, audiostyle, bg_code, shape_code, appea_code, batch_Rmats,
                 batch_Tvecs, batch_inv_inmats, dist_expr, t_rand=None, fine_u=None):
        batch_size, tv, n_r = batch_xy.size()
        assert tv == 2
        assert bg_code is None
        if n_r != self.featmap_size ** 2:
            raise ValueError("forward() renders an image, so N_r must equal featmap_size^2; "
                             "use render_features() for free ray sets")
        if for_train and t_rand is None:
            # same generator consumption as the reference's torch.rand_like(zvals) (NetWorks/utils.py:77)
            t_rand = torch.rand(batch_size, n_r, self.num_sample_coarse + 1, device=batch_xy.device, dtype=torch.float32)
        fs, C = self.featmap_size, self.featmap_nc
        needs_grad = torch.is_grad_enabled() and (
            any(p.requires_grad for p in self.parameters()) or
            any(torch.is_tensor(t) and t.requires_grad for t in (audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs)))
        if needs_grad:
            return self._forward_train(batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats, t_rand,
                                       for_train=for_train, fine_u=fine_u)
        if self._graph_usable(for_train, t_rand, batch_xy):
            return self._forward_graph(batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats)
        n_pass = 2 if self.hier_sampling else 1
        nb = n_pass * batch_size
        maps = self._maps_with_background(nb, batch_xy.device)
        if self.hier_sampling and for_train and fine_u is None:  # the reference's torch.rand(num_temp, NFsample) (NetWorks/utils.py:227)
            fine_u = torch.rand(batch_size * n_r, self.num_sample_fine + 1, device=batch_xy.device, dtype=torch.float32)
        imgs = self._infer_launch(batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats, t_rand, fine_u, maps,
                                  bufs={"bg_in_maps": True})
        return self._result(imgs, batch_size, nb)

    def _maps_with_background(self, nb, dev):
        """The renderer's input batch [nb + 1, fs, fs, C] (merged maps, then the background map) as scratch kept per (device,
        stream), grow-only: the background slot is filled once per parameter version and batch size instead of by a copy
        kernel in every forward (it is 4 us of a 0.64 ms one-head step).  The merged slots are overwritten by every call; the
        images returned to the caller are separate."""
        bg = self._bg_hwc()
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ent = self._maps_cache.get(key)
        fs, C = self.featmap_size, self.featmap_nc
        if ent is None or ent[1].shape[0] < nb + 1:
            ent = (None, torch.empty(nb + 1, fs, fs, C, dtype=torch.float32, device=dev))
        tag = (self._bg_serial, nb)
        if ent[0] != tag:
            ent[1][nb].view(fs * fs, C).copy_(bg)
            self._maps_cache[key] = ent = (tag, ent[1])
        return ent[1][:nb + 1]

    def _result(self, imgs, batch_size, nb):
        res = {"coarse_dict": {"merge_img": imgs[:batch_size], "bg_img": imgs[nb:]}}
        if self.hier_sampling:
            res["fine_dict"] = {"merge_img": imgs[batch_size:nb], "bg_img": imgs[nb:]}
        return res

    def _infer_launch(self, batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats, t_rand, fine_u,
                      maps, imgs=None, bufs=None):
        """Enqueue one inference forward.  The merged maps (coarse, then fine) and the background map go through the 2-D
        renderer in one call; the render kernel writes its merged maps straight into that batch.  With `imgs` and `bufs`
        (caller-owned workspaces / intermediates) nothing is allocated, so the sequence can be recorded into a hipGraph."""
        batch_size, _, n_r = batch_xy.size()
        fs, C = self.featmap_size, self.featmap_nc
        nb = (2 if self.hier_sampling else 1) * batch_size
        bufs = bufs or {}
        coarse = self.render_features(batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats,
                                      t_rand=t_rand, want_weight=self.hier_sampling, want_merge=True, want_fg=False,
                                      merge_out=maps[:batch_size].view(batch_size, fs * fs, C), weight_out=bufs.get("weight"),
                                      workspace=bufs.get("render_ws"))
        if self.hier_sampling:
            planes = self.fine_planes(batch_xy, coarse["weight"], batch_Tvecs, t_rand=t_rand, fine_u=fine_u, out=bufs.get("planes"))
            self.render_features(batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats,
                                 z_planes=planes, want_merge=True, want_fg=False,
                                 merge_out=maps[batch_size:nb].view(batch_size, fs * fs, C), workspace=bufs.get("fine_ws"))
        if "nr_ws" not in bufs and not bufs.get("bg_in_maps"):  # (a recorded graph keeps the background map in its static batch: see _forward_graph)
            maps[nb].view(fs * fs, C).copy_(self._bg_hwc())
        return self.neural_render.render_hwc(maps, self.precision, img=imgs, ws=bufs.get("nr_ws"))

    # ---- hipGraph replay of the inference forward ---------------------------------------------------------------
    def _graph_usable(self, for_train, t_rand, batch_xy):
        """Replay is used for plain mode="test" forwards (the reference's validation / fitting / sweep call shape,
        talker_trainer.py:1119, Utils/RenderUtils.py:120) on a GPU, unless switched off (use_graph=False or N3DT_GRAPH=0) or
        the bench's kernel-timing hook is active (its events belong to the launching call, which a replay never runs)."""
        return (self.use_graph and not for_train and t_rand is None and batch_xy.is_cuda and not _lib.PROF_ACTIVE and
                not self.include_vd)  # (include_vd allocates its per-ray bias per call: not recorded)

    def _param_signature(self):
        return tuple(p.data_ptr() for p in self.parameters())

    def _forward_graph(self, batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats):
        B, _, n_r = batch_xy.size()
        dev = batch_xy.device
        cur = torch.cuda.current_stream(dev)
        key = (B, n_r, self.precision, dev.index, cur.cuda_stream)
        sig = self._param_signature()
        e = self._graphs.get(key)
        if e is not None and e["sig"] != sig:  # parameters were re-allocated (e.g. .to(), a new bg_featmap): record again
            self._drop_graph(self._graphs.pop(key))
            e = None
        xy = batch_xy if batch_xy.dtype == torch.float32 else batch_xy.float()
        small = [ops._f32c(t) for t in (batch_Rmats, batch_Tvecs, batch_inv_inmats, shape_code, appea_code)]
        if self.audio_dim > 0:
            small.append(ops._f32c(audiostyle))
        if e is None:
            if len(self._graphs) >= 8:  # a handful of call shapes per model in practice; do not grow without bound
                self._drop_graph(self._graphs.pop(next(iter(self._graphs))))
            e = self._graphs[key] = self._record_graph(B, n_r, dev, sig, xy, small)
        # packed weights follow the parameters' version counters; re-packed in place, so the recorded address stays valid.
        # Should an address differ all the same (the buffer was replaced behind the graph's back), the graph is recorded again.
        for fine in ((False, True) if self.hier_sampling else (False,)):
            params, ws, bs = self._mlp_params(fine=fine)
            if self._packed(e["geom"], _lib.PRECISIONS[self.precision], params, ws, bs).data_ptr() != e["packed"][fine] or \
                    self._bg_hwc().data_ptr() != e["bg_ptr"]:
                self._drop_graph(self._graphs.pop(key))
                e = self._graphs[key] = self._record_graph(B, n_r, dev, sig, xy, small)
                break
        self._refresh_graph_constants(e)
        ops.stage_inputs([(xy, e["xy"])] + list(zip(small, e["small"])), view=xy)
        ops.graph_launch(e["graph"])
        imgs = e["imgs"] if self.graph_static_outputs else e["imgs"].clone()
        return self._result(imgs, B, e["nb"])

    def _refresh_graph_constants(self, e):
        """What a replay reads but does not compute: the background map's slot of the renderer batch and the renderer's packed
        block weights, refreshed (on the caller's stream, ahead of the replay) when their parameters' versions moved."""
        bg = self.neural_render.bg_featmap
        ver = (bg.data_ptr(), bg._version)
        if e.get("bg_ver") != ver:
            fs, C = self.featmap_size, self.featmap_nc
            e["maps"][e["nb"]].view(fs * fs, C).copy_(self._bg_hwc())
            e["bg_ver"] = ver
        self.neural_render.ensure_packed(e["nb"] + 1, self.precision, e["bufs"]["nr_ws"])

    def _record_graph(self, B, n_r, dev, sig, xy, small):
        fs, C, P = self.featmap_size, self.featmap_nc, self.pred_img_size
        prec = _lib.PRECISIONS[self.precision]
        nb = (2 if self.hier_sampling else 1) * B
        f32 = dict(dtype=torch.float32, device=dev)
        e = {"sig": sig, "nb": nb, "xy": torch.empty(B, 2, n_r, **f32), "small": [torch.empty_like(t) for t in small],
             "maps": torch.empty(nb + 1, fs, fs, C, **f32), "imgs": torch.empty(nb + 1, 3, P, P, **f32)}
        geom = e["geom"] = self._geom(B, n_r, e["xy"])
        bufs = {"render_ws": torch.empty(ops.render_workspace_bytes(geom, prec), dtype=torch.uint8, device=dev),
                "nr_ws": torch.empty(ops.neural_render_workspace_bytes(self.neural_render._geom(nb + 1), nb + 1), dtype=torch.uint8, device=dev)}
        # a freshly allocated workspace may sit where a released one sat: whatever was recorded for that address is void
        self.neural_render._packed_sig.pop(bufs["nr_ws"].data_ptr(), None)
        packed = {}
        if self.hier_sampling:
            n_fine = self.num_sample_coarse + self.num_sample_fine
            bufs["weight"] = torch.empty(B, n_r, self.num_sample_coarse, **f32)
            bufs["planes"] = torch.empty(B, n_r, n_fine + 1, **f32)
            gfine = self._geom(B, n_r, e["xy"], n_samples=n_fine, z_planes_given=1)
            bufs["fine_ws"] = torch.empty(ops.render_workspace_bytes(gfine, prec), dtype=torch.uint8, device=dev)
        for fine in ((False, True) if self.hier_sampling else (False,)):
            params, ws, bs = self._mlp_params(fine=fine)
            packed[fine] = self._packed(geom, prec, params, ws, bs).data_ptr()  # packs now, on the caller's stream
        e["packed"], e["bufs"] = packed, bufs
        e["bg_ptr"] = self._bg_hwc().data_ptr()
        self._refresh_graph_constants(e)
        R, T, Kinv, shape, appea = e["small"][:5]
        audio = e["small"][5] if self.audio_dim > 0 else None
        # record on a private stream (the legacy default stream cannot be captured); replays go to the caller's stream
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            ops.graph_begin(side)
            try:
                self._infer_launch(e["xy"], audio, shape, appea, R, T, Kinv, None, None, e["maps"], imgs=e["imgs"], bufs=bufs)
            except BaseException:
                try:  # end the capture so the stream is usable again, and let the ORIGINAL error through
                    ops.graph_destroy(ops.graph_end(side))
                except Exception:
                    pass
                raise
            e["graph"] = ops.graph_end(side)
        return e

    def _drop_graph(self, e):
        ops.graph_destroy(e["graph"])
        self.neural_render._packed_sig.pop(e["bufs"]["nr_ws"].data_ptr(), None)  # the workspace goes back to the allocator

    def release_graphs(self):
        """Destroy the recorded hipGraphs and their static buffers (they are re-recorded on demand)."""
        for e in self._graphs.values():
            self._drop_graph(e)
        self._graphs.clear()

    def __del__(self):
        try:
            self.release_graphs()
        except Exception:
            pass

    def _forward_train(self, batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats, t_rand,
                       for_train=False, fine_u=None):
        """Differentiable forward: gradients reach every parameter, audiostyle, shape_code, appea_code and (when they require
        grad) batch_Rmats / batch_Tvecs.  With hier_sampling the fine pass is differentiated too; its sample planes come
        from the DETACHED coarse weights, as in the reference (NetWorks/utils.py:219), so they are constants of the backward."""
        B, _, n_r = batch_xy.size()
        fs, C = self.featmap_size, self.featmap_nc
        xy = batch_xy if batch_xy.dtype == torch.float32 else batch_xy.float()
        geom = self._geom(B, n_r, xy)
        if self._grad_arena is not None and parallel_graph_task_id() == -1:
            self._grad_arena.end_pass()  # (a backward that raised never told the arena that its pass was over)
        layers = self.fg_CD_predictor.layers()
        mlp = [m.weight for m in layers] + [m.bias for m in layers]
        ray_bias = fine_ray_bias = None
        if self.include_vd:
            # RGB_layer_1 without its view-direction columns goes to the kernels (a differentiable cat: its gradient flows back
            # into the parameter's other columns); the 27 columns act through the per-ray bias
            h, vd = self.mlp_h_channel, self._vd_dim()
            w10 = layers[10].weight
            mlp[10] = torch.cat([w10[:, :h], w10[:, h + vd:]], dim=1)
            ray_bias = self._vd_ray_bias(xy.detach(), batch_Rmats, batch_inv_inmats)
        audio = audiostyle if self.audio_dim > 0 else torch.zeros(B, 0, device=xy.device)
        n_pass = 2 if self.hier_sampling else 1
        nb = B * n_pass
        # the renderer's input batch, allocated per call (it is saved for the backward): the volumetric passes write their merged
        # maps straight into it, the renderer's forward fills the last slot from bg_featmap
        maps = torch.empty(nb + 1, fs, fs, C, dtype=torch.float32, device=xy.device)
        Kinv = ops._f32c(batch_inv_inmats)
        merge = _RenderFn.apply(self, geom, _Slot(maps[:B].view(B, n_r, C)), xy.detach(), Kinv, None if t_rand is None else ops._f32c(t_rand),
                                batch_Rmats, batch_Tvecs, shape_code, appea_code, audio, self.neural_render.bg_featmap, ray_bias, *mlp)
        merged = [merge]
        if self.hier_sampling:
            with torch.no_grad():
                w = self.render_features(batch_xy, audiostyle, shape_code, appea_code, batch_Rmats, batch_Tvecs, batch_inv_inmats,
                                         t_rand=t_rand, want_weight=True, want_merge=False, precision=self.train_precision)["weight"]
                if for_train and fine_u is None:  # the reference's torch.rand(num_temp, NFsample) (NetWorks/utils.py:227)
                    fine_u = torch.rand(B * n_r, self.num_sample_fine + 1, device=xy.device, dtype=torch.float32)
                planes = self.fine_planes(batch_xy, w, batch_Tvecs, t_rand=t_rand, fine_u=fine_u)
            gfine = self._geom(B, n_r, xy, n_samples=planes.shape[-1] - 1, z_planes_given=1)
            flayers = self.fine_fg_CD_predictor.layers()
            fmlp = [m.weight for m in flayers] + [m.bias for m in flayers]
            if self.include_vd:
                fw10 = flayers[10].weight
                fmlp[10] = torch.cat([fw10[:, :h], fw10[:, h + vd:]], dim=1)
                fine_ray_bias = self._vd_ray_bias(xy.detach(), batch_Rmats, batch_inv_inmats, fine=True)
            merged.append(_RenderFn.apply(self, gfine, _Slot(maps[B:nb].view(B, n_r, C)), xy.detach(), Kinv, planes, batch_Rmats, batch_Tvecs,
                                          shape_code, appea_code, audio, self.neural_render.bg_featmap, fine_ray_bias, *fmlp))
        flat = []
        for m in self.neural_render._flat_modules():
            flat += [m.weight, m.bias]
        imgs = _NeuralRenderFn.apply(self.neural_render, _Slot(maps), len(merged), None, self.neural_render.bg_featmap, *merged, *flat)
        res = {"coarse_dict": {"merge_img": imgs[0], "bg_img": imgs[-1]}}
        if self.hier_sampling:
            res["fine_dict"] = {"merge_img": imgs[1], "bg_img": imgs[-1]}
        return res

    def forward(self, mode, batch_xy, batch_uv, audiostyle=None, bg_code=None, shape_code=None, appea_code=None,
                batch_Rmats=None, batch_Tvecs=None, batch_inv_inmats=None, dist_expr=False, **kwargs):
        assert mode in ["train", "test"]
        return self._forward(mode == "train", batch_xy, batch_uv, audiostyle, bg_code, shape_code, appea_code,
                             batch_Rmats, batch_Tvecs, batch_inv_inmats, dist_expr, t_rand=kwargs.get("t_rand"),
                             fine_u=kwargs.get("fine_u"))
