#!/usr/bin/env python3
"""Maximum-size property check: a batch rendered in ONE call equals the same frames rendered one at a time, bit for bit.

Frames are independent, so this holds at any size -- and breaks if an index, a workspace offset or a launch dimension wraps
at 2^31 / 2^32.  Sizes (reading N of config 2: 512^2 rays x 64 samples per frame; fg_feat 268 MB per frame):
    usage: tools/big_batch_probe.py [frames = 20] [precision = bf16]
20 frames = 5.2 M rays, 336 M sample points, fg_feat 5.4 GB (1.34 G floats: past 2^30 elements and 2^32 bytes).
Also the R reading at a batch of 64 heads through the whole forward (64 x 3 x 512^2 image, renderer workspace 64 frames wide).
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nerf-3dtalker-code_amd"))
import torch  # noqa: E402
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402


def train_section(dev, opt, B):
    """Config-3 geometry, B heads in ONE training step (fused bf16 path): the gradient of frame b's latent codes depends on frame b
    alone, so B x (its row in the batched step) must equal the one-frame step's (the loss is a batch mean).  Saved tensors:
    3.7 GB + 3.5 GB per head."""
    from n3dt.train import fused_data_losses as data_losses, disk_mask
    net = HeadNeRFNet(opt, include_vd=False, hier_sampling=False, precision="fp32", train_precision="bf16").to(dev)
    net.load_state_dict(syn.make_state_dict(opt, seed=0, bg_noise=0.1), strict=True)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
    P = opt.pred_img_size
    t_rand = torch.rand(B, opt.featmap_size ** 2, opt.num_sample_coarse + 1, device=dev, generator=torch.Generator(dev).manual_seed(7))

    def grads(sl):
        n = sl.stop - sl.start
        lat = {k: inp[k][sl].clone().requires_grad_(True) for k in ("audiostyle", "shape_code", "appea_code")}
        net.zero_grad()
        out = net._forward(True, inp["batch_xy"][sl], None, lat["audiostyle"], None, lat["shape_code"], lat["appea_code"],
                           inp["batch_Rmats"][sl], inp["batch_Tvecs"][sl], inp["batch_inv_inmats"][sl], False, t_rand=t_rand[sl])
        gt = torch.full((n, 3, P, P), 0.5, device=dev)
        loss = data_losses(out["coarse_dict"], gt, disk_mask(n, P).to(dev))["total_loss"]
        loss.backward()
        return {k: v.grad.detach().clone() for k, v in lat.items()}, float(loss.detach())

    t0 = time.time()
    big, loss_big = grads(slice(0, B))
    torch.cuda.synchronize()
    print("training step, %d heads in one call (%d sample points): loss %.6f, %.2f s" % (
        B, B * opt.featmap_size ** 2 * opt.num_sample_coarse, loss_big, time.time() - t0), flush=True)
    bad = 0
    for b in (0, B // 2, B - 1):
        one, _ = grads(slice(b, b + 1))
        for k in one:
            a, c = big[k][b] * B, one[k][0]
            scale = float(c.abs().max()) + 1e-30
            err = float((a - c).abs().max()) / scale
            cos = float(torch.nn.functional.cosine_similarity(a.flatten(), c.flatten(), dim=0))
            ok = err < 2e-3 and cos > 0.99999   # same kernels on the same blocks; only the order of fp32 atomic sums differs
            print("  head %3d d %-11s max err %.2e of scale, cosine %.7f  %s" % (b, k, err, cos, "ok" if ok else "DIFFERS"), flush=True)
            bad += 0 if ok else 1
    return bad


def render_sections(dev, opt, frames, prec, heads=64):
    net = HeadNeRFNet(opt, include_vd=False, hier_sampling=False, precision=prec).to(dev)
    net.load_state_dict(syn.make_state_dict(opt, seed=0, bg_noise=0.1), strict=True)
    bad = 0
    with torch.no_grad():
        # --- reading N: the feature stage on pred^2 rays per frame
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, frames, n_side=opt.pred_img_size).items()}
        args = lambda sl: (inp["batch_xy"][sl], inp["audiostyle"][sl], inp["shape_code"][sl], inp["appea_code"][sl],  # noqa: E731
                           inp["batch_Rmats"][sl], inp["batch_Tvecs"][sl], inp["batch_inv_inmats"][sl])
        t0 = time.time()
        big = net.render_features(*args(slice(0, frames)), want_merge=False, want_depth=True)
        torch.cuda.synchronize()
        print("reading N, %d frames in one call: fg_feat %s (%.2f G floats), %.2f s" % (
            frames, tuple(big["fg_feat"].shape), big["fg_feat"].numel() / 2 ** 30, time.time() - t0), flush=True)
        for b in sorted({0, 1, frames // 2, frames - 2, frames - 1}):
            one = net.render_features(*args(slice(b, b + 1)), want_merge=False, want_depth=True)
            for k in ("fg_feat", "bg_alpha", "depth"):
                if k in one and one[k] is not None:
                    same = torch.equal(one[k][0], big[k][b])
                    d = float((one[k][0].float() - big[k][b].float()).abs().max())
                    print("  frame %3d %-9s %s (max diff %.3g)" % (b, k, "identical" if same else "DIFFERS", d), flush=True)
                    bad += 0 if same else 1
        del big
        torch.cuda.empty_cache()
        # --- reading R, whole forward, `heads` heads in one call
        B = heads
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
        call = lambda sl: net("test", inp["batch_xy"][sl], None, inp["audiostyle"][sl], None, inp["shape_code"][sl],  # noqa: E731
                              inp["appea_code"][sl], inp["batch_Rmats"][sl], inp["batch_Tvecs"][sl],
                              inp["batch_inv_inmats"][sl])["coarse_dict"]["merge_img"]
        big = call(slice(0, B)).clone()
        print("reading R, %d heads in one call: merge_img %s" % (B, tuple(big.shape)), flush=True)
        for b in sorted({0, B // 2 - 1, B // 2, B - 1}):
            one = call(slice(b, b + 1))
            same = torch.equal(one[0], big[b])
            print("  head %3d merge_img %s (max diff %.3g)" % (b, "identical" if same else "DIFFERS", float((one[0] - big[b]).abs().max())), flush=True)
            bad += 0 if same else 1
    return bad


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    dev = torch.device("cuda", 0)
    opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
    bad = render_sections(dev, opt, frames, prec)
    bad += train_section(dev, opt, int(os.environ.get("N3DT_BIG_TRAIN_BATCH", "16")))
    print("big-batch probe: %s" % ("OK" if bad == 0 else "%d MISMATCHES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
