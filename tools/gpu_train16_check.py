#!/usr/bin/env python3
"""Gradient comparison of the fused bf16 training path against the exact fp32 path (run on the GPU box).

usage: tools/gpu_train16_check.py [fs] [n_samples] [batch]
Prints per-tensor max |error| relative to the tensor's scale and the cosine between the two gradients.
"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402
from n3dt.train import data_losses, disk_mask  # noqa: E402

dev = torch.device("cuda:0")


def grads(opt, sd, B, precision, t_rand, nr_precision="fp32"):
    net = HeadNeRFNet(opt, False, False, train_precision=precision).to(dev)
    net.load_state_dict(sd)
    net.neural_render.train_precision = nr_precision
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
    for k in ("audiostyle", "shape_code", "appea_code"):
        d[k] = d[k].clone().requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.time()
    out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
              d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
    gt = torch.full_like(out["merge_img"], 0.5)
    terms = data_losses(out, gt, disk_mask(B, opt.pred_img_size).to(dev))
    total = terms["bg_loss"] + terms["head_loss"] + terms["nonhead_loss"]
    total.backward()
    torch.cuda.synchronize()
    dt = time.time() - t0
    g = {k: d[k].grad.detach().clone() for k in ("audiostyle", "shape_code", "appea_code")}
    g.update({n: p.grad.detach().clone() for n, p in net.named_parameters()})
    return float(total), out["merge_img"].detach(), g, dt


def main():
    fs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": fs * 4, "num_sample_coarse": ns})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    t_rand = syn.stratified_noise(B, fs * fs, ns, 7).to(dev)
    l32, img32, g32, _ = grads(opt, sd, B, "fp32", t_rand)
    l16, img16, g16, _ = grads(opt, sd, B, "bf16", t_rand)
    print("fs %d ns %d B %d: loss fp32 %.6f bf16 %.6f, img max diff %.2e" % (fs, ns, B, l32, l16, float((img32 - img16).abs().max())))
    worst = 0.0
    for k in g32:
        a, b = g32[k].double().flatten(), g16[k].double().flatten()
        scale = float(a.abs().max()) + 1e-30
        rel = float((a - b).abs().max()) / scale
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        worst = max(worst, rel)
        flag = "  <<<" if (rel > 0.1 or cos < 0.99) else ""
        print("  %-52s rel %.2e cos %.5f scale %.2e%s" % (k, rel, cos, scale, flag))
    print("  worst rel %.2e" % worst)
    for prec in ("fp32", "bf16"):
        ts = [grads(opt, sd, B, prec, t_rand)[3] for _ in range(3)]
        print("  %s fwd+bwd wall %.1f ms" % (prec, 1e3 * min(ts)))


if __name__ == "__main__":
    main()
