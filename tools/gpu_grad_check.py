#!/usr/bin/env python3
"""Print gradient errors of the HIP training path against the golden fixtures (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden, synthetic_case  # noqa: E402
from n3dt import HeadNeRFNet, synthetic as syn  # noqa: E402
from n3dt.train import data_losses, disk_mask  # noqa: E402

dev = torch.device("cuda:0")


def run(name):
    g, m = load_golden(name)
    opt, sd, inp = synthetic_case(m)
    net = HeadNeRFNet(opt, False, False).to(dev)
    net.load_state_dict(sd, strict=True)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    for k in ("audiostyle", "shape_code", "appea_code", "batch_Rmats", "batch_Tvecs"):
        d[k] = d[k].clone().requires_grad_(True)
    t_rand = None
    if m["mode"] == "train":
        t_rand = syn.stratified_noise(m["batch"], opt.featmap_size ** 2, opt.num_sample_coarse, m["t_rand_seed"]).to(dev)
    out = net(m["mode"], d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
              d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
    print(name, "fwd img err %.2e bg %.2e" % (np.abs(out["merge_img"].detach().cpu().numpy() - g["merge_img"]).max(),
                                               np.abs(out["bg_img"].detach().cpu().numpy() - g["bg_img"]).max()))
    gt = torch.full_like(out["merge_img"], 0.5)
    mask = disk_mask(m["batch"], opt.pred_img_size).to(dev)
    terms = data_losses(out, gt, mask)
    total = terms["bg_loss"] + terms["head_loss"] + terms["nonhead_loss"]
    print("  loss terms", [float(terms[k]) for k in ("bg_loss", "head_loss", "nonhead_loss")], "golden", g["loss_terms"])
    total.backward()
    for k in ("audiostyle", "shape_code", "appea_code", "batch_Rmats", "batch_Tvecs"):
        ref = g["grad_in." + k]
        got = d[k].grad.cpu().numpy()
        print("  d%-11s max|err| %.2e  (max|ref| %.2e)" % (k, np.abs(got - ref).max(), np.abs(ref).max()))
    worst = 0.0
    for pname, p in net.named_parameters():
        idx = g["grad_p.%s.idx" % pname]
        val = g["grad_p.%s.val" % pname]
        got = p.grad.detach().reshape(-1)[torch.from_numpy(idx).to(dev)].cpu().numpy()
        scale = np.abs(val).max() + 1e-12
        rel = np.abs(got - val).max() / scale
        srel = abs(float(p.grad.double().sum()) - float(g["grad_p.%s.sum" % pname])) / (float(g["grad_p.%s.abs" % pname]) + 1e-12)
        worst = max(worst, rel)
        print("  %-50s rel %.2e  sumrel %.2e  (max|ref| %.2e)" % (pname, rel, srel, scale))
    print("  worst param rel err %.2e" % worst)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["tiny_test", "tiny_train"]:
        run(n)
