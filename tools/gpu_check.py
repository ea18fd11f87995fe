#!/usr/bin/env python3
"""Print HIP-vs-golden error magnitudes for every fixture and precision (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden, synthetic_case  # noqa: E402
from n3dt import HeadNeRFNet, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")


def run(name, precision):
    g, m = load_golden(name)
    opt, sd, inp = synthetic_case(m)
    net = HeadNeRFNet(opt, False, False, precision=precision).to(dev)
    net.load_state_dict(sd, strict=True)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    t_rand = None
    if m.get("mode") == "train":
        t_rand = syn.stratified_noise(m["batch"], opt.featmap_size ** 2, opt.num_sample_coarse, m["t_rand_seed"]).to(dev)
    with torch.no_grad():
        feats = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"],
                                    d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand, want_depth=True, want_weight=True)
        out = net("train" if t_rand is not None else "test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None,
                  d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)
    torch.cuda.synchronize()
    fg = feats["fg_feat"].permute(0, 2, 1).cpu().numpy()  # [B,C,Nr]
    ba = feats["bg_alpha"].cpu().numpy()[:, None, :]
    step = int(g["ray_index_step"]) if "ray_index_step" in g else 1
    msg = ["%-10s %-5s" % (name, precision)]
    msg.append("fg %.2e" % np.abs(fg[:, :, ::step] - g["fg_feat"]).max())
    msg.append("ba %.2e" % np.abs(ba - g["bg_alpha"]).max())
    if "depth" in g:
        msg.append("dep %.2e" % np.abs(feats["depth"].cpu().numpy()[:, None, :] - g["depth"]).max())
    if "weight" in g:
        msg.append("w %.2e" % np.abs(feats["weight"].cpu().numpy()[:, None] - g["weight"]).max())
    img = out["coarse_dict"]["merge_img"].cpu().numpy()
    bg = out["coarse_dict"]["bg_img"].cpu().numpy()
    if "merge_img" in g:
        msg.append("img %.2e bg %.2e" % (np.abs(img - g["merge_img"]).max(), np.abs(bg - g["bg_img"]).max()))
    elif "merge_img_q16" in g:
        msg.append("img %.2e" % np.abs(img - g["merge_img_q16"].astype(np.float32) / 65535.0).max())
    else:
        c0, cs = int(g["crop_origin"]), g["merge_img_crop_q16"].shape[-1]
        msg.append("imgcrop %.2e rowsum %.2e" % (
            np.abs(img[:, :, c0:c0 + cs, c0:c0 + cs] - g["merge_img_crop_q16"].astype(np.float32) / 65535.0).max(),
            np.abs(img.astype(np.float64).sum(-1) - g["merge_img_rowsum"]).max()))
    print("  ".join(msg), flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or ["tiny_test", "tiny_train", "cfg1", "cfg2r", "hr"]
    for n in names:
        for p in ("fp32", "bf16", "fp16"):
            try:
                run(n, p)
            except Exception as e:  # keep going: this is a diagnostic
                print("%-10s %-5s FAILED: %r" % (n, p, e), flush=True)
