#!/usr/bin/env python3
"""How many Adam steps of the build's own trainer make a seed-0 head SHARP?  Config 4's geometry (32 x 32 rays x 64 samples ->
256^2), B frames, exact fp32 training path, target = a disk of a seeded colour pattern on a white background (the reference's
loss: bg + head + nonhead, HeadNeRFLossUtils.py:125-146).  Prints the alpha statistics the `contrast` fixture's manifest holds
(share of rays with bg_alpha < 0.01 = saturated, > 0.5 = mostly transparent) every few steps.
usage: train_sharp_probe.py [steps=600] [lr=1e-3] [B=2] [train_precision=fp32]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    tp = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    dev = torch.device("cuda:0")
    opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 64})
    net, info = syn.train_sharp_head(opt, dev, steps=steps, lr=lr, batch=B, train_precision=tp, log_every=50, log=print, want_share=float(sys.argv[5]) if len(sys.argv) > 5 else 0.2)
    print(info)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("%.1f s" % (time.time() - t0))
