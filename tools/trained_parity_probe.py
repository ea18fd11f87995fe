#!/usr/bin/env python3
"""How the inference precisions hold up as the build's own trainer makes a head sharper: train_sharp_head for a fixed number of
Adam steps (no early stop), then frame 0 in bf16 / fp16 / bf16x3 / fp32 against the CPU oracle on those weights.
usage: trained_parity_probe.py [steps ...] (default 100 200 400 800) [lr=1e-3 via N3DT_PROBE_LR]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, REPO)
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 64})
lr = float(os.environ.get("N3DT_PROBE_LR", "1e-3"))
for steps in [int(a) for a in sys.argv[1:]] or [100, 200, 400, 800]:
    net, info = syn.train_sharp_head(opt, dev, steps=steps, lr=lr, batch=2, want_share=2.0)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    del net
    one = syn.frame_inputs(opt, 1)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in one.items()}
    ref = orc.forward(sd, opt, one)
    errs = {}
    for prec in ("bf16", "fp16", "bf16x3", "fp32"):
        n2 = HeadNeRFNet(opt, False, False, precision=prec).to(dev)
        n2.load_state_dict(sd, strict=True)
        with torch.no_grad():
            r = n2("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                   d["batch_Tvecs"], d["batch_inv_inmats"])["coarse_dict"]
        errs[prec] = float(np.abs(r["merge_img"].cpu().numpy() - ref["merge_img"]).max())
    print("steps %4d lr %g: loss %.5f one-sample rays %.2f feat max %.0f | RGB L-inf bf16 %.2e fp16 %.2e bf16x3 %.2e fp32 %.2e" % (
        steps, lr, info["loss_last"], info["one_sample_rays_share"], info["fg_feat_abs_max"], errs["bf16"], errs["fp16"], errs["bf16x3"], errs["fp32"]))
    sys.stdout.flush()
