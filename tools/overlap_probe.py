#!/usr/bin/env python3
"""Does overlapping the neural renderer of step i with the fused MLP kernel of step i+1 (two streams) raise frames/s?
Run on the GPU box.  Serial loop vs two-stream pipeline, same work per step (16 frames + background image)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn, ops  # noqa: E402

dev = torch.device("cuda:0")
B, fs, C = 16, 64, 256
opt = BaseOptions({"featmap_size": fs, "featmap_nc": C, "pred_img_size": 512, "num_sample_coarse": 64})
net = HeadNeRFNet(opt, False, False, precision="bf16").to(dev)
net.load_state_dict(syn.make_state_dict(opt, seed=0, bg_noise=0.1))
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
bg = net.neural_render.bg_featmap.detach().view(C, fs * fs)


def mlp(maps):
    net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                        d["batch_inv_inmats"], want_merge=True, want_fg=False, merge_out=maps[:B].view(B, fs * fs, C))
    ops.chw_to_hwc(bg, C, fs * fs, maps[B].view(fs * fs, C))


def serial(steps):
    maps = torch.empty(B + 1, fs, fs, C, device=dev)
    for _ in range(steps):
        mlp(maps)
        net.neural_render.render_hwc(maps, "bf16")


def piped(steps):
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    maps = [torch.empty(B + 1, fs, fs, C, device=dev) for _ in range(2)]
    ev_mlp = [torch.cuda.Event() for _ in range(2)]
    ev_nr = [torch.cuda.Event() for _ in range(2)]
    for e in ev_nr:
        e.record(sB)
    for i in range(steps):
        k = i & 1
        with torch.cuda.stream(sA):
            sA.wait_event(ev_nr[k])
            mlp(maps[k])
            ev_mlp[k].record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev_mlp[k])
            net.neural_render.render_hwc(maps[k], "bf16")
            ev_nr[k].record(sB)
    sA.synchronize()
    sB.synchronize()


with torch.no_grad():
    for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
        fn(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(30)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-6s %.3f ms/step  %.1f frames/s" % (name, 1e3 * dt / 30, 30 * B / dt))
