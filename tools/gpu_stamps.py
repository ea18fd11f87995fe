#!/usr/bin/env python3
"""Read the phase stamps of the diagnostic fused-kernel build (N3DT_LIB=.../libn3dt_stamp.so)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
net = HeadNeRFNet(opt, False, False, precision="bf16").to(dev)
net.load_state_dict(syn.make_state_dict(opt, seed=0, bg_noise=0.1))
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, 8).items()}
with torch.no_grad():
    for _ in range(3):
        out = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"],
                                  d["batch_Tvecs"], d["batch_inv_inmats"], want_weight=True)
torch.cuda.synchronize()
from n3dt import ops  # noqa: E402
ws = ops.WORKSPACE.buf[("render", 0)]
# the wlocal region is the tail of the render workspace: blocks * 32 floats
blocks = 8 * 4096 * 2
wl = ws.view(torch.float32)[-(blocks * 32 + 64):]
# find by scanning: stamps are big integers stored as floats in the first 4 slots of each 32-float record
import ctypes  # noqa: E402
from n3dt import _lib  # noqa: E402
g = net._geom(8, 4096, d["batch_xy"])
total = _lib.lib().n3dt_render_workspace_bytes(ctypes.byref(g), 1)
wl_bytes = ((blocks * 32 * 4 + 255) // 256) * 256
full = ws[total - wl_bytes: total - wl_bytes + blocks * 32 * 4].view(torch.float32).view(blocks, 32).double()
wl = full[:, :4]
m = wl.mean(0)
tot = m.sum()
print("per wave mean cycles: bias-init %.0f  mfma-loop %.0f  epilogue %.0f  rendezvous %.0f  (sum %.0f)" % (*m.tolist(), tot))
print("shares: bias %.1f%%  mfma %.1f%%  epilogue %.1f%%  rendezvous %.1f%%" % tuple((100 * m / tot).tolist()))
print("per tile (107 tiles): bias %.0f mfma %.0f epi %.0f rv %.0f" % tuple((m / 107).tolist()))

# timeline of one workgroup (8 consecutive waves): MFMA-loop [start, end] of stream tiles 20..33, relative to wave 0's first
for wg in (0, 1000):
    rows = full[wg * 8: wg * 8 + 8, 4:32]
    base = rows[0, 0]
    print("workgroup %d: per wave (start,end) of the MFMA loop for tiles 20..25, cycles relative to wave 0" % wg)
    for w in range(8):
        r = ((rows[w] - base) % (1 << 24)).tolist()
        print("  wave %d: " % w + "  ".join("(%5d,%5d)" % (r[2 * i], r[2 * i + 1]) for i in range(6)))
