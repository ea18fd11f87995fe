"""Per-forward wall time of mode="test" renders at small batches, hipGraph replay against the kernel-by-kernel path.
usage: latency_probe.py [steps]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
for name, (fs, ns, pred), B in (("cfg2 B=1", (64, 64, 512), 1), ("cfg2 B=4", (64, 64, 512), 4), ("cfg4 B=4", (32, 64, 256), 4),
                                ("cfg5 B=4", (32, 96, 1024), 4), ("cfg2 B=16", (64, 64, 512), 16)):
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": pred, "num_sample_coarse": ns})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
    row = []
    for use_graph, static in ((False, False), (True, False), (True, True)):
        net = HeadNeRFNet(opt, False, False, precision="bf16", use_graph=use_graph, graph_static_outputs=static).to(dev)
        net.load_state_dict(sd)
        with torch.no_grad():
            for _ in range(10):
                net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                    d["batch_Tvecs"], d["batch_inv_inmats"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                    d["batch_Tvecs"], d["batch_inv_inmats"])
            torch.cuda.synchronize()
            row.append(1e3 * (time.perf_counter() - t0) / steps)
    print("%-10s kernel-by-kernel %.3f ms | graph replay %.3f ms | replay, static outputs %.3f ms  -> %.0f frames/s" % (
        name, row[0], row[1], row[2], B / row[1] * 1e3))
