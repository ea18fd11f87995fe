"""Render one fixture geometry with whatever kernel variant the environment selects (N3DT_X16_TILING, N3DT_NR_FUSED,
N3DT_GRAPH ... are read once per process, hence a child process per variant) and save the images / feature maps.
usage: variant_check.py <fixture name> <precision> <out.npz>"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden, synthetic_case  # noqa: E402
from n3dt import HeadNeRFNet, synthetic as syn  # noqa: E402

name, precision, out = sys.argv[1], sys.argv[2], sys.argv[3]
g, m = load_golden(name)
opt, sd, inp = synthetic_case(m)
dev = torch.device("cuda", 0)
net = HeadNeRFNet(opt, include_vd=False, hier_sampling=False, precision=precision).to(dev)
net.load_state_dict(sd, strict=True)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
t_rand = None
if m.get("mode") == "train":
    t_rand = syn.stratified_noise(m["batch"], opt.featmap_size ** 2, opt.num_sample_coarse, m["t_rand_seed"]).to(dev)
with torch.no_grad():
    f = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                            d["batch_inv_inmats"], t_rand=t_rand, want_weight=True)
    o = net(m.get("mode", "test"), d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"],
            d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
torch.cuda.synchronize()
np.savez(out, fg_feat=f["fg_feat"].cpu().numpy(), bg_alpha=f["bg_alpha"].cpu().numpy(), weight=f["weight"].cpu().numpy(),
         merge_img=o["merge_img"].cpu().numpy(), bg_img=o["bg_img"].cpu().numpy())
print("variant_check: %s %s tiling=%s -> %s" % (name, precision, os.environ.get("N3DT_X16_TILING", "default"), out))
