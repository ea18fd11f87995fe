"""Diagnostic for the withdrawn tiling 2 (docs/tuning_log.md, round 4): does the garbage depend on what ran on the CUs before?
Renders a large batch first (every CU's registers hold leftovers), then the tiny_train fixture, in ONE process with
N3DT_X16_TILING=2 N3DT_X16_TILING2_DIAG=1 and the diagnostic library in N3DT_LIB."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from conftest import load_golden, synthetic_case  # noqa: E402
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
g, m = load_golden("tiny_train")
opt, sd, inp = synthetic_case(m)
net = HeadNeRFNet(opt, False, False, precision="bf16").to(dev)
net.load_state_dict(sd, strict=True)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
t_rand = syn.stratified_noise(m["batch"], opt.featmap_size ** 2, opt.num_sample_coarse, m["t_rand_seed"]).to(dev)


def check(tag):
    with torch.no_grad():
        f = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                                d["batch_inv_inmats"], t_rand=t_rand)
    e = np.abs(f["fg_feat"].permute(0, 2, 1).cpu().numpy() - g["fg_feat"]).max(axis=1)
    print("%s: max err per frame %s; rays off frame0 %s frame1 %s" % (tag, e.max(axis=1), np.nonzero(e[0] > 1e-2)[0][:40], np.nonzero(e[1] > 1e-2)[0][:40]))


check("fresh process")
if len(sys.argv) > 1 and sys.argv[1] == "pollute":
    big = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
    nb = HeadNeRFNet(big, False, False, precision="bf16").to(dev)
    nb.load_state_dict(syn.make_state_dict(big, seed=0, bg_noise=0.1))
    db = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(big, 8).items()}
    with torch.no_grad():
        nb.render_features(db["batch_xy"], db["audiostyle"], db["shape_code"], db["appea_code"], db["batch_Rmats"], db["batch_Tvecs"], db["batch_inv_inmats"])
    torch.cuda.synchronize()
    check("after a chip-filling launch")
    check("again")

with torch.no_grad():
    f = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                            d["batch_inv_inmats"], t_rand=t_rand, want_weight=True, want_depth=True)
w = f["weight"].cpu().numpy()            # [B,Nr,Ns]
wr = g["weight"][:, 0]                    # [B,Nr,Ns]
ba, bar = f["bg_alpha"].cpu().numpy(), g["bg_alpha"][:, 0]
fe = np.abs(f["fg_feat"].permute(0, 2, 1).cpu().numpy() - g["fg_feat"]).max(axis=1)
for r in (0, 1, 2, 3):
    print("frame 0 ray %d: weight err %.3e  bg_alpha %.5f (ref %.5f)  feat err %.3e  w[:4] %s ref %s" % (
        r, np.abs(w[0, r] - wr[0, r]).max(), ba[0, r], bar[0, r], fe[0, r], w[0, r, :4], wr[0, r, :4]))
bad = fe[0] > 1e-2
ff = f["fg_feat"].cpu().numpy()[0]       # [Nr, C]
gf = g["fg_feat"][0].T                    # [Nr, C]
r = int(np.nonzero(bad)[0][0])
diff = np.abs(ff[r] - gf[r])
print("ray %d: channels off %d of 256; first bad channels %s; ratio got/ref on them %s" % (r, int((diff > 1e-2).sum()), np.nonzero(diff > 1e-2)[0][:12],
      (ff[r] / (gf[r] + 1e-12))[np.nonzero(diff > 1e-2)[0][:6]]))
