import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "nerf-3dtalker-code_amd"); sys.path.insert(0, ".")
import torch
from test_gpu_parity import dev, to_dev, build_net, fwd, feats
from n3dt import BaseOptions, synthetic as syn
opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
d = to_dev(syn.frame_inputs(opt, 8))
ref = build_net(opt, sd, "fp32")
img_ref = fwd(ref, d)
for prec in ("bf16x3", "fp16", "bf16"):
    net = build_net(opt, sd, prec)
    o = fwd(net, d)
    e = (o["merge_img"] - img_ref["merge_img"]).abs()
    print(prec, "merge max %.2e per-image" % float(e.max()), [round(float(e[i].max()), 6) for i in range(8)], "bg %.2e" % float((o["bg_img"] - img_ref["bg_img"]).abs().max()))
    o2 = fwd(net, d)
    print("   second call equal:", torch.equal(o["merge_img"], o2["merge_img"]))
