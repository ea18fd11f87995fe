import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "nerf-3dtalker-code_amd"))
from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn, _lib
dev = torch.device("cuda:0")
opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
for label, scale in (("random", 1.0), ("zeros", 0.0), ("random", 1.0), ("zeros", 0.0)):
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    sd = {k: v * scale if v.dtype.is_floating_point and "blur" not in k and ".f" not in k else v for k, v in sd.items()}
    net = HeadNeRFNet(opt, False, False, precision="bf16").to(dev)
    net.load_state_dict(sd)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, 8).items()}
    L = _lib.lib()
    with torch.no_grad():
        for _ in range(5):
            net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"], want_merge=False)
        torch.cuda.synchronize()
        L.n3dt_prof_enable(20)
        for _ in range(20):
            net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"], want_merge=False)
        torch.cuda.synchronize()
    ms = (ctypes.c_float * 20)(); n = ctypes.c_int(0)
    L.n3dt_prof_collect(ms, 20, ctypes.byref(n)); L.n3dt_prof_enable(0)
    print(label, "weights: kernel ms", round(float(np.mean([ms[i] for i in range(n.value)])), 3))
