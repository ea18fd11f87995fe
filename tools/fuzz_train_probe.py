"""Replay single cases of tools/fuzz_train.py (same generator stream) and vary one thing at a time: is a case outside the band
because of the network drawn (weight seed), the sample jitter, the batch, the camera gradients?   usage: fuzz_train_probe.py seed case [case ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import fuzz_train as ft  # noqa: E402
from n3dt import BaseOptions, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
KEYS = ["audiostyle", "shape_code", "fg_CD_predictor.FeaExt_module_0.weight", "fg_CD_predictor.FeaExt_module_5.weight", "fg_CD_predictor.density_module.weight"]


def run(fs, ns, B, variant, cam, wseed, tseed):
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": fs * 4, "num_sample_coarse": ns})
    kw = ft.variant_kw(variant)
    sd = syn.make_state_dict(opt, seed=wseed, bg_noise=0.1, **kw)
    t_rand = syn.stratified_noise(B, fs * fs, ns, tseed).to(dev)
    i32, g32 = ft.grads(opt, sd, B, "fp32", t_rand, cam, variant, dev)
    i16, g16 = ft.grads(opt, sd, B, "bf16", t_rand, cam, variant, dev)
    out = ["img %.1e" % float((i32 - i16).abs().max())]
    for k in KEYS:
        if k in g32:
            a, b = g32[k].double().flatten(), g16[k].double().flatten()
            out.append("%s %.4f (|g| %.1e)" % (k.replace("fg_CD_predictor.", "")[:18], float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float(a.abs().max())))
    return "  ".join(out)


def main():
    seed, want = int(sys.argv[1]), set(int(x) for x in sys.argv[2:])
    rng = np.random.default_rng(seed)
    for case in range(max(want) + 1):
        fs = int(rng.choice([4, 6, 8, 10, 12, 16, 20]))
        ns = int(rng.choice([3, 16, 20, 31, 32, 33, 40, 64, 65, 96, 100]))
        B = int(rng.choice([1, 2, 3, 5]))
        variant = str(rng.choice(["plain", "plain", "gaze", "noaudio"]))
        cam = bool(rng.random() < 0.33)
        wseed, tseed = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        if case not in want:
            continue
        print("case %d: fs %d ns %d B %d %s cam %d wseed %d tseed %d" % (case, fs, ns, B, variant, cam, wseed, tseed))
        print("   as drawn        ", run(fs, ns, B, variant, cam, wseed, tseed))
        print("   other jitter    ", run(fs, ns, B, variant, cam, wseed, tseed + 1))
        print("   other weights   ", run(fs, ns, B, variant, cam, wseed + 1, tseed))
        print("   no camera grads ", run(fs, ns, B, variant, False, wseed, tseed))
        print("   B = 3           ", run(fs, ns, 3, variant, cam, wseed, tseed))
        print("   featmap 16      ", run(16, ns, B, variant, cam, wseed, tseed))
        print("   64 samples      ", run(fs, 64, B, variant, cam, wseed, tseed))


if __name__ == "__main__":
    main()
