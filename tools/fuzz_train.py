"""Randomised check of the fused mixed-precision training path (nerf_fwd_x16_train / nerf_bwd_x16 / dw_x16 kernels) against the
exact-fp32 training path on the same inputs (GPU; both paths go through the C ABI).

usage: fuzz_train.py [cases=40] [seed=1]
Draws featmap sizes (incl. ones whose block count leaves dead waves in the last workgroup: the saved-tile dump record), sample
counts with ragged last blocks, batch sizes, the gaze / audio-less module variants and, for a third of the cases, camera
gradients; the loss is the three MSE terms against a seeded random target image.  Band asserted on the WELL-CONDITIONED cases:
cosine >= 0.985 and per tensor max error <= 50 % of the tensor's scale (seen over seeds 1, 5, 7, 9, 117 such cases: >= 0.9936,
<= 20 %; the seed-0 cases of tests/test_gpu_train.py sit at <= 1 % / >= 0.9989).  A case whose network is almost transparent on
the drawn rays has an almost vanishing gradient into the MLP (|d density weight| < 1e-2; about a quarter of the draws), and so
does one with fewer than 512 sample points in all: there
the two paths differ by rounding noise on a cancelled sum (cosine down to 0.87, tools/fuzz_train_probe.py); those are reported
and held to cosine >= 0.85.  bf16 operand rounding through ten chained layers keeps the descent direction, not every entry.
THE CLASSIFICATION RULE IS FROZEN (round 4): a case is in the loose band iff  max|d density_module.weight| (fp32 path) < 1e-2
OR  B * fs^2 * ns < 512;  nothing else enters it, and a run fails when more than 40 % of its draws land there.  A case outside
the rule that misses the tight band is a finding to be explained by the kernels, not by another clause here.
Camera gradients are REPORTED, not asserted: their 2^k-weighted cancellation over a few hundred samples makes the bf16 path's
d R / d T a direction of varying quality on tiny geometries (cosine 0.75 - 1.0 seen), which is why train_precision="fp32" is the
mode for fitting cameras.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))


LOOSE_MAX = 0.40  # share of the draws the loose band may take (seen over seeds 1, 5, 7, 9, 11: 20 - 33 %)


def variant_kw(variant):
    return {"include_gaze": True, "eye_gaze_dim": 64} if variant == "gaze" else ({"audio_dim": 0} if variant == "noaudio" else {})


def grads(opt, sd, B, precision, t_rand, cam, variant, dev):
    from n3dt import HeadNeRFNet, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    kw = variant_kw(variant)
    net = HeadNeRFNet(opt, False, False, train_precision=precision, **kw).to(dev)
    net.load_state_dict(sd)
    net.neural_render.train_precision = "fp32"  # isolate the volumetric stage
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B, **kw).items()}
    names = ["audiostyle", "shape_code", "appea_code"] + (["batch_Rmats", "batch_Tvecs"] if cam else [])
    if variant == "noaudio":
        names.remove("audiostyle")
    for k in names:
        d[k] = d[k].clone().requires_grad_(True)
    out = net("train", d["batch_xy"], d["batch_uv"], d.get("audiostyle"), None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
              d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
    # A seeded RANDOM target image: against a constant 0.5 some random networks produce images so close to the target that the
    # loss gradient nearly vanishes (|d audiostyle| ~ 1e-6) and what is compared is rounding noise on a cancelled sum -- seed 7
    # found three such cases (cosine 0.87 - 0.97, a one-element bias gradient with the wrong sign), identical before and after
    # the kernel changes of round 3; tools/fuzz_train_probe.py shows the cosine tracking the gradient's magnitude.
    gen = torch.Generator().manual_seed(1234)
    target = torch.rand(out["merge_img"].shape, generator=gen).to(dev)
    terms = data_losses(out, target, disk_mask(B, opt.pred_img_size).to(dev))
    (terms["bg_loss"] + terms["head_loss"] + terms["nonhead_loss"]).backward()
    g = {k: d[k].grad.detach().clone() for k in names}
    g.update({n: p.grad.detach().clone() for n, p in net.named_parameters() if n.startswith("fg_CD_predictor")})
    return out["merge_img"].detach(), g


def main():
    from n3dt import BaseOptions, synthetic as syn
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    bad = loose = 0
    worst = {"max": 0.0, "cos": 1.0, "cam_cos": 1.0}
    for case in range(cases):
        fs = int(rng.choice([4, 6, 8, 10, 12, 16, 20]))
        ns = int(rng.choice([3, 16, 20, 31, 32, 33, 40, 64, 65, 96, 100]))
        B = int(rng.choice([1, 2, 3, 5]))
        variant = str(rng.choice(["plain", "plain", "gaze", "noaudio"]))
        cam = bool(rng.random() < 0.33)
        opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": fs * 4, "num_sample_coarse": ns})
        kw = variant_kw(variant)
        sd = syn.make_state_dict(opt, seed=int(rng.integers(0, 1000)), bg_noise=0.1, **kw)
        t_rand = syn.stratified_noise(B, fs * fs, ns, int(rng.integers(0, 1000))).to(dev)
        img32, g32 = grads(opt, sd, B, "fp32", t_rand, cam, variant, dev)
        img16, g16 = grads(opt, sd, B, "bf16", t_rand, cam, variant, dev)
        blocks = B * fs * fs * ((ns + 31) // 32)
        msg = []
        # A network whose head is almost transparent on the drawn rays has an almost vanishing gradient into the MLP (|d density
        # weight| 1e-4 .. 3e-3 where the usual draw gives 1e-2 .. 2e-1), and what the two paths then differ by is rounding noise on
        # a cancelled sum: the cosine tracks the gradient's magnitude and recovers with any change that lets the head show (another
        # jitter seed, a larger batch or map: tools/fuzz_train_probe.py).  Such cases are REPORTED and held to a loose band only.
        vanishing = float(g32["fg_CD_predictor.density_module.weight"].abs().max()) < 1e-2
        # ... and so is a case with only a few hundred sample points in all (4 x 4 rays x 3 samples x 3 frames = 144: cosine 0.977,
        # seed 11): the rounding noise of the bf16 chain averages out over the points a gradient is summed over
        vanishing = vanishing or B * fs * fs * ns < 512
        loose += int(vanishing)
        if float((img32 - img16).abs().max()) > 4e-3:
            msg.append("image %.2e" % float((img32 - img16).abs().max()))
        for k in g32:
            a, b = g32[k].double().flatten(), g16[k].double().flatten()
            scale = float(a.abs().max())
            if scale == 0.0 and float(b.abs().max()) == 0.0:
                continue
            err = float((a - b).abs().max()) / (scale + 1e-30)
            cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            camk = k in ("batch_Rmats", "batch_Tvecs")
            if camk:
                worst["cam_cos"] = min(worst["cam_cos"], cos)
                continue
            if not vanishing:
                worst["max"] = max(worst["max"], err)
                worst["cos"] = min(worst["cos"], cos)
            # band = the envelope observed over seeds 1, 5, 9 (110 cases): the DIRECTION holds (cosine >= 0.981) while single entries
            # of the first layers' gradients can be off by a large fraction of the tensor's scale; the same figures, case by case,
            # from the library before and after round 2's kernel changes (the arithmetic did not change)
            if vanishing:
                if a.numel() > 1 and cos < 0.85:
                    msg.append("%s err %.3f cos %.5f (vanishing-gradient case)" % (k, err, cos))
            elif err > 0.5 or cos < 0.985:
                msg.append("%s err %.3f cos %.5f" % (k, err, cos))
        tag = "fs %2d ns %3d B %d %-7s cam %d blocks %5d (%%8 = %d)" % (fs, ns, B, variant, cam, blocks, blocks % 8)
        if msg:
            bad += 1
            print("FAIL", tag, "; ".join(msg[:4]))
        else:
            print("ok  ", tag, "(vanishing gradient or < 512 sample points: |d density weight| %.1e, loose band)" % float(g32["fg_CD_predictor.density_module.weight"].abs().max()) if vanishing else "")
        sys.stdout.flush()
    print("%d / %d cases failed; worst non-camera tensor: max error %.3f of scale, cosine %.5f; worst camera-gradient cosine %.3f" %
          (bad, cases, worst["max"], worst["cos"], worst["cam_cos"]))
    # the loose band is an exception, not a refuge: the run fails when more than LOOSE_MAX of the draws land in it
    print("%d / %d cases in the loose band (limit %.0f %%)" % (loose, cases, 100 * LOOSE_MAX))
    if loose > LOOSE_MAX * cases:
        print("FAIL: too many cases classified into the loose band")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
