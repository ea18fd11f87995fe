"""Where does the bf16 training path's camera-gradient error come from?  (VERDICT r2 #3a proposed
splitting the operands of the two d-PE products.)

For a few geometries: d R, d T of (a) the exact fp32 path, (b) the SAME path with the d-PE products' operands rounded to bf16
(N3DT_DIAG_PE_BF16 = 1: dZ, 2: weights, 3: both -- i.e. the exact path's dZ through the bf16 path's LAST product), (c) the fused bf16
path.  If (b) reproduces (c)'s error, splitting the last product's operands would fix the camera gradients; if (b) is far smaller,
the error is the rounding dZ has collected through the seven bf16 stages above it, which a split of the last product cannot touch.

usage: python tools/cam_error_probe.py            (runs its five configurations as sub-processes: the switch is read once)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
CASES = [(8, 8, 2, 0), (8, 32, 2, 1), (16, 64, 2, 2), (16, 40, 1, 3), (32, 64, 1, 4), (12, 20, 3, 5), (32, 64, 2, 6)]


def one(precision):
    import torch
    from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    dev = torch.device("cuda:0")
    out = []
    for fs, ns, B, seed in CASES:
        opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": fs * 4, "num_sample_coarse": ns})
        sd = syn.make_state_dict(opt, seed=seed, bg_noise=0.1)
        net = HeadNeRFNet(opt, False, False, train_precision=precision).to(dev)
        net.load_state_dict(sd)
        net.neural_render.train_precision = "fp32"
        d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
        for k in ("batch_Rmats", "batch_Tvecs"):
            d[k] = d[k].clone().requires_grad_(True)
        t_rand = syn.stratified_noise(B, fs * fs, ns, 7).to(dev)
        o = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
        t = data_losses(o, torch.full_like(o["merge_img"], 0.5), disk_mask(B, opt.pred_img_size).to(dev))
        (t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]).backward()
        out.append({"R": d["batch_Rmats"].grad.flatten().tolist(), "T": d["batch_Tvecs"].grad.flatten().tolist()})
    print("RESULT " + json.dumps(out))


def main():
    import numpy as np
    runs = {}
    for name, prec, diag in (("exact", "fp32", "0"), ("dZ->bf16", "fp32", "1"), ("W->bf16", "fp32", "2"), ("both->bf16", "fp32", "3"), ("bf16 path", "bf16", "0")):
        env = dict(os.environ, N3DT_DIAG_PE_BF16=diag)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", prec], env=env, capture_output=True, text=True, timeout=600)
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        assert line, r.stderr[-2000:]
        runs[name] = json.loads(line[0][7:])
    print("%-22s" % "case (fs, ns, B)" + "".join("%-30s" % n for n in list(runs)[1:]))
    for ci, case in enumerate(CASES):
        row = "%-22s" % str(case[:3])
        for name in list(runs)[1:]:
            cell = []
            for k in ("R", "T"):
                a, b = np.array(runs["exact"][ci][k]), np.array(runs[name][ci][k])
                cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
                err = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-30))
                cell.append("%s cos %.5f err %.3f" % (k, cos, err))
            row += "%-30s" % (cell[0] + " |")
            row = row[:-1] + " "
        print(row)
        row2 = "%-22s" % ""
        for name in list(runs)[1:]:
            a, b = np.array(runs["exact"][ci]["T"]), np.array(runs[name][ci]["T"])
            cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            err = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-30))
            row2 += "%-30s" % ("T cos %.5f err %.3f" % (cos, err))
        print(row2)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        one(sys.argv[2])
    else:
        main()
