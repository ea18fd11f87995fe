#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/rocpd_pmc.py output of tools/prof_r04_kead.sh).

usage: tools/make_traffic.py <pmc_bf16_b16.json> <lib_sha16.txt> <profile name the judge can open> [train pmc json] [out = profiles/traffic.json]
With a training-step PMC summary (tools/prof_r04.sh: pmc_train_bf16_b2.json) the HBM bytes of the volumetric stage of one config-3
step go in as `train_cfg2_bf16_b2` (the sum over the training forward, the dX chain and the three weight-gradient launches).
HBM bytes per launch of the headline kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE come from their own
--pmc passes and FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads).  Also writes the
MFMA-busy share, SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), of every kernel of the step, and the
sha256[:16] of the libn3dt.so the counters were taken on: bench.py prints it next to the hash of the library it is running.
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGORITHMIC_B16 = 105551872  # 16 frames: partials 196 fp32 per (ray, 32-sample block) + packed weights + bias tables


def main():
    pmc = json.load(open(sys.argv[1]))
    sha = open(sys.argv[2]).read().strip()
    src = sys.argv[3]
    train = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4].endswith(".json") and "train" in sys.argv[4] else None
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(REPO, "profiles", "traffic.json")
    kern = {}
    for name, c in pmc.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        rec = {"hbm_bytes_per_launch": (2 * c["FETCH_SIZE"]["mean_per_dispatch"] + c["WRITE_SIZE"]["mean_per_dispatch"]) * 1024,
               "fetch_KiB": c["FETCH_SIZE"]["mean_per_dispatch"], "write_KiB": c["WRITE_SIZE"]["mean_per_dispatch"]}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            rec["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / (1024 * c["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8)
        if "SQ_LDS_BANK_CONFLICT" in c:
            rec["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]["mean_per_dispatch"]
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            rec["wave_cycles_waiting"] = c["SQ_WAIT_ANY"]["mean_per_dispatch"] / c["SQ_WAVE_CYCLES"]["mean_per_dispatch"]
        kern[name] = rec
    head = next(v for k, v in kern.items() if k.startswith("nerf_fwd_x16_kernel"))
    res = {"R_bf16_b16": head["hbm_bytes_per_launch"], "_algorithmic_bytes_b16": ALGORITHMIC_B16, "_lib_sha16": sha, "_source": src,
           "_how": "rocprofv3 -i tools/pmc/render_r02.txt (FETCH_SIZE and WRITE_SIZE in separate passes) over `bench.py --no-extras "
                   "--no-cpu-baseline --steps 5 --warmup 2` (tools/prof_r04.sh), summarised by tools/rocpd_pmc.py; bytes per "
                   "launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "_kernels": kern}
    if train:
        tp = json.load(open(train))
        vol = {}
        for name, c in tp.items():
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            is_vol = name.startswith(("nerf_fwd_x16_train_kernel", "nerf_bwd_x16_kernel", "dw_x16_flat_kernel", "dw_x16_multi_kernel")) or \
                name.startswith("dw_x16_kernel<2, 4, 4, 3")
            if is_vol:
                vol[name] = {"fetch_GB": 2 * c["FETCH_SIZE"]["mean_per_dispatch"] * 1024 / 1e9, "write_GB": c["WRITE_SIZE"]["mean_per_dispatch"] * 1024 / 1e9}
        res["train_cfg2_bf16_b2"] = sum(v["fetch_GB"] + v["write_GB"] for v in vol.values()) * 1e9
        res["_train_kernels_GB"] = vol
        res["_train_lib_sha16"] = sha
        res["_train_source"] = os.path.basename(train)
        # algorithmic bytes of the decomposition (per config-3 step, 16 384 blocks of 32 samples): saved layer inputs written once
        # (98 + 12 tiles of 2 KiB + gate words) and read once by the weight gradients, dZ written once (103 tiles) and read once
        res["_train_algorithmic_bytes"] = 16384 * ((98 + 12) * 2048 + 12288 + 103 * 2048 + 98 * 2048 + 103 * 2048)
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("traffic R_bf16_b16 = %.1f MB per launch (algorithmic %.1f), MFMA busy %.3f, library %s" %
          (head["hbm_bytes_per_launch"] / 1e6, ALGORITHMIC_B16 / 1e6, head.get("mfma_busy", float("nan")), sha))


if __name__ == "__main__":
    main()
