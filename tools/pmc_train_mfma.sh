#!/bin/bash
# MFMA-busy share and wait share of every kernel of the config-3 training step on the CURRENT build (tools/pmc/mfma_clock.txt).
# usage: tools/pmc_train_mfma.sh <tag>  -> gpurun_out/<tag>/pmc_train_mfma_bf16_b2.json
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r04}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tm
rocprofv3 -i $R/tools/pmc/mfma_clock.txt --kernel-trace -d /tmp/p_tm -o run -- python3 $R/bench.py --mode train --steps 3 --warmup 1 > $O/pmc_train_mfma.log 2>&1
python3 $R/tools/rocpd_pmc.py $O/pmc_train_mfma_bf16_b2.json $(find /tmp/p_tm -name "*.db" | sort)
python3 - $O/pmc_train_mfma_bf16_b2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, c in d.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] > 0:
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / (1024 * c["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8)
        wait = c["SQ_WAIT_ANY"]["mean_per_dispatch"] / max(c["SQ_WAVE_CYCLES"]["mean_per_dispatch"], 1)
        rows.append((c["GRBM_GUI_ACTIVE"]["mean_per_dispatch"], k, busy, wait))
for cyc, k, busy, wait in sorted(rows, reverse=True)[:16]:
    print("%-64s MFMA busy %5.1f %%  waves waiting %5.1f %%  (%.0f k GUI cycles per launch)" % (k[:64], 100 * busy, 100 * wait, cyc / 1e3))
PY
