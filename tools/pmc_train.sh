#!/bin/bash
# counters of the training step's kernels for a counter file.  usage: tools/pmc_train.sh <tag> <pmc file> [filter]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PMC=$2; FIL=${3:-dw_x16}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$TAG
rocprofv3 -i $R/$PMC --kernel-trace -d /tmp/pm_$TAG -o run -- python3 $R/bench.py --mode train --steps 3 --warmup 1 > $R/gpurun_out/$TAG/pmc.log 2>&1
python3 $R/tools/rocpd_pmc.py $R/gpurun_out/$TAG/pmc.json $(find /tmp/pm_$TAG -name "*.db" | sort)
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/$TAG/pmc.json"))
for k,x in d.items():
    if "$FIL" in k: print(k[:60], {c: round(v["mean_per_dispatch"]) for c,v in x.items()})
PY
