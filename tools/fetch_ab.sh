#!/bin/bash
# FETCH_SIZE per kernel of the training step under two values of an environment switch.  usage: tools/fetch_ab.sh <tag> VAR a b [filter]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; VAR=$2; A=$3; B=$4; FIL=${5:-dw_x16}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  rm -rf /tmp/f_$v
  export $VAR=$v
  rocprofv3 -i $R/tools/pmc/fetch_only.txt --kernel-trace -d /tmp/f_$v -o run -- python3 $R/bench.py --mode train --steps 3 --warmup 1 > $R/gpurun_out/$TAG/fetch_$v.log 2>&1
  python3 $R/tools/rocpd_pmc.py $R/gpurun_out/$TAG/fetch_$VAR.$v.json $(find /tmp/f_$v -name "*.db" | sort)
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/$TAG/fetch_$VAR.$v.json"))
for k,x in d.items():
    if "$FIL" in k: print("$VAR=$v", k[:50], "FETCH x2 = %.1f MB" % (x["FETCH_SIZE"]["mean_per_dispatch"]*2*1024/1e6))
PY
done
