#!/bin/bash
# like ab_train_env.sh but each configuration is a full "VAR=val VAR2=val2" string; usage: tools/ab_train_env2.sh <tag> "A=1 B=2" "A=3" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python3 $R/bench.py --mode train --steps 40 --warmup 8 > $R/gpurun_out/$TAG/tmp.json 2>/dev/null
    python3 -c "import json; r=json.load(open('$R/gpurun_out/$TAG/tmp.json')); print('[$cfg] rep $rep: %.3f ms' % r['ms_per_step'])"
  done
done
