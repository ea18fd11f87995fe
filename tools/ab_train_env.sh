#!/bin/bash
# A/B of an environment switch on the config-3 training step, alternating runs on one box.
# usage: tools/ab_train_env.sh <tag> VAR val_a val_b [val_c ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; VAR=$2; shift 2
mkdir -p $R/gpurun_out/$TAG
for rep in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v python3 $R/bench.py --mode train --steps 40 --warmup 8 > $R/gpurun_out/$TAG/$VAR.$v.$rep.json 2>/dev/null
    python3 -c "import json; r=json.load(open('$R/gpurun_out/$TAG/$VAR.$v.$rep.json')); print('$VAR=$v rep $rep: %.3f ms' % r['ms_per_step'])"
  done
done
