#!/bin/bash
# A/B of environment switches on the training step's three volumetric stages (bench.py --mode train prints fwd / dX / dW stage times
# from hipEvent spans).  usage: tools/ab_train_stage.sh <out.log> <config> "<ENV=1 ...>" ["<ENV=...>" ...]   ("-" = no switch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; CFG=$2; shift 2
mkdir -p "$(dirname "$OUT")"
for rep in 1 2; do
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  line=$(env $e python3 $R/bench.py --mode train --config $CFG --steps 30 --warmup 6 2>/dev/null | grep '"metric"')
  python3 - "$e" "$line" <<'PY' >> $OUT
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-40s step %.3f ms  fwd %.3f  dX %.3f  dW %.3f" % (sys.argv[1] or "(default)", d["ms_per_step"], r["fwd_kernel_ms"], r["dx_chain_ms"], r["dw_stage_ms"]))
PY
done
done
