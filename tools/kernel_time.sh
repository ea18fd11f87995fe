#!/bin/bash
# Average time of the kernels matching a pattern in one profiled config-3 training run, per environment setting.
# usage: tools/kernel_time.sh <pattern> "<ENV=..>" ...   ("-" = none)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  rm -rf /tmp/kt
  export $e >/dev/null 2>&1
  rocprofv3 --kernel-trace -d /tmp/kt -o run -- python3 $R/bench.py --mode train --steps 10 --warmup 3 > /tmp/kt.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) /tmp/kt.csv
  echo "== ${e:-default}"
  python3 - "$PAT" <<'PY'
import csv, sys
for r in csv.reader(open("/tmp/kt.csv")):
    if sys.argv[1] in r[0]:
        print("  %-72s calls %4s  total %9.1f us  avg %8.1f us" % (r[0][:72], r[1], float(r[2]) / 1e3, float(r[3]) / 1e3))
PY
  grep -m1 -o '"ms_per_step": [0-9.]*' /tmp/kt.log
  if [ -n "$e" ]; then unset ${e%%=*}; fi
done
