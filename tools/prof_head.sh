#!/bin/bash
# rocprofv3 kernel trace of the headline command (run on the GPU box).  usage: tools/prof_head.sh <tag>  -> gpurun_out/<tag>/head_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph_$TAG
rocprofv3 --kernel-trace -d /tmp/ph_$TAG -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 20 > $R/gpurun_out/$TAG/head.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/ph_$TAG -name "*.db" | head -1) $R/gpurun_out/$TAG/head_stats.csv
grep -E "blur|nr_level|ray_head|to_rgb|fold" $R/gpurun_out/$TAG/head_stats.csv | awk -F'",' '{print substr($1,7,44), $2}'
grep -m1 '"metric"' $R/gpurun_out/$TAG/head.log | cut -c1-130
