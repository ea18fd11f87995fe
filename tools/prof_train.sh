#!/bin/bash
# rocprofv3 kernel trace of the config-3 training step (run on the GPU box): stats csv + the last step's launch sequence.
# usage: tools/prof_train.sh <tag>     -> gpurun_out/<tag>/train_stats.csv, train_seq.txt, train.log
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o run -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $R/gpurun_out/$TAG/train.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $R/gpurun_out/$TAG/train_stats.csv
python3 $R/tools/rocpd_seq.py $DB 13 $R/gpurun_out/$TAG/train_seq.txt
tail -1 $R/gpurun_out/$TAG/train.log
