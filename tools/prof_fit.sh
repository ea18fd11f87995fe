#!/bin/bash
# launch sequence of one single-image fitting iteration (config 4 geometry).  usage: tools/prof_fit.sh <tag> [bf16|fp32]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PREC=${2:-bf16}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$TAG
rocprofv3 --kernel-trace -d /tmp/pf_$TAG -o run -- python3 $R/bench.py --mode fit --config cfg4 --precision $PREC --steps 10 --warmup 3 > $R/gpurun_out/$TAG/fit_$PREC.log 2>&1
DB=$(find /tmp/pf_$TAG -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $R/gpurun_out/$TAG/fit_stats_$PREC.csv
python3 $R/tools/rocpd_seq.py $DB 13 $R/gpurun_out/$TAG/fit_seq_$PREC.txt
tail -1 $R/gpurun_out/$TAG/fit_seq_$PREC.txt
