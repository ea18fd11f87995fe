#!/bin/bash
# Round-3 evidence, run on the GPU box: kernel stats of the headline command and of the config-3 training step (+ its launch
# sequence), HBM counters of the training step.  usage: tools/prof_r03.sh <tag>   -> gpurun_out/<tag>/...
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_head /tmp/p_train /tmp/p_pmc
rocprofv3 --kernel-trace -d /tmp/p_head -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_head.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p_head -name "*.db" | head -1) $O/kernel_stats_bf16_b16.csv
grep -m1 '"metric"' $O/bench_head.log > $O/bench_bf16_b16.json
echo "headline profiled"
rocprofv3 --kernel-trace -d /tmp/p_train -o run -- python3 $R/bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.log 2>&1
DB=$(find /tmp/p_train -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $O/kernel_stats_train_bf16_b2.csv
python3 $R/tools/rocpd_seq.py $DB 13 $O/train_seq_bf16_b2.txt
grep -m1 '"metric"' $O/bench_train.log > $O/bench_train_bf16_b2.json
echo "train profiled"
rocprofv3 -i $R/tools/pmc/train_hbm.txt --kernel-trace -d /tmp/p_pmc -o run -- python3 $R/bench.py --mode train --steps 3 --warmup 1 > $O/pmc_train.log 2>&1
python3 $R/tools/rocpd_pmc.py $O/pmc_train_bf16_b2.json $(find /tmp/p_pmc -name "*.db" | sort)
echo "train counters collected"
