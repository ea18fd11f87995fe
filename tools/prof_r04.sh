#!/bin/bash
# Round-4 evidence on the CURRENT build, run on the GPU box:  usage: tools/prof_r04.sh <tag>  -> gpurun_out/<tag>/...
#   kernel stats of the headline command and of the config-3 / config-4 training steps (+ launch sequence of config 3),
#   counters of the headline command (FETCH / WRITE / MFMA busy / LDS: tools/pmc/render_r02.txt) and of the config-3 training
#   step (FETCH / WRITE), the sha256[:16] of the library they were taken on.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_head /tmp/p_train /tmp/p_train4 /tmp/p_pmc /tmp/p_hpmc
sha256sum $R/nerf-3dtalker-code_amd/lib/libn3dt.so | cut -c1-16 > $O/lib_sha16.txt
rocprofv3 --kernel-trace -d /tmp/p_head -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_head.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p_head -name "*.db" | head -1) $O/kernel_stats_bf16_b16.csv
grep -m1 '"metric"' $O/bench_head.log > $O/bench_bf16_b16.json
echo "headline profiled"
rocprofv3 --kernel-trace -d /tmp/p_train -o run -- python3 $R/bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.log 2>&1
DB=$(find /tmp/p_train -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $O/kernel_stats_train_bf16_b2.csv
python3 $R/tools/rocpd_seq.py $DB 13 $O/train_seq_bf16_b2.txt
grep -m1 '"metric"' $O/bench_train.log > $O/bench_train_bf16_b2.json
rocprofv3 --kernel-trace -d /tmp/p_train4 -o run -- python3 $R/bench.py --mode train --config cfg4 --steps 10 --warmup 3 > $O/bench_train4.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/p_train4 -name "*.db" | head -1) $O/kernel_stats_train_cfg4_bf16_b4.csv
grep -m1 '"metric"' $O/bench_train4.log > $O/bench_train_cfg4_bf16_b4.json
echo "train profiled"
rocprofv3 -i $R/tools/pmc/train_hbm.txt --kernel-trace -d /tmp/p_pmc -o run -- python3 $R/bench.py --mode train --steps 3 --warmup 1 > $O/pmc_train.log 2>&1
python3 $R/tools/rocpd_pmc.py $O/pmc_train_bf16_b2.json $(find /tmp/p_pmc -name "*.db" | sort)
echo "train counters collected"
rocprofv3 -i $R/tools/pmc/render_r02.txt --kernel-trace -d /tmp/p_hpmc -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_head.log 2>&1
python3 $R/tools/rocpd_pmc.py $O/pmc_bf16_b16.json $(find /tmp/p_hpmc -name "*.db" | sort)
echo "headline counters collected"
