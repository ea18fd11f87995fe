#!/bin/bash
# Every fuzzer on the CURRENT build, seeds never run before this call: usage tools/fuzz_battery.sh <tag> <seed0>
# One process at a time; a failing fuzzer stops the battery (no GPU step after a failed one).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-fuzz}; S=${2:-100}
O=$R/gpurun_out/$TAG; mkdir -p $O
sha256sum $R/nerf-3dtalker-code_amd/lib/libn3dt.so | cut -c1-16 > $O/lib_sha16.txt
cd $R
python tools/fuzz_parity.py 40 $S > $O/fuzz_parity_seed$S.log 2>&1 && tail -2 $O/fuzz_parity_seed$S.log
N3DT_FUZZ_VD=1 python tools/fuzz_parity.py 24 $((S+1)) > $O/fuzz_parity_vd_seed$((S+1)).log 2>&1 && tail -2 $O/fuzz_parity_vd_seed$((S+1)).log
python tools/fuzz_train.py 40 $((S+2)) > $O/fuzz_train_seed$((S+2)).log 2>&1 && tail -3 $O/fuzz_train_seed$((S+2)).log
python tools/fuzz_train.py 40 $((S+3)) > $O/fuzz_train_seed$((S+3)).log 2>&1 && tail -3 $O/fuzz_train_seed$((S+3)).log
python tools/fuzz_train_nr.py 30 $((S+4)) > $O/fuzz_train_nr_seed$((S+4)).log 2>&1 && tail -2 $O/fuzz_train_nr_seed$((S+4)).log
