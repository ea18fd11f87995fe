#!/bin/bash
# Round-4 evidence: counters of the headline command on the CURRENT build (tools/pmc/render_r02.txt: FETCH / WRITE / MFMA busy / LDS),
# every kernel of the step -- nerf_fwd_x16_kernel, blur_mfma_*, ray_head_mfma_kernel, the block kernels.
# usage: tools/prof_r04_head.sh <tag>  -> gpurun_out/<tag>/pmc_bf16_b16.json (+ the build hash of the library it measured)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_hpmc
rocprofv3 -i $R/tools/pmc/render_r02.txt --kernel-trace -d /tmp/p_hpmc -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_head.log 2>&1
python3 $R/tools/rocpd_pmc.py $O/pmc_bf16_b16.json $(find /tmp/p_hpmc -name "*.db" | sort)
sha256sum $R/nerf-3dtalker-code_amd/lib/libn3dt.so | cut -c1-16 > $O/lib_sha16.txt
echo "headline counters collected"
