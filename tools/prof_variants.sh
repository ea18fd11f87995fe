# rocprofv3 kernel statistics of `bench.py <args>` for several library builds on ONE box (GPU box: run through gpurun).
# usage: prof_variants.sh "<bench args>" name1 name2 ...   (name "default" = shipped lib, others = lib/variants/libn3dt_<name>.so);
# prints the top kernels per variant and leaves gpurun_out/r02/var_<name>.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
mkdir -p $R/gpurun_out/r02
for v in "$@"; do
  unset N3DT_LIB
  if [ $v != default ]; then export N3DT_LIB=$R/nerf-3dtalker-code_amd/lib/variants/libn3dt_$v.so; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace -d /tmp/prof_$v -o run -- python3 $R/bench.py $ARGS --no-extras --no-cpu-baseline > $R/gpurun_out/r02/var_$v.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/prof_$v -name "*.db" | head -1) $R/gpurun_out/r02/var_$v.csv
  echo "== $v"; python3 - <<PY
import csv
for i,r in enumerate(csv.DictReader(open("$R/gpurun_out/r02/var_$v.csv"))):
    if i<6: print("   %-60s %5s %10.1f us" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
