#!/bin/bash
# Build lib/variants/libn3dt_NAME.so: the shipped objects with some sources recompiled under extra flags (diagnostic / A-B builds,
# selected at run time with N3DT_LIB; tools/ab_libs.py and tools/prof_variants.sh take them).  Run `make` first.
# usage: mkvariant.sh NAME "EXTRA FLAGS" file1 [file2...]   (files = csrc basenames to recompile with the flags, e.g. nerf_fwd_x16)
set -e
cd "$(dirname "$0")/../nerf-3dtalker-code_amd"
NAME=$1; EXTRA=$2; shift 2
mkdir -p build/var_$NAME lib/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o)
  use=$o
  for f in "$@"; do
    if [ "$f" = "$b" ]; then
      /opt/rocm/bin/hipcc $FLAGS $EXTRA -c csrc/$b.hip -o build/var_$NAME/$b.o &
      use=build/var_$NAME/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libn3dt_$NAME.so $OBJS
echo built lib/variants/libn3dt_$NAME.so
