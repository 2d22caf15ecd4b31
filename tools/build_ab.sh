#!/bin/bash
# Builds a variant of the HIP library for A/B runs (tools/exp_libs.py): tools/build_ab.sh <name> [extra hipcc flags...]
# Every source file gets the flags csrc/FLAGS lists for it, like in the Makefile; NOFLAGS=1 leaves those out.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/ab/obj_$NAME
OBJS=()
for f in ray-tracing_amd/csrc/*.hip; do
  b=$(basename "$f" .hip)
  extra=""
  [ -z "${NOFLAGS:-}" ] && extra=$(sed -n "s/^$b\.hip //p" ray-tracing_amd/csrc/FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden $extra "$@" \
    -c -o tools/ab/obj_$NAME/$b.o "$f" &
  OBJS+=(tools/ab/obj_$NAME/$b.o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/ab/$NAME.so "${OBJS[@]}"
rm -rf tools/ab/obj_$NAME
echo built tools/ab/$NAME.so
