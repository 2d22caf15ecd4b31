#!/bin/bash
# A/B builds that differ in one source file's flags: tools/build_file_variant.sh <name> <file.hip> <flags...>
# (the other objects are taken from the last `make`)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILE=$2; shift 2
B=$(basename "$FILE" .hip)
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden "$@" \
  -c -o tools/ab/$NAME.o ray-tracing_amd/csrc/$B.hip
OBJS=""
for o in ray-tracing_amd/lib/obj/*.o; do [ "$(basename $o .o)" = "$B" ] || OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/ab/$NAME.so tools/ab/$NAME.o $OBJS
rm -f tools/ab/$NAME.o
echo built tools/ab/$NAME.so
