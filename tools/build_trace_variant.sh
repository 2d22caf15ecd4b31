#!/bin/bash
# A/B builds that differ in the flags of rt_trace.hip only: tools/build_trace_variant.sh <name> <flags for rt_trace.hip...>
# (the other objects are taken from the last `make`)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden "$@" \
  -c -o tools/ab/$NAME.o ray-tracing_amd/csrc/rt_trace.hip
OBJ=ray-tracing_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/ab/$NAME.so tools/ab/$NAME.o $OBJ/rt_api.o $OBJ/rt_kernels.o $OBJ/rt_trace_dense.o $OBJ/rt_wavefront.o
rm -f tools/ab/$NAME.o
echo built tools/ab/$NAME.so
