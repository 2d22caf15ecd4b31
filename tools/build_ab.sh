#!/bin/bash
# Builds a variant of the HIP library for A/B runs (tools/exp_libs.py): tools/build_ab.sh <name> [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden "$@" \
  -shared -o tools/ab/$NAME.so ray-tracing_amd/csrc/*.hip
echo built tools/ab/$NAME.so
