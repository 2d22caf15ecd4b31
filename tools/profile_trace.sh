#!/bin/bash
# Profiles the trace kernel on one MI355X: kernel-trace stats, then PMC passes (each in its own run).
# Usage (from the repository root, on the GPU box): tools/profile_trace.sh <outdir> [rt_render args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
BIN=./ray-tracing_amd/lib/rt_render
ARGS=${@:-"--scene cover --width 1200 --height 675 --spp 64"}
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- $BIN $ARGS > "$OUT/stats.log" 2>&1 || exit 1
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/$name" -o run --output-format csv -- $BIN $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
pass pmc2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE
pass pmc3 FETCH_SIZE
pass pmc4 WRITE_SIZE
# matrix-core counters (their own pass; names differ between ROCm releases, so a failure here is not fatal)
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 -d "$OUT/pmc5" -o run --output-format csv -- $BIN $ARGS > "$OUT/pmc5.log" 2>&1 || echo "pass pmc5 failed (counter names?)"
echo done
