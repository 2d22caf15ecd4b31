#!/bin/bash
# rocprofv3 runs of the default bench command on one MI355X: kernel-trace stats, then HBM traffic and VALU counters
# (each PMC pass in its own run).  Usage: tools/profile_bench.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu-baseline --valu-steps 0"}
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 bench.py $ARGS > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" -d "$OUT/$name" -o run --output-format csv -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass pmc_valu SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
pass pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SMEM SQ_INSTS_VMEM
# cycles in which vector and matrix instructions execute together (its own pass: the name is not in every ROCm release)
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES -d "$OUT/pmc_coexec" -o run --output-format csv -- python3 bench.py $ARGS > "$OUT/pmc_coexec.log" 2>&1 || echo "pass pmc_coexec failed (counter names?)"
echo done
