#!/bin/bash
# PMC passes (separate rocprofv3 runs per counter group, as MI355X_MICROARCH.md prescribes;
# never combined with sys/hip/hsa traces).  Usage: tools/pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
