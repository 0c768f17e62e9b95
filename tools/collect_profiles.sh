#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command at batch 1 and 8,
# then PMC passes (separate runs per counter group) at batch 1.  Output under gpurun_out/profiles_raw/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_raw
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace_b1" -o t -- python "$R/bench.py" --no-cpu-baseline --no-extra > "$OUT/trace_b1.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/trace_b8" -o t -- python "$R/bench.py" --no-cpu-baseline --no-extra --batch 8 --steps 20 --warmup 3 > "$OUT/trace_b8.log" 2>&1
pmc() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/pmc_$name.log" 2>&1; }
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_UNALIGNED_STALL
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum
tail -1 "$OUT/trace_b1.log" | cut -c1-200
ls "$OUT"
