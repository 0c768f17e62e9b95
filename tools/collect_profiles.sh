#!/bin/bash
# Run on the GPU box (via gpurun): per configuration, the bench JSON line, the rocprofv3 kernel-trace stats of the same command and the
# PMC passes (separate runs per counter group, never combined with sys/hip/hsa traces).  Usage: tools/collect_profiles.sh <tag> [configs]
# Output under gpurun_out/<tag>/; tools/summarise_profiles.sh turns it into the tracked files under profiles/.
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_profiles}
CONFIGS=${2:-"large_b1_f32 large_b8_f32 normal_b8_bf16"}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in $CONFIGS; do
  IFS=_ read -r variant b dt <<< "$cfg"; b=${b#b}
  ARGS="--variant $variant --batch $b --dtype $dt --no-cpu-baseline --no-extra"
  python "$R/bench.py" $ARGS --layers "$OUT/classes_$cfg.txt" > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_$cfg" -o t -- python "$R/bench.py" $ARGS --steps 40 --warmup 5 > "$OUT/trace_$cfg.log" 2>&1
  pmc() { local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$cfg/pmc_$name" -o pmc -- \
      python "$R/bench.py" $ARGS --steps 2 --warmup 1 > "$OUT/pmc_$cfg.$name.log" 2>&1; }
  pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc l2 TCC_HIT_sum TCC_MISS_sum
done
ls "$OUT"
