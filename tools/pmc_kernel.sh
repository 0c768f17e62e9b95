#!/bin/bash
# PMC counters of the kernels whose name contains <pattern> inside bench.py (GPU box).  Usage: tools/pmc_kernel.sh <outdir> <pattern> [bench args]
set -u
OUT=$(realpath -m "$1"); PAT=$2; shift 2
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra "${ARGS[@]}" > "$OUT/$name.log" 2>&1; }
ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL
run sq3 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM
python - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for path in glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k)
    for name, v in sorted(d.items()): print("   %-32s %14.0f per launch (%d launches)" % (name, v / n[k][name], n[k][name]))
PY
