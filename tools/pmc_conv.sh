#!/bin/bash
# PMC counters of ONE conv shape through tools/time_conv.py (GPU box).  Usage: tools/pmc_conv.sh <outdir> <time_conv args...>
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$R/tools/time_conv.py" "${ARGS[@]}" > "$OUT/$name.log" 2>&1; }
ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL
python - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob(sys.argv[1] + "/*/*/pmc_counter_collection.csv") + glob.glob(sys.argv[1] + "/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"): n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "fullk" not in k and "igemm" not in k: continue
    c = max(n[(k, "SQ_WAVE_CYCLES")], 1)
    print(k, "launches", c)
    for name, v in sorted(d.items()): print("   %-32s %14.0f per launch" % (name, v / (c if name in ("SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY","SQ_INSTS_VALU_MFMA_MOPS_F32","SQ_VALU_MFMA_BUSY_CYCLES","GRBM_GUI_ACTIVE") else max(n[(k,"SQ_INSTS_LDS")],1))))
PY
