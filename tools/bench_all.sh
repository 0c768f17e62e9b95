#!/bin/bash
# Run on the GPU box: the bench line of every configuration quoted in README.md / DESIGN.md, one JSON file each under gpurun_out/<tag>/.
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_bench_all}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
python bench.py > "$OUT/default.json" 2> "$OUT/default.err"
for cfg in "large 8 f32" "normal 1 f32" "normal 8 f32" "normal 8 bf16" "large 8 bf16" "large 1 bf16" "normal 1 bf16"; do
  set -- $cfg
  python bench.py --variant $1 --batch $2 --dtype $3 --no-cpu-baseline --no-extra > "$OUT/$1_b$2_$3.json" 2> "$OUT/$1_b$2_$3.err"
done
python - "$OUT" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = d["roofline"]
    print("%-20s %9.1f frames/s %8.4f ms  dominant %s frac %.3f  whole %.3f" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel"].split(":")[0], r["frac"], r["whole_forward"]["frac"]))
PY
