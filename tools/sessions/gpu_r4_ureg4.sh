#!/bin/bash
# full per-class tables of both arms (UR forms on / off), two runs each
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4ureg; mkdir -p $OUT
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-extra --steps 100 --batch 1 2>/dev/null | tail -1 > $OUT/full_ureg_$i.json
  LSP_HIP_WINO_UREG=0 python bench.py --no-cpu-baseline --no-extra --steps 100 --batch 1 2>/dev/null | tail -1 > $OUT/full_lds_$i.json
done
python - <<'PY'
import json
for arm in ("ureg", "lds"):
    for i in (1, 2):
        d = json.load(open("gpurun_out/r4ureg/full_%s_%d.json" % (arm, i)))
        print(arm, i, d["value"], d["ms_per_step"], "sum of classes", d["roofline"]["sum_of_classes_ms"], "device", d["roofline"]["whole_forward"]["ms_device"])
        print("   " + " | ".join("%s x%d %.4f" % (c["kernel"][:18], c["launches"], c["ms"]) for c in d["roofline"]["per_class"]))
PY
