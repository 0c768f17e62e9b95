#!/bin/bash
# round-5 closing session #3: whole GPU suite + smoke at HEAD, the default bench line (with the new extras), box info
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5close3; mkdir -p $OUT
bash tools/sessions/box_info.sh > $OUT/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc; tail -6 $OUT/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $OUT/smoke.log
T0=$(date +%s); timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; tail -3 $OUT/bench.err
python - $OUT/bench_default.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["extra"]
print("bench default: %.1f frames/s, %.4f ms/step, frac %.3f, sum of classes %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["sum_of_classes_ms"]))
for k in ("config2_normal_b8_bf16", "small_unet_native_plan", "concurrent_batch1_forwards", "render_loop_end_to_end"):
    print(" ", k, json.dumps({a: b for a, b in e[k].items() if a not in ("note", "workload", "vs_fp32_oracle")}))
print("  pcie:", e.get("pcie_inclusive_frames_per_s_batch1_uint8"), "cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
P
