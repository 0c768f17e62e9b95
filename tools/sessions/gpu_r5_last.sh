#!/bin/bash
# round-5 last session: box info + the default bench line at HEAD once more (one more sample of the pool's boxes)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5last; mkdir -p $OUT
bash tools/sessions/box_info.sh > $OUT/box.txt 2>&1
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT/bench_default.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["extra"]
print("bench default: %.1f frames/s, %.4f ms/step, frac %.3f, sum of classes %.4f ms, cfg2 %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["sum_of_classes_ms"], e["config2_normal_b8_bf16"]["frames_per_s"]))
print("  concurrent:", json.dumps({k: v for k, v in e["concurrent_batch1_forwards"].items() if k != "note"}))
print("  render loop:", {k: v for k, v in e["render_loop_end_to_end"].items() if k != "note"}, "small:", {k: v["frames_per_s"] for k, v in e["small_unet_native_plan"].items() if isinstance(v, dict)})
P
