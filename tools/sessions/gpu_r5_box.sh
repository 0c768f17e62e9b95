#!/bin/bash
# round-5: which kind of box + the default bench line (with extra.concurrent_batch1_forwards) + the concurrent-streams probe
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5box_$(date +%H%M%S); mkdir -p $OUT
bash tools/sessions/box_info.sh > $OUT/box.txt 2>&1; head -40 $OUT/box.txt
timeout 400 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
python - $OUT/bench_default.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench default: %.1f frames/s, %.4f ms/step, sum of classes %.4f ms, cfg2 %.0f; concurrent: %s; small: %s" % (d["value"], d["ms_per_step"], d["roofline"]["sum_of_classes_ms"],
      d["extra"]["config2_normal_b8_bf16"]["frames_per_s"], {k: v["frames_per_s"] for k, v in d["extra"]["concurrent_batch1_forwards"].items() if k.startswith("streams")},
      {k: v["frames_per_s"] for k, v in d["extra"]["small_unet_native_plan"].items() if isinstance(v, dict)}))
P
timeout 300 python tools/multistream_probe.py large f32 4 2>&1 | grep -v amdgpu.ids | tee $OUT/probe.txt
