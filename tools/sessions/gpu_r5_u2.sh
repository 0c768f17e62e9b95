#!/bin/bash
# round-5 session: native `small` plan with the direct last layer and the weight-streaming kernel of the <= 16-position levels (tests, timing, per-launch table); then the default
# bench line of THIS tree against the previous commit's tree on the same box, A-B-A-B (session u1 measured 580 frames/s with every kernel class at its old time: box or build?)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5u2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_unet_small.py -m gpu -x -q -s > $OUT/pytest_unet.log 2>&1; echo "pytest unet rc=$?"; grep -E "max-abs|passed|failed|Error" $OUT/pytest_unet.log | tail -40
timeout 600 python tools/unet_small_time.py --no-oracle 2>&1 | grep -v amdgpu.ids > $OUT/unet_small_time.txt; echo "time rc=$?"; cat $OUT/unet_small_time.txt
mkdir -p /tmp/prevtree && tar xzf tools/ab/prev_tree.tgz -C /tmp/prevtree
for rep in 1 2; do
  for arm in new prev; do
    if [ $arm = new ]; then d=$GRAFT_REPO_ROOT; else d=/tmp/prevtree; fi
    (cd $d && timeout 400 python bench.py --steps 40 --warmup 10 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/$OUT/bench_${arm}_$rep.json)
    python - $OUT/bench_${arm}_$rep.json $arm $rep <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench %s #%s: %.1f frames/s, %.4f ms/step, sum of classes %.4f ms, cfg2 %.0f frames/s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["sum_of_classes_ms"], d["extra"]["config2_normal_b8_bf16"]["frames_per_s"]))
P
  done
done
