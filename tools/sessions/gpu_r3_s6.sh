#!/bin/bash
# round-3 GPU session 6: fp16 storage path (opt.fp16) -- layer tests, autocast-oracle plan tests, bench lines
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s6; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k fp16 -q > $OUT/fp16_conv.log 2>&1; echo "fp16 conv rc=$?"; tail -5 $OUT/fp16_conv.log
timeout 600 python -m pytest tests/test_gpu_plans.py -k "fp16" -q -s > $OUT/fp16_plans.log 2>&1; echo "fp16 plans rc=$?"; grep -v amdgpu.ids $OUT/fp16_plans.log | tail -40
for cfg in "normal 8 f16" "large 8 f16" "normal 1 f16" "large 1 f16" "normal 8 bf16"; do
  set -- $cfg
  timeout 300 python bench.py --variant $1 --batch $2 --dtype $3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'whole', d['roofline']['whole_forward']['frac'])"
done
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
