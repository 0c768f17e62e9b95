#!/bin/bash
# round-3 GPU session 13: the 16-bit row / band kernels in fp16 storage
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s13; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "fp16 or bf16 or 16bit" > $OUT/conv16.log 2>&1; echo "16-bit conv tests rc=$?"; tail -4 $OUT/conv16.log
timeout 600 python -m pytest tests/test_gpu_plans.py -q -s -k "fp16 or bf16" > $OUT/plans16.log 2>&1; echo "16-bit plan tests rc=$?"; grep -E "HIP fp16|autocast oracle vs|passed|failed" $OUT/plans16.log | tail -12
for cfg in "normal 8 f16" "large 8 f16" "normal 8 bf16" "large 8 bf16"; do
  set -- $cfg
  timeout 300 python bench.py --variant $1 --batch $2 --dtype $3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'whole', d['roofline']['whole_forward']['frac'])"
  cp /dev/null /dev/null
done
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
