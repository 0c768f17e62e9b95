#!/bin/bash
# round-3 GPU session 7: hand-off retry + residency checks, InstanceNorm vs float64 at 512x512 and its throughput, whole suite
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s7; mkdir -p $OUT
timeout 600 python -m pytest tests/test_instance_norm.py -m gpu -q -s > $OUT/in_tests.log 2>&1; echo "instance-norm rc=$?"; grep -E "max-abs|float64|itself|passed|failed|Error|assert" $OUT/in_tests.log | head -40
timeout 600 python -m pytest tests/test_gpu_stress.py tests/test_gpu_a2h.py tests/test_rnn.py -m gpu -q > $OUT/handoff.log 2>&1; echo "hand-off rc=$?"; tail -5 $OUT/handoff.log
for cfg in "normal 1" "normal 8" "large 1" "large 8"; do
  set -- $cfg
  timeout 300 python tools/in_bench.py $1 $2 2>/dev/null | tail -12
done | tee $OUT/in_bench.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
