#!/bin/bash
# round-5 session: the tiny-M kernel on inputs up to 128 KB (smallm_kb=128: L6.down of a one-frame plan leaves the 36-way split-K implicit GEMM + reduce launch), A-B-A-B against the
# default (64) in one process; then golden / hazard / conv / plans tests with the LDS-DMA staging
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5sm2; mkdir -p $OUT
for cfg in "large 1 f32 5" "normal 1 f32 5" "large 3 f32 3" "large 4 f32 3"; do
  timeout 300 python tools/ab_tune.py smallm_kb=128 $cfg 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab.txt
timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_hazards.py tests/test_gpu_conv.py tests/test_gpu_plans.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
