#!/bin/bash
# round-5 session: conv3x3_smallm stages its input tensor by LDS-DMA (every piece in flight at once) instead of register pairs, each pair waited for (four dependent round trips inside a
# 6.6-us launch for the 4x4 levels).  Bit-identical by construction; A-B-A-B against smallm_dma=0 in one process, fp32 batch 1 of both variants; golden + hazard tests.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5sm; mkdir -p $OUT
for cfg in "large 1 f32 5" "normal 1 f32 5" "large 2 f32 3"; do
  timeout 300 python tools/ab_tune.py smallm_dma=0 $cfg 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_hazards.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
