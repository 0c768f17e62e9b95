#!/bin/bash
# round-5 session: native `small` plan -- unet_tiny with LDS-DMA staging, the fp16 storage plan (opt.fp16 for size small) against the autocast oracle; timing of both
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5u3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_small.py -m gpu -q -s > $OUT/pytest_unet.log 2>&1; echo "pytest unet rc=$?"; grep -E "small generator fp16|autocast|HIP fp16|opt.fp16|passed|failed|Error|error" $OUT/pytest_unet.log | tail -30
timeout 900 python tools/unet_small_time.py --no-oracle 2>&1 | grep -v amdgpu.ids > $OUT/unet_small_time.txt; echo "time rc=$?"; cat $OUT/unet_small_time.txt
