#!/bin/bash
# round-5 session: native `small` plan with block 0 on its live taps (quarters padded to a K-tile), vector loads in the input pass, the fp16 plan's last layer on rowlast128;
# then the whole GPU suite (kernels.h / igemm.hip / edge_layers.hip were touched)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5u4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_small.py -m gpu -q -s > $OUT/pytest_unet.log 2>&1; echo "pytest unet rc=$?"; grep -E "HIP fp16|autocast oracle|opt.fp16|passed|failed|Error|error" $OUT/pytest_unet.log | tail -30
timeout 900 python tools/unet_small_time.py --no-oracle 2>&1 | grep -v amdgpu.ids > $OUT/unet_small_time.txt; echo "time rc=$?"; cat $OUT/unet_small_time.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -8 $OUT/pytest_all.log
