#!/bin/bash
# round-3 GPU session 10: software-pipelined transform (LSP_HIP_WINO_SP=0/1)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s10; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q > $OUT/wino_tests.log 2>&1; echo "wino tests rc=$?"; tail -3 $OUT/wino_tests.log
timeout 300 python -m pytest tests/test_gpu_network.py -q -x -k "golden or batch" > $OUT/net_tests.log 2>&1; echo "network tests rc=$?"; tail -2 $OUT/net_tests.log
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_SP large 1 f32 0
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_SP normal 1 f32 0
timeout 400 bash tools/sessions/wino_stamps_job.sh 2>&1 | grep -E "workgroups|K loop:"
