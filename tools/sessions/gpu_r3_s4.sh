#!/bin/bash
# round-3 GPU session 4: copies issued between the MFMA groups (LSP_HIP_WINO_IL: 0 block issue + ring 2, 1 interleaved (+ ring 3 for nb 1), 2 interleaved + ring 2)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s4; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q > $OUT/wino_tests.log 2>&1; echo "wino tests rc=$?"; tail -3 $OUT/wino_tests.log
LSP_HIP_WINO_IL=2 timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q > $OUT/wino_tests_il2.log 2>&1; echo "wino tests (IL=2) rc=$?"; tail -3 $OUT/wino_tests_il2.log
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_IL large 1 f32 0
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_IL large 1 f32 0 2
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_IL large 8 f32 0
timeout 600 bash tools/sessions/wino_stamps_job.sh 2>&1 | grep -v XCD | tail -60
