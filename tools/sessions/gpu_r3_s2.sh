#!/bin/bash
# round-3 GPU session 2: Winograd phase stamps, A-B of the epilogue-operand prefetch and of the block order, full suite
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s2; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q > $OUT/wino_tests.log 2>&1; echo "wino tests rc=$?"; tail -3 $OUT/wino_tests.log
timeout 600 bash tools/sessions/wino_stamps_job.sh 2>&1 | tail -80
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_PRE large 1 f32
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_XCD large 1 f32 0
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_PRE large 8 f32
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
