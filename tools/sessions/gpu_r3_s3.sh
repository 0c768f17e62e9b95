#!/bin/bash
# round-3 GPU session 3: 3-deep ring (one channel block per wave), pre-ticket operand prefetch, 16x16 level at >= 4 frames
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s3; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q > $OUT/wino_tests.log 2>&1; echo "wino tests rc=$?"; tail -3 $OUT/wino_tests.log
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_STAGES large 1 f32 2
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_STAGES normal 1 f32 2
timeout 200 python bench.py --no-cpu-baseline --no-extra --batch 8 2>/dev/null | cut -c1-200
timeout 600 bash tools/sessions/wino_stamps_job.sh 2>&1 | tail -120
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
