#!/bin/bash
# UR form: the remaining conv shapes, the whole GPU suite, profile of the default bench
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4ureg; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -s -k "winograd and not upconv and not winograd4" > $OUT/pytest_conv.log 2>&1; echo "conv tests rc=$?"; grep "nb3\|4003\|^wino (.*, 3," $OUT/pytest_conv.log | head -12; tail -2 $OUT/pytest_conv.log
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_all.log
