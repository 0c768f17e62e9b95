#!/bin/bash
# round-5 opener 4 (GPU box): the register form of wino3x3<1> with four register sets (operands three K-steps ahead instead of two; 204 VGPRs, tune key wino_ureg=2,
# test-hook tile 4004).  Parity first (its cases of test_conv3x3_winograd + the golden network through it), then the whole forward A-B-A-B, then its phase stamps.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5ur4; mkdir -p $OUT
LSP_TEST_UR4=1 timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "winograd and nb4" 2>&1 | tail -3 | tee $OUT/parity.txt
LSP_HIP_WINO_UREG=2 timeout 300 python -m pytest tests/test_gpu_network.py -m gpu -q -k "golden" 2>&1 | tail -3 | tee -a $OUT/parity.txt
for cfg in "large 1 f32" "large 8 f32"; do timeout 200 python tools/ab_tune.py wino_ureg=2 $cfg 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ab.txt
