#!/bin/bash
cd $GRAFT_REPO_ROOT
LSP_HIP_WINO_ROT=1 timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q 2>&1 | tail -2
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_ROT large 1 f32 0 1
LSP_HIP_WINO_SP=0 timeout 300 tools/ab_switch.sh LSP_HIP_WINO_ROT large 1 f32 0 1
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_ROT large 8 f32 0 1
