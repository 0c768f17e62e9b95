#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s5; mkdir -p $OUT
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_PRIO large 1 f32 0 1
timeout 300 tools/ab_switch.sh LSP_HIP_WINO_PRIO large 8 f32 0 1
