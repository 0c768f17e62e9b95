#!/bin/bash
# round-5 probe: N concurrent batch-1 forwards on N streams (N engines, one blob) against one stream
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5ms; mkdir -p $OUT
for cfg in "large f32 4" "normal f32 4" "normal bf16 4"; do timeout 300 python tools/multistream_probe.py $cfg 2>&1 | grep -v amdgpu.ids; done | tee $OUT/probe.txt
