#!/bin/bash
# round-5 probe: concurrent streams at batch 8 (do two 8-frame forwards in flight beat one?), fp32 and bf16; box info first
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5ms2; mkdir -p $OUT
bash tools/sessions/box_info.sh > $OUT/box.txt 2>&1
for cfg in "large f32 3 8" "normal bf16 3 8" "large f32 2 4"; do timeout 300 python tools/multistream_probe.py $cfg 2>&1 | grep -v amdgpu.ids; done | tee $OUT/probe.txt
