#!/bin/bash
# round-3 GPU session 27: the K-split full-K kernel against the unsplit one, layer by layer (eager back-to-back launches) and in the plan
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s27; mkdir -p $OUT
for a in "512 0 512 8 0 16 16 1 -1 0 1 0" "512 0 512 8 0 16 16 1 -1 0 1 2" "512 512 512 4 1 16 16 1 -1 0 0 0" "512 512 512 4 1 16 16 1 -1 0 0 2"; do
  timeout 120 python tools/time_conv.py $a 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $OUT/fullk_split_time.txt
LSP_HIP_FULLK_SPLIT=1 timeout 300 python tools/layer_table.py large 1 2>/dev/null | grep -E "fullk|sum" | tee $OUT/layers_split.txt
timeout 300 python tools/layer_table.py large 1 2>/dev/null | grep -E "fullk|sum" | tee $OUT/layers_nosplit.txt
