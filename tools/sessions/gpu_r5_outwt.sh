#!/bin/bash
# round-5 opener 2 (GPU box): do write-through output stores (tune key out_wt=1: wino3x3 / winoup3x3 store their finished tile with sc1) shorten the dependent kernel
# boundaries?  The guide prices a boundary at + B / 6 TB/s behind B bytes the predecessor leaves dirty in L2 (2.1-16.8 MB per Winograd layer here, ~280 MB per forward = up
# to ~45 us of 1.52 ms).  Outputs must be bit-identical (only the store instruction changes); timing A-B-A-B in one process, fp32 batch 1 and 8, then the InstanceNorm plan.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5outwt; mkdir -p $OUT
for cfg in "large 1 f32" "large 8 f32" "normal 1 f32"; do
  timeout 200 python tools/ab_tune.py out_wt=1 $cfg 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab.txt
