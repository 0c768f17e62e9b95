#!/bin/bash
# round 5, session 4 (GPU box): the 16-bit full-K kernel (csrc/fullk16.hip) -- per-kernel parity, network parity of the 16-bit plans through it, per-layer timing against the
# planner's previous choice, whole-forward A-B of configs[2]
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s4; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "16bit" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $OUT/parity.txt
timeout 900 python -m pytest tests/test_gpu_plans.py -m gpu -q -x -k "bf16 or fp16" 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $OUT/parity.txt
{
# c0 c1 cout hs up tile_m tile_n batch k_group dtype residual split stride        (tile 16 16 = conv3x3_fullk16, 0 0 = what the planner's tiling rule picks for the implicit GEMM)
for shape in "512 0 512 4 0 B -1 1 1 0 1" "512 0 512 2 0 B -1 1 1 0 1" "512 0 512 8 0 B -1 1 0 0 2" "512 0 512 4 0 B -1 1 0 0 2" "512 0 512 2 1 B -1 1 0 0 1" "512 512 512 4 1 B -1 1 0 0 1" "512 0 512 16 0 B -1 1 0 0 2" "512 0 512 8 0 B -1 1 1 0 1"; do
  set -- $shape
  for b in 8 4 2; do
    timeout 100 python tools/time_conv.py $1 $2 $3 $4 $5 16 16 $b -1 $8 $9 0 ${11} 2>&1 | grep "us per launch"
    timeout 100 python tools/time_conv.py $1 $2 $3 $4 $5 0 0 $b 0 $8 $9 0 ${11} 2>&1 | grep "us per launch"
  done
done
} | tee $OUT/time_conv.txt
for cfg in "normal 8 bf16" "large 8 bf16" "normal 8 f16" "normal 4 bf16"; do timeout 200 python tools/ab_tune.py fullk16=0 $cfg 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ab.txt
timeout 200 python tools/ab_tune.py fullk16=7 normal 8 bf16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
