#!/bin/bash
# round 5, session 7 (GPU box): tilings of the two 16x16-level layers the 16-bit plans still run as implicit GEMM + splitk_reduce (L5.up: 9-tap upsample gather over the concat; L4.down: stride 2)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s7; mkdir -p $OUT
{
echo "# L5.up  1024 -> 512, 8x8 -> 16x16 (9-tap gather form: 64x64 / 32x64 tiles only), bf16; args: split"
for b in 8 4 2; do for sp in 0 2 4 8; do timeout 100 python tools/time_conv.py 512 512 512 8 1 64 64 $b 0 1 0 $sp 2>&1 | grep "us per launch" | sed "s/$/  [split $sp]/"; done; done
echo "# L4.down 512 -> 512, 32x32 -> 16x16 stride 2, bf16"
for b in 8 4; do for t in "64 64 0" "64 64 2" "64 64 4" "64 128 2" "64 128 4" "64 128 6" "128 128 4" "128 128 8" "128 64 4"; do set -- $t; timeout 100 python tools/time_conv.py 512 0 512 32 0 $1 $2 $b 0 1 0 $3 2 2>&1 | grep "us per launch" | sed "s/$/  [split $3]/"; done; done
} | tee $OUT/tilings.txt
