#!/bin/bash
# round 5, session 9 (GPU box): the 8-frame 16-bit layers that run 128x128 tiles with 2 K-splits + a splitk_reduce launch (M = 8192): is a 64x128 / 128x64 tile WITHOUT a split better?
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s9; mkdir -p $OUT
{
run() { timeout 100 python tools/time_conv.py "$@" 2>&1 | grep "us per launch" | sed "s/$/  [split ${12}]/"; }
echo "# L3.down 256 -> 512, 64x64 -> 32x32 stride 2 (K = 2304)"
for t in "0 0 0" "128 128 2" "64 128 1" "128 64 1" "128 128 1"; do set -- $t; run 256 0 512 64 0 $1 $2 8 0 1 0 $3 2; done
echo "# L3 / L4.u res convs 512 -> 512 at 32x32 (K = 4608), residual"
for t in "0 0 0" "128 128 2" "64 128 1" "128 64 1" "128 128 1"; do set -- $t; run 512 0 512 32 0 $1 $2 8 0 1 1 $3 1; done
echo "# L4.up 1024 -> 512, 16x16 -> 32x32 sub-pixel form (4 parities x K = 4096)"
for t in "0 0 0" "128 128 2" "64 128 1" "128 64 1" "128 128 1"; do set -- $t; run 512 512 512 16 2 $1 $2 8 0 1 0 $3 1; done
echo "# the same three at 4 frames (M = 4096)"
for t in "0 0 0" "64 128 1" "128 128 1" "128 128 2"; do set -- $t; run 256 0 512 64 0 $1 $2 4 0 1 0 $3 2; run 512 0 512 32 0 $1 $2 4 0 1 1 $3 1; run 512 512 512 16 2 $1 $2 4 0 1 0 $3 1; done
} | tee $OUT/tilings.txt
