#!/bin/bash
# round-3 GPU session 26: phase stamps of the full-K kernel on the 8x8 and 16x16 layers of the batch-1 plan (library built with -DLSPF2F_FULLK_STAMPS here)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s26; mkdir -p $OUT
cp livespeechportraits_amd/liblspf2f.so /tmp/shipped.so
cp tools/ablate_builds/liblspf2f_FKSTAMPS.so livespeechportraits_amd/liblspf2f.so
for a in "512 0 512 8 0 16 16 1 -1 0 1" "512 0 512 16 0 32 16 1 -1 0 1" "512 512 512 4 1 16 16 1 -1"; do
  timeout 120 python tools/time_conv.py $a 2>&1 | grep -v amdgpu.ids
done | tee $OUT/fullk_stamps.txt
cp /tmp/shipped.so livespeechportraits_amd/liblspf2f.so
