#!/bin/bash
# round-5 opener (GPU box; the box copy is scratch): where does the slow tenth of wino3x3<1>'s waves come from?  Round 4 left the K loop at 96 % of MFMA issue (median)
# while the p90 wave leaves it 7 000 cycles (18 %) after the median one -- the launch ends with the LAST wave.  The stamp build now records where each wave ran
# (HW_REG_XCC_ID / HW_REG_HW_ID), and tools/wino_stamps.py says whether the tail is a late start, whole CUs, whole XCDs or scattered SIMDs.
set -e
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5tail; mkdir -p $OUT
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_WINO_STAMPS" > $OUT/build.log 2>&1
for a in "128 128 3 1" "128 128 4 1" "128 128 3 1" "64 256 2 1" "256 64 3 2" "512 32 3 4" "64 256 2 1 8"; do
  timeout 120 python tools/wino_stamps.py $a 2>&1 | grep -v amdgpu.ids
done | tee $OUT/stamps.txt
make -C livespeechportraits_amd/csrc -B -j32 > $OUT/rebuild.log 2>&1
