# GPU box job (the box copy is scratch): rebuild the library with the Winograd kernels' phase stamps, print where the gated workgroups of wino3x3_chain wait, rebuild the shipped library
set -e
mkdir -p gpurun_out/wino_chain_stamps
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_WINO_STAMPS" > gpurun_out/wino_chain_stamps/build.log 2>&1
for a in "128 128 1 4 3" "128 128 1 4 1" "128 128 1 4 4" "128 128 1 2 3" "256 64 2 4 3" "512 32 4 4 3"; do
  timeout 120 python tools/probes/wino_chain_stamps.py $a 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/wino_chain_stamps/stamps.txt
make -C livespeechportraits_amd/csrc -B -j32 > gpurun_out/wino_chain_stamps/rebuild.log 2>&1
