# debug job (GPU box; the box copy is scratch): rebuild the library with the Winograd kernel's phase stamps and print them for the plan's shapes
# (nb 3 = tile 4003: one channel block per wave with the U fragments in registers)
set -e
mkdir -p gpurun_out/wino_stamps
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_WINO_STAMPS" > gpurun_out/wino_stamps/build.log 2>&1
for a in "64 256 2 1" "128 128 1 1" "128 128 3 1" "256 64 3 2" "512 32 3 4" "64 256 2 1 8" "512 32 2 1 8"; do
  python tools/wino_stamps.py $a 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/wino_stamps/stamps.txt
make -C livespeechportraits_amd/csrc -B -j32 > gpurun_out/wino_stamps/rebuild.log 2>&1
