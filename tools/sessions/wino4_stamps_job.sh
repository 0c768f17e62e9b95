# debug job (GPU box; the box copy is scratch): rebuild the library with the Winograd kernels' phase stamps and print them for the plan's shapes
set -e
mkdir -p gpurun_out/wino4_stamps
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_WINO_STAMPS" > gpurun_out/wino4_stamps/build.log 2>&1
for a in "64 256 1" "128 128 2" "64 256 1 8"; do
  python tools/wino4_stamps.py $a 2>&1 | grep -v "amdgpu.ids\|XCD"
done | tee gpurun_out/wino4_stamps/stamps2.txt
make -C livespeechportraits_amd/csrc -B -j32 > gpurun_out/wino4_stamps/rebuild.log 2>&1
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "winograd4" 2>&1 | tail -2
