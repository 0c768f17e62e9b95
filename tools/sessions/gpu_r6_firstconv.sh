# GPU box job (round 6): the first conv at one frame -- copies only / MFMAs only / stores only (-DLSPF2F_ABLATE build; bits: 1 no MFMAs, 2 no stores, 4 no window copies, 8 no weight copies)
mkdir -p gpurun_out/firstconv
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" > gpurun_out/firstconv/build.log 2>&1
for d in 0 1 2 3 4 8 12 13 14 15 7 11; do
  LSP_HIP_DBG=$d python tools/first_conv_time.py large 1 f32 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/firstconv/ablate.txt
