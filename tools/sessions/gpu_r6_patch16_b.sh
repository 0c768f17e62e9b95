# GPU box job (round 6): the intermittent wrong piece of the patch-staged kernel -- ablation build, failure rate per arm
mkdir -p gpurun_out/patch16
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" > gpurun_out/patch16/build_b.log 2>&1
for d in 0 32 64 256 96; do
  for a in "1 128 128 32 32 128" "8 128 128 32 32 128" "1 128 128 64 64 128" "8 256 256 64 64 128" "8 512 512 32 32 64" "8 512 512 32 32 128"; do
    LSP_HIP_DBG=$d python tools/probes/patch16_debug4.py $a 10 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/patch16/debug4.txt
