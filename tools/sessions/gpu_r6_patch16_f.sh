# GPU box job (round 6): segment timers of the patch-staged kernel's K loop (-DLSPF2F_PATCH_STAMPS build)
mkdir -p gpurun_out/patch16
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_PATCH_STAMPS -DLSPF2F_ABLATE" > gpurun_out/patch16/build_f.log 2>&1
for d in 0 1 2 4 512; do
  echo "dbg=$d"; LSP_HIP_DBG=$d python tools/probes/patch16_stamps.py 256 256 64 64 128 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/patch16/stamps.txt
LSP_HIP_DBG=0 python tools/probes/patch16_stamps.py 512 512 32 32 64 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/patch16/stamps.txt
