# GPU box job (round 6): patch-staged kernel, second ablation round: 512 no s_setprio, 1024 tap-invariant fragment addresses (no address arithmetic per tap), 2048 no lgkmcnt(0) before the barrier
mkdir -p gpurun_out/patch16
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" > gpurun_out/patch16/build_e.log 2>&1
for shape in "256 0 256 64 0 7064 128 8 0 1 1" "512 0 512 32 0 7032 64 8 0 1 1"; do
  for d in 0 512 1024 1536 2048 1 1025 4 516; do
    echo -n "dbg=$d  "; LSP_HIP_DBG=$d timeout 120 python tools/time_conv.py $shape 2>&1 | grep "us per launch"
  done
done | tee gpurun_out/patch16/ablate2.txt
