# GPU box job (round 6): where the time of the 16-bit 128x128 implicit GEMM goes, per layer shape of configs[2] -- ablation build (-DLSPF2F_ABLATE; results are wrong by construction for dbg != 0)
# bits: 1 no refetch in the K loop, 16 no epilogue, 32 no K loop
set -e
mkdir -p gpurun_out/igemm16_ablate
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" > gpurun_out/igemm16_ablate/build.log 2>&1
for shape in "256 0 256 64 0 128 128 8 0 1 1" "512 0 512 32 0 128 128 8 0 1 1 2" "256 256 128 64 2 128 128 8 0 1 0" "64 0 128 256 0 128 128 8 0 1 0 0 2" "512 512 256 32 2 128 128 8 0 1 0"; do
  for d in 0 1 16 32 48; do
    echo -n "dbg=$d  "; LSP_HIP_DBG=$d timeout 120 python tools/time_conv.py $shape 2>&1 | grep "us per launch"
  done
done | tee gpurun_out/igemm16_ablate/ablate.txt
make -C livespeechportraits_amd/csrc -B -j32 > gpurun_out/igemm16_ablate/rebuild.log 2>&1
