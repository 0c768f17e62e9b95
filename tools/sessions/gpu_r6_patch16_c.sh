# GPU box job (round 6): where the time of the patch-staged kernel goes -- ablation build (-DLSPF2F_ABLATE; results wrong by construction for dbg != 0)
# bits: 1 no copies in the K loop, 2 no fragment reads, 4 no MFMAs, 16 no epilogue, 32 no stagger (both groups in lockstep)
mkdir -p gpurun_out/patch16
make -C livespeechportraits_amd/csrc -B -j32 CXXFLAGS="-O3 -std=c++17 -fPIC -DLSPF2F_ABLATE" > gpurun_out/patch16/build_c.log 2>&1
for shape in "256 0 256 64 0 7064 128 8 0 1 1" "512 0 512 32 0 7032 64 8 0 1 1"; do
  for d in 0 1 2 3 4 7 16 23 32; do
    echo -n "dbg=$d  "; LSP_HIP_DBG=$d timeout 120 python tools/time_conv.py $shape 2>&1 | grep "us per launch"
  done
done | tee gpurun_out/patch16/ablate.txt
