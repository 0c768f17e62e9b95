# GPU box job (round 6): the sub-pixel up-conv form of the patch-staged kernel (conv3x3_patchup16): parity, per-layer time against the 128x128 implicit GEMM (up4), whole forward A-B
mkdir -p gpurun_out/patchup
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "patch_kernel" 2>&1 | tail -6 | tee gpurun_out/patchup/tests.txt
# time_conv: c0 c1 cout hs up tile_m tile_n batch k_group dtype
for shape in "512 512 512 16 2 128 128 8 0 1 0 2" "512 512 512 16 2 7116 64 8 0 1" "512 512 256 32 2 128 128 8 0 1" "512 512 256 32 2 7132 128 8 0 1" "256 256 128 64 2 128 128 8 0 1" "256 256 128 64 2 7164 128 8 0 1" "256 256 128 64 2 7164 64 8 0 1"; do
  timeout 120 python tools/time_conv.py $shape 2>&1 | grep "us per launch"
done | tee gpurun_out/patchup/time.txt
for cfg in "normal 8 bf16" "large 8 bf16"; do python tools/ab_tune.py patchup16=0 $cfg 2 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/patchup/ab.txt
