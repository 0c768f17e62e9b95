# GPU box job (round 6): first run of the patch-staged 16-bit kernel (patch16.hip): parity, then per-layer time against the 128x128 implicit GEMM
set -x
mkdir -p gpurun_out/patch16
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "patch_kernel" 2>&1 | tail -15 | tee gpurun_out/patch16/tests.txt
for shape in "256 0 256 64 0 128 128 8 0 1 1" "256 0 256 64 0 7064 128 8 0 1 1" "256 0 256 64 0 7064 64 8 0 1 1" "512 0 512 32 0 128 128 8 0 1 1 2" "512 0 512 32 0 7032 64 8 0 1 1" "512 0 512 32 0 7032 128 8 0 1 1" "128 0 128 128 0 7064 128 8 0 1 1"; do
  timeout 120 python tools/time_conv.py $shape 2>&1 | grep "us per launch"
done | tee gpurun_out/patch16/time.txt
