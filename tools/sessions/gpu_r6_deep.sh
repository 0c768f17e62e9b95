# GPU box job (round 6): conv3x3_patch16d -- 64 channels per workgroup with the copies between the MFMAs and the weights four K-tiles ahead (6-slot ring): parity, race screen, per layer, whole forward A-B
mkdir -p gpurun_out/deep
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "patch_kernel" 2>&1 | tail -6 | tee gpurun_out/deep/tests.txt
for a in "8 512 512 32 32 64" "8 256 256 64 64 64" "1 128 128 32 32 64"; do python tools/probes/patch16_debug4.py $a 10 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/deep/race.txt
for shape in "512 0 512 32 0 128 128 8 0 1 1 2" "512 0 512 32 0 7032 65 8 0 1 1" "512 0 512 32 0 7032 64 8 0 1 1" "256 0 256 64 0 7064 128 8 0 1 1" "256 0 256 64 0 7064 65 8 0 1 1" "256 0 256 64 0 7064 64 8 0 1 1"; do
  timeout 120 python tools/time_conv.py $shape 2>&1 | grep "us per launch"
done | tee gpurun_out/deep/time.txt
for cfg in "normal 8 bf16" "large 8 bf16"; do python tools/ab_tune.py patch16_deep=0 $cfg 2 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/deep/ab.txt
