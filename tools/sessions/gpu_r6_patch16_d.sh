# GPU box job (round 6): the patch-staged kernel inside the plans -- 16-bit network tests, then A-B-A-B of the tune key on configs[2] and its neighbours
mkdir -p gpurun_out/patch16
timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_conv.py -m gpu -x -q -k "bf16 or f16 or fp16 or patch" 2>&1 | tail -8 | tee gpurun_out/patch16/net_tests.txt
for cfg in "normal 8 bf16" "large 8 bf16" "normal 8 f16" "normal 4 bf16"; do
  python tools/ab_tune.py patch16=0 $cfg 3 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/patch16/ab.txt
