#!/bin/bash
# round-3 GPU session 21: scalar-stream last conv (parity through the network tests, A-B against the matrix-core kernel); phase stamps of the rasteriser
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s21; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -m gpu -q -x > $OUT/pytest_net.log 2>&1; echo "network+plans rc=$?"; tail -4 $OUT/pytest_net.log
last() { python bench.py --no-cpu-baseline --no-extra --steps 50 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-36s %8.1f fps  last_conv %.2f us' % ('$1', d['value'], pc['last_conv']['us_per_launch']))"; }
for i in 1 2; do
  last "b1 scalar-stream (default)"
  LSP_HIP_LASTCONV_MFMA=1 last "b1 matrix-core"
  last "b8 scalar-stream (default)" "--batch 8 --steps 20"
  LSP_HIP_LASTCONV_MFMA=1 last "b8 matrix-core" "--batch 8 --steps 20"
done | tee $OUT/lastconv_ab.txt
cp livespeechportraits_amd/liblspf2f.so /tmp/shipped.so
cp tools/ablate_builds/liblspf2f_RSTAMPS.so livespeechportraits_amd/liblspf2f.so
timeout 120 python tools/raster_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/raster_stamps.txt
cp /tmp/shipped.so livespeechportraits_amd/liblspf2f.so
