#!/bin/bash
# round-4 GPU session: wino3x3 UR form (U fragments in registers) -- parity on the conv shapes and the goldens, then A-B-A-B against the LDS form
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4ureg; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "winograd and not upconv and not winograd4" > $OUT/pytest_conv.log 2>&1; echo "conv tests rc=$?"; tail -3 $OUT/pytest_conv.log
timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "golden or batch8" > $OUT/pytest_net.log 2>&1; echo "network tests rc=$?"; tail -3 $OUT/pytest_net.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-10s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us f %.3f' % (k[:12], c['launches'], c['us_per_launch'], c['frac_mfma']) for k,c in pc.items() if k.startswith('wino3'))))"; }
for i in 1 2 3; do
  run "ureg" 1
  LSP_HIP_WINO_UREG=0 run "lds" 1
done 2>&1 | tee $OUT/ab_b1.txt
for i in 1 2; do
  run "ureg" 8
  LSP_HIP_WINO_UREG=0 run "lds" 8
done 2>&1 | tee $OUT/ab_b8.txt
