#!/bin/bash
# round-4 GPU session 4: weights-stationary last conv -- parity (goldens, uint8, every variant), A-B against the LDS-fed matrix-core kernel, then the rest of the suite
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4s4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -x -q -s -k "golden or uint8 or last_conv" > $OUT/pytest_net.log 2>&1; echo "network tests rc=$?"; grep "max-abs\|passed\|failed\|Error" $OUT/pytest_net.log | head -20
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-14s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.1f us hbm %.3f' % (k[:14], c['launches'], c['us_per_launch'], c['frac_hbm']) for k,c in pc.items() if k.startswith('last') or k.startswith('first'))))"; }
for i in 1 2; do
  run "8-wave" 1
  LSP_HIP_LASTCONV_MFMA=1 run "4-wave" 1; LSP_HIP_LASTCONV_WS=1 run "w-stationary" 1
done 2>&1 | tee $OUT/ab_b1.txt
run "8-wave" 8 2>&1 | tee $OUT/ab_b8.txt
LSP_HIP_LASTCONV_MFMA=1 run "4-wave" 8 2>&1 | tee -a $OUT/ab_b8.txt
timeout 900 python -m pytest tests/test_gpu_plans.py -m gpu -q > $OUT/pytest_rest.log 2>&1; echo "network + plans + multidevice rc=$?"; tail -4 $OUT/pytest_rest.log
