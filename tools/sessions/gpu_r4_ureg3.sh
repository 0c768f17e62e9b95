#!/bin/bash
# winoup3x3 UR form: parity on the conv shapes and goldens, A-B-A-B against the LDS forms (LSP_HIP_WINO_UREG=0 switches both Winograd kernels back)
# (record of a session: the register forms of winoup3x3 and the tiles 5003-5006 lived in that session's working tree only -- profiles/r04_winoup_ureg_ab.txt; the script still runs, against the shipped kernels)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4ureg; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -s -k "winograd_upconv" > $OUT/pytest_upconv.log 2>&1; echo "upconv tests rc=$?"; grep "^winoup (.*, [34], [0-9], " $OUT/pytest_upconv.log | head -14; tail -2 $OUT/pytest_upconv.log
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_instance_norm.py -m gpu -x -q -k "golden or batch8" > $OUT/pytest_net2.log 2>&1; echo "network tests rc=$?"; tail -2 $OUT/pytest_net2.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-10s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us f %.3f' % (k[:12], c['launches'], c['us_per_launch'], c['frac_mfma']) for k,c in pc.items() if k.startswith('wino'))))"; }
for b in 1 8; do for i in 1 2; do
  run "ureg" $b
  LSP_HIP_WINO_UREG=0 run "lds" $b
done; done 2>&1 | tee $OUT/ab_up.txt
