#!/bin/bash
# full-K kernel for the 4x4 / 2x2 / 8x8 levels at batch > 1: parity (batch-8 tests, plan tests, InstanceNorm), then the bench lines at batch 2, 4, 8
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4fullkb; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py tests/test_instance_norm.py tests/test_multidevice.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
for b in 2 4 8; do python bench.py --no-cpu-baseline --no-extra --steps 50 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('large b$b %8.1f fps %8.4f ms | %s' % (d['value'], d['ms_per_step'], ' '.join('%s x%d %.1f' % (c['kernel'][:16], c['launches'], c['us_per_launch']) for c in d['roofline']['per_class'] if c['kernel'].startswith(('conv3x3_fullk','igemm','splitk','conv3x3_smallm')))))"; done | tee $OUT/bench.txt
python bench.py --variant normal --no-cpu-baseline --no-extra --steps 50 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('normal b8 %8.1f fps %8.4f ms' % (d['value'], d['ms_per_step']))" | tee -a $OUT/bench.txt
