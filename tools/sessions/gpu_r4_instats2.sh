#!/bin/bash
# InstanceNorm plans: winoup3x3 leaves its statistics too -- parity, the class lines, and the BatchNorm default bench (its epilogue carries the new code behind a null pointer)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4instats2; mkdir -p $OUT
timeout 300 python -m pytest tests/test_instance_norm.py -m gpu -q -s > $OUT/pytest.log 2>&1; echo "IN tests rc=$?"; grep "max-abs vs the reference module\|passed\|failed" $OUT/pytest.log | cut -c1-110
for b in 1 8; do python tools/in_bench.py large $b 2>/dev/null | head -1 | cut -c1-100; done | tee $OUT/bench.txt
python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BatchNorm default: %.1f fps | %s' % (d['value'], ' '.join('%s %.2f' % (c['kernel'][:12], c['us_per_launch']) for c in d['roofline']['per_class'] if c['kernel'].startswith('winoup'))))" | tee -a $OUT/bench.txt
