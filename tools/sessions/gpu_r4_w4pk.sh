#!/bin/bash
# Winograd F(4x4,3x3) with packed transform arithmetic against F(2x2,3x3) (register form): parity of the opt-in route, then A-B-A-B at batch 1 and 8
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4w4pk; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_network.py -m gpu -x -q -k "winograd4" > $OUT/pytest.log 2>&1; echo "F(4x4) tests rc=$?"; tail -2 $OUT/pytest.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-6s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us' % (k[:12], c['launches'], c['us_per_launch']) for k,c in pc.items() if k.startswith('wino3') or k.startswith('wino4'))))"; }
for b in 1 8; do for i in 1 2; do
  run "F2x2" $b
  LSP_HIP_WINO4=1 run "F4x4" $b
done; done 2>&1 | tee $OUT/ab.txt
