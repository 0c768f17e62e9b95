#!/bin/bash
# round-3 GPU session 28: the K-split full-K kernel in the plan (the switch is now read before the plan is made), A-B-A-B
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s28; mkdir -p $OUT
LSP_HIP_FULLK_SPLIT=1 timeout 300 python tools/layer_table.py large 1 2>/dev/null | grep -E "L5.d.res0.a|L6.up|sum" | tee $OUT/layers_split.txt
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
def us(k):
    return ' '.join('%s x%d %.2f' % (n.split('<')[0][:14], c['launches'], c['us_per_launch']) for n,c in pc.items() if n.startswith(k))
print('%-28s %8.1f fps %8.4f ms | %s' % ('$1', d['value'], d['ms_per_step'], us('conv3x3_fullk')))"; }
for i in 1 2 3; do
  LSP_HIP_FULLK_SPLIT=1 run "b1 K split"
  LSP_HIP_FULLK_SPLIT=0 run "b1 no split"
done | tee $OUT/ab.txt
LSP_HIP_FULLK_SPLIT=1 run "normal b1 K split" "--variant normal"
LSP_HIP_FULLK_SPLIT=0 run "normal b1 no split" "--variant normal"
LSP_HIP_FULLK_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -m gpu -q -x 2>&1 | tail -2
