#!/bin/bash
# round-3 GPU session 25: K-split form of the full-K kernel for the 8x8 levels at batch 1 (parity, A-B)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s25; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "full_k" > $OUT/fullk_tests.log 2>&1; echo "full-K tests rc=$?"; tail -6 $OUT/fullk_tests.log
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py tests/test_instance_norm.py -m gpu -q -x > $OUT/net.log 2>&1; echo "network+plans+instance-norm rc=$?"; tail -4 $OUT/net.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
def us(k):
    return ' '.join('%s x%d %.2f' % (n.split('<')[0][:14], c['launches'], c['us_per_launch']) for n,c in pc.items() if n.startswith(k))
print('%-28s %8.1f fps %8.4f ms | %s | %s' % ('$1', d['value'], d['ms_per_step'], us('conv3x3_fullk'), us('conv3x3_smallm')))"; }
for i in 1 2 3; do
  run "b1 default (K split)"
  LSP_HIP_FULLK_SPLIT=0 run "b1 no split"
done | tee $OUT/ab.txt
run "normal b1 default" "--variant normal"
LSP_HIP_FULLK_SPLIT=0 run "normal b1 no split" "--variant normal"
