#!/bin/bash
# round-3 GPU session 30 (the last of the budget): stride-2 convs of the small levels on the K-split full-K kernel -- parity, golden, A-B
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s30; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "stride2 or k_split" 2>&1 | tail -3
timeout 200 python -m pytest tests/test_gpu_network.py -m gpu -q -k "golden" 2>&1 | tail -1
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-14s %8.1f fps %8.4f ms | fullk x%d %.2f us | %s' % ('$1', d['value'], d['ms_per_step'], pc['conv3x3_fullk']['launches'], pc['conv3x3_fullk']['us_per_launch'], ' '.join('%s x%d %.1f' % (k[:22], c['launches'], c['us_per_launch']) for k,c in pc.items() if k.startswith('igemm') or k.startswith('splitk'))))"; }
for i in 1 2; do
  run "s2 on"
  LSP_HIP_FULLK_S2=0 run "s2 off"
done | tee $OUT/ab.txt
