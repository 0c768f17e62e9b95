#!/bin/bash
# last conv, K split over the waves (route 6): parity on the goldens (fp32 + uint8 output), then A-B against the shipped eight-wave kernel
# (record of a session: the K-split kernel `last_conv_ks` / route 6 it drives is not in the library; its source is archived, not built, in tools/sessions/experiments/last_conv_experiments.inc -- profiles/r04_lastconv_ab.txt)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4lcks; mkdir -p $OUT
LSP_HIP_LASTCONV=6 timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "golden and (large_512 or normal_512) or uint8 or batch8" > $OUT/pytest.log 2>&1; echo "tests (route 6) rc=$?"; tail -3 $OUT/pytest.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-8s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us hbm %.3f' % (k[:12], c['launches'], c['us_per_launch'], c['frac_hbm']) for k,c in pc.items() if k.startswith('last') or k.startswith('first'))))"; }
for b in 1 8; do for i in 1 2; do
  run "8-wave" $b
  LSP_HIP_LASTCONV=6 run "k-split" $b
done; done 2>&1 | tee $OUT/ab.txt
