#!/bin/bash
# last conv on the vector ALU with scalar weights (route 7): parity on the goldens (fp32 + uint8 output, InstanceNorm bias), then A-B against the shipped eight-wave matrix-core kernel
# (record of a session: the scalar-weight kernel `last_conv_sv` / route 7 it drives is not in the library; its source is archived, not built, in tools/sessions/experiments/last_conv_experiments.inc -- profiles/r04_lastconv_ab.txt)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4lcsv; mkdir -p $OUT
LSP_HIP_LASTCONV=${LC_ROUTE:-7} timeout 600 python -m pytest tests/test_gpu_network.py tests/test_instance_norm.py -m gpu -x -q -k "(golden and (large_512 or normal_512)) or uint8 or batch8" > $OUT/pytest.log 2>&1; echo "tests (route ${LC_ROUTE:-7}) rc=$?"; tail -3 $OUT/pytest.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-8s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.2f us hbm %.3f' % (k[:12], c['launches'], c['us_per_launch'], c['frac_hbm']) for k,c in pc.items() if k.startswith('last') or k.startswith('first'))))"; }
for b in 1 8; do for i in 1 2; do
  run "mfma8" $b
  LSP_HIP_LASTCONV=${LC_ROUTE:-7} run "valu-s" $b
done; done 2>&1 | tee $OUT/ab.txt
