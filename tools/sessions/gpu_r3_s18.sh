#!/bin/bash
# round-3 GPU session 18: matrix-core last conv (parity on the whole suite, A-B against the vector-ALU kernels at batch 1 / 8), tiny-M kernel with its
# epilogue operands requested up front
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -m gpu -q -x > $OUT/pytest_net.log 2>&1; echo "network+plans rc=$?"; tail -4 $OUT/pytest_net.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
def us(k):
    return ' '.join('%s %.2f' % (n.split('<')[0][:14], c['us_per_launch']) for n,c in pc.items() if n.startswith(k))
print('%-34s %8.1f fps %8.4f ms | %s | %s | %s | %s' % ('$1', d['value'], d['ms_per_step'], us('first'), us('conv3x3_smallm'), us('conv3x3_fullk'), us('last')))"; }
for i in 1 2; do
  run "b1 default"
  LSP_HIP_LASTCONV_VALU=1 run "b1 vector-ALU last conv"
done
for i in 1 2; do
  run "b8 default" "--batch 8 --steps 30"
  LSP_HIP_LASTCONV_VALU=1 run "b8 vector-ALU last conv" "--batch 8 --steps 30"
done
run "normal b1 default" "--variant normal"
run "normal b8 default" "--variant normal --batch 8 --steps 30"
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "whole suite rc=$?"; tail -4 $OUT/pytest.log
