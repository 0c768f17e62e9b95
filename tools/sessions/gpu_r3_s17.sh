#!/bin/bash
# round-3 GPU session 17: next-launch weight prefetch in conv3x3_smallm and the LDS-DMA staged first conv -- parity, then A-B timings with the class table;
# layout probe of v_mfma_f32_4x4x1_16b_f32 for the last-conv rewrite
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s17; mkdir -p $OUT
tools/probes/mfma4x4_probe 2>&1 | grep -v amdgpu.ids | head -70
timeout 900 python -m pytest tests/test_gpu_plans.py tests/test_gpu_network.py tests/test_gpu_conv.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
def us(k):
    return ' '.join('%s %.2f' % (n.split('<')[0][:14], c['us_per_launch']) for n,c in pc.items() if n.startswith(k))
print('%-34s %8.1f fps %8.4f ms | %s | %s | %s | %s' % ('$1', d['value'], d['ms_per_step'], us('first'), us('conv3x3_smallm'), us('conv3x3_fullk'), us('last')))"; }
for i in 1 2; do
  run "b1 default"
  LSP_HIP_PREFETCH=0 run "b1 no prefetch"
  LSP_HIP_FIRSTCONV_REGSTAGE=1 run "b1 register-staged first conv"
done
run "b8 default" "--batch 8 --steps 30"
LSP_HIP_FIRSTCONV_REGSTAGE=1 run "b8 register-staged first conv" "--batch 8 --steps 30"
run "normal b1 default" "--variant normal"
LSP_HIP_PREFETCH=0 run "normal b1 no prefetch" "--variant normal"
