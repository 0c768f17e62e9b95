#!/bin/bash
# round 5, session 6 (GPU box): last_conv_vl (vector ALU, weights broadcast from LDS, 4 pixels per lane, K split over the waves) -- parity on the goldens (fp32 + uint8), A-B against last_conv_mfma<8>;
# the masked-K tile sweep
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s6; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -x -k "last_conv or image or tensor2im or u8" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/parity.txt
LSP_HIP_LASTCONV_VL=1 timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -x -k "large_512 or normal_512 or image or u8" 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $OUT/parity.txt
timeout 300 python -m pytest tests/test_unet_small.py -m gpu -q -x -k "masked" 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/parity.txt
for cfg in "large 1 f32" "large 8 f32"; do timeout 200 python tools/ab_tune.py lastconv=6 $cfg 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ab.txt
for r in 0 6; do LSP_HIP_LASTCONV=$r timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
d=json.loads(l[-1])
print('lastconv=$r large 1 f32: %.1f frames/s %.4f ms | ' % (d['value'], d['ms_per_step']) + ' | '.join('%s x%d %.1f us' % (c['kernel'], c['launches'], c['ms']*1e3) for c in d['roofline']['per_class'] if 'conv' in c['kernel'] and ('last' in c['kernel'] or 'first' in c['kernel'])))"; done | tee $OUT/bench.txt
