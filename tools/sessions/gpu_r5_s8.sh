#!/bin/bash
# round 5, session 8 (GPU box): rowlast128 with pixel shuffle + tanh in its epilogue (bit-identical to the two-launch form?), the long-K split rule of the 16-bit plans; configs[2] after both
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_plans.py tests/test_gpu_network.py tests/test_gpu_hazards.py -m gpu -q -x -k "bf16 or fp16 or f16 or rowlast or image or 16bit" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/parity.txt
for cfg in "normal 8 bf16" "large 8 bf16" "normal 8 f16" "normal 3 bf16"; do timeout 200 python tools/ab_tune.py rowlast_fused=0 $cfg 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ab.txt
timeout 300 python bench.py --variant normal --batch 8 --dtype bf16 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
d=json.loads(l[-1])
print('normal 8 bf16: %.1f frames/s %.4f ms | ' % (d['value'], d['ms_per_step']) + ' | '.join('%s x%d %.1f us' % (c['kernel'], c['launches'], c['ms']*1e3) for c in d['roofline']['per_class']))" | tee $OUT/bench.txt
