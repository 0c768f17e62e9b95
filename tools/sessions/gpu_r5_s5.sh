#!/bin/bash
# round 5, session 5 (GPU box): first_conv_feat walking frames per pixel position (parity + its class time at 8 frames), the LDS broadcast probe (last-conv VALU form bound)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/parity.txt
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bcast_probe tools/probes/lds_bcast_probe.hip 2>&1 | tail -3 && timeout 120 /tmp/lds_bcast_probe ) 2>&1 | tee $OUT/lds_bcast_probe.txt
for cfg in "normal 8 bf16" "large 8 f32" "large 1 f32"; do
  set -- $cfg
  timeout 300 python bench.py --variant $1 --batch $2 --dtype $3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
d=json.loads(l[-1])
print('$cfg: %.1f frames/s %.4f ms | ' % (d['value'], d['ms_per_step']) + ' | '.join('%s x%d %.1f us' % (c['kernel'], c['launches'], c['ms']*1e3) for c in d['roofline']['per_class']))"
done | tee $OUT/bench.txt
