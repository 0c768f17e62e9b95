#!/bin/bash
# round-3 GPU session 19: rasteriser in its thread-per-item form (bit-exactness, launch time), Winograd kernels under InstanceNorm plans (parity, throughput),
# issue rate of v_mfma_f32_4x4x1_16b_f32, the vector-ALU route of the last conv
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s19; mkdir -p $OUT
tools/probes/mfma_rate_probe 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_raster.py -m gpu -q > $OUT/raster_tests.log 2>&1; echo "raster rc=$?"; tail -4 $OUT/raster_tests.log
timeout 120 python tools/time_raster.py 2>&1 | grep -v amdgpu.ids | tee $OUT/raster_time.txt
timeout 600 python -m pytest tests/test_instance_norm.py -m gpu -q > $OUT/in_tests.log 2>&1; echo "instance-norm rc=$?"; tail -4 $OUT/in_tests.log
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -k "last_conv" > $OUT/last_tests.log 2>&1; echo "last-conv routes rc=$?"; tail -3 $OUT/last_tests.log
for cfg in "normal 1" "normal 8" "large 1" "large 8"; do
  set -- $cfg
  timeout 300 python tools/in_bench.py $1 $2 2>/dev/null | tail -14
done | tee $OUT/in_bench.txt
LSP_HIP_LASTCONV_VALU=1 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pc={c['kernel']:c for c in d['roofline']['per_class']}
print('b1 vector-ALU last conv', d['value'], [ (k, round(c['us_per_launch'],2)) for k,c in pc.items() if k.startswith('last')])"
python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('b1 default', d['value'], d['ms_per_step']); e=d.get('extra',{})
print({k:v for k,v in e.items() if 'pcie' in k or 'pipeline' in k})"
