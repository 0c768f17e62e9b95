#!/bin/bash
# round-3 GPU session 23: rasteriser with its steps dealt evenly over 1024 threads (bit-exactness, launch time, phase stamps)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s23; mkdir -p $OUT
timeout 600 python -m pytest tests/test_raster.py -m gpu -q > $OUT/raster_tests.log 2>&1; echo "raster rc=$?"; tail -4 $OUT/raster_tests.log
timeout 120 python tools/time_raster.py 2>&1 | grep -v amdgpu.ids | tee $OUT/raster_time.txt
cp livespeechportraits_amd/liblspf2f.so /tmp/shipped.so
cp tools/ablate_builds/liblspf2f_RSTAMPS.so livespeechportraits_amd/liblspf2f.so
timeout 120 python tools/raster_stamps.py 2>&1 | grep -v amdgpu.ids | tee $OUT/raster_stamps.txt
cp /tmp/shipped.so livespeechportraits_amd/liblspf2f.so
python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('b1 default', d['value'], d['ms_per_step']); e=d.get('extra',{})
print({k:v for k,v in e.items() if 'pcie' in k})"
