#!/bin/bash
# round-3 GPU session 20: rasteriser with its steps dealt evenly to the threads (bit-exactness, launch time); where the matrix-core last conv's time goes
# (timing-only builds of tools/sessions/lastconv_ablate.sh swapped over the scratch copy's library)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s20; mkdir -p $OUT
timeout 600 python -m pytest tests/test_raster.py -m gpu -q > $OUT/raster_tests.log 2>&1; echo "raster rc=$?"; tail -4 $OUT/raster_tests.log
timeout 120 python tools/time_raster.py 2>&1 | grep -v amdgpu.ids | tee $OUT/raster_time.txt
last() { python bench.py --no-cpu-baseline --no-extra --steps 50 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-28s %8.1f fps  last_conv %.2f us' % ('$1', d['value'], pc['last_conv']['us_per_launch']))"; }
cp livespeechportraits_amd/liblspf2f.so /tmp/shipped.so
for v in shipped NOMFMA NODMA PF2; do
  if [ $v = shipped ]; then cp /tmp/shipped.so livespeechportraits_amd/liblspf2f.so; else cp tools/ablate_builds/liblspf2f_$v.so livespeechportraits_amd/liblspf2f.so; fi
  last "b1 $v"
  last "b8 $v" "--batch 8 --steps 20"
done | tee $OUT/lastconv_ablation.txt
cp /tmp/shipped.so livespeechportraits_amd/liblspf2f.so
