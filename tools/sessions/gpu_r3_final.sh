#!/bin/bash
# round-3 closing session: whole GPU suite, smoke, the profile collection of tools/sessions/gpu_r3_prof.sh, then the side tables (InstanceNorm plans, rasteriser, per-layer times)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3final; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/sessions/gpu_r3_prof.sh
for cfg in "normal 1" "normal 8" "large 1" "large 8"; do
  set -- $cfg
  timeout 300 python tools/in_bench.py $1 $2 2>/dev/null | tail -14
done > $OUT/in_bench.txt; grep InstanceNorm $OUT/in_bench.txt
timeout 120 python tools/time_raster.py 2>&1 | grep -v amdgpu.ids | tee $OUT/raster_time.txt
timeout 300 python tools/layer_table.py large 1 2>/dev/null > $OUT/layers_large_b1_f32.txt; tail -3 $OUT/layers_large_b1_f32.txt
