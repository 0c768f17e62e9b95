#!/bin/bash
# round-6 closing session (GPU box): the whole GPU suite + smoke() at HEAD, then the evidence -- bench lines of every quoted configuration, rocprofv3 kernel stats + class tables + PMC of the
# three profiled configurations, per-layer tables
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r6final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc; tail -6 $OUT/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $OUT/smoke.log
timeout 1500 bash tools/bench_all.sh r6final/bench > $OUT/bench_all.txt 2>&1; tail -12 $OUT/bench_all.txt
for cfg in "large 1 f32" "large 8 f32" "normal 8 bf16"; do set -- $cfg; timeout 300 python tools/layer_table.py $1 $2 $3 2>&1 | grep -v amdgpu.ids > $OUT/layers_$1_b$2_$3.txt; done
timeout 1500 bash tools/collect_profiles.sh r6final/prof "large_b1_f32 large_b8_f32 normal_b8_bf16" > $OUT/collect.log 2>&1
for v in "normal 1" "normal 8" "large 1" "large 8"; do timeout 200 python tools/in_bench.py $v 2>&1 | grep -v amdgpu.ids | tail -14; done > $OUT/in_bench.txt
ls $OUT $OUT/prof | head -60
