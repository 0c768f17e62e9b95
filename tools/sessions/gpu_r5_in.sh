#!/bin/bash
# round-5 session: InstanceNorm plans -- in_small with its rows resident in registers, the tiny-M kernel normalising in its own epilogue, the one-launch route only up to 256 pixels;
# tests first (goldens, the arms), then each arm A-B-A-B against the default in one process (large / normal, one frame), then the bench table
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5in; mkdir -p $OUT
timeout 900 python -m pytest tests/test_instance_norm.py -m gpu -q -s -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "max-abs vs the reference module|passed|failed|Error" $OUT/pytest.log | tail -30
for arm in in_small_regs=0 in_smallm_fused=0 in_small_max_hw=256 in_small_regs=0,in_smallm_fused=0; do
  for cfg in "large 1" "normal 1"; do set -- $cfg
    timeout 300 python tools/ab_tune.py $arm $1 $2 f32 3 instance 2>&1 | grep -v amdgpu.ids
  done
done | tee $OUT/ab.txt
for v in "normal 1" "normal 8" "large 1" "large 8"; do timeout 200 python tools/in_bench.py $v 2>&1 | grep -v amdgpu.ids | tail -14; done > $OUT/in_bench.txt; grep "frames/s" $OUT/in_bench.txt
