#!/bin/bash
# round 5, session 2 (GPU box): the new hazard suite + the multi-device / two-rank tests, the wino_prio A-B arm, and the 16-bit plan's evidence regenerated BEFORE its kernels change
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hazards.py tests/test_multidevice.py tests/test_gpu_rccl.py -m gpu -q -x --durations=8 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/hazards.txt
for arm in wino_prio=1 wino_prio=2; do timeout 200 python tools/ab_tune.py $arm large 1 f32 2>&1 | grep -v amdgpu.ids; done | tee $OUT/prio_ab.txt
timeout 200 python tools/ab_tune.py wino_prio=1 large 8 f32 2>&1 | grep -v amdgpu.ids | tee -a $OUT/prio_ab.txt
timeout 300 python tools/layer_table.py normal 8 bf16 2>&1 | grep -v amdgpu.ids > $OUT/layers_normal_b8_bf16.txt
timeout 900 bash tools/collect_profiles.sh r5s2/prof "normal_b8_bf16" > $OUT/collect.log 2>&1
ls $OUT $OUT/prof | head -40
