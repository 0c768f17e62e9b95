#!/bin/bash
# round 5, session 3 (GPU box): the wino_prio schemes (ProgressPrio, wino_common.h) -- fair / fair + tail skew, register form only or every Winograd loop
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5s3; mkdir -p $OUT
for arm in wino_prio=1 wino_prio=3 wino_prio=4 wino_prio=6; do timeout 200 python tools/ab_tune.py $arm large 1 f32 2>&1 | grep -v amdgpu.ids; done | tee $OUT/prio_ab.txt
for arm in wino_prio=4; do timeout 200 python tools/ab_tune.py $arm normal 1 f32 2>&1 | grep -v amdgpu.ids; timeout 200 python tools/ab_tune.py $arm large 8 f32 2>&1 | grep -v amdgpu.ids; done | tee -a $OUT/prio_ab.txt
