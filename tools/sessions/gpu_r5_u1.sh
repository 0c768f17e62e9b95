#!/bin/bash
# round-5 session: the native `small` U-Net plan (include/lspunet.h) -- its GPU tests, timing against the host-sequenced form with every A-B arm and a per-launch table --
# and the default bench line (the implicit-GEMM kernels were rebuilt with three more kernarg fields: the shipped instances must be where they were)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5u1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_unet_small.py -m gpu -x -q > $OUT/pytest_unet.log 2>&1; echo "pytest unet rc=$?"; tail -12 $OUT/pytest_unet.log
timeout 600 python tools/unet_small_time.py 2>&1 | grep -v amdgpu.ids > $OUT/unet_small_time.txt; echo "time rc=$?"; cat $OUT/unet_small_time.txt
timeout 400 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; python - <<'P'
import json
d=json.load(open("gpurun_out/r5u1/bench_default.json"))
print("bench default:", d["value"], d["unit"], "ms/step", d["ms_per_step"], "cfg2", d.get("extra",{}).get("config2_normal_b8_bf16"))
P
