#!/bin/bash
# round-4 GPU session 3: full GPU suite after blob slimming + MultiDeviceParallel; default bench line
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4s3; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s3/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "alg", r["algorithmic_frac"], "clock", r["clock_ghz_observed"], "whole", r["whole_forward"])
print({k: v for k, v in d.get("extra", {}).items() if not isinstance(v, dict)})
PY
