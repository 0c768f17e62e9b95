#!/bin/bash
# round-4 GPU session 2: full GPU suite after the switchboard refactor (no getenv in the library), bench line with executed fractions + observed clock
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4s2; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s2/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "alg", r["algorithmic_frac"], "clock", r["clock_ghz_observed"], "at clock", r["frac_at_observed_clock"], "whole", r["whole_forward"])
for c in r["per_class"]: print("  %-30s x%-3d %7.2f us  frac %.3f alg %.3f hbm %.3f" % (c["kernel"], c["launches"], c["us_per_launch"], c["frac_mfma"], c["algorithmic_mfma"], c["frac_hbm"]))
print({k: v for k, v in d.get("extra", {}).items() if not isinstance(v, dict)})
PY
