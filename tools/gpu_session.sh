#!/bin/bash
# One gpurun call = tests + bench + kernel traces.  Usage (on the GPU box): tools/gpu_session.sh <tag> [tests|notests]
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-s}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
if [ "${2:-tests}" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
  tail -5 "$OUT/pytest.log"
fi
timeout 600 python bench.py > "$OUT/bench_b1.json" 2> "$OUT/bench_b1.err"; echo "bench rc=$?"
cut -c1-400 "$OUT/bench_b1.json"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_b1" -o t -- python "$R/bench.py" --no-cpu-baseline --no-extra --layers "$OUT/layers_b1.txt" > "$OUT/trace_b1.log" 2>&1
python "$R/tools/rocprof_summary.py" "$OUT"/trace_b1/*/t_results.db --frames 65 > "$OUT/kernel_stats_large_b1.txt" 2>&1 || python "$R/tools/rocprof_summary.py" "$OUT"/trace_b1/t_results.db --frames 65 > "$OUT/kernel_stats_large_b1.txt" 2>&1
head -30 "$OUT/kernel_stats_large_b1.txt"
