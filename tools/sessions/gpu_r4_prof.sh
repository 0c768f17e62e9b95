#!/bin/bash
# round-4 profile collection: bench lines of every quoted configuration + rocprofv3 kernel-trace stats + PMC passes + the InstanceNorm class tables
cd $GRAFT_REPO_ROOT
bash tools/bench_all.sh r04_bench_all 2>&1 | tail -12
for cfg in "large 8 f16" "normal 8 f16"; do set -- $cfg; python bench.py --variant $1 --batch $2 --dtype $3 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_all/$1_b$2_$3.json 2>/dev/null; done
bash tools/collect_profiles.sh r04_profiles "large_b1_f32 large_b8_f32" 2>&1 | tail -5
# keep the merge small: the raw traces stay on the box, the summaries travel
python - <<'PY'
import glob, os, subprocess, sys
R = os.environ["GRAFT_REPO_ROOT"]
for db in glob.glob(R + "/gpurun_out/r04_profiles/trace_*/**/t_results.db", recursive=True) + glob.glob(R + "/gpurun_out/r04_profiles/trace_*/t_results.db"):
    cfg = db.split("trace_")[1].split("/")[0]
    out = open(R + "/gpurun_out/r04_profiles/kernel_stats_%s.txt" % cfg, "w")
    subprocess.call([sys.executable, R + "/tools/rocprof_summary.py", db], stdout=out)
PY
for cfg in large_b1_f32 large_b8_f32; do
  nf=$(python - gpurun_out/r04_profiles/pmc_$cfg/pmc_fetch <<'PY'
import csv, glob, sys
n = 0
for p in glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True):
    n += sum(1 for r in csv.DictReader(open(p)) if "first_conv" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE")
print(max(n, 1))
PY
)
  python tools/pmc_summary.py gpurun_out/r04_profiles/pmc_$cfg --forwards "$nf" --json gpurun_out/r04_profiles/pmc_$cfg.json --label "bench.py $cfg" > gpurun_out/r04_profiles/pmc_$cfg.txt
done
rm -rf gpurun_out/r04_profiles/trace_* gpurun_out/r04_profiles/pmc_*/pmc_*
for v in normal large; do for b in 1 8; do python tools/in_bench.py $v $b 2>/dev/null; done; done > gpurun_out/r04_profiles/instance_norm_bench.txt
python tools/layer_table.py large 1 f32 2>/dev/null > gpurun_out/r04_profiles/layers_large_b1_f32.txt
python tools/layer_table.py large 8 f32 2>/dev/null > gpurun_out/r04_profiles/layers_large_b8_f32.txt
ls gpurun_out/r04_profiles; head -12 gpurun_out/r04_profiles/kernel_stats_large_b1_f32.txt; head -8 gpurun_out/r04_profiles/pmc_large_b1_f32.txt
