#!/bin/bash
# Build container: gpurun_out/<tag>/ (tools/collect_profiles.sh) -> tracked summaries under profiles/ named <round>_*.   tools/summarise_profiles.sh [tag] [round]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r03_profiles}
R=${2:-r03}
IN=gpurun_out/$TAG
for cfg in large_b1_f32 large_b8_f32 normal_b8_bf16; do
  [ -f "$IN/bench_$cfg.json" ] || continue
  db=$(ls $IN/trace_$cfg/*/t_results.db $IN/trace_$cfg/t_results.db 2>/dev/null | head -1)
  # forwards in the trace: warm-up 5 + steps 40 + the class-timing replays are excluded by counting last_conv launches
  n=$(python - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
print(c.execute("select count(*) from kernels where name like '%first_conv%' and name not like '%first_conv_feat%'").fetchone()[0])   # one per forward (the feature-map pass of a shared-candidate batch is a second kernel)
PY
)
  python tools/rocprof_summary.py "$db" > profiles/${R}_kernel_stats_$cfg.txt
  cp "$IN/classes_$cfg.txt" profiles/${R}_kernel_classes_$cfg.txt
  cp "$IN/bench_$cfg.json" profiles/${R}_bench_$cfg.json
  # PMC runs: bench.py --steps 2 --warmup 1 = 3 replays + 1 eager forward + 10 class replays + 1 warm-up each; normalise by the first_conv count
  nf=$(python - "$IN/pmc_$cfg/pmc_fetch" <<'PY'
import csv, glob, sys
n = 0
for p in glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True):
    n += sum(1 for r in csv.DictReader(open(p)) if "first_conv" in r["Kernel_Name"] and "first_conv_feat" not in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE")
print(max(n, 1))
PY
)
  python tools/pmc_summary.py "$IN/pmc_$cfg" --forwards "$nf" --json profiles/${R}_pmc_$cfg.json --label "bench.py $cfg" > profiles/${R}_pmc_$cfg.txt
done
ls profiles/ | grep ${R}
