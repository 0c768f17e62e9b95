#!/bin/bash
# round-3 GPU session 29: K split of the full-K kernel extended to the 16x16 layers (32-pixel tiles) -- parity, A-B-A-B-A-B against the 8x8-only rule --
# then the bench lines of every configuration and the batch-1 trace / PMC passes under whichever rule won (printed; the library default is set to it)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s29; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "k_split" > $OUT/split_tests.log 2>&1; echo "K-split tests rc=$?"; tail -3 $OUT/split_tests.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-22s %8.1f fps %8.4f ms | fullk x%d %.2f us' % ('$1', d['value'], d['ms_per_step'], pc['conv3x3_fullk']['launches'], pc['conv3x3_fullk']['us_per_launch']))"; }
for i in 1 2 3; do
  LSP_HIP_FULLK_SPLIT_TILES=256 run "tiles<=256"
  LSP_HIP_FULLK_SPLIT_TILES=128 run "tiles<=128"
done | tee $OUT/ab.txt
CHOICE=$(python - $OUT/ab.txt <<'PY'
import sys
a, b = [], []
for line in open(sys.argv[1]):
    v = float(line.split()[1])
    (a if line.startswith("tiles<=256") else b).append(v)
print(256 if sum(a) / len(a) > 1.004 * sum(b) / len(b) else 128)
PY
)
echo "rule chosen: tiles <= $CHOICE"
export LSP_HIP_FULLK_SPLIT_TILES=$CHOICE
LSP_HIP_FULLK_SPLIT_TILES=$CHOICE timeout 300 python -m pytest tests/test_gpu_network.py -m gpu -q -k "golden or batch8" 2>&1 | tail -1
bash tools/bench_all.sh r03_bench_all 2>&1 | tail -9
bash tools/collect_profiles.sh r03_profiles "large_b1_f32" 2>&1 | tail -2
python - <<'PY'
import glob, os, subprocess, sys
R = os.environ["GRAFT_REPO_ROOT"]
for db in glob.glob(R + "/gpurun_out/r03_profiles/trace_*/**/t_results.db", recursive=True) + glob.glob(R + "/gpurun_out/r03_profiles/trace_*/t_results.db"):
    cfg = db.split("trace_")[1].split("/")[0]
    out = open(R + "/gpurun_out/r03_profiles/kernel_stats_%s.txt" % cfg, "w")
    subprocess.call([sys.executable, R + "/tools/rocprof_summary.py", db], stdout=out)
PY
cfg=large_b1_f32
nf=$(python - gpurun_out/r03_profiles/pmc_$cfg/pmc_fetch <<'PY'
import csv, glob, sys
n = 0
for p in glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True):
    n += sum(1 for r in csv.DictReader(open(p)) if "first_conv" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE")
print(max(n, 1))
PY
)
python tools/pmc_summary.py gpurun_out/r03_profiles/pmc_$cfg --forwards "$nf" --json gpurun_out/r03_profiles/pmc_$cfg.json --label "bench.py $cfg" > gpurun_out/r03_profiles/pmc_$cfg.txt
rm -rf gpurun_out/r03_profiles/trace_* gpurun_out/r03_profiles/pmc_*/pmc_*
head -14 gpurun_out/r03_profiles/kernel_stats_large_b1_f32.txt
