#!/bin/bash
# FETCH_SIZE / L2 hit per LSP_HIP_XCD mode (separate rocprofv3 --pmc passes, no trace domains).
# Usage: tools/xcd_fetch.sh <outdir> [bench args...]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for m in ${XCD_MODES:-0 1 2 auto}; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    if [ $m = auto ]; then unset LSP_HIP_XCD; else export LSP_HIP_XCD=$m; fi
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/${tag}_$m" -o pmc -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" > "$OUT/${tag}_$m.log" 2>&1
  done
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for m in os.environ.get("XCD_MODES", "0 1 2 auto").split():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for tag in ("FETCH_SIZE", "TCC_HIT_sum"):
        for path in glob.glob(os.path.join(out, "%s_%s" % (tag, m), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lspf2f::", "")
                fam = "igemm3x3" if k.startswith("igemm3x3") else k.split("<")[0]
                agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    tot = sum(d["FETCH_SIZE"] for d in agg.values())
    print("mode %s  total FETCH_SIZE %.1f MB (raw, all forwards of the run)" % (m, tot / 1024))
    for fam, d in sorted(agg.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])[:6]:
        h, ms = d["TCC_HIT_sum"], d["TCC_MISS_sum"]
        print("   %-22s fetch %9.1f MB   L2 hit %5.1f %%" % (fam, d["FETCH_SIZE"] / 1024, 100 * h / max(h + ms, 1)))
PY
