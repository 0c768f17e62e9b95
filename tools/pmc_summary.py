#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (tools/collect_profiles.sh) per kernel: counter sums per forward.
  python tools/pmc_summary.py gpurun_out/profiles_raw --forwards 8
FETCH_SIZE/WRITE_SIZE are in KB as reported by rocprofv3; per MI355X_MICROARCH.md (HBM section)
FETCH_SIZE under-counts wide (16 B/lane) streaming reads by exactly 2x on gfx950, so the
'fetch x2' column is the corrected upper estimate for those kernels."""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--forwards", type=int, default=8, help="forwards in each PMC run (warmup+steps+5 timed passes)")
    ap.add_argument("--json", default=None, help="also write per-forward HBM-side bytes per kernel family (read by bench.py)")
    ap.add_argument("--label", default="bench.py large batch 1 fp32")
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for path in sorted(glob.glob(os.path.join(a.root, "pmc_*", "pmc_counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("lspf2f::", "").replace("(lspf2f::IgemmParams)", "").replace("void ", "")
            k = k.split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if "sq1" in path and (r["Dispatch_Id"], r["Counter_Name"]) not in seen and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                calls[k] += 1
    names = sorted({c for d in agg.values() for c in d})
    print("# PMC sums per forward (%d forwards per run); FETCH/WRITE in MB" % a.forwards)
    tot_f = tot_w = 0.0
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) - kv[1].get("FETCH_SIZE", 0)):
        f = d.get("FETCH_SIZE", 0) / 1024 / a.forwards
        w = d.get("WRITE_SIZE", 0) / 1024 / a.forwards
        tot_f += f
        tot_w += w
        hit = d.get("TCC_HIT_sum", 0)
        miss = d.get("TCC_MISS_sum", 0)
        mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / a.forwards
        print("%-44s calls/fwd %5.1f  fetch %8.1f MB (x2: %8.1f)  write %7.1f MB  L2 hit %5.1f%%  mfma_busy %.3e cyc  lds_bank_conflict %.3g  lds_unaligned %.3g"
              % (k[:44], calls[k] / a.forwards, f, 2 * f, w, 100 * hit / max(hit + miss, 1), mf,
                 d.get("SQ_LDS_BANK_CONFLICT", 0) / a.forwards, d.get("SQ_LDS_UNALIGNED_STALL", 0) / a.forwards))
    print("# total per forward: fetch %.1f MB (x2 %.1f MB), write %.1f MB" % (tot_f, 2 * tot_f, tot_w))
    if a.json:
        import json
        fam = {"conv_family": ("igemm3x3", "wino3x3", "winoup3x3", "splitk_reduce", "conv3x3_smallm", "conv3x3_fullk", "rowconv", "rowup", "bandconv"), "igemm3x3": ("igemm3x3",),
               "wino3x3": ("wino3x3",), "wino3x3<1>": ("wino3x3<1,",), "wino3x3<2>": ("wino3x3<2,",), "winoup3x3": ("winoup3x3",), "rowconv": ("rowconv", "rowup"), "bandconv": ("bandconv",),
               "splitk_reduce": ("splitk_reduce",), "conv3x3_fullk": ("conv3x3_fullk",), "conv3x3_smallm": ("conv3x3_smallm",),
               "first_conv": ("first_conv",), "last_conv": ("last_conv", "pixel_shuffle", "rowlast")}
        out = {}
        for name, prefixes in fam.items():
            fr = sum(d.get("FETCH_SIZE", 0) for k, d in agg.items() if k.startswith(prefixes)) * 1024 / a.forwards
            wr = sum(d.get("WRITE_SIZE", 0) for k, d in agg.items() if k.startswith(prefixes)) * 1024 / a.forwards
            out[name] = {"fetch_raw": fr, "fetch_x2": 2 * fr, "write": wr}
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), %s, %d forwards per pass; tools/collect_profiles.sh + tools/pmc_summary.py --json" % (a.label, a.forwards),
                   "correction": "FETCH_SIZE doubled for 16-B/lane streaming reads on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected",
                   "per_forward_bytes": out}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
