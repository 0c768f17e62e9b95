#!/bin/bash
# in-graph kernel durations (rocprofv3 kernel trace of the whole forward, cold weights) with the UR forms on / off; the four winoup launches of a forward apart
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4ureg; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra --steps 40 --warmup 5"
rocprofv3 --kernel-trace --stats -d $OUT/trace_ureg -o t -- python $R/bench.py $ARGS > $OUT/trace_ureg.log 2>&1
LSP_HIP_WINO_UREG=0 rocprofv3 --kernel-trace --stats -d $OUT/trace_lds -o t -- python $R/bench.py $ARGS > $OUT/trace_lds.log 2>&1
for a in ureg lds; do
  db=$(find $OUT/trace_$a -name "t_results.db" | head -1)
  python $R/tools/rocprof_summary.py $db --by-grid > $OUT/kernel_stats_$a.txt
  echo "== $a"; grep -i "wino" $OUT/kernel_stats_$a.txt | cut -c1-140
  python - $db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
for key in ("winoup3x3", "wino3x3<1"):
    d = [(e - s) / 1e3 for n, s, e in rows if key in n]
    per = 4 if key == "winoup3x3" else 24
    d = d[len(d) % per:]
    n = len(d) // per
    d = d[(n // 2) * per:]                      # second half of the run (graph replays)
    n = len(d) // per
    print(key, "per position in the forward (us, mean of %d forwards):" % n, " ".join("%.1f" % (sum(d[i::per]) / n) for i in range(per)))
PY
done
rm -rf $OUT/trace_ureg $OUT/trace_lds
