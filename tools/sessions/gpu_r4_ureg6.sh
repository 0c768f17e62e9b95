#!/bin/bash
# winoup3x3 register forms (2 | 3 sets) against the LDS form: parity, whole-forward A-B, in-graph kernel durations from a kernel trace
# (record of a session: the `winoup_ureg` key and the register forms of winoup3x3 lived in that session's working tree only -- profiles/r04_winoup_ureg_ab.txt; with the shipped library the three arms are the same kernel)
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4ureg; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "winograd_upconv" > $OUT/pytest_upconv.log 2>&1; echo "upconv tests rc=$?"; tail -2 $OUT/pytest_upconv.log
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-10s b%s %8.1f fps %8.4f ms (sum of classes %.4f) | %s' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['sum_of_classes_ms'], ' '.join('%s x%d %.2f us' % (k[:12], c['launches'], c['us_per_launch']) for k,c in pc.items() if k.startswith('winoup'))))"; }
for b in 1 8; do for i in 1 2; do
  LSP_HIP_WINOUP_UREG=0 run "up-lds" $b
  LSP_HIP_WINOUP_UREG=2 run "up-reg2" $b
  LSP_HIP_WINOUP_UREG=3 run "up-reg3" $b
done; done 2>&1 | tee $OUT/ab_up2.txt
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra --steps 40 --warmup 5"
for a in 0 2 3; do
  LSP_HIP_WINOUP_UREG=$a rocprofv3 --kernel-trace --stats -d $OUT/trace_up$a -o t -- python $R/bench.py $ARGS > $OUT/trace_up$a.log 2>&1
  db=$(find $OUT/trace_up$a -name "t_results.db" | head -1)
  python - $db $a <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if "winoup3x3" in n]
d = d[len(d) % 4:]; n = len(d) // 4; d = d[(n // 2) * 4:]; n = len(d) // 4
print("winoup_ureg=%s: winoup3x3 per position in the forward (us, mean of %d traced forwards): %s" % (sys.argv[2], n, " ".join("%.1f" % (sum(d[i::4]) / n) for i in range(4))))
PY
  rm -rf $OUT/trace_up$a
done 2>&1 | tee $OUT/trace_up.txt
