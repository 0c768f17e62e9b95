#!/bin/bash
# InstanceNorm plans: statistics from the wino3x3 epilogue (default) against the separate in_reduce_stats pass (LSP_HIP_IN_WINO_STATS=0): parity, then the class tables
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4instats; mkdir -p $OUT
timeout 900 python -m pytest tests/test_instance_norm.py -m gpu -q -s > $OUT/pytest.log 2>&1; echo "IN tests rc=$?"; grep "max-abs vs the reference module\|passed\|failed" $OUT/pytest.log | cut -c1-150
for v in large normal; do for b in 1 8; do
  python tools/in_bench.py $v $b 2>/dev/null | head -1
  LSP_HIP_IN_WINO_STATS=0 python tools/in_bench.py $v $b 2>/dev/null | head -1 | sed 's/^/   (separate pass) /'
done; done | tee $OUT/ab.txt
python tools/in_bench.py large 1 2>/dev/null > $OUT/classes_large_b1.txt
