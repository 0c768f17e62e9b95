#!/bin/bash
# round-3 GPU session 1: Winograd kernel parity + timing, then the whole GPU suite and the bench with the kernel on / off
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s1; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k winograd -q -s > $OUT/wino_tests.log 2>&1; echo "wino tests rc=$?"; tail -25 $OUT/wino_tests.log
timeout 200 python tools/wino_debug.py > $OUT/wino_debug.log 2>&1; tail -60 $OUT/wino_debug.log
timeout 400 python tools/wino_sweep.py 1 8 > $OUT/sweep.log 2>&1; cat $OUT/sweep.log
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json; tail -5 $OUT/bench.err
LSP_HIP_WINO=0 timeout 300 python bench.py --no-cpu-baseline --no-extra > $OUT/bench_nowino.json 2>&1; cut -c1-300 $OUT/bench_nowino.json
timeout 300 python bench.py --no-cpu-baseline --no-extra --layers $OUT/classes_b1.txt > $OUT/bench_b1.json 2>&1; cat $OUT/classes_b1.txt
timeout 300 python bench.py --no-cpu-baseline --no-extra --batch 8 --layers $OUT/classes_b8.txt > $OUT/bench_b8.json 2>&1; cut -c1-300 $OUT/bench_b8.json; cat $OUT/classes_b8.txt
LSP_HIP_WINO=0 timeout 300 python bench.py --no-cpu-baseline --no-extra --batch 8 > $OUT/bench_b8_nowino.json 2>&1; cut -c1-300 $OUT/bench_b8_nowino.json
python tools/layer_table.py > $OUT/layers_b1.txt 2>&1; head -80 $OUT/layers_b1.txt
