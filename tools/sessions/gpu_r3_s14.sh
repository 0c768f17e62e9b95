#!/bin/bash
# round-3 GPU session 14: up-conv Winograd kernel -- parity, network tests, A-B (LSP_HIP_WINO=0 turns both Winograd kernels off)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3s14; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_conv.py -k "upconv" -q -s > $OUT/winoup_tests.log 2>&1; echo "winoup tests rc=$?"; grep -E "^winoup|passed|failed|Error" $OUT/winoup_tests.log | tail -16
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -q -x > $OUT/net.log 2>&1; echo "network tests rc=$?"; tail -3 $OUT/net.log
for i in 1 2; do
  LSP_HIP_WINOUP=0 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('winoup off', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('winoup on ', d['value'], d['ms_per_step'])"
done
for b in 8; do
  LSP_HIP_WINOUP=0 python bench.py --no-cpu-baseline --no-extra --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b8 winoup off', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-extra --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b8 winoup on ', d['value'], d['ms_per_step'])"
done
python bench.py --no-cpu-baseline --no-extra --layers $OUT/classes_b1.txt > /dev/null 2>&1; cat $OUT/classes_b1.txt
python tools/layer_table.py 2>/dev/null | grep -E "\.up|down " | head -20
