#!/bin/bash
# round-4 GPU session 1: Winograd F(4x4,3x3) kernel -- per-kernel parity, network goldens through it, whole-forward A-B against F(2x2,3x3), per-layer table
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4s1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -s -k "winograd4" > $OUT/pytest_wino4.log 2>&1; echo "wino4 conv tests rc=$?"; grep "^wino4\|passed\|failed\|Error\|error" $OUT/pytest_wino4.log | head -40
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -s -k "golden or live" > $OUT/pytest_net.log 2>&1; echo "network tests rc=$?"; grep "max-abs\|passed\|failed" $OUT/pytest_net.log | head -20
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --batch $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc={c['kernel']:c for c in d['roofline']['per_class']}
print('%-14s b%s %8.1f fps %8.4f ms | %s' % ('$1', '$2', d['value'], d['ms_per_step'], ' '.join('%s x%d %.1f' % (k[:14], c['launches'], c['us_per_launch']) for k,c in pc.items() if k.startswith('wino'))))"; }
for i in 1 2; do
  run "wino4 on" 1
  LSP_HIP_WINO4=0 run "wino4 off" 1
done 2>&1 | tee $OUT/ab_b1.txt
run "wino4 on" 8 2>&1 | tee $OUT/ab_b8.txt
LSP_HIP_WINO4=0 run "wino4 off" 8 2>&1 | tee -a $OUT/ab_b8.txt
timeout 300 python tools/layer_table.py large 1 f32 > $OUT/layers_wino4_b1.txt 2>&1; grep "wino\|sum" $OUT/layers_wino4_b1.txt | head -50
LSP_HIP_WINO4=0 timeout 300 python tools/layer_table.py large 1 f32 > $OUT/layers_wino2_b1.txt 2>&1; grep "wino3\|sum" $OUT/layers_wino2_b1.txt | head -40
