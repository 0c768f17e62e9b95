#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_conv.py -k "upconv" -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  run "default (nb 2, target 512)"
  LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=1024 run "nb 1, target 1024"
  LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=512 run "nb 1, target 512"
  LSP_HIP_WINOUP_NB=2 LSP_HIP_WINOUP_TARGET=1024 run "nb 2, target 1024"
done
LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=1024 python tools/layer_table.py 2>/dev/null | grep -E "winoup"
