#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=1024 run "b1 nb 1, target 1024"
  LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=1536 run "b1 nb 1, target 1536"
  LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=2048 run "b1 nb 1, target 2048"
done
for i in 1 2; do
  LSP_HIP_WINOUP_NB=2 run "b8 nb 2" "--batch 8 --steps 30"
  LSP_HIP_WINOUP_NB=1 run "b8 nb 1" "--batch 8 --steps 30"
done
for v in normal; do
  LSP_HIP_WINOUP_NB=2 run "normal b1 nb 2 t512" "--variant normal"
  LSP_HIP_WINOUP_NB=1 LSP_HIP_WINOUP_TARGET=1024 run "normal b1 nb 1 t1024" "--variant normal"
done
