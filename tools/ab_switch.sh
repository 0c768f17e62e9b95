#!/bin/bash
# A-B-A-B of one library switch inside one session (GPU box): tools/ab_switch.sh <ENV_NAME> <variant> <batch> <dtype> [off-value [on-value]]
# e.g. tools/ab_switch.sh LSP_HIP_ROWCONV normal 8 bf16     -> frames/s and ms per step with the switch at 0 / unset, twice each.
# The switches are read once per handle at create, so every arm is a fresh bench.py process.
set -u
NAME=$1; VAR=$2; B=$3; DT=$4; OFF=${5:-0}; ON=${6:-}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for arm in off on off on; do
  if [ $arm = off ]; then export $NAME=$OFF; elif [ -n "$ON" ]; then export $NAME=$ON; else unset $NAME; fi
  python bench.py --variant $VAR --batch $B --dtype $DT --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$NAME %-3s  %9.1f frames/s  %8.4f ms' % ('$arm', d['value'], d['ms_per_step']))"
done
