#!/bin/bash
# round-5 opener 3 (GPU box): HIP-runtime settings that touch what a batch-1 forward is made of besides kernels -- 79 dependent kernel boundaries (~3 us each behind
# 2-17 MB of dirty output, the guide's `boundary` row) and their kernarg fetches.  Each is a process-level environment variable of libamdhip64 (strings of the
# library, ROCm 7.2); the same bench command under each, one box, default first and last.  Anything that wins becomes a documented platform setting of bench.py /
# INTEGRATION.md (like HSA_ENABLE_IPC_MODE_LEGACY), not a library switch.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5env; mkdir -p $OUT
run() {   # name, env assignments...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-extra --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('%-44s %s' % ('$name', 'FAILED' if d is None else '%.1f frames/s  %.4f ms  sum of classes %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['sum_of_classes_ms'])))"
}
{
run "default" LSP_R5=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "AMD_OPT_FLUSH=0" AMD_OPT_FLUSH=0
run "AMD_OPT_FLUSH=1" AMD_OPT_FLUSH=1
run "AMD_OPT_FLUSH=3" AMD_OPT_FLUSH=3
run "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0
run "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1
run "default (again)" LSP_R5=0
} | tee $OUT/env_ab.txt
