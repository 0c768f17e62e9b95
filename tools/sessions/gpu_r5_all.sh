#!/bin/bash
# the three prepared openers of round 5 in one gpurun call (~10 min of box time): out_wt A-B, runtime-environment A-B, K-loop tail by placement (rebuilds the library twice:
# last, so that a timeout cannot leave a stamp build behind for the other two)
cd $GRAFT_REPO_ROOT
bash tools/sessions/gpu_r5_outwt.sh
mkdir -p gpurun_out/r5outwt; ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/boundary_probe tools/probes/boundary_probe.hip 2>/dev/null && timeout 120 /tmp/boundary_probe ) 2>&1 | tee gpurun_out/r5outwt/boundary_probe.txt
bash tools/sessions/gpu_r5_env.sh
bash tools/sessions/gpu_r5_ur4.sh
bash tools/sessions/gpu_r5_tail.sh
