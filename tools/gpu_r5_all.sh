#!/bin/bash
# the three prepared openers of round 5 in one gpurun call (~10 min of box time): out_wt A-B, runtime-environment A-B, K-loop tail by placement (rebuilds the library twice:
# last, so that a timeout cannot leave a stamp build behind for the other two)
cd $GRAFT_REPO_ROOT
bash tools/gpu_r5_outwt.sh
bash tools/gpu_r5_env.sh
bash tools/gpu_r5_ur4.sh
bash tools/gpu_r5_tail.sh
