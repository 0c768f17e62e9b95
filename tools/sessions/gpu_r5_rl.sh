#!/bin/bash
# round-5 session: the render loop with several batches in flight (test + timing), box info
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5rl; mkdir -p $OUT
bash tools/sessions/box_info.sh > $OUT/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_network.py -m gpu -q -x -k "render_loop" 2>&1 | tail -5 | tee $OUT/pytest.txt
for cfg in "large 0 8 256" "normal 1 8 512" "large 0 1 128"; do timeout 300 python tools/render_loop_time.py $cfg 2>&1 | grep -v "amdgpu.ids\|initialized\|^-----"; done | tee $OUT/render_loop.txt
