#!/bin/bash
# round-3 closing session: whole GPU suite, smoke, then the profile collection of tools/gpu_r3_prof.sh
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3final; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/gpu_r3_prof.sh
