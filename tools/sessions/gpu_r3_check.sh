#!/bin/bash
# round-3 last check of the committed state: whole GPU suite + smoke + the default bench line
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3check; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
