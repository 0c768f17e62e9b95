#!/bin/bash
# round-4 closing check: the whole GPU suite + smoke() at HEAD
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4final; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $OUT/smoke.log
