#!/bin/bash
# PMC breakdown of one bench configuration: tools/pmc_kernel.sh <outname> <bench args...>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/b.log 2>&1
