#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_issue; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_(ACTIVE_INST|INSTS|INST_CYCLES|WAIT)[A-Z0-9_]*" | sort -u | tr "\n" " " > $OUT/counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o pmc -- python $R/tools/tune_conv.py --only "128>128@128" --batch 8 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/b -o pmc -- python $R/tools/tune_conv.py --only "128>128@128" --batch 8 > $OUT/b.log 2>&1
cat $OUT/counters.txt
