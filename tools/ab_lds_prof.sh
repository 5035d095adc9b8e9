#!/bin/bash
# Developer tool (GPU box): kernel trace + PMC passes of tools/ab_lds_table.py, one run per variant
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/s4/ldsprof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for w in lds grid; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o kt_$w --output-format csv -- python $ROOT/tools/ab_lds_table.py 1 $w 60 > $OUT/kt_$w.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o p1_$w --output-format csv -- python $ROOT/tools/ab_lds_table.py 1 $w 20 > $OUT/p1_$w.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT -o p2_$w --output-format csv -- python $ROOT/tools/ab_lds_table.py 1 $w 20 > $OUT/p2_$w.log 2>&1
done
ls $OUT
