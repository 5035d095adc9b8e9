#!/bin/bash
# PMC counter passes for the bench kernel (run on the GPU box through gpurun).  Counters are collected in their own
# runs with --kernel-trace only (no sys/hip/hsa traces), one pass per --pmc group; outputs go to gpurun_out/pmc_<tag>/.
#   bash tools/pmc_profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-single-env --no-parity-gate --prewarm 1000 $@"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OUT -o pass$i --output-format csv -- $BENCH > $OUT/pass$i.log 2>&1
done
ls $OUT
