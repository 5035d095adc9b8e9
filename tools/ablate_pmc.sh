#!/bin/bash
# Developer experiment (GPU box, through gpurun): dynamic instruction counts of the step kernel per ablation build.
# One rocprofv3 --pmc pass (kernel-trace only) per library in build_variants/ and launch mode; summary lines go to
# gpurun_out/ablate_<tag>.txt.     bash tools/ablate_pmc.sh <tag> "<variant names>" "<modes: one roll>" [bench args...]
TAG=${1:-abl}; VARS=${2:-base}; MODES=${3:-one roll}; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/ablate_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  for mode in $MODES; do
    extra=""; [ $mode = roll ] && extra="--rollout 20 --warmup 40"
    BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-single-env --no-parity-gate --prewarm 1000 --lib $ROOT/build_variants/libatcstep_$v.so $extra $@"
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/${v}_$mode -o p --output-format csv -- $BENCH > $OUT/${v}_$mode.log 2>&1
    echo "== $v $mode" >> $ROOT/gpurun_out/ablate_$TAG.txt
    python $ROOT/tools/pmc_summary.py $OUT/${v}_$mode k_step >> $ROOT/gpurun_out/ablate_$TAG.txt 2>&1
    rm -rf $OUT/${v}_$mode
  done
done
cat $ROOT/gpurun_out/ablate_$TAG.txt
