#!/bin/bash
# Developer experiment (GPU box, through gpurun): step time of A/B libraries in build_variants/ on a list of workloads.
#   bash tools/ab_time.sh "<variant names>" "<N>x<B>:<one|roll> ..." [rounds]        (bench.py --lib, HIP-event timing)
VARS=$1; WORK=$2; ROUNDS=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
B="python bench.py --repeats 3 --no-cpu-baseline --no-single-env --no-parity-gate --no-collective --no-baseline-configs"
for r in $(seq $ROUNDS); do
for w in $WORK; do
  shape=${w%%:*}; mode=${w##*:}; n=${shape%%x*}; b=${shape##*x}
  if [ $mode = roll ]; then args="--steps 400 --warmup 40 --rollout 20 --prewarm 4000"; else args="--steps 2000 --warmup 100 --prewarm 6000"; fi
  for v in $VARS; do
    $B --aircraft $n --envs $b $args --lib build_variants/libatcstep_$v.so 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('$v ${b}x$n $mode %.3f us' % (j['ms_per_step']*1e3))"
  done
done
done
