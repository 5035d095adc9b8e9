#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE; separate --pmc runs with --kernel-trace) for any bench workload.
#   bash tools/pmc_traffic.sh <tag> [bench args...]      -> gpurun_out/pmc_<tag>/
TAG=${1:-traffic}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-single-env --no-parity-gate $@"
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $OUT -o pass$i --output-format csv -- $BENCH > $OUT/pass$i.log 2>&1
done
