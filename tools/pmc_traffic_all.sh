#!/bin/bash
# HBM traffic of every workload bench.py reports, on the library in the tree (GPU box, through gpurun):
# two separate rocprofv3 --pmc passes per workload (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) and a table in the format of
# profiles/pmc_traffic.json keyed by the library's ABI version.     bash tools/pmc_traffic_all.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
run() {  # name, bench args...
  local name=$1; shift
  bash $ROOT/tools/pmc_traffic.sh ${TAG}_$name --prewarm 1000 --no-collective --no-baseline-configs "$@" > /dev/null 2>&1
}
run n16_65536
run n16_65536_nohint --no-held-hint
run n16_65536_roll --rollout 20 --warmup 40
run n1_65536 --aircraft 1
run n1_65536_roll --aircraft 1 --rollout 20 --warmup 40
run n16_8192 --envs 8192
run n16_8192_roll --envs 8192 --rollout 20 --warmup 40
run n64_4096 --aircraft 64 --envs 4096
run n64_4096_roll --aircraft 64 --envs 4096 --rollout 20 --warmup 40
run n64_32768 --aircraft 64 --envs 32768
run n16_262144 --envs 262144
run n1_65536_g0125 --aircraft 1 --grid-cell 0.125
python $ROOT/tools/pmc_traffic_table.py $TAG > $ROOT/gpurun_out/pmc_traffic_$TAG.json
cat $ROOT/gpurun_out/pmc_traffic_$TAG.json
