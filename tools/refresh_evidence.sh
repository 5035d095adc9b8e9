#!/bin/bash
# Round-end evidence on the library in the tree (GPU box, through gpurun): default and driver-style bench lines, the HBM
# traffic table of every reported workload, the GPU test log.     bash tools/refresh_evidence.sh <tag>
TAG=${1:-r03f}
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
bash tools/pmc_traffic_all.sh ${TAG} > gpurun_out/${TAG}_traffic.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.log
cat gpurun_out/${TAG}_bench_default.json
