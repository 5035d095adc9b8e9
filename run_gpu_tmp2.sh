#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05a/pytest_$TAG.txt
cat gpurun_out/r05a/pytest_$TAG.txt
bash run_gpu_tmp.sh
