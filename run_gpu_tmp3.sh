#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 300 build_variants/write_bw 2>&1 | tee gpurun_out/r05a/write_bw_$TAG.txt
bash tools/ablate_pmc.sh $TAG "base skip0 new abl2" "roll" --no-collective --no-baseline-configs > /dev/null 2>&1
cat gpurun_out/ablate_$TAG.txt
TAG=${TAG}b bash tools/ablate_pmc.sh ${TAG}n64 "base new h16_6 hl2" "roll" --no-collective --no-baseline-configs --aircraft 64 --envs 4096 > /dev/null 2>&1
cat gpurun_out/ablate_${TAG}n64.txt
bash run_gpu_tmp.sh
