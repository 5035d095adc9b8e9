#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 120 build_variants/write_bw 2>&1 | grep "pattern, stores only\|300 dep\|400 dep\|one 16-B store per lane, plain" | tee gpurun_out/r05a/write_bw_$TAG.txt
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L 2>&1 | grep -i "^\s*\(TA_\|TCP_\|TD_\|SQ_INST\|SQ_WAIT\|SQ_ACTIVE\|SQ_BUSY\|GRBM\)" | cut -c1-150 > $GRAFT_REPO_ROOT/gpurun_out/r05a/counters.txt; timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r05a/counters_all.txt 2>&1)
bash run_gpu_tmp.sh
