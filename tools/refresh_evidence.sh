#!/bin/bash
# Round-end evidence on the library in the tree (GPU box, through gpurun): default and driver-style bench lines, the HBM
# traffic table of every reported workload, rocprofv3 kernel statistics and counter summaries of the single-step and the fused
# launch, the GPU test log, a fuzz sweep, the VALU issue-rate micro-benchmark.     bash tools/refresh_evidence.sh <tag> [fuzz cases]
TAG=${1:-r03f}; FUZZ=${2:-10000}
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest_gpu.log 2>&1
tail -3 $O/${TAG}_pytest_gpu.log
bash tools/pmc_traffic_all.sh ${TAG} > $O/${TAG}_traffic.log 2>&1
cp $O/pmc_traffic_${TAG}.json $O/${TAG}_pmc_traffic.json
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2> $O/${TAG}_bench_driver.err
B="python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --repeats 1 --no-cpu-baseline --no-single-env --no-parity-gate --no-collective --no-baseline-configs"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_one -o p -- $B > $O/prof_${TAG}_one.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_roll -o p -- $B --rollout 20 --warmup 40 > $O/prof_${TAG}_roll.log 2>&1)
for m in one roll; do f=$(ls $O/prof_${TAG}_$m/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && awk 'length($0) < 600' $f > $O/${TAG}_${m}_kernel_stats.csv; rm -rf $O/prof_${TAG}_$m; done
bash tools/pmc_profile.sh ${TAG} --no-collective --no-baseline-configs > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_${TAG} k_step > $O/${TAG}_pmc_summary.txt 2>&1
bash tools/pmc_profile.sh ${TAG}_roll --no-collective --no-baseline-configs --rollout 20 --warmup 40 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_${TAG}_roll k_step > $O/${TAG}_roll_pmc_summary.txt 2>&1
rm -rf $O/pmc_${TAG} $O/pmc_${TAG}_roll $O/pmc_${TAG}_n*
[ -x build_variants/valu_rates ] && timeout 60 build_variants/valu_rates > $O/${TAG}_valu_rates.txt 2>&1
if [ "$FUZZ" -gt 0 ]; then
  ATC_FUZZ_CASES=$FUZZ ATC_FUZZ_SEED=4000000 timeout 1500 python -m pytest tests/test_fuzz_parity.py -q -m gpu -n 8 > $O/${TAG}_fuzz.txt 2>&1
  tail -2 $O/${TAG}_fuzz.txt
fi
cat $O/${TAG}_bench_default.json
