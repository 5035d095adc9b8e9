mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r01c.log 2>&1; echo rc=$? >> gpurun_out/pytest_gpu_r01c.log
tail -4 gpurun_out/pytest_gpu_r01c.log
timeout 600 python bench.py > gpurun_out/r01c_bench_n16_65536.json 2>/dev/null; cat gpurun_out/r01c_bench_n16_65536.json
timeout 300 python bench.py --aircraft 1 --no-cpu-baseline > gpurun_out/r01c_bench_n1_65536.json 2>/dev/null
timeout 300 python bench.py --envs 8192 --no-cpu-baseline > gpurun_out/r01c_bench_n16_8192.json 2>/dev/null
timeout 300 python bench.py --envs 4096 --aircraft 64 --no-cpu-baseline > gpurun_out/r01c_bench_n64_4096.json 2>/dev/null
timeout 300 python bench.py --rollout 20 --no-cpu-baseline > gpurun_out/r01c_bench_n16_rollout20.json 2>/dev/null
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_r01c -o r01c --output-format csv -- python $ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline > $ROOT/gpurun_out/prof_r01c.log 2>&1
cd $ROOT
bash tools/pmc_profile.sh r01c > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r01c k_step | tee gpurun_out/pmc_r01c_summary.txt
ls gpurun_out/prof_r01c
