for rep in 1 2 3; do
ATC_LIBATCSTEP=$PWD/build_variants/libatcstep_nts0.so timeout 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nts0', round(d['ms_per_step']*1000,2), round(d['roofline']['frac'],4))"
timeout 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nts1', round(d['ms_per_step']*1000,2), round(d['roofline']['frac'],4))"
done
timeout 1500 python -m pytest tests -m gpu -q -x -k "not full_size" > gpurun_out/pytest_gpu10.log 2>&1; echo rc=$? >> gpurun_out/pytest_gpu10.log
tail -3 gpurun_out/pytest_gpu10.log
