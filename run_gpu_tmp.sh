timeout 1500 python -m pytest tests -m gpu -q -x -k "not full_size" > gpurun_out/pytest_gpu14.log 2>&1; echo rc=$? >> gpurun_out/pytest_gpu14.log
tail -3 gpurun_out/pytest_gpu14.log
for args in "--sep-nm 3" "--sep-nm 0" "--envs 4096 --aircraft 64" "--rollout 20"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 300 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$args', round(d['ms_per_step']*1000,2), round(d['roofline']['frac'],4), d['config']['episodes_finished'])"
done
