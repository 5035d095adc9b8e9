mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu4.log 2>&1; echo rc=$? >> gpurun_out/pytest_gpu4.log
tail -15 gpurun_out/pytest_gpu4.log
for args in "" "--aircraft 1" "--envs 4096 --aircraft 64" "--rollout 20"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 1000 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$args', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['episodes_finished'])"
done
