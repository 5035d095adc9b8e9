for args in "--sep-nm 3" "--envs 4096 --aircraft 64" "--aircraft 1" "--envs 8192"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 2000 --warmup 300 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$args', round(d['ms_per_step']*1000,2), round(d['roofline']['frac'],4), d['config']['episodes_finished'])"
done
