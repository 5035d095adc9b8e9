timeout 900 python -m pytest tests -m gpu -q -x -k "atcgym or g7 or seeded or adapter or rgb" > gpurun_out/pytest_gym.log 2>&1; tail -3 gpurun_out/pytest_gym.log
python tools/single_env_fps.py 30000 2>&1 | tail -2
