timeout 900 python -m pytest tests/test_hip_edge_cases.py -m gpu -q -k "huge" > gpurun_out/pytest_huge.log 2>&1; tail -15 gpurun_out/pytest_huge.log
