timeout 900 python -m pytest tests/test_hip_edge_cases.py -m gpu -q -k "variants" > gpurun_out/pytest_var.log 2>&1; tail -15 gpurun_out/pytest_var.log
