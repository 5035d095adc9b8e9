timeout 1500 python -m pytest tests/test_hip_edge_cases.py -m gpu -q > gpurun_out/pytest_edge.log 2>&1; echo rc=$? >> gpurun_out/pytest_edge.log
tail -40 gpurun_out/pytest_edge.log
