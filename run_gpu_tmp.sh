timeout 900 python -m pytest tests/test_sb_adapter.py -m gpu -q > gpurun_out/pytest_sb.log 2>&1; echo rc=$? >> gpurun_out/pytest_sb.log
tail -12 gpurun_out/pytest_sb.log
