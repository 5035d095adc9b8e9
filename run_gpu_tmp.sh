timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu12.log 2>&1; echo rc=$? >> gpurun_out/pytest_gpu12.log
tail -3 gpurun_out/pytest_gpu12.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 400 --warmup 100 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
