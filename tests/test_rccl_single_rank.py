"""RCCL on the one GPU of the test box: the multi-GPU path's exact calls — init_process_group("nccl", device_id=...),
all_gather_into_tensor and all_reduce on DEVICE tensors (atc_hip/dist.py) — executed in a world of one rank, on the statistics
of a real env batch.  Counterpart of the reference's only parallelism, SubprocVecEnv x 8 + Monitor
(learning/atc-gym-stable-baselines.py:76-80).  No scaling is measured here (one GPU); what is proven is that the RCCL branch
runs on MI355X and returns the right values."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_allgather_of_episode_statistics_one_rank():
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch
        from atc_hip import dist as D
        from atc_hip.vec_env import AtcVecEnv
        from envs.atc import scenarios
        rank, ws, local = D.init(backend="nccl", force=True)
        assert (rank, ws) == (0, 1) and D.backend_name() == "nccl"
        env = AtcVecEnv(4096, 16, scenario=scenarios.LOWW(random_entrypoints=True), device=local, auto_reset=True, seed=5)
        a = torch.rand((4096, 16, 3), device=env.device) * 2 - 1
        for t in range(300):
            env.step(a, held=t > 0)
        torch.cuda.synchronize()
        ret, length = D.all_gather_stats(env.ep_return, env.ep_length, force=True)
        assert ret.is_cuda and ret.shape == (1, 4096) and length.shape == (1, 4096) and length.dtype == torch.int32
        assert torch.equal(ret[0], env.ep_return) and torch.equal(length[0], env.ep_length)
        assert int((length[0] > 0).sum()) > 0            # some episodes did finish: the statistics are not all zero
        assert D.max_over_ranks(2.5, env.device, force=True) == 2.5 and D.sum_over_ranks(7.0, env.device, force=True) == 7.0
        D.barrier()
        env.close()
        D.shutdown()
        print("rccl-ok", tuple(ret.shape))
    """ % os.path.join(ROOT, "atc-reinforcement-learning_amd"))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "ATC_DIST_BACKEND"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0 and "rccl-ok (1, 4096)" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
