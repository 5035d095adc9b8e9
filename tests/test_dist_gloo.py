"""World-size-2 gloo test of the env sharding + the episode-return all-gather (the multi-GPU path's only exchange)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

import helpers as H  # noqa: F401


def _worker(rank, ws, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "atc-reinforcement-learning_amd"))
    from atc_hip import dist as D
    r, w, _ = D.init(backend="gloo")
    lo, hi = D.shard_range(total, r, w)
    # equal shard sizes are required by all_gather_into_tensor: pad like the bench (equal per-rank batch)
    n = total // w
    ret = torch.arange(r * n, (r + 1) * n, dtype=torch.float32) * 0.5
    length = torch.arange(r * n, (r + 1) * n, dtype=torch.int32)
    g_ret, g_len = D.all_gather_stats(ret, length)
    mx = D.max_over_ranks(10.0 + r, torch.device("cpu"))
    sm = D.sum_over_ranks(1.0 + r, torch.device("cpu"))
    D.barrier()
    q.put((r, lo, hi, g_ret.reshape(-1).tolist(), g_len.reshape(-1).tolist(), mx, sm, D.rank_seed(5, r)))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_shard_and_allgather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, ws, port = 64, 2, 29617
    procs = [ctx.Process(target=_worker, args=(r, ws, port, total, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(ws))
    for p in procs:
        p.join(30)
    assert [(r[1], r[2]) for r in res] == [(0, 32), (32, 64)]
    for r in res:
        assert r[3] == [0.5 * i for i in range(64)] and r[4] == list(range(64))
        assert r[5] == 11.0 and r[6] == 3.0
    assert res[0][7] != res[1][7]


def test_shard_range_remainders():
    from atc_hip import dist as D
    spans = [D.shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert D.shard_range(524288, 7, 8) == (458752, 524288)
    assert [t.shape[0] for t in D.all_gather_stats(torch.zeros(5))] == [1]
