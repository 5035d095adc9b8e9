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


def _worker_unequal(rank, ws, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "atc-reinforcement-learning_amd"))
    from atc_hip import dist as D
    D.init(backend="gloo")
    try:
        D.all_gather_stats(torch.zeros(8 + rank))   # rank 1 owns one env more: every rank must notice, none may hang
        q.put((rank, "no error"))
    except AssertionError as exc:
        q.put((rank, str(exc)))
    D.shutdown()


@pytest.mark.timeout(120)
def test_unequal_shards_raise_on_every_rank():
    """the equal-shard guard has to be a decision all ranks take alike — here rank 0 and rank 1 hold different
    shard sizes and both must raise (before: one could enter all_reduce while the other entered all_gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unequal, args=(r, 2, 29631, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
    assert all("equally sized shards" in res[r] for r in (0, 1)), res


def _worker_single(q):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        os.environ.pop(k, None)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "atc-reinforcement-learning_amd"))
    from atc_hip import dist as D
    assert D.backend_name() is None
    plain = D.all_gather_stats(torch.arange(6.0))[0]          # no group: identity
    D.init(backend="gloo", force=True)                          # a world of one rank, real backend
    skipped = D.all_gather_stats(torch.arange(6.0))[0]         # group exists but world size 1: still the identity ...
    forced = D.all_gather_stats(torch.arange(6.0), force=True)[0]   # ... unless the collective is asked for
    q.put((D.backend_name(), plain.tolist(), skipped.tolist(), forced.tolist(), D.max_over_ranks(3.0, torch.device("cpu"), force=True),
           D.sum_over_ranks(4.0, torch.device("cpu"), force=True)))
    D.shutdown()


@pytest.mark.timeout(120)
def test_forced_collective_in_a_world_of_one():
    """The code path bench.py and tests/test_rccl_single_rank.py take on a one-GPU box, here on gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_single, args=(q,))
    p.start()
    name, plain, skipped, forced, mx, sm = q.get(timeout=90)
    p.join(30)
    want = [[0.0, 1.0, 2.0, 3.0, 4.0, 5.0]]
    assert name == "gloo" and plain == want and skipped == want and forced == want and mx == 3.0 and sm == 4.0


def _worker_exchange(rank, ws, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "atc-reinforcement-learning_amd"))
    import torch.distributed as dist
    from atc_hip import dist as D
    D.init(backend="gloo")
    calls = {"all_gather": 0, "all_reduce": 0}
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

    def count_ag(*a, **k):
        calls["all_gather"] += 1
        return real_ag(*a, **k)

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)
    dist.all_gather_into_tensor, dist.all_reduce = count_ag, count_ar
    n = 16
    # the per-episode env record of atc_state_t: ep_length and ep_return are neighbouring 32-bit words of an 8-word record
    stats = torch.zeros((n, 8), dtype=torch.int32)
    ep_len = stats[:, 1]
    ep_ret = stats[:, 2:3].view(torch.float32).squeeze(1)
    xch = D.StatsExchange()
    log = []
    for block in range(4):       # four "rollouts": statistics of rollout b are reported while rollout b + 1 is queued
        ep_len += 1 + rank       # ... the rollout's steps change the live statistics ...
        ep_ret += 0.5 * (block + 1) + 100.0 * rank
        if block:
            xch.issue()                       # rollout b - 1's snapshot, asynchronously
            n_after_issue = calls["all_gather"]
            ep_len += 1000                    # the next rollout is already overwriting the live record
            ret, length = xch.wait()
            ep_len -= 1000
            log.append((block - 1, ret.tolist(), length.tolist(), n_after_issue))
        xch.snapshot(ep_ret, ep_len)
    xch.issue()
    ret, length = xch.wait()
    log.append((3, ret.tolist(), length.tolist(), calls["all_gather"]))
    q.put((rank, log, dict(calls), xch.collectives))
    dist.all_gather_into_tensor, dist.all_reduce = real_ag, real_ar
    D.shutdown()


@pytest.mark.timeout(120)
def test_stats_exchange_one_async_collective_per_report():
    """Round-3 review, next #1: the episode statistics travel as ONE packed collective per report, issued asynchronously, and
    what a report returns is the SNAPSHOT of the finished rollout — not whatever the next rollout has written meanwhile."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, 29647, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
    for rank, log, calls, n_coll in res:
        assert n_coll == 4 and calls["all_gather"] == 4          # four reports, four collectives: one each
        assert calls["all_reduce"] == 2                          # the equal-shard check of the first exchange, once
        for k, (block, ret, length, n_ag) in enumerate(log):
            assert block == k and n_ag == k + 1                  # issue() is where the collective is issued
            for r in range(2):                                   # [world][rows]: rank r's statistics after rollout `block`
                want_len = (block + 1) * (1 + r)
                want_ret = sum(0.5 * (b + 1) + 100.0 * r for b in range(block + 1))
                assert length[r] == [want_len] * 16 and ret[r] == [want_ret] * 16, (rank, block, r)
