"""Multi-GPU: env instances never interact, so the batch shards by contiguous env-index range, one process per GPU,
with NO collective on the step path.  The only exchange is an all-gather of per-env episode statistics (returns,
lengths) once per report interval — `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm); 256 KiB per rank at
65 536 envs, i.e. latency-bound, so it is issued once per rollout, never per step.  (The reference's counterpart is
SubprocVecEnv's pipe per worker + Monitor CSVs, learning/atc-gym-stable-baselines.py:69-80.)

Round 4: the exchange is ONE collective per report (the statistics packed into one [rows, k] 32-bit tensor) and it is
ASYNCHRONOUS — `StatsExchange`: a snapshot at the end of a rollout, the all-gather issued from a side stream so that it runs
beside whatever the caller does next (the bench: the closing barrier and the next rollout's start-up), the wait where the
result is needed.  A 20-step rollout is 0.4 ms of GPU work; two blocking all-gathers and a barrier inside that window would
have cost a fifth of it on a path that has no step-path communication at all.

Works on CPU tensors with the gloo backend too (tests/test_dist_gloo.py, world_size 2)."""
import os

import torch
import torch.distributed as dist


def world():
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def init(backend=None, force=False):
    """Initialises the default process group when WORLD_SIZE > 1 (nccl on GPU, gloo otherwise).
    force=True does so for a single process too — a world of one rank whose collectives still go through the backend (RCCL on a
    GPU box): how `bench.py` and the GPU tests exercise the exact `init_process_group("nccl", device_id=...)` / device-tensor
    collective calls of the multi-GPU path on a box with one GPU."""
    rank, ws, local = world()
    if backend is None:  # ATC_DIST_BACKEND=gloo lets the multi-rank control flow be exercised on a box with one GPU
        backend = os.environ.get("ATC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        # one process per GPU: bind before the communicator is created.  If the launcher narrowed the visible devices to one
        # per rank (HIP_VISIBLE_DEVICES), LOCAL_RANK exceeds the device count and the rank's GPU is device 0.
        n_dev = torch.cuda.device_count()
        if local >= n_dev:
            if n_dev != 1:
                raise RuntimeError("LOCAL_RANK %d but only %d visible devices: one process per GPU" % (local, n_dev))
            local = 0
        torch.cuda.set_device(local)
    if (ws > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (required by the host driver for RCCL)
        kw = {}
        if ws == 1 and "MASTER_PORT" not in os.environ:   # a world of one: nobody to meet, any free local port will do
            kw["init_method"] = "tcp://127.0.0.1:%d" % _free_port()
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws, **kw)
    return rank, ws, local


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def backend_name():
    """Backend of the initialised default group ("nccl" = RCCL on ROCm, "gloo"), or None."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def shard_range(total_envs, rank, world_size):
    """Contiguous env-index range [lo, hi) owned by `rank`; remainders go to the lowest ranks."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(seed, rank):
    """Distinct RNG key per rank for the entry draws (envs are indexed locally on each GPU)."""
    return (int(seed) + 0x9E3779B97F4A7C15 * (rank + 1)) & (2 ** 64 - 1) if rank else int(seed)


_shard_rows = {}   # call signature (dtypes + trailing dims: the same on every rank) -> shard sizes verified equal on every rank


def _collective(force):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)


def all_gather_stats(*tensors, force=False):
    """All-gathers equally-shaped per-env tensors from every rank into [world, ...] tensors (one collective each).
    Identity (with a leading axis of 1) when not distributed — unless force=True and a (one-rank) group exists: then the
    backend's collective runs all the same.

    all_gather_into_tensor needs equal shards (shard_range() gives them only when the world size divides the env count).
    That is verified collectively by the FIRST call with a given signature (number of tensors, dtypes, trailing dimensions —
    everything but the shard size itself), i.e. by a decision every rank takes alike whatever its own shard size: mismatching
    ranks meet in the same MIN / MAX reductions and all of them raise.  Later calls with that signature must keep their
    shard sizes (a local assertion)."""
    if not _collective(force):
        return [t.unsqueeze(0) for t in tensors]
    ws = dist.get_world_size()
    host_staged = dist.get_backend() == "gloo"  # gloo gathers host tensors; RCCL gathers device tensors in place
    key = tuple((str(t.dtype), tuple(t.shape[1:])) for t in tensors)
    mine = [int(t.shape[0]) for t in tensors]
    if key not in _shard_rows:
        rows = torch.tensor(mine, dtype=torch.int64, device="cpu" if host_staged else tensors[0].device)
        lo, hi = rows.clone(), rows.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise AssertionError("all_gather_stats needs equally sized shards on every rank (%s .. %s)" % (lo.tolist(), hi.tolist()))
        _shard_rows[key] = mine
    assert mine == _shard_rows[key], "shard sizes changed since the first all_gather_stats call of this kind"
    out = []
    for t in tensors:
        src = t.contiguous()
        if host_staged and src.is_cuda:
            src = src.cpu()
        g = torch.empty((ws * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(g, src)  # concatenation along dim 0 (the layout both nccl/RCCL and gloo accept)
        out.append(g.view((ws,) + tuple(src.shape)).to(t.device))
    return out


class StatsExchange:
    """The path's only exchange as ONE asynchronous collective per report.

        xch = StatsExchange()
        ...rollout r...        xch.snapshot(env.ep_return, env.ep_length)   # packs the statistics: one [rows, k] int32 tensor
        ...rollout r + 1 ...   <queue the rollout's step launches>          # the GPU is busy from here on
                               xch.issue()          # all_gather_into_tensor(async_op=True) from a SIDE stream: the host cost of
                                                    #   the call (tens of microseconds) and the collective itself run beside
                                                    #   the step kernels already queued; nothing waits for them
                               ret, length = xch.wait()                    # [world, rows] tensors of rollout r's statistics

    The k statistics must be 1-D tensors of one length with 4-byte elements (float32 / int32: `ep_return`, `ep_length` are
    neighbouring words of the per-episode env record); they travel as their bit patterns and come back in their own dtypes.
    The collective is issued under a private side stream, so the backend's stream synchronises with THAT (idle) stream and not
    with the stream the step kernels were queued on — an event recorded by snapshot() orders it after the packing copies alone —
    (torch's process group makes its stream wait for the caller's current
    stream: issued on the compute stream after the launches, the collective would start when the last step has finished;
    issued before them, the host time of the call delays the first launch — measured 41 us per 20-step block on the one-rank
    RCCL group).  `wait()` makes the side stream, then the CURRENT stream wait for the collective — it does not block the
    host — so it belongs after the launches that should overlap it.  Not distributed (and not forced): snapshot / wait degrade to a local copy with a leading
    axis of 1 and no collective is issued.  `collectives` counts the collectives issued (tests: one per report)."""

    def __init__(self, force=False):
        self.force = bool(force)
        self.collectives = 0
        self._src = self._dst = self._host = self._work = self._side = self._packed = None
        self._dtypes = None
        self._pending = False

    def snapshot(self, *tensors):
        t0 = tensors[0]
        assert all(t.dim() == 1 and t.shape == t0.shape and t.element_size() == 4 for t in tensors), \
            "statistics must be 1-D, equally long, 4 bytes per element"
        if self._src is None or self._src.shape != (t0.shape[0], len(tensors)) or self._src.device != t0.device:
            self._src = torch.empty((t0.shape[0], len(tensors)), dtype=torch.int32, device=t0.device)
            self._dst = None
        self._dtypes = [t.dtype for t in tensors]
        for c, t in enumerate(tensors):   # (strided views of the env record are fine: the copy gathers them)
            self._src[:, c].copy_(t.view(torch.int32))
        # The packing copies are kernels on the CURRENT (compute) stream; the collective is issued from a side stream that the
        # process group synchronises with — an idle stream that knows nothing of them.  This event is what orders the collective
        # after the copies (and only after them: launches queued on the compute stream after the snapshot stay unordered with it,
        # which is the overlap).  Without it the all-gather could read `_src` before the copies — queued behind a whole rollout —
        # had run.
        self._packed = None
        if self._src.is_cuda:
            self._packed = torch.cuda.Event()
            self._packed.record(torch.cuda.current_stream(self._src.device))
        self._pending = True

    def rearm(self):
        """Marks the last snapshot as unreported again (measurement loops that exchange the same snapshot repeatedly)."""
        assert self._src is not None, "snapshot() first"
        self._pending = True

    def issue(self):
        assert self._pending, "snapshot() first"
        self._pending = False
        if not _collective(self.force):
            self._work = None
            return
        ws = dist.get_world_size()
        src = self._src
        if dist.get_backend() == "gloo" and src.is_cuda:   # gloo gathers host tensors (test flows on one GPU)
            self._host = src.cpu()   # (a synchronising copy on the compute stream: ordered after the packing copies)
            src = self._host
        key = (("packed", src.shape[1]),)
        if key not in _shard_rows:   # equal shards: decided by every rank alike on the first call (see all_gather_stats)
            rows = torch.tensor([int(src.shape[0])], dtype=torch.int64, device=src.device)
            lo, hi = rows.clone(), rows.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                raise AssertionError("StatsExchange needs equally sized shards on every rank (%s .. %s)" % (lo.tolist(), hi.tolist()))
            _shard_rows[key] = int(src.shape[0])
        assert int(src.shape[0]) == _shard_rows[key], "shard size changed since the first exchange"
        if self._dst is None or self._dst.device != src.device or self._dst.shape[0] != ws * src.shape[0]:
            self._dst = torch.empty((ws * src.shape[0], src.shape[1]), dtype=torch.int32, device=src.device)
        if src.is_cuda:
            if self._side is None or self._side.device != src.device:
                self._side = torch.cuda.Stream(device=src.device)
            if self._packed is not None:
                self._side.wait_event(self._packed)   # the collective reads `_src`: after the snapshot's packing copies
            with torch.cuda.stream(self._side):
                self._work = dist.all_gather_into_tensor(self._dst, src, async_op=True)
        else:
            self._work = dist.all_gather_into_tensor(self._dst, src, async_op=True)
        self.collectives += 1

    def wait(self):
        if self._work is not None:
            if self._dst.is_cuda:
                with torch.cuda.stream(self._side):
                    self._work.wait()
                torch.cuda.current_stream(self._dst.device).wait_stream(self._side)
            else:
                self._work.wait()
            self._work = None
            g = self._dst.view(-1, self._src.shape[0], self._src.shape[1])
        else:
            g = self._src.unsqueeze(0)
        g = g.to(self._src.device)
        # copies, not views: `_dst` / `_src` are overwritten by the next issue() / snapshot() — possibly queued before the caller
        # has looked at this report
        return [g[:, :, c].contiguous().view(dt).clone() for c, dt in enumerate(self._dtypes)]


def barrier(force=False):
    """dist.barrier() of the default group (a no-op in a world of one rank unless force=True: then the backend's barrier runs
    all the same, which is how its cost is measured on a one-GPU box)."""
    if _collective(force):
        dist.barrier()


def max_over_ranks(value, device, force=False):
    if not _collective(force):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device, force=False):
    if not _collective(force):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
