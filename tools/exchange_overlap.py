#!/usr/bin/env python3
"""Developer experiment (GPU box): what does a concurrent stream cost a block of 20 dependent step launches?
Variants: the block alone; + a small copy kernel on a side stream; + the packed all-gather issued from a side stream (one-rank
RCCL group); + the same all-gather issued AFTER the block has drained (serial)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import torch
from atc_hip import dist as D
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios

D.init(backend="nccl", force=True)
dev = torch.device("cuda", 0)
B, N = 65536, 16
env = AtcVecEnv(B, N, scenario=scenarios.LOWW(random_entrypoints=True), auto_reset=True, seed=1)
a = torch.rand((B, N, 3), device=dev) * 2 - 1
first, rest = env.make_launcher(a), env.make_launcher(a, held=True)


def block():
    first()
    for _ in range(19):
        rest()


for _ in range(300):
    block()
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
src = torch.zeros((B, 2), dtype=torch.int32, device=dev)
dst = torch.zeros((B, 2), dtype=torch.int32, device=dev)
xch = D.StatsExchange(force=True)
xch.snapshot(env.ep_return, env.ep_length)


def v_alone():
    block()


def v_copy():
    block()
    with torch.cuda.stream(side):
        dst.copy_(src, non_blocking=True)
    torch.cuda.current_stream().wait_stream(side)


def v_xch():
    block()
    xch.rearm()
    xch.issue()
    xch.wait()


def v_serial():
    block()
    torch.cuda.synchronize()
    xch.rearm()
    xch.issue()
    xch.wait()


def v_xch_first():
    xch.rearm()
    xch.issue()
    block()
    xch.wait()


for name, f in (("alone", v_alone), ("side copy", v_copy), ("exchange after launches", v_xch), ("exchange before launches", v_xch_first),
                ("serial", v_serial), ("alone", v_alone), ("exchange after launches", v_xch)):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print("%-28s median %.1f us  min %.1f  p90 %.1f" % (name, ts[len(ts) // 2], ts[0], ts[int(0.9 * len(ts))]))
D.shutdown()
