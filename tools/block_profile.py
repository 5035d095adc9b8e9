#!/usr/bin/env python3
"""Developer tool: per-launch durations inside a short timed block that starts on an idle, synchronised device (what
bench.py --steps 20 measures): HIP events between consecutive atc_step launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import torch
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios

B, N, K = 65536, 16, 20
env = AtcVecEnv(B, N, scenario=scenarios.LOWW(random_entrypoints=True), auto_reset=True)
acts = [(torch.rand((B, N, 3), device="cuda") * 2 - 1) for _ in range(2)]
la = [env.make_launcher(a) for a in acts]
for t in range(400):
    la[(t // 20) % 2]()
torch.cuda.synchronize()
for idle_ms in (0.0, 0.2, 2.0, 20.0):
    rows = []
    for rep in range(6):
        torch.cuda.synchronize()
        time.sleep(idle_ms * 1e-3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for t in range(K):
            la[(t // 20) % 2]()
            ev[t + 1].record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)], (t1 - t0) * 1e6, (t2 - t0) * 1e6))
    r = rows[-1]
    print("idle %5.1f ms: host enqueue %.0f us, wall %.0f us, per-launch us: %s" % (
        idle_ms, r[1], r[2], " ".join("%.1f" % v for v in r[0])))
