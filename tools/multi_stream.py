#!/usr/bin/env python3
"""Developer tool: the headline batch as S independent sub-batches stepped on S HIP streams (no join between steps), so
that one sub-batch's launch ramp / tail overlaps the other's body.
  python tools/multi_stream.py [envs_total] [aircraft] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import torch
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
HOLD = 20
ORDER = [int(v) for v in os.environ.get("ATC_MS_ORDER", "1,2,4,8,1,2").split(",")]
for S in ORDER:
    envs = [AtcVecEnv(B // S, N, scenario=scenarios.LOWW(random_entrypoints=True), auto_reset=True, seed=s) for s in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    acts = [[(torch.rand((B // S, N, 3), device="cuda") * 2 - 1) for _ in range(4)] for _ in range(S)]
    torch.cuda.synchronize()

    from atc_hip.vec_env import make_multi_launcher
    calls = [make_multi_launcher(envs, [acts[s][k] for s in range(S)], streams) for k in range(4)]

    def run(n):
        for t in range(n):
            calls[(t // HOLD) % 4]()
    run(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams %d: %.2f us per step of all %d envs, %.3f G env-steps/s" % (S, dt / K * 1e6, B, B * K / dt / 1e9))
    for e in envs:
        e.close()
