#!/usr/bin/env python3
"""Developer tool: period of back-to-back launches of a trivial kernel on one stream (what one launch per step costs before
the step kernel does anything), and the step period as a function of the batch size (fixed cost vs per-env cost)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import torch
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios


def period(launch, n=4000):
    for _ in range(500):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


scn = scenarios.LOWW(random_entrypoints=True)
tiny = AtcVecEnv(4, 16, scenario=scn, auto_reset=True)
a = torch.zeros((4, 16, 3), device="cuda")
print("trivial launch (4 envs x 16): %.2f us per launch" % period(tiny.make_launcher(a)))
x = torch.zeros(64, device="cuda")
print("torch elementwise add on 64 floats: %.2f us per launch" % period(lambda: x.add_(1.0)))
for B in (4096, 16384, 32768, 65536, 131072, 262144, 524288):
    env = AtcVecEnv(B, 16, scenario=scn, auto_reset=True)
    acts = (torch.rand((B, 16, 3), device="cuda") * 2 - 1)
    la, la_held = env.make_launcher(acts), env.make_launcher(acts, held=True)   # the one action tensor is repeated for ever
    for _ in range(3000 if B <= 65536 else 800):
        la()
    n = 2000 if B <= 65536 else 500
    print("%7d envs: %.2f us per step, %.2f with ATC_M_ACTIONS_HELD" % (B, period(la, n), period(la_held, n)))
    env.close()
