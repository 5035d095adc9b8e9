#!/usr/bin/env python3
"""Closed-loop throughput: a small torch MLP policy (2 x 64 tanh, the size of stable-baselines' MlpPolicy used by the
reference, learning/atc-gym-stable-baselines.py:109-121) acts on the device observations every step — no host copies.

    python tools/closed_loop.py [--envs 65536] [--aircraft 16] [--steps 500]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "atc-reinforcement-learning_amd")]
import torch  # noqa: E402
from atc_hip.vec_env import AtcVecEnv  # noqa: E402
from envs.atc import scenarios  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--aircraft", type=int, default=16)
    ap.add_argument("--steps", type=int, default=500)
    a = ap.parse_args()
    B, N = a.envs, a.aircraft
    env = AtcVecEnv(B, N, scenario=scenarios.LOWW(random_entrypoints=N > 1), auto_reset=True)
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(10, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                 torch.nn.Linear(64, 3), torch.nn.Tanh()).cuda()   # one shared per-aircraft policy
    obs = env.reset()
    with torch.no_grad():
        for _ in range(20):
            obs, rew, done, info = env.step(policy(obs.view(B * N, 10)).view(B, N, 3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            obs, rew, done, info = env.step(policy(obs.view(B * N, 10)).view(B, N, 3))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("closed loop, %d envs x %d aircraft, MLP 10-64-64-3 per aircraft: %.1f M env-steps/s (%.1f us per step), "
          "%d episodes finished" % (B, N, B * a.steps / dt / 1e6, dt / a.steps * 1e6, int(env.episodes.sum()) - B))
    env.close()


if __name__ == "__main__":
    main()
