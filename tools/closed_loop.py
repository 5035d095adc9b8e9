#!/usr/bin/env python3
"""Closed-loop throughput: a small torch MLP policy (2 x 64 tanh, the size of stable-baselines' MlpPolicy used by the
reference, learning/atc-gym-stable-baselines.py:109-121) acts on the device observations every step — no host copies.

    python tools/closed_loop.py [--envs 65536] [--aircraft 16] [--steps 500] [--frame-skip 20]

--frame-skip k: the policy acts every k-th step and its action is held in between (the reference's demo loop holds an action for
20 steps, learning/atc-gym-demo.py:18-19) — as k single steps of which k - 1 carry the held-action promise, and as one
atc_rollout_hold launch per decision.
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
    ap.add_argument("--frame-skip", type=int, default=1)
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
    k = a.frame_skip
    if k > 1:
        n_dec = max(1, a.steps // k)
        with torch.no_grad():
            for mode in ("single steps, held-action hint", "single steps, no hint", "one atc_rollout_hold launch per decision"):
                out = None
                for rep in range(2):   # first pass warms up
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n_dec):
                        act = policy(obs.view(B * N, 10)).view(B, N, 3)
                        if mode.startswith("one"):
                            out = env.rollout(act[None], out, hold=k)
                            obs = out["obs"][k - 1]
                        else:
                            for j in range(k):
                                obs, rew, done, info = env.step(act, held=(j > 0 and "no hint" not in mode))
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                print("frame skip %d, %s: %.1f M env-steps/s (%.1f us per env step, %.1f us per decision)" % (
                    k, mode, B * n_dec * k / dt / 1e6, dt / (n_dec * k) * 1e6, dt / n_dec * 1e6))
    env.close()


if __name__ == "__main__":
    main()
