#!/usr/bin/env python3
"""Single-env throughput of the drop-in AtcGym under the reference's benchmark protocol
(learning/atc-gym-compute-performance.py:7-19: ONE sampled action repeated, no reset on done, frames / wall-clock).

    python tools/single_env_fps.py [--frames 100000] [--warmup 200]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "atc-reinforcement-learning_amd")]


def measure(frames, warmup):
    from envs.atc.atc_gym import AtcGym
    gym_env = AtcGym()
    gym_env.reset()
    fixed = gym_env.action_space.sample()
    for _ in range(warmup):
        gym_env.step(fixed)
    began = time.perf_counter()
    for _ in range(frames):
        gym_env.step(fixed)
    elapsed = time.perf_counter() - began
    gym_env.close()
    return frames / elapsed


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("frames", nargs="?", type=int, default=100000)
    ap.add_argument("--warmup", type=int, default=200)
    a = ap.parse_args()
    print("FPS: %.1f  (%d frames, single AtcGym env on the GPU)" % (measure(a.frames, a.warmup), a.frames))
