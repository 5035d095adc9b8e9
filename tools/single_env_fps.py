#!/usr/bin/env python3
"""The reference's own benchmark protocol (learning/atc-gym-compute-performance.py:7-19) on the drop-in AtcGym:
100 000 x env.step(one fixed sampled action), no reset on done, FPS = N / wall time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import envs.atc.atc_gym as atc_gym  # noqa: E402

env = atc_gym.AtcGym()
env.reset()
nextaction = env.action_space.sample()
num = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for i in range(200):
    env.step(nextaction)
t0 = time.time()
for i in range(num):
    state, reward, done, info = env.step(nextaction)
t1 = time.time()
print("Finished!")
print("FPS: %f" % (num / (t1 - t0)))
