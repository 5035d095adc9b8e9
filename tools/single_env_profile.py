import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import numpy as np
from envs.atc import atc_gym
env = atc_gym.AtcGym(); env.reset()
a = np.array([0.1, -0.2, 0.3], np.float32)
for _ in range(3000): env.step(a)
n = 50000
t0 = time.perf_counter()
for _ in range(n): env.step(a)
full = (time.perf_counter() - t0) / n * 1e6
seq = env._seq
t0 = time.perf_counter()
for i in range(n):
    seq = (seq + 1) & 0x7fffffff
    env._step_packet(env._raw_stream(), seq)
call = (time.perf_counter() - t0) / n * 1e6
env._seq = seq
t0 = time.perf_counter()
for i in range(n):
    env._raw_stream()
rs = (time.perf_counter() - t0) / n * 1e6
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20000): env.step(a)
pr.disable()
print("full step %.2f us | launch+poll call %.2f us | raw stream %.2f us" % (full, call, rs))
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
