import sys, time
sys.path[:0] = [".", "atc-reinforcement-learning_amd"]
import torch
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios
scn = scenarios.LOWW(random_entrypoints=True)
for B in (8192, 65536):
    env = AtcVecEnv(B, 16, scenario=scn, auto_reset=True)
    a = torch.rand((B, 16, 3), device="cuda") * 2 - 1
    la = env.make_launcher(a)
    for name, fn in (("env.step(tensor)", lambda: env.step(a)), ("launcher", la)):
        for _ in range(3000): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5000): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%6d envs %-18s host %.2f us per call, end-to-end %.2f us per step" % (B, name, (t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
    env.close()
