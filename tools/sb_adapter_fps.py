#!/usr/bin/env python3
"""PCIe-inclusive throughput of the stable-baselines-shaped adapter (numpy in / numpy out every step).
    python tools/sb_adapter_fps.py [num_envs] [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "atc-reinforcement-learning_amd")]
import numpy as np  # noqa: E402
from atc_hip.sb_adapter import AtcSBVecEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
venv = AtcSBVecEnv(B)
venv.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
for _ in range(20):
    venv.step(acts)
t0 = time.perf_counter()
for _ in range(steps):
    venv.step(acts)
dt = time.perf_counter() - t0
print("AtcSBVecEnv %d envs: %.0f env-steps/s (%.2f ms per vector step, host arrays + info dicts every step)"
      % (B, B * steps / dt, dt / steps * 1e3))
venv.close()
