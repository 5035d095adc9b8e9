#!/usr/bin/env python3
"""Developer tool: per-wavefront, per-step phase timing of the multi-step launch (ATC_TRACE build, s_memtime stamps).
  hipcc ... -DATC_TRACE=1 -o build_variants/libatcstep_trace.so ; python tools/trace_rollout.py [envs] [aircraft] [T]"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import numpy as np
import torch
from atc_hip import lib as _binding
_binding.use_library(os.environ.get("ATC_TRACE_LIB") or os.path.join(ROOT, "build_variants", "libatcstep_trace.so"))
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
env = AtcVecEnv(B, N, scenario=scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=N > 1), auto_reset=True)
acts = [(torch.rand((1, B, N, 3), device="cuda") * 2 - 1) for _ in range(4)]
out = None
for t in range(30):
    out = env.rollout(acts[t % 4], out, hold=T)
W = 1
while W < N:
    W *= 2
n_waves = (B * W + 255) // 256 * 4
trace = torch.zeros((n_waves, T, 8), dtype=torch.int64, device="cuda")
ptr = trace.data_ptr()
env.params.reserved0 = ptr & 0xffffffff
env.params.reserved1 = struct.unpack("f", struct.pack("I", (ptr >> 32) & 0xffffffff))[0]
torch.cuda.synchronize()
env.rollout(acts[0], out, hold=T)
torch.cuda.synchronize()
raw = trace.cpu().numpy().astype(np.float64)
env.params.reserved0 = 0
env.params.reserved1 = 0.0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for t in range(50):
    env.rollout(acts[t % 4], out, hold=T)
ev1.record()
torch.cuda.synchronize()
launch_us = ev0.elapsed_time(ev1) / 50 * 1e3
# s_memtime counts shader-clock cycles; the clock under this load is ~2.05-2.1 GHz (SQ_BUSY_CYCLES / kernel duration of the PMC
# passes).  Durations are printed in microseconds at GHZ and are proportional to cycles; per-XCD counters are not aligned, so only
# differences within one wavefront are used.
GHZ = float(os.environ.get("ATC_TRACE_GHZ", "2.1"))
tick = 1e-3 / GHZ
print("launch %.1f us = %.2f us per step (HIP events); stamps in cycles, converted at %.2f GHz" % (launch_us, launch_us / T, GHZ))
names = ["top: decode + kinematics (+ first loads in step 0)", "mva resolve + pair scan", "overrides + corridor",
         "obs + shaping + normalise", "reductions, flag/reward stores, auto-reset (+ next action fetch)", "obs transpose + store"]
if os.environ.get("ATC_TRACE_MODE") == "2":   # -DATC_TRACE_MODE=2: stamps 1..5 dissect the step's last phase
    names = ["everything up to the lookup resolve incl. it (stamp 0 -> 1)", "override chain: quiet test (+ the rare block) (1 -> 2)",
             "(observation if not first) normalisation, inactive fix-up (2 -> 3)", "reductions, done logic, packet, auto-reset (3 -> 4)",
             "observation transpose + store (4 -> 5)", "flag / reward / done stores (5 -> 6)"]
if os.environ.get("ATC_TRACE_MODE") == "1":   # a library built with -DATC_TRACE_MODE=1: stamps 1..5 dissect the first phase
    names = ["loop top: output bases, decode, masks (stamp 0 -> 1)", "first half up to the rate limits (1 -> 2)",
             "float64 kinematics (2 -> 3)", "position -> fp32, lookup cell address, gather issued (3 -> 4)",
             "second half up to the lookup resolve: scan, observation-first arithmetic (4 -> 5)",
             "resolve, override chain, normalisation, reductions, stores, auto-reset (5 -> 6)"]
life = (raw[:, T - 1, 7] - raw[:, 0, 0]) * tick
print("wave lifetime: mean %.1f us, per step %.2f us" % (life.mean(), life.mean() / T))
for label, sl in (("step 0", slice(0, 1)), ("steps 1..T-1", slice(1, T))):
    r = raw[:, sl, :]
    d = np.diff(r[:, :, :7], axis=2) * tick
    print(label)
    for k, nme in enumerate(names):
        print("  %-70s mean %6.3f  median %6.3f  p90 %6.3f us" % (nme, d[:, :, k].mean(), np.median(d[:, :, k]), np.percentile(d[:, :, k], 90)))
    print("  step total (stamp 0 -> 6): mean %.3f us" % ((r[:, :, 6] - r[:, :, 0]) * tick).mean())
gap = (raw[:, 1:, 0] - raw[:, :-1, 6]) * tick
print("between steps (stamp 6 -> next stamp 0): mean %.3f us" % gap.mean())
print("state store (last stamp 6 -> 7): mean %.3f us" % ((raw[:, T - 1, 7] - raw[:, T - 1, 6]) * tick).mean())
