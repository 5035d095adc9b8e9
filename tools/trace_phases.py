#!/usr/bin/env python3
"""Developer tool: per-wavefront phase timing of k_step with the ATC_TRACE build (s_memtime stamps).
  hipcc ... -DATC_TRACE=1 -o build_variants/libatcstep_trace.so ; python tools/trace_phases.py [envs] [aircraft]"""
import ctypes as C
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

B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
W = 1 << max(0, (N - 1).bit_length())
env = AtcVecEnv(B, N, scenario=scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=N > 1), auto_reset=True)
acts = [(torch.rand((B, N, 3), device="cuda") * 2 - 1) for _ in range(4)]
HELD = os.environ.get("ATC_TRACE_HELD", "1") != "0"   # launches 2..20 of an action block carry ATC_M_ACTIONS_HELD
for t in range(300):
    env.step(acts[(t // 20) % 4], held=HELD and t % 20 != 0)
n_waves = (B * W + 255) // 256 * 4
trace = torch.zeros((n_waves, 8), dtype=torch.int64, device="cuda")
ptr = trace.data_ptr()
env.params.reserved0 = ptr & 0xffffffff
env.params.reserved1 = struct.unpack("f", struct.pack("I", (ptr >> 32) & 0xffffffff))[0]
env.refresh_params()   # step(held=True) passes the held-action twin of `params`: it has to carry the trace pointer too
torch.cuda.synchronize()
env.step(acts[0])                 # first launch of a block
env.step(acts[0], held=HELD)      # the traced launch: a repeat
torch.cuda.synchronize()
raw = trace.cpu().numpy()
good = (raw > 0).all(axis=1) & (np.diff(raw, axis=1) >= 0).all(axis=1)
print("wavefronts traced: %d, usable rows: %d" % (len(raw), int(good.sum())))
names = ["loads -> decode+kinematics", "mva resolve + pair scan", "overrides + corridor", "obs + shaping + normalise",
         "reductions, flag/reward stores, auto-reset", "obs transpose + store", "state store"]
# s_memtime counts a per-XCD clock (the counters of different XCDs are not aligned): workgroups are dealt round-robin to the
# 8 XCDs, so the timeline is drawn per XCD and the tick is calibrated against the kernel time measured with HIP events
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for t in range(200):
    env.step(acts[(t // 20) % 4], held=HELD and t % 20 != 0)
ev1.record()
torch.cuda.synchronize()
kernel_us = ev0.elapsed_time(ev1) / 200 * 1e3
wave = np.arange(len(raw))
xcd = (wave // 4) % 8
GHZ = float(os.environ.get("ATC_TRACE_GHZ", "2.1"))   # s_memtime counts shader-clock cycles (~2.05-2.1 GHz under this load)
tick_us = 1e-3 / GHZ
print("kernel %.2f us per launch (HIP events, launch to launch); stamps in cycles, converted at %.2f GHz" % (kernel_us, GHZ))
d = np.diff(raw.astype(np.float64), axis=1) * tick_us
print("phase durations per wavefront [us]: mean / median / p90 / share of lifetime")
life = (raw[:, 7] - raw[:, 0]) * tick_us
for k, nme in enumerate(names):
    print("%-44s %6.2f %6.2f %6.2f   %4.1f %%" % (nme, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90),
                                                  100 * d[:, k].sum() / life.sum()))
print("wave lifetime [us]: mean %.2f median %.2f p90 %.2f" % (life.mean(), np.median(life), np.percentile(life, 90)))
if B * W < 65536 * 16:
    sys.exit(0)
x = 0
r = raw[xcd == x].astype(np.float64)
t0 = r[:, 0].min()
r = (r - t0) * tick_us
print("XCD %d timeline (1 us bins): waves waiting for their loads / computing / in the store phases / started / finished" % x)
for b in np.arange(0.0, r[:, 7].max(), 1.0):
    mid = b + 0.5
    loading = int(((r[:, 0] <= mid) & (mid < r[:, 1])).sum())
    compute = int(((r[:, 1] <= mid) & (mid < r[:, 5])).sum())
    storing = int(((r[:, 5] <= mid) & (mid < r[:, 7])).sum())
    print("%5.0f us  load %4d  compute %4d  store %4d   +%4d -%4d" % (
        b, loading, compute, storing, int(((r[:, 0] >= b) & (r[:, 0] < b + 1)).sum()), int(((r[:, 7] >= b) & (r[:, 7] < b + 1)).sum())))
