#!/usr/bin/env python3
"""How far do the positions of the fp32 specification (include/atc_step.h, ABI 18 / 19: fixed-point speed / heading state, float64
kinematics, dithered rounding on the 2^-25 nm grid) drift from the float64 reference?  (CPU, the two instantiations of the test
oracle side by side over every episode of tests/golden/g9_wide.npz: 650 963 steps, episodes of up to 6 000 steps.)

The bearing to the FAF (obs[8], atc_gym.py:289-292) moves by e / d_faf radians for a position error e: 1e-5 in observation units
at the closest approach of the fixture (d_faf = 0.027 nm) takes e < 8.5e-7 nm — what the near-FAF exception of rounds 1-3 was
about (tools/faf_conditioning.py measured 1e-6 median / 5e-6 max after 700 steps for the ABI-17 specification)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd"), os.path.join(ROOT, "tests")]
import numpy as np

import helpers as H
import test_oracle_golden as T

fx = H.WideFixture(sys.argv[1] if len(sys.argv) > 1 else "g9_wide.npz")   # e.g. g11_unbounded.npz: headings beyond the 32-bit field (ABI 19)
errs, perr, verr, ages = [], [], [], []
for (scen, dt, shaping, normalize, discrete), eps in fx.groups().items():
    B = len(eps)
    bes = [T._OracleLockstep(d, scen, dt, shaping, normalize, discrete, B) for d in (np.float64, np.float32)]
    for b, ep in enumerate(eps):
        for be in bes:
            be.place(b, ep["init_state"], ep["init_timesteps"], ep["init_last_action"])
    steps = np.array([ep["steps"] for ep in eps])
    starts = np.array([ep["start"] for ep in eps])
    for t in range(int(steps.max())):
        live = t < steps
        rows = np.where(live, starts + t, starts)
        a = fx.action[rows].astype(np.float32).reshape(B, 1, 3)
        st = [be.step(a)[5] for be in bes]
        errs.append(np.hypot(st[0][live, 0] - st[1][live, 0], st[0][live, 1] - st[1][live, 1]))
        perr.append(np.abs(st[0][live, 3] - st[1][live, 3]))
        verr.append(np.abs(st[0][live, 4] - st[1][live, 4]))
        ages.append(np.full(int(live.sum()), t))
e, p, v, age = (np.concatenate(x) for x in (errs, perr, verr, ages))
print("steps compared: %d" % len(e))
print("position error [nm]: median %.2e  p99 %.2e  p99.9 %.2e  max %.2e   (grid step 2^-25 nm = %.2e)" % (
    np.median(e), np.quantile(e, .99), np.quantile(e, .999), e.max(), 2.0 ** -25))
for lo, hi in ((0, 200), (200, 800), (800, 3000), (3000, 6001)):
    m = (age >= lo) & (age < hi)
    if m.any():
        print("  steps %4d..%4d of an episode: median %.2e  max %.2e nm  (%d samples)" % (lo, hi - 1, np.median(e[m]), e[m].max(), int(m.sum())))
print("heading error [deg]: max %.2e   speed error [kt]: max %.2e" % (p.max(), v.max()))
