#!/usr/bin/env python3
"""Where does the fp32 path's position error against the float64 reference come from?  (CPU, numpy, ~10 s.)

ROUND 3 / ABI 17 — kept as the measurement behind that round's conclusion ("the near-FAF exception is a property of fp32 aircraft
STATE").  Round 4 acted on it: since ABI 18 speed and heading are 32-bit fixed point and the displacement is float64
(include/atc_step.h); tools/position_error.py measures that specification (profiles/r04_position_error.txt) and the exception
is retired.  "K" below is the SUPERSEDED ABI-17 specification.

The bearing to the FAF (obs[8], atc_gym.py:289-292) moves by e / d_faf radians for a position error e, so keeping it within
1e-5 (normalised: 1.8e-3 deg = 3.1e-5 rad) at d_faf = 0.027 nm — the one step of the wide fixture that needs the stated
exception of tests/helpers.py:replay_wide — takes e < 8.5e-7 nm after the ~700 steps an approach lasts.  This script flies
4 096 aircraft for 1 000 steps under the reference's action protocol (a new U(-1,1) action every 20 steps,
learning/atc-gym-demo.py:18-19) three ways and prints the position deviation from the float64 reference:

  R  float64 everything (the reference's arithmetic, atc_gym.py:318-335, model.py:60-129);
  S  fp32 STATE — targets decoded, rate limits applied and v / phi held in float32 exactly as any fp32 implementation must
     (they are fp32 observations) — but the displacement evaluated in float64 from that state (exact sin / cos, exact
     v / 3600 dt) and accumulated in float64: the best ANY implementation with fp32 aircraft state can do;
  K  the fp32 specification of include/atc_step.h: polynomial sin / cos, fp32 products, positions on the 2^-25 nm grid.

If S is already beyond 8.5e-7 nm, the exception is a property of fp32 state, not of the kinematics specification."""
import numpy as np

rng = np.random.default_rng(0)
n, steps, dt = 4096, 1000, 1.0
f32, f64 = np.float32, np.float64


def clamp(d, lo, hi):
    return np.minimum(np.maximum(d, lo), hi)


def sincos_spec(phi):   # include/atc_step.h: fp32 heading kinematics
    S = [f32(-0.16666631400585175), f32(0.008331366814672947), f32(-0.00019439239986240864)]
    Cc = [f32(-0.5), f32(0.04166661575436592), f32(-0.001388648059219122), f32(2.436429167573806e-05)]
    k = np.rint(phi * f32(1.0 / 90.0)).astype(f32)
    t = (f64(phi) - 90.0 * f64(k)).astype(f32)            # one fma: exact
    r = t * f32(np.pi / 180.0)
    r2 = r * r
    fma = lambda a, b, c: (f64(a) * f64(b) + f64(c)).astype(f32)  # noqa: E731
    sp = fma(fma(fma(S[2], r2, S[1]), r2, S[0]), r2, f32(1.0))
    s = sp * r
    c = fma(fma(fma(fma(Cc[3], r2, Cc[2]), r2, Cc[1]), r2, Cc[0]), r2, f32(1.0))
    q = k.astype(np.int64) & 3
    s1 = np.where(q & 1, c, s)
    c1 = np.where(q & 1, s, c)
    return np.where(q & 2, -s1, s1), np.where((q + 1) & 2, -c1, c1)


x0, y0, phi0, v0 = 10.0, 51.0, 90.0, 250.0
R = dict(x=np.full(n, x0), y=np.full(n, y0), phi=np.full(n, phi0), v=np.full(n, v0))
S = dict(x=np.full(n, x0), y=np.full(n, y0), phi=np.full(n, phi0, f32), v=np.full(n, v0, f32))
K = dict(x=np.zeros(n, np.int64), y=np.zeros(n, np.int64), phi=np.full(n, phi0, f32), v=np.full(n, v0, f32))
scale = 2.0 ** 25
print("steps   S: fp32 state, float64 kinematics [nm]      K: the fp32 specification [nm]")
print("        median     p90        max                   median     p90        max")
for t in range(1, steps + 1):
    if (t - 1) % 20 == 0:
        a = rng.uniform(-1, 1, (n, 3)).astype(f32)
    # reference: float64 decode of the float32 action
    tv = f64(a[:, 0]) * 200 / 2 + 200 / 2 + 100
    tp = f64(a[:, 2]) * 360 / 2 + 360 / 2 + 0
    R["v"] = R["v"] + clamp(tv - R["v"], -5 * dt, 5 * dt)
    R["phi"] = R["phi"] + clamp(tp - R["phi"], -3 * dt, 3 * dt)
    d = R["v"] / 3600 * dt
    R["x"] += np.sin(np.radians(R["phi"])) * d
    R["y"] += np.cos(np.radians(R["phi"])) * d
    # fp32 state (the same operations in float32)
    tv32 = a[:, 0] * f32(200) / f32(2) + f32(100) + f32(100)
    tp32 = a[:, 2] * f32(360) / f32(2) + f32(180) + f32(0)
    for M in (S, K):
        M["v"] = M["v"] + clamp(tv32 - M["v"], f32(-5 * dt), f32(5 * dt))
        M["phi"] = M["phi"] + clamp(tp32 - M["phi"], f32(-3 * dt), f32(3 * dt))
    d = f64(S["v"]) / 3600 * dt
    S["x"] += np.sin(np.radians(f64(S["phi"]))) * d
    S["y"] += np.cos(np.radians(f64(S["phi"]))) * d
    dist = (K["v"] / f32(3600.0)) * f32(dt)
    sn, cs = sincos_spec(K["phi"])
    K["x"] += np.rint(f64(sn * dist) * scale).astype(np.int64)
    K["y"] += np.rint(f64(cs * dist) * scale).astype(np.int64)
    if t in (100, 300, 500, 700, 1000):
        eS = np.hypot(S["x"] - R["x"], S["y"] - R["y"])
        eK = np.hypot(x0 + K["x"] / scale - R["x"], y0 + K["y"] / scale - R["y"])
        print("%5d   %.2e   %.2e   %.2e              %.2e   %.2e   %.2e" % (
            t, np.median(eS), np.percentile(eS, 90), eS.max(), np.median(eK), np.percentile(eK, 90), eK.max()))
print("bound for a 1e-5 bearing at d_faf = 0.027 nm: 8.5e-7 nm")
