"""Second half of the random-sector probe (see random_sectors_record.py): replays the recorded reference episodes through both oracle instantiations."""
import os, sys, json, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import helpers as H
import test_oracle_golden as T
defs = json.load(open(os.path.join(ROOT, "tests/golden/_probe_defs.json")))
_orig = H.make_scenario
def make_scenario(name):
    if name in defs:
        from envs.atc import model, scenarios
        d = defs[name]
        s = scenarios.Scenario()
        s.mvas = [model.MinimumVectoringAltitude([tuple(p) for p in ring], int(h)) for ring, h in d["mvas"]]
        s.runway = model.Runway(*d["runway"])
        s.airspace = model.Airspace(s.mvas, s.runway)
        s.entrypoints = [model.EntryPoint(x, y, phi, lv) for x, y, phi, lv in d["entries"]]
        s.noise_areas = []
        return s
    return _orig(name)
H.make_scenario = make_scenario
fx = H.WideFixture("_probe_random_sectors.npz")
print("groups", len(fx.groups()), "steps", len(fx.flags))
for dtype in (np.float64, np.float32):
    f64 = dtype == np.float64
    bad = 0
    for key, eps in fx.groups().items():
        class One:
            def groups(self): return {key: eps}
        sub = One()
        for a in ("flags", "done", "actions_taken", "reward", "samp_rows", "obs", "state", "action", "samp_index"):
            setattr(sub, a, getattr(fx, a))
        try:
            H.replay_wide(sub, lambda *a: T._OracleLockstep(dtype, *a), obs_tol=2e-6 if f64 else 1e-5,
                          state_tol=1e-11 if f64 else 1e-5, rew_tol=1e-10 if f64 else 1e-5)
        except AssertionError as e:
            tb = traceback.extract_tb(sys.exc_info()[2])[-1]
            if tb.lineno == 193:
                continue
            bad += 1
            print(" FAIL", dtype.__name__, key, "line", tb.lineno, str(e)[:160])
    print(dtype.__name__, "failing groups:", bad, "of", len(fx.groups()))
