"""One-off probe (build container only, needs /root/reference): the REFERENCE stepped on three random polygon sets (tests/test_random_sectors.py:
random_sector — overlapping, concave, shared and axis-aligned borders; five random entry points with three flight levels each) at dt 1 / 0.3 / 2 s,
shaping / normalisation off and discrete once each, random actions held 5 / 20 / 60 steps: 24 628 steps, 67 below-MVA and 281 outside terminals.
Writes tests/golden/_probe_random_sectors.npz + _probe_defs.json (git-ignored scratch); random_sectors_replay.py — a second process: this one
has the reference's `envs` package imported — replays them through both oracle instantiations at g9's bars.  Round 6: 0 failing groups of 12, twice."""
import os, sys, json, random, importlib.util, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
spec = importlib.util.spec_from_file_location("gg", os.path.join(ROOT, "tests/golden/generate_golden.py"))
gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
# NOTE: gg put the reference's `envs` first on sys.path and imported it: the product's envs package is shadowed in THIS process, so
# the replay runs in a second process (replay2.py)
from test_random_sectors import random_sector
f32 = gg.f32
ref_model, shape, ref_scen = gg.ref_model, gg.shape, gg.ref_scen
defs = {}
for seed, n_poly in ((31, 6), (32, 10), (33, 14)):
    mvas, runway, _ = random_sector(seed, n_poly)
    rng = np.random.default_rng(seed)
    entries = [(float(np.float32(x)), float(np.float32(y)), float(rng.integers(0, 360)), [int(l) for l in rng.choice([60, 90, 150, 250, 370, 380], 3, replace=False)])
               for x, y in rng.uniform(15, 45, (5, 2))]
    name = "Rand%d" % seed
    defs[name] = dict(mvas=[([list(map(float, p)) for p in np.asarray(ring)], float(h)) for ring, h in mvas], runway=list(map(float, runway)), entries=entries)
    def factory(mv=defs[name]["mvas"], rw=runway, en=entries):
        class S(ref_scen.Scenario):
            def __init__(self):
                self.mvas = [ref_model.MinimumVectoringAltitude(shape.Polygon([tuple(p) for p in ring]), int(h)) for ring, h in mv]
                self.runway = ref_model.Runway(*rw)
                self.airspace = ref_model.Airspace(self.mvas, self.runway)
                self.entrypoints = [ref_model.EntryPoint(x, y, phi, lv) for x, y, phi, lv in en]
        return S()
    gg.SCENARIOS[name] = factory
json.dump(defs, open(os.path.join(ROOT, "tests/golden/_probe_defs.json"), "w"))
rec = gg.WideRecorder(stride=8)
rng = np.random.default_rng(99)
for name in defs:
    for dt, shaping, normalize, discrete in ((1.0, True, True, False), (0.3, True, True, False), (2.0, False, False, False), (1.0, True, True, True)):
        env = gg.make_env(name, dt=dt, shaping=shaping, normalize=normalize, discrete=discrete)
        random.seed(len(name) + int(dt * 10))
        for k in range(10):
            hold = int(rng.choice([5, 20, 60]))
            acts = []
            for b in range(1500 // hold + 1):
                if discrete:
                    a = np.array([rng.integers(0, 20), rng.integers(0, 380), rng.integers(0, 360)], dtype=np.float64)
                else:
                    a = f32([rng.uniform(-1, 1), rng.uniform(-1, 1) if k % 2 else rng.uniform(-1, -0.5), rng.uniform(-1, 1)])
                acts.append(np.tile(a, (hold, 1)))
            rec.run(env, np.concatenate(acts)[:1500], name, dt, shaping, normalize, discrete, extra_after_done=2)
    print(name, len(rec.flags), flush=True)
rec.save(os.path.join(ROOT, "tests/golden/_probe_random_sectors.npz"))
fl, dn = np.asarray(rec.flags), np.asarray(rec.done).astype(bool)
print("episodes", len(rec.ep), "steps", len(fl), "below", int(((fl & 1) != 0)[dn].sum()), "outside", int(((fl & 2) != 0)[dn].sum()), "won", int(((fl & 4) != 0)[dn].sum()))
