#!/usr/bin/env python3
"""Developer soak test (GPU box): the persistent step server against the launch path over millions of steps — two AtcGym envs fed
the same actions, compared bit for bit after every step, with resets, pauses longer than the server's lease, and reads of the
device state in between.     python tools/soak_step_server.py [steps]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
from envs.atc import atc_gym, model, scenarios  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
atc_gym._TIGHT_GAP_S = 200e-6   # two envs stepped alternately: count that as a tight loop
envs = [atc_gym.AtcGym(model.SimParameters(0.3), scenarios.LOWW(random_entrypoints=True), persistent=p) for p in (True, False)]
srv, ref = envs
rng = np.random.default_rng(1)
draws = 0


def reset_both():
    global draws
    draws += 1
    out = []
    for e in envs:
        random.seed(draws)
        out.append(e.reset())
    assert np.array_equal(out[0], out[1])


reset_both()
served = restarts = 0
was = False
t0 = time.perf_counter()
a = rng.uniform(-1, 1, 3).astype(np.float32)
for t in range(n):
    if t % 20 == 0:
        a = rng.uniform(-1.05, 1.05, 3).astype(np.float32)
    rs, rr = srv.step(a), ref.step(a)
    assert np.array_equal(rs[0], rr[0]) and rs[1] == rr[1] and rs[2] == rr[2] and np.array_equal(rs[3]["original_state"], rr[3]["original_state"]), t
    assert (srv.timesteps, srv.actions_taken) == (ref.timesteps, ref.actions_taken), t
    served += srv._serving
    restarts += srv._serving and not was
    was = srv._serving
    if rs[2] or t % 5003 == 5002:
        reset_both()
    if t % 9973 == 0:
        time.sleep(0.002)                      # longer than the lease
    if t % 7919 == 0:
        assert srv._airplane.h == ref._airplane.h and srv._airplane.phi == ref._airplane.phi
dt = time.perf_counter() - t0
print("%d steps bit-identical (server vs launches), %.1f %% served, %d server starts, %d resets, %.1f s" % (n, 100.0 * served / n, restarts, draws, dt))
for e in envs:
    e.close()
