"""Shared test helpers: golden loading, scenario factory, episode replay against any env adapter."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

F_BELOW_MVA, F_OUTSIDE, F_WON, F_TIMEOUT, F_INVALID_V, F_INVALID_H, F_CONFLICT, F_NOISE, F_INACTIVE = \
    1, 2, 4, 8, 16, 32, 64, 128, 256


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


_npz_cache = {}


def golden_npz(name):
    """Fully materialised dict (NpzFile re-decompresses an array on every [] access)."""
    if name not in _npz_cache:
        with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
            _npz_cache[name] = {k: z[k] for k in z.files}
    return _npz_cache[name]


def episodes_of(npz):
    return json.loads(str(npz["episodes"]))


def make_scenario(name):
    """Scenario objects of the product's host mirror for the golden scenario names."""
    from envs.atc import model, scenarios
    if name == "LOWW":
        return scenarios.LOWW()
    if name == "LOWW_random":
        return scenarios.LOWW(random_entrypoints=True)
    if name == "Simple":
        return scenarios.SimpleScenario()
    if name == "UnitTest":  # fixture world of the reference's model_test.py:94-113
        s = scenarios.SimpleScenario()
        s.runway = model.Runway(20, 20, 0, 180)
        s.airspace = model.Airspace(s.mvas, s.runway)
        return s
    raise KeyError(name)


_compiled = {}


def compiled(name, grid_cell=None):
    key = (name, grid_cell)
    if key not in _compiled:
        from envs.atc import scenarios
        _compiled[key] = scenarios.compile_scenario(make_scenario(name), grid_cell=grid_cell)
    return _compiled[key]


class StepRecord:
    __slots__ = ("obs", "raw", "reward", "done", "flags", "timesteps", "actions_taken", "total_reward", "state")


def replay_episode(adapter, npz, ep, check, max_steps=None):
    """Drives `adapter` (see OracleAdapter / HipAdapter) through golden episode `ep`, calling
    check(step_index, golden_row_index, StepRecord) after each step."""
    adapter.configure(ep["scen"], ep["dt"], ep["shaping"], ep["normalize"], ep["discrete"])
    adapter.reset()
    adapter.set_state(*ep["init_state"])
    adapter.set_counters(ep["init_timesteps"], ep["init_last_action"])
    n = ep["steps"] if max_steps is None else min(ep["steps"], max_steps)
    s = ep["start"]
    for t in range(n):
        rec = adapter.step(npz["action"][s + t])
        check(t, s + t, rec)
    return n
