"""CPU tests of the build-defined extension semantics (multi-aircraft envs, separation scan, noise areas, auto-reset):
there is no reference code for these (README.md:51,60,62 of the reference are prose) — parity is UNPINNED and the oracle
is the definition.  These tests pin the definition with hand-computed known answers and check that it reduces to the
reference's single-aircraft behaviour."""
import numpy as np
import pytest

import helpers as H
from oracle import oracle as O


def _env(scen_name_or_obj, B, N, dtype=np.float64, **kw):
    from envs.atc import scenarios
    comp = H.compiled(scen_name_or_obj) if isinstance(scen_name_or_obj, str) else scenarios.compile_scenario(scen_name_or_obj)
    return O.OracleEnv(comp, B, N, O.make_params(**kw), dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_separation_known_answers(dtype):
    """2.99 / 3.00 / 3.01 nm x 999 / 1000 / 1001 ft: conflict iff d < 3 nm and |dh| < 1000 ft (strict), both aircraft
    get -200 (+ shaping), env done, min separation reported."""
    cases = [(d, dh) for d in (2.99, 3.0, 3.01) for dh in (999.0, 1000.0, 1001.0)]
    env = _env("LOWW_random", len(cases), 2, dtype)
    for b, (d, dh) in enumerate(cases):
        env.set_state(b, 0, 30.0, 60.0, 15000.0, 0.0, 250.0)
        env.set_state(b, 1, 30.0 + d, 60.0, 15000.0 + dh, 0.0, 250.0)
    a = np.zeros((len(cases), 2, 3))
    a[:, :, 0] = 0.5
    a[:, 0, 1] = 2 * 15000.0 / 38000.0 - 1
    for b, (d, dh) in enumerate(cases):
        a[b, 1, 1] = 2 * (15000.0 + dh) / 38000.0 - 1
    a[:, :, 2] = -1.0
    env.step(a)
    for b, (d, dh) in enumerate(cases):
        expect = d < 3.0 and dh < 1000.0
        assert bool(env.flags[b, 0] & H.F_CONFLICT) == expect == bool(env.flags[b, 1] & H.F_CONFLICT), (d, dh)
        assert bool(env.done[b]) == expect
        assert abs(env.min_sep[b] - d) < 1e-4
        if expect:
            assert -400.0 <= env.reward[b] < -390.0 and np.all(env.ac_reward[b] < -195.0)


def test_n1_in_multi_aircraft_env_equals_single_env():
    """Aircraft far apart never interact: each aircraft of a 4-aircraft env behaves exactly like a single-aircraft env
    fed the same actions (rewards, flags, observations), and the env reward is their sum."""
    rng = np.random.default_rng(3)
    starts = [(20.0, 60.0, 15000.0, 90.0, 250.0), (55.0, 70.0, 17000.0, 230.0, 240.0), (60.0, 30.0, 19000.0, 300.0, 230.0),
              (30.0, 15.0, 21000.0, 20.0, 220.0)]
    multi = _env("LOWW_random", 1, 4)
    singles = [_env("LOWW", 1, 1) for _ in range(4)]
    for k, st in enumerate(starts):
        multi.set_state(0, k, *st)
        singles[k].set_state(0, 0, *st)
    for t in range(120):
        if t % 15 == 0:
            a = rng.uniform(-1, 1, (1, 4, 3))
        multi.step(a)
        tot = 0.0
        for k in range(4):
            singles[k].step(a[:, k:k + 1])
            assert multi.flags[0, k] == singles[k].flags[0, 0]
            assert np.array_equal(multi.obs[0, k], singles[k].obs[0, 0])
            assert multi.ac_reward[0, k] == singles[k].reward[0]
            tot += singles[k].reward[0]
        assert abs(multi.reward[0] - tot) < 1e-9
        if multi.done[0]:
            break


def test_win_hands_over_and_env_continues_until_all_done():
    env = _env("LOWW_random", 1, 2)
    env.set_state(0, 0, 48.9, 31.9, 3300.0, 345.0, 200.0)
    env.set_state(0, 1, 20.0, 60.0, 15000.0, 90.0, 250.0)
    a = np.zeros((1, 2, 3))
    a[0, 0] = [0.0, 2 * 2700.0 / 38000.0 - 1, 2 * 345.0 / 360.0 - 1]
    for t in range(60):
        env.step(a)
        if env.flags[0, 0] & H.F_WON:
            break
    assert env.flags[0, 0] & H.F_WON and not env.done[0] and env.reward[0] > 10000
    assert int(env.active_mask[0]) == 2
    env.step(a)
    assert env.flags[0, 0] == H.F_INACTIVE and np.all(env.obs[0, 0] == 0) and env.ac_reward[0, 0] == 0


def test_noise_area_penalty_and_flag():
    from envs.atc import scenarios
    env = _env(scenarios.LOWWDense(), 3, 1)
    for b, st in enumerate([(43.0, 38.0, 5000.0, 90.0, 200.0), (43.0, 38.0, 9000.0, 90.0, 200.0), (20.0, 60.0, 5000.0, 90.0, 200.0)]):
        env.set_state(b, 0, *st)
    base = _env("LOWW_random", 3, 1)
    for b, st in enumerate([(43.0, 38.0, 5000.0, 90.0, 200.0), (43.0, 38.0, 9000.0, 90.0, 200.0), (20.0, 60.0, 5000.0, 90.0, 200.0)]):
        base.set_state(b, 0, *st)
    a = np.zeros((3, 1, 3))
    a[:, 0, 1] = [2 * 5000 / 38000 - 1, 2 * 9000 / 38000 - 1, 2 * 5000 / 38000 - 1]
    a[:, 0, 2] = -0.5
    env.step(a)
    base.step(a)
    assert [bool(f & H.F_NOISE) for f in env.flags[:, 0]] == [True, False, False]
    assert abs((base.reward[0] - env.reward[0]) - 0.02) < 1e-12   # penalty of the first noise area
    assert env.reward[1] == base.reward[1] and not env.done.any()


def test_auto_reset_bookkeeping_and_random_entry_determinism():
    env = _env("LOWW_random", 64, 1, np.float32, auto_reset=True, random_entry=True, seed=11)
    env2 = _env("LOWW_random", 64, 1, np.float32, auto_reset=True, random_entry=True, seed=11)
    env3 = _env("LOWW_random", 64, 1, np.float32, auto_reset=True, random_entry=True, seed=12)
    assert np.array_equal(env.x, env2.x) and not np.array_equal(env.x, env3.x)
    entries = {(float(np.float32(e[0])), float(np.float32(e[1]))) for e in H.compiled("LOWW_random").entrypoints}
    assert {(float(x), float(y)) for x, y in zip(env.x, env.y)} <= entries
    rng = np.random.default_rng(0)
    ends = 0
    for t in range(1500):
        if t % 20 == 0:
            a = rng.uniform(-1, 1, (64, 1, 3)).astype(np.float32)
        before_t = env.timesteps.copy()
        before_ret = env.total_reward.copy()
        env.step(a)
        d = env.done.astype(bool)
        if d.any():
            ends += int(d.sum())
            assert np.all(env.timesteps[d] == 0) and np.all(env.total_reward[d] == 0) and np.all(env.actions_taken[d] == 0)
            assert np.array_equal(env.ep_length[d], before_t[d] + 1)
            assert np.allclose(env.ep_return[d], before_ret[d] + env.reward[d], rtol=1e-6)
            # the returned observation of a reset env is the RAW reset state (x in nm, h in ft ...)
            assert np.all(env.obs[d, 0, 2] >= 13000.0)
        assert np.all(env.timesteps[~d] == before_t[~d] + 1)
    assert ends > 30 and np.all(env.episodes >= 1)


def test_thread_count_does_not_change_results():
    a = np.random.default_rng(1).uniform(-1, 1, (256, 4, 3)).astype(np.float32)
    outs = []
    for threads in (1, 4):
        O.set_threads(threads)
        env = _env("LOWW_random", 256, 4, np.float32, auto_reset=True)
        for t in range(50):
            env.step(a)
        outs.append((env.obs.copy(), env.reward.copy(), env.flags.copy(), env.x.copy()))
    O.set_threads(1)
    for u, v in zip(outs[0], outs[1]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_extension_known_answers(dtype):
    """Hand-computed one-step outcomes of the multi-aircraft semantics (helpers.extension_known_answers): override order,
    per-aircraft rewards and their env sum, termination, hand-over masks — on both instantiations of the oracle."""
    for name, params, t0, aircraft, want in H.extension_known_answers():
        env = _env("LOWW_random", 1, len(aircraft), dtype, shaping=False, **params)
        for k, (st, _) in enumerate(aircraft):
            env.set_state(0, k, *st)
        env.timesteps[0] = t0
        a = np.array([[act for _, act in aircraft]], dtype=np.float32)
        env.step(a)
        assert [int(f) for f in env.flags[0]] == want["flags"], name
        assert np.allclose(env.ac_reward[0], want["ac_reward"], rtol=0, atol=2e-3 if dtype == np.float32 else 1e-9), (name, env.ac_reward[0])
        assert abs(env.reward[0] - sum(want["ac_reward"])) <= (4e-3 if dtype == np.float32 else 1e-9), name
        assert bool(env.done[0]) == want["done"] and int(env.active_mask[0]) == want["mask_after"], name
        if name == "win hands over, env continues":   # the next step: a handed-over aircraft is inactive — zero observation, zero reward
            env.step(a)
            assert int(env.flags[0, 0]) == H.F_INACTIVE and np.all(env.obs[0, 0] == 0) and env.ac_reward[0, 0] == 0
            assert abs(env.reward[0] - (-0.10)) < 1e-6 and not env.done[0]
