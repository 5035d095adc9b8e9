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
    if name == "Dyadic":  # tie-break sector of tests/golden/g8_tiebreak.npz (integer / dyadic vertices, shared borders)
        g = golden_npz("g8_tiebreak.npz")
        s = scenarios.Scenario()
        rings, heights = json.loads(str(g["rings"])), g["heights"]
        s.mvas = [model.MinimumVectoringAltitude([tuple(p) for p in ring], int(h)) for ring, h in zip(rings, heights)]
        s.runway = model.Runway(*[float(v) for v in g["runway"]])
        s.airspace = model.Airspace(s.mvas, s.runway)
        s.entrypoints = [model.EntryPoint(2, 30, 90, [150])]
        s.noise_areas = []
        return s
    if name == "Sliver":  # near-horizontal LONG shared edges (dy = 0.03 over dx = 50): the fp32 x-intersection of
        s = scenarios.Scenario()   # such an edge is ~2.5e-3 nm off its float64 line — the lookup grid's margins must cover it
        rings = [[(5, 5), (55, 5), (55, 20.03), (5, 20), (5, 5)],
                 [(5, 20), (55, 20.03), (55, 39.98), (5, 40.01), (5, 20)],
                 [(5, 40.01), (55, 39.98), (55, 60), (5, 60), (5, 40.01)]]
        s.mvas = [model.MinimumVectoringAltitude(r, h) for r, h in zip(rings, (3000, 4200, 2500))]
        s.runway = model.Runway(30, 12, 400, 90)
        s.airspace = model.Airspace(s.mvas, s.runway)
        s.entrypoints = [model.EntryPoint(8, 30, 90, [120, 140])]
        s.noise_areas = []
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


# ---------------------------------------------------------------------------------------------- compact wide fixture (g9)
class WideFixture:
    """tests/golden/g9_wide.npz: per-step flags / done / actions_taken / reward of every reference step, observation and
    float64 state on sampled rows, actions stored at their change points."""

    def __init__(self, name="g9_wide.npz"):
        z = golden_npz(name)
        self.episodes = json.loads(str(z["episodes"]))
        self.flags, self.done, self.actions_taken = z["flags"], z["done"], z["actions_taken"]
        self.reward = z["reward"].astype(np.float64)
        self.samp_rows, self.obs, self.state = z["samp_rows"], z["obs"], z["state"]
        n = len(self.flags)
        idx = np.zeros(n, np.int64)          # forward-fill the action change points
        idx[z["act_rows"]] = np.arange(len(z["act_rows"]))
        idx = np.maximum.accumulate(idx)
        self.action = z["act_vals"][idx]
        self.samp_index = np.full(n, -1, np.int64)
        self.samp_index[self.samp_rows] = np.arange(len(self.samp_rows))

    def groups(self):
        out = {}
        for ep in self.episodes:
            out.setdefault((ep["scen"], ep["dt"], ep["shaping"], ep["normalize"], ep["discrete"]), []).append(ep)
        return out


WRAP_ROWS = [0]   # observation rows accepted on the other side of the +-180 deg wrap (see obs_close)


def obs_close(obs, gold, tol, normalize):
    """|obs - gold| <= tol per component, with ONE stated equivalence: observation word 9, relative_angle(phi_to_runway, phi)
    (atc_gym.py:284-287, model.py:340-342), jumps from +180 to -180 deg where the heading is EXACTLY opposite the runway heading.
    At timesteps like 0.05 s a turning aircraft passes through that heading exactly in decimal arithmetic (integer heading + k x 0.15
    deg) and the side the reference reports is decided by the rounding noise of its float64 heading accumulation (1e-14 deg), which
    the fp32 path's fixed-point heading (2^-23 deg steps, include/atc_step.h) cannot and need not follow: -180 and +180 are the
    same angle.  Such rows — both values within 2e-5 of the wrap, on opposite sides — count as equal (and are counted)."""
    d = np.abs(np.asarray(obs, dtype=np.float64) - gold)
    edge = 1.0 if normalize else 180.0
    o9, g9 = np.asarray(obs, dtype=np.float64)[..., 9], gold[..., 9]
    wrap = (np.abs(np.abs(o9) - edge) <= 2e-5 * edge) & (np.abs(np.abs(g9) - edge) <= 2e-5 * edge) & (o9 * g9 < 0)
    if wrap.any():
        d[..., 9] = np.where(wrap, np.abs(np.abs(o9) - np.abs(g9)), d[..., 9])
        WRAP_ROWS[0] += int(wrap.sum())
    return np.all(d <= tol)


def replay_wide(fx, make_backend, obs_tol, state_tol, rew_tol, max_envs=4096):
    """Runs every episode of the wide fixture through a lock-step backend (the episodes of one configuration side by
    side as the envs of one batch).  make_backend(scen, dt, shaping, normalize, discrete, B) returns an object with
    place(b, init_state, init_timesteps, init_last_action) and step(actions[B,1,3]) -> (obs[B,10], reward[B], done[B],
    flags[B], actions_taken[B], state[B,5]).  Integer outputs are compared exactly on EVERY reference step.

    The plain 1e-5 bar applies to EVERY comparison, next to the FAF as well: rounds 1-3 widened the tolerance of the reward and
    of obs[8] inside 0.25 nm of the FAF (the bearing to it is ill-conditioned there and fp32 speed / heading state put the
    position ~2e-6 nm off the float64 reference); since ABI 18 speed and heading are 32-bit fixed point and the displacement
    is float64 with dithered rounding (include/atc_step.h) — positions stay within ~1e-6 nm over 6 000 steps, 2e-7 typically —
    and the exception is retired.  Returns the number of steps compared."""
    total = 0
    for (scen, dt, shaping, normalize, discrete), eps in fx.groups().items():
        half = 0.5 * compiled(scen).norm_max.astype(np.float64)
        for lo in range(0, len(eps), max_envs):
            ge = eps[lo:lo + max_envs]
            B = len(ge)
            be = make_backend(scen, dt, shaping, normalize, discrete, B)
            for b, ep in enumerate(ge):
                be.place(b, ep["init_state"], ep["init_timesteps"], ep["init_last_action"])
            steps = np.array([ep["steps"] for ep in ge])
            starts = np.array([ep["start"] for ep in ge])
            for t in range(int(steps.max())):
                live = t < steps
                rows = np.where(live, starts + t, starts)
                obs, rew, done, flags, acts, state = be.step(fx.action[rows].astype(np.float32).reshape(B, 1, 3))
                lr = rows[live]
                assert np.array_equal(np.asarray(flags)[live].astype(np.uint8), fx.flags[lr]), (scen, t)
                assert np.array_equal(np.asarray(done)[live].astype(np.uint8), fx.done[lr]), (scen, t)
                assert np.array_equal(np.asarray(acts)[live], fx.actions_taken[lr]), (scen, t)
                gw = fx.reward[lr]
                st = np.asarray(state, dtype=np.float64)[live]
                # the fixture stores rewards as float32 (6e-8 relative)
                assert np.all(np.abs(np.asarray(rew, dtype=np.float64)[live] - gw)
                              <= (rew_tol + 1e-7) * np.maximum(1.0, np.abs(gw))), (scen, t)
                si = fx.samp_index[lr]
                has = si >= 0
                if has.any():
                    go = fx.obs[si[has]].astype(np.float64)
                    tol = (obs_tol if normalize else obs_tol * half) * np.ones((int(has.sum()), 10))
                    assert obs_close(np.asarray(obs, dtype=np.float64)[live][has], go, tol, normalize), (scen, t)
                    gs = fx.state[si[has]]
                    assert np.all(np.abs(st[has] - gs) <= state_tol * np.maximum(1.0, np.abs(gs))), (scen, t)
                total += int(live.sum())
            be.close()
    assert total == len(fx.flags)
    return total


def replay_wide_interleaved(fx, make_backend, N, obs_tol, state_tol, rew_tol, max_envs=1024, chunk=1):
    """The reference pins MULTI-aircraft envs too, where the extension's own rules are switched off: with a separation minimum of 0
    nobody is ever in conflict, and with the reference's episode rule (ATC_M_KEEP_ACTIVE: no hand-over, any terminal aircraft ends the
    episode) an env of N aircraft IS N reference episodes flown side by side on one clock.  So N episodes of the wide fixture — same
    configuration, same starting time step, similar length — become the N aircraft of one env, each fed its own recorded actions, and
    every aircraft must reproduce its episode's record step by step (flags exact; sampled observation and state within the bars of
    replay_wide), the env the sums: reward = the episodes' rewards added up, actions_taken = their counters added up, done = any of
    them done — until the shortest episode ends.  That is the lane -> aircraft mapping, the per-env reductions and the shared clock
    of the batched step checked against the reference over whole episodes, which the one-aircraft replays cannot see.

    make_backend(scen, dt, shaping, normalize, discrete, B, N) -> object with place(b, k, init_state, init_last_action),
    set_timesteps(b, t) and step(actions[B, N, 3]) -> (obs[B, N, 10], reward[B], done[B], flags[B, N], actions_taken[B],
    state[B, N, 5]).  chunk > 1: the backend's rollout(actions[chunk, B, N, 3]) flies `chunk` steps per call (a multi-step launch)
    and returns the same tuple with a leading step axis, counters and state as they are after the LAST step of the chunk.
    Returns (aircraft-steps compared, envs flown)."""
    total = envs = 0
    for (scen, dt, shaping, normalize, discrete), eps in fx.groups().items():
        half = 0.5 * compiled(scen).norm_max.astype(np.float64)
        by_t0 = {}
        for ep in eps:
            by_t0.setdefault(ep["init_timesteps"], []).append(ep)
        packs = []
        for t0, ge in by_t0.items():
            ge = sorted(ge, key=lambda ep: ep["steps"])
            packs += [ge[i:i + N] for i in range(0, len(ge) - N + 1, N)]   # (a remainder of fewer than N episodes stays out)
        for lo in range(0, len(packs), max_envs):
            pk = packs[lo:lo + max_envs]
            B = len(pk)
            be = make_backend(scen, dt, shaping, normalize, discrete, B, N)
            for b, env_eps in enumerate(pk):
                be.set_timesteps(b, env_eps[0]["init_timesteps"])
                for k, ep in enumerate(env_eps):
                    be.place(b, k, ep["init_state"], ep["init_last_action"])
            starts = np.array([[ep["start"] for ep in env_eps] for env_eps in pk])       # [B, N]
            length = np.array([min(ep["steps"] for ep in env_eps) for env_eps in pk])     # the env's episode: its shortest
            def rows_at(t):
                live = t < length                                                          # [B]
                return live, np.where(live[:, None], starts + t, starts)                  # (finished envs replay a harmless row)

            def compare(t, live, rows, obs, rew, done, flags, acts, state):
                lr = rows[live]                                                            # [L, N]
                assert np.array_equal(np.asarray(flags)[live].astype(np.uint8), fx.flags[lr]), (scen, N, t)
                assert np.array_equal(np.asarray(done)[live].astype(bool), fx.done[lr].astype(bool).any(axis=1)), (scen, N, t)
                gw = fx.reward[lr]
                assert np.all(np.abs(np.asarray(rew, dtype=np.float64)[live] - gw.sum(axis=1))
                              <= (rew_tol + 1e-7) * np.maximum(1.0, np.abs(gw)).sum(axis=1)), (scen, N, t)
                si = fx.samp_index[lr]
                has = si >= 0
                if has.any():
                    go = fx.obs[si[has]].astype(np.float64)
                    tol = (obs_tol if normalize else obs_tol * half) * np.ones((int(has.sum()), 10))
                    assert obs_close(np.asarray(obs, dtype=np.float64)[live][has], go, tol, normalize), (scen, N, t)
                if acts is not None:
                    assert np.array_equal(np.asarray(acts)[live], fx.actions_taken[lr].sum(axis=1)), (scen, N, t)
                    if has.any():
                        gs = fx.state[si[has]]
                        assert np.all(np.abs(np.asarray(state, dtype=np.float64)[live][has] - gs) <= state_tol * np.maximum(1.0, np.abs(gs))), (scen, N, t)
                return int(live.sum()) * N

            T = int(length.max())
            t = 0
            while t < T:
                if chunk == 1:
                    live, rows = rows_at(t)
                    total += compare(t, live, rows, *be.step(fx.action[rows].astype(np.float32)))
                    t += 1
                    continue
                lr = [rows_at(t + c) for c in range(chunk)]
                obs, rew, done, flags, acts, state = be.rollout(np.stack([fx.action[r].astype(np.float32) for _, r in lr]))
                for c, (live, rows) in enumerate(lr):
                    last = c == chunk - 1
                    total += compare(t + c, live, rows, obs[c], rew[c], done[c], flags[c], acts if last else None, state if last else None)
                t += chunk
            envs += B
            be.close()
    return total, envs


# ---------------------------------------------------------------------------------------------- extension known answers
# Hand-computed outcomes of ONE step for the build-defined multi-aircraft semantics (separation, hand-over,
# override order, per-aircraft rewards summed per env), with reward shaping OFF so that every reward is a small exact sum:
# base -0.05 dt per aircraft (atc_gym.py:137), -1 per refused target (:312-315), -50 outside (:156-161), -200 below the MVA
# (:149-153), the conflict reward for a lost separation (extension), 10000 + 5 (limit - t) for the corridor (:163-169), -200
# for a time-out (:171-173), in the chain's order.  Nothing here comes from the oracle or the kernels; the one borrowed fact is
# the REFERENCE's verdict that WIN_STATE + WIN_ACTION is inside the corridor after the step (row 523257 of tests/golden/g9_wide.npz).
WIN_STATE = (48.49859896879164, 33.44283053975357, 2700.0003457069397, 344.0000009536743, 200.0)
WIN_ACTION = (0.0, -0.8578947186470032, 0.9111111164093018)
FAR_A = (20.0, 60.0, 15000.0, 90.0, 250.0)      # inside the airspace (the reference's lattice G3: MVA 3500 / 2700 ft there),
FAR_B = (55.0, 70.0, 21000.0, 20.0, 220.0)      # far above it and > 20 nm from each other and from everything placed below


def hold_action(state):
    """The continuous action whose targets are the state's own speed / altitude / heading (nothing changes but the position)."""
    x, y, h, phi, v = state
    return ((v - 200.0) / 100.0, h / 19000.0 - 1.0, phi / 180.0 - 1.0)


def extension_known_answers():
    """[(name, params, timesteps before the step, [(state, action)] per aircraft, expected)] with expected =
    {flags: [...], ac_reward: [...], done: bool, mask_after: int}; params = dict(timestep_limit, conflict_reward)."""
    W, C, O_, B, T = F_WON, F_CONFLICT, F_OUTSIDE, F_BELOW_MVA, F_TIMEOUT
    near0, near1 = (30.0, 60.0, 15000.0, 0.0, 250.0), (32.0, 60.0, 15500.0, 0.0, 250.0)          # 2 nm, 500 ft
    low0, low1 = (30.0, 60.0, 1000.0, 0.0, 250.0), (32.0, 60.0, 1200.0, 0.0, 250.0)                # below every MVA (>= 2600 ft)
    out0 = (1.0, 1.0, 15000.0, 90.0, 250.0)                                                       # outside LOWW's bounds
    k = []
    k.append(("conflict pair + bystander", {}, 0, [(near0, hold_action(near0)), (near1, hold_action(near1)), (FAR_A, hold_action(FAR_A))],
              dict(flags=[C, C, 0], ac_reward=[-200.0, -200.0, -0.05], done=True, mask_after=0b111)))
    k.append(("conflict overrides below-MVA", dict(conflict_reward=-123.0), 0,
              [(low0, hold_action(low0)), (low1, hold_action(low1)), (FAR_A, hold_action(FAR_A))],
              dict(flags=[B | C, B | C, 0], ac_reward=[-123.0, -123.0, -0.05], done=True, mask_after=0b111)))
    k.append(("below-MVA alone", {}, 0, [(low0, hold_action(low0)), (FAR_A, hold_action(FAR_A)), (FAR_B, hold_action(FAR_B))],
              dict(flags=[B, 0, 0], ac_reward=[-200.0, -0.05, -0.05], done=True, mask_after=0b111)))
    k.append(("outside the airspace", {}, 0, [(out0, hold_action(out0)), (FAR_A, hold_action(FAR_A)), (FAR_B, hold_action(FAR_B))],
              dict(flags=[O_, 0, 0], ac_reward=[-50.0, -0.05, -0.05], done=True, mask_after=0b111)))
    k.append(("time-out ends every aircraft", dict(timestep_limit=5), 5,
              [(near0, hold_action(near0)), (FAR_A, hold_action(FAR_A)), (FAR_B, hold_action(FAR_B))],
              dict(flags=[T, T, T], ac_reward=[-200.0, -200.0, -200.0], done=True, mask_after=0b111)))
    k.append(("win + conflict: two aircraft on the same winning state", {}, 0, [(WIN_STATE, WIN_ACTION), (WIN_STATE, WIN_ACTION), (FAR_A, hold_action(FAR_A))],
              dict(flags=[W | C, W | C, 0], ac_reward=[10000.0 + 5 * 5999, 10000.0 + 5 * 5999, -0.05], done=True, mask_after=0b100)))
    k.append(("win hands over, env continues", {}, 0, [(WIN_STATE, WIN_ACTION), (FAR_A, hold_action(FAR_A)), (FAR_B, hold_action(FAR_B))],
              dict(flags=[W, 0, 0], ac_reward=[10000.0 + 5 * 5999, -0.05, -0.05], done=False, mask_after=0b110)))
    k.append(("time-out overrides the win", dict(timestep_limit=40), 40, [(WIN_STATE, WIN_ACTION), (FAR_A, hold_action(FAR_A)), (FAR_B, hold_action(FAR_B))],
              dict(flags=[W | T, T, T], ac_reward=[-200.0, -200.0, -200.0], done=True, mask_after=0b110)))
    k.append(("win bonus runs out at the limit", dict(timestep_limit=6000), 5999, [(WIN_STATE, WIN_ACTION), (FAR_A, hold_action(FAR_A)), (FAR_B, hold_action(FAR_B))],
              dict(flags=[W, 0, 0], ac_reward=[10000.0, -0.05, -0.05], done=False, mask_after=0b110)))
    k.append(("refused targets cost 1 each and change nothing", {}, 0, [(FAR_A, (1.5, -1.2, 0.0)), (FAR_B, hold_action(FAR_B)), (near0, hold_action(near0))],
              dict(flags=[F_INVALID_V | F_INVALID_H, 0, 0], ac_reward=[-2.05, -0.05, -0.05], done=False, mask_after=0b111)))
    return k
