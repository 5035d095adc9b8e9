"""GPU parity tests: the HIP path (through the C-ABI of libatcstep.so) against
  (a) the golden vectors captured from the reference (tests/golden/), and
  (b) the fp32 CPU oracle on identical seeded inputs (integer outputs bit-exact, fp32 values within 1e-5).
Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TERMINAL = H.F_BELOW_MVA | H.F_OUTSIDE | H.F_TIMEOUT | H.F_CONFLICT


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _sector(name, grid_cell=None):
    from atc_hip import lib
    return lib.Scenario(H.compiled(name, grid_cell=grid_cell))


# ------------------------------------------------------------------------------------------------ reference's own tests
class TestReferenceModelTests:
    """The 8 cases of the reference's envs/atc/model_test.py, written against the mirrored classes (which evaluate on
    the GPU)."""

    def setup_method(self):
        from envs.atc import model
        self.model = model
        self.mvas = [
            model.MinimumVectoringAltitude([(15, 0), (35, 0), (35, 26)], 3500),
            model.MinimumVectoringAltitude([(15, 0), (35, 26), (35, 30), (15, 30), (15, 27.8)], 2400),
            model.MinimumVectoringAltitude([(15, 30), (35, 30), (35, 40), (15, 40)], 4000),
            model.MinimumVectoringAltitude([(0, 10), (15, 0), (15, 28.7), (0, 17)], 8000),
            model.MinimumVectoringAltitude([(0, 17), (15, 28.7), (15, 40), (0, 32)], 6500),
        ]
        self.runway = model.Runway(20, 20, 0, 180)
        self.airspace = model.Airspace(self.mvas, self.runway)

    def test_airspace_get_mvas(self):
        assert self.airspace.get_mva_height(34, 1) == 3500

    def test_inside_corridor_true_when_on_correct_side(self):
        mva = self.airspace.get_mva_height(self.runway.corridor.faf[0][0], self.runway.corridor.faf[1][0])
        assert self.runway.inside_corridor(19, 10, mva + 300, 30) is True

    def test_inside_corridor_false_when_right_position_with_wrong_heading(self):
        mva = self.airspace.get_mva_height(self.runway.corridor.faf[0][0], self.runway.corridor.faf[1][0])
        assert self.runway.inside_corridor(19, 10, mva, 330) is False

    def test_inside_corridor_angle_cases(self):
        c = self.runway.corridor
        assert c._inside_corridor_angle(21, 10, 30) is False
        assert c._inside_corridor_angle(19, 10, 340) is False
        assert c._inside_corridor_angle(19, 10, 190) is False
        assert c._inside_corridor_angle(21, 10, 340) is True

    def test_bounding_box_airspace_multiple_mva(self):
        assert self.airspace.get_bounding_box() == (0.0, 0.0, 35.0, 40.0)

    def test_outside_raises(self):
        with pytest.raises(ValueError):
            self.airspace.get_mva_height(-5, -5)

    def test_ray_tracing_function(self):
        assert self.model.ray_tracing(30, 10, [(15, 0), (35, 0), (35, 26)]) is True
        assert self.model.ray_tracing(16, 20, [(15, 0), (35, 0), (35, 26)]) is False
        # the polygon's device sector is compiled once and reused; arrays of points go in one launch
        assert len(self.model._ray_sectors) == 1
        hit = self.model.ray_tracing(np.array([30.0, 16.0, 34.9]), np.array([10.0, 20.0, 0.1]), [(15, 0), (35, 0), (35, 26)])
        assert hit.tolist() == [True, False, True] and len(self.model._ray_sectors) == 1
        assert self.airspace.get_mva_heights([34.0, -5.0], [1.0, -5.0]).tolist() == [3500, -1]


# ------------------------------------------------------------------------------------------------ G3 / G4 / G5 lattices
@pytest.mark.parametrize("grid_cell", [None, 0.0625, 0.125, 0.5, 1.0])
@pytest.mark.parametrize("scen", ["LOWW", "Simple"])
def test_mva_lattice_golden(scen, grid_cell):
    g = H.golden_npz("g3_mva.npz")
    s = _sector(scen, grid_cell)
    xs, ys = g[scen + "_xs"], g[scen + "_ys"]
    X, Y = np.meshgrid(xs, ys)
    got = s.query_mva(X.ravel(), Y.ravel(), use_grid=grid_cell is not None).reshape(len(ys), len(xs))
    assert np.array_equal(got, g[scen + "_lattice"])


@pytest.mark.parametrize("grid_cell", [None, 0.0625, 0.125, 0.25, 0.5, 2.0])
@pytest.mark.parametrize("scen", ["LOWW", "Simple", "Sliver"])
def test_mva_bitexact_vs_oracle_dense_and_edges(scen, grid_cell):
    """1.5 M random points + points hugging every polygon edge/vertex (within a few fp32 ulps): the polygon index must be
    identical to the fp32 oracle's ordered scan, with and without the lookup grid."""
    from oracle import oracle as O
    comp = H.compiled(scen, grid_cell)
    s = _sector(scen, grid_cell)
    rng = np.random.default_rng(3)
    b = comp.bbox
    pts = [np.stack([rng.uniform(b[0] - 2, b[2] + 2, 1500000), rng.uniform(b[1] - 2, b[3] + 2, 1500000)], 1)]
    for ring in comp.mva_rings:
        for k in range(len(ring) - 1):
            t = rng.uniform(0, 1, 400)[:, None]
            p = ring[k][None, :] * (1 - t) + ring[k + 1][None, :] * t
            jitter = rng.integers(-3, 4, p.shape) * np.spacing(np.abs(p).astype(np.float32)).astype(np.float64)
            pts.append(p + jitter)
            pts.append(np.repeat(ring[k][None, :], 49, 0) + np.stack(np.meshgrid(np.arange(-3, 4), np.arange(-3, 4)), -1)
                       .reshape(-1, 2) * np.spacing(np.abs(ring[k]).astype(np.float32)).astype(np.float64))
    pts = np.concatenate(pts).astype(np.float32)
    q = O.OracleQueries(comp, np.float32)
    exp = q.mva(pts[:, 0], pts[:, 1])
    got = s.query_mva(pts[:, 0], pts[:, 1], use_grid=grid_cell is not None)
    assert np.array_equal(got, exp)
    assert (exp >= 0).sum() > 100000 and (exp < 0).sum() > 1000


@pytest.mark.parametrize("scen", ["LOWW", "UnitTest", "Simple"])
def test_corridor_tables_golden(scen):
    g = H.golden_npz("g4_corridor.npz")
    s = _sector(scen)
    xs, ys, hs, phis = (g[scen + k] for k in ("_xs", "_ys", "_hs", "_phis"))
    shape = tuple(g[scen + "_inside_shape"])
    exp = np.unpackbits(g[scen + "_inside"])[:int(np.prod(shape))].reshape(shape)
    Y, X, Hh, P = np.meshgrid(ys, xs, hs, phis, indexing="ij")
    got = s.query_corridor(X.ravel(), Y.ravel(), Hh.ravel(), P.ravel()).reshape(shape)
    assert np.array_equal(got, exp)
    ashape = tuple(g[scen + "_angle_shape"])
    aexp = np.unpackbits(g[scen + "_angle"])[:int(np.prod(ashape))].reshape(ashape)
    Y, X, P = np.meshgrid(ys, xs, phis, indexing="ij")
    agot = s.query_corridor(X.ravel(), Y.ravel(), np.zeros(X.size), P.ravel(), angle_only=True).reshape(ashape)
    assert np.array_equal(agot, aexp)
    p2 = g[scen + "_phis_unwrapped"]
    Y, X, P = np.meshgrid(ys, xs, p2, indexing="ij")
    ugot = s.query_corridor(X.ravel(), Y.ravel(), np.full(X.size, hs[1]), P.ravel()).reshape(Y.shape)
    assert np.array_equal(ugot, g[scen + "_inside_unwrapped"])


def test_corridor_random_vs_oracle():
    from oracle import oracle as O
    comp = H.compiled("LOWW")
    s = _sector("LOWW")
    rng = np.random.default_rng(11)
    n = 400000
    fx, fy = comp.corridor["faf"]
    x = (fx + rng.uniform(-5, 5, n)).astype(np.float32)
    y = (fy + rng.uniform(-6, 3, n)).astype(np.float32)
    h = rng.uniform(2000, 5000, n).astype(np.float32)
    phi = np.where(rng.random(n) < 0.3, np.round(rng.uniform(-400, 800, n)), rng.uniform(-400, 800, n)).astype(np.float32)
    exp = O.OracleQueries(comp, np.float32).corridor(x, y, h, phi)
    got = s.query_corridor(x, y, h, phi)
    assert np.array_equal(got, exp)
    assert exp.sum() > 1000


def test_shaping_golden():
    g = H.golden_npz("g5_shaping.npz")
    s = _sector("LOWW")
    out = s.query_shaping(g["d_faf"], g["phi_rel_faf"], g["phi_plane"], g["h"], g["on_gp"])
    assert np.max(np.abs(out[:, 0] - g["pos"])) <= 1e-5
    assert np.max(np.abs(out[:, 1] - g["ang"])) <= 1e-5
    assert np.max(np.abs(out[:, 2] - g["gs"])) <= 1e-5


@pytest.mark.parametrize("grid_cell", [None, 0.0625, 0.125, 0.25, 0.5, 1.0])
def test_tiebreak_points_exact(grid_cell):
    """G8: sector with integer / dyadic vertices (the fp32 blob holds exactly the reference's polygons): vertices, edge
    points, shared borders, overlapping polygons and points 2^-10 nm either side of every edge must get the REFERENCE's
    answer (model.py:282-289,318-337), with the ordered scan and through every lookup grid."""
    g = H.golden_npz("g8_tiebreak.npz")
    s = _sector("Dyadic", grid_cell)
    got = s.query_mva(g["pts"][:, 0], g["pts"][:, 1], use_grid=grid_cell is not None)
    assert np.array_equal(got, g["h"])


# ------------------------------------------------------------------------------------------------ trajectories (G2, G6)
class HipGymAdapter:
    """Replays golden episodes through the single-env AtcGym mirror (what a user of the reference would call)."""

    def __init__(self):
        self.env = None
        self.key = None

    def configure(self, scen, dt, shaping, normalize, discrete):
        from envs.atc import atc_gym, model
        key = (scen, dt, shaping, normalize, discrete)
        if key != self.key:
            if self.env is not None:
                self.env.close()
            sp = model.SimParameters(dt, reward_shaping=shaping, normalize_state=normalize,
                                     discrete_action_space=discrete)
            self.env = atc_gym.AtcGym(sim_parameters=sp, scenario=H.make_scenario(scen))
            self.key = key

    def reset(self):
        return self.env.reset()

    def set_state(self, x, y, h, phi, v):
        ap = self.env._airplane
        ap.x, ap.y, ap.h, ap.phi, ap.v = x, y, h, phi, v

    def set_counters(self, timesteps, last_action):
        self.env._vec.timesteps[0] = int(timesteps)
        self.env.timesteps = int(timesteps)
        self.env.last_action = last_action

    def step(self, action):
        e = self.env
        obs, rew, done, info = e.step(action)
        r = H.StepRecord()
        r.obs, r.raw = np.asarray(obs), np.asarray(info["original_state"])
        r.reward, r.done = float(rew), int(done)
        r.flags = int(e._vec.flags[0, 0])
        r.timesteps, r.actions_taken, r.total_reward = e.timesteps, e.actions_taken, float(e.total_reward)
        r.state = np.array(e._vec.get_state(0, 0))
        return r


def _hip_check(npz, ep, stats):
    half_range = 0.5 * H.compiled(ep["scen"]).norm_max.astype(np.float64)

    def check(t, row, rec):
        assert rec.flags == int(npz["flags"][row]), (t, rec.flags, int(npz["flags"][row]))
        assert rec.done == int(npz["done"][row])
        assert rec.timesteps == int(npz["timesteps"][row])
        assert rec.actions_taken == int(npz["actions_taken"][row])
        go, gr = npz["obs"][row].astype(np.float64), npz["raw"][row].astype(np.float64)
        if ep["normalize"]:
            assert np.all(np.abs(rec.obs - go) <= 1e-5), (t, rec.obs, go)
        else:
            assert np.all(np.abs(rec.obs - go) <= 1e-5 * half_range), (t, rec.obs, go)
        assert np.all(np.abs(rec.raw - gr) <= 1e-5 * half_range), (t, rec.raw, gr)
        gw = float(npz["reward"][row])
        assert abs(rec.reward - gw) <= 1e-5 * max(1.0, abs(gw)), (t, rec.reward, gw)
        stats["steps"] += 1

    return check


def test_scripted_trajectories_through_atcgym():
    npz = H.golden_npz("g2_scripted.npz")
    stats = {"steps": 0}
    ad = HipGymAdapter()
    for ep in H.episodes_of(npz):
        H.replay_episode(ad, npz, ep, _hip_check(npz, ep, stats))
    assert stats["steps"] == len(npz["reward"])


def _group_key(ep):
    return (ep["scen"], ep["dt"], ep["shaping"], ep["normalize"], ep["discrete"])


@pytest.mark.parametrize("fixture", ["g6_rollouts.npz", "g2_scripted.npz"])
def test_golden_rollouts_batched(fixture):
    """All golden episodes of one configuration run side by side as the envs of ONE AtcVecEnv (each env fed its own
    recorded actions) — the batched kernel must reproduce every per-step record of the reference."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model
    npz = H.golden_npz(fixture)
    eps = H.episodes_of(npz)
    groups = {}
    for ep in eps:
        groups.setdefault(_group_key(ep), []).append(ep)
    total = 0
    for (scen, dt, shaping, normalize, discrete), geps in groups.items():
        B = len(geps)
        sp = model.SimParameters(dt, reward_shaping=shaping, normalize_state=normalize, discrete_action_space=discrete)
        env = AtcVecEnv(B, 1, sim_parameters=sp, scenario=H.make_scenario(scen), auto_reset=False, spawn="lattice",
                        want_raw_obs=True, keep_active=True)   # the reference's aircraft is never handed over
        half_range = torch.as_tensor(0.5 * H.compiled(scen).norm_max.astype(np.float64))
        for b, ep in enumerate(geps):
            env.set_state(b, 0, *ep["init_state"])
            env.timesteps[b] = ep["init_timesteps"]
            env.set_last_action(b, 0, ep["init_last_action"])
        T = max(ep["steps"] for ep in geps)
        steps = np.array([ep["steps"] for ep in geps])
        starts = np.array([ep["start"] for ep in geps])
        for t in range(T):
            live = t < steps
            rows = np.where(live, starts + t, starts)  # finished envs just replay a harmless action
            act = npz["action"][rows].astype(np.float32).reshape(B, 1, 3)
            obs, rew, done, info = env.step(act)
            obs_c, raw_c = obs.cpu().double(), info["original_state"].cpu().double()
            rew_c, done_c, fl_c = rew.cpu().double().numpy(), done.cpu().numpy(), info["flags"].cpu().numpy()[:, 0]
            ts_c, at_c = env.timesteps.cpu().numpy(), env.actions_taken.cpu().numpy()
            lr = rows[live]
            assert np.array_equal(fl_c[live], npz["flags"][lr].astype(np.int32)), t
            assert np.array_equal(done_c[live], npz["done"][lr]), t
            assert np.array_equal(ts_c[live], npz["timesteps"][lr]), t
            assert np.array_equal(at_c[live], npz["actions_taken"][lr]), t
            go = torch.as_tensor(npz["obs"][lr].astype(np.float64))
            gr = torch.as_tensor(npz["raw"][lr].astype(np.float64))
            lt = torch.as_tensor(live)
            tol_o = 1e-5 if normalize else 1e-5 * half_range
            assert bool(((obs_c[lt] - go).abs() <= tol_o).all()), t
            assert bool(((raw_c[lt] - gr).abs() <= 1e-5 * half_range).all()), t
            gw = npz["reward"][lr]
            assert np.all(np.abs(rew_c[live] - gw) <= 1e-5 * np.maximum(1.0, np.abs(gw))), t
            total += int(live.sum())
        env.close()
    assert total == len(npz["reward"])


class _HipLockstep:
    def __init__(self, scen, dt, shaping, normalize, discrete, B):
        from atc_hip.vec_env import AtcVecEnv
        from envs.atc import model
        sp = model.SimParameters(dt, reward_shaping=shaping, normalize_state=normalize, discrete_action_space=discrete)
        self.env = AtcVecEnv(B, 1, sim_parameters=sp, scenario=H.make_scenario(scen), auto_reset=False, spawn="lattice",
                             keep_active=True)

    def place(self, b, init_state, init_timesteps, init_last_action):
        self.env.set_state(b, 0, *init_state)
        self.env.timesteps[b] = init_timesteps
        self.env.set_last_action(b, 0, init_last_action)

    def step(self, actions):
        e = self.env
        torch = e.torch
        obs, rew, done, info = e.step(actions)
        pack = torch.cat([obs.double(), rew.double()[:, None], done.double()[:, None], info["flags"].double(),
                          e.actions_taken.double()[:, None], e.x[:, None], e.y[:, None], e.h.double()[:, None],
                          e.phi.double()[:, None], e.v.double()[:, None]], dim=1).cpu().numpy()   # one device->host hop
        return pack[:, :10], pack[:, 10], pack[:, 11], pack[:, 12], pack[:, 13].astype(np.int64), pack[:, 14:19]

    def close(self):
        self.env.close()


class _HipInterleaved:
    def __init__(self, scen, dt, shaping, normalize, discrete, B, N):
        from atc_hip.vec_env import AtcVecEnv
        from envs.atc import model
        sp = model.SimParameters(dt, reward_shaping=shaping, normalize_state=normalize, discrete_action_space=discrete)
        self.env = AtcVecEnv(B, N, sim_parameters=sp, scenario=H.make_scenario(scen), auto_reset=False, spawn="lattice",
                             keep_active=True, sep_nm=0.0)
        self.B, self.N = B, N

    def place(self, b, k, init_state, init_last_action):
        self.env.set_state(b, k, *init_state)
        self.env.set_last_action(b, k, init_last_action)

    def set_timesteps(self, b, t):
        self.env.timesteps[b] = t

    def _state(self):
        e = self.env
        torch = e.torch
        st = torch.stack([e.x, e.y, e.h.double(), e.phi, e.v], dim=1).cpu().numpy().reshape(self.B, self.N, 5)
        return e.actions_taken.cpu().numpy(), st

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions)
        acts, st = self._state()
        return (obs.cpu().numpy().reshape(self.B, self.N, 10), rew.cpu().numpy(), done.cpu().numpy(),
                info["flags"].cpu().numpy().reshape(self.B, self.N), acts, st)

    def rollout(self, actions):
        out = self.env.rollout(self.env.torch.as_tensor(actions).to(self.env.device), hold=1)
        T = actions.shape[0]
        acts, st = self._state()
        return (out["obs"].cpu().numpy().reshape(T, self.B, self.N, 10), out["reward"].cpu().numpy(), out["done"].cpu().numpy(),
                out["flags"].cpu().numpy().reshape(T, self.B, self.N), acts, st)

    def close(self):
        self.env.close()


@pytest.mark.parametrize("N,chunk,fixture", [(16, 1, "g9_wide.npz"), (64, 1, "g9_wide.npz"), (5, 1, "g9_wide.npz"), (16, 10, "g9_wide.npz"),
                                              (64, 10, "g9_wide.npz"), (32, 5, "g9_wide.npz"), (8, 1, "g11_unbounded.npz"), (8, 10, "g11_unbounded.npz"),
                                              (8, 1, "g12_timesteps.npz"), (8, 10, "g12_timesteps.npz"), (16, 5, "g12_timesteps.npz"),
                                              (5, 1, "g13_timestep_sweep.npz"), (5, 10, "g13_timestep_sweep.npz")])
def test_reference_episodes_as_the_aircraft_of_one_env(N, chunk, fixture):
    """helpers.replay_wide_interleaved through the batched kernels: N reference episodes of g9 are the N aircraft of one env
    (separation minimum 0, the reference's episode rule), single steps and multi-step launches (the 32- / 64-aircraft ones under the
    separation-scan horizon) — the multi-aircraft step checked against the REFERENCE over whole episodes."""
    _torch()
    fx = H.WideFixture(fixture)   # (g11: actions outside the action space, WIDE headings in several aircraft of one env)
    n, envs = H.replay_wide_interleaved(fx, _HipInterleaved, N, obs_tol=1e-5, state_tol=1e-5, rew_tol=1e-5, chunk=chunk)
    least = {"g9_wide.npz": (8, 100000), "g13_timestep_sweep.npz": (30, 10000)}.get(fixture, (10, 20000))
    assert envs >= least[0] and n > least[1], (n, envs)


def test_wide_fixture_batched():
    """G9: 650 963 reference steps (LOWW / random entries / Simple / UnitTest / Dyadic, dt 1-2-5, discrete, shaping and
    normalisation off, >= 50 wins / MVA busts / timeouts, episodes stepped on past done) replayed through the batched
    kernel: every integer output of every step exact, rewards and sampled observations within 1e-5 — everywhere (the
    near-FAF exception of rounds 1-3 is retired, helpers.replay_wide)."""
    fx = H.WideFixture()
    n = H.replay_wide(fx, _HipLockstep, obs_tol=1e-5, state_tol=1e-5, rew_tol=1e-5)
    assert n == len(fx.flags) > 500000


def test_unbounded_heading_fixture_batched():
    """G11: actions outside the action space as the REFERENCE replays them (tests/golden/generate_golden.py:gen_g11) — sustained
    a_phi up to +-3 (headings to 720 / -360 deg), headings wound to +-5 500 deg and back, un-clipped random actions, discrete
    indices beyond 360 — through the batched kernel at g9's bars.  ABI 18's heading format ended at 436 / -76 deg; ABI 19 keeps
    the exact counts behind the saturating 32-bit field (include/atc_step.h)."""
    fx = H.WideFixture("g11_unbounded.npz")
    n = H.replay_wide(fx, _HipLockstep, obs_tol=1e-5, state_tol=1e-5, rew_tol=1e-5)
    assert n == len(fx.flags) > 93000
    phi = fx.state[:, 3]
    assert phi.max() >= 5000 and phi.min() <= -5000


def test_timestep_sweep_fixture_batched():
    """G13: ten more timesteps, 0.01 ... 47 s, ordinary random-held actions, every configuration switch, three sectors (122 805
    reference steps) through the batched kernel at g9's bars."""
    fx = H.WideFixture("g13_timestep_sweep.npz")
    H.WRAP_ROWS[0] = 0
    n = H.replay_wide(fx, _HipLockstep, obs_tol=1e-5, state_tol=1e-5, rew_tol=1e-5)
    assert n == len(fx.flags) > 120000 and H.WRAP_ROWS[0] <= 10


def test_timestep_fixture_batched():
    """G12: SimParameters.timestep in {0.05, 0.1, 0.15, 0.3, 0.7, 1.3, 3.7} s as the REFERENCE steps them — sustained descents into
    the MVAs, altitude ties decided by the reference's own float64 rounding (111 flagged at step n, 89 one step later), landings on
    targets, 3 000-step episodes — through the batched kernel: flags / done / action counters exact on every one of the 376 632
    steps, values within 1e-5.  ABI 20: the altitude is the reference's float64, the timestep a double (include/atc_step.h)."""
    fx = H.WideFixture("g12_timesteps.npz")
    H.WRAP_ROWS[0] = 0
    n = H.replay_wide(fx, _HipLockstep, obs_tol=1e-5, state_tol=1e-5, rew_tol=1e-5)
    assert n == len(fx.flags) > 370000 and H.WRAP_ROWS[0] <= 40


def test_timestep_fixture_single_env():
    """The same through the drop-in AtcGym (one env, packet polling): descents and ties of G12 at dt = 0.1 / 0.15 / 0.3 step by
    step, the altitude read back from the device equal to the reference's float64 BIT FOR BIT on every sampled row."""
    from envs.atc import atc_gym, model
    fx = H.WideFixture("g12_timesteps.npz")
    n_eps = 0
    for dt in (0.1, 0.15, 0.3):
        eps = [ep for ep in fx.episodes if ep["scen"] == "LOWW" and ep["dt"] == dt and not ep["discrete"] and ep["shaping"]
               and ep["steps"] <= 3100]
        eps = eps[:2] + eps[-3:]        # two descents from the entry point, three placed episodes (ties / landings)
        env = atc_gym.AtcGym(sim_parameters=model.SimParameters(dt))
        for ep in eps:
            env.reset()
            ap = env._airplane
            ap.x, ap.y, ap.h, ap.phi, ap.v = ep["init_state"]
            env._vec.timesteps[0] = int(ep["init_timesteps"])
            env.timesteps = int(ep["init_timesteps"])
            env.last_action = ep["init_last_action"]
            for t in range(ep["steps"]):
                row = ep["start"] + t
                obs, rew, done, info = env.step(fx.action[row].astype(np.float32))
                assert bool(done) == bool(fx.done[row]) and env.actions_taken == fx.actions_taken[row], (dt, row, t)
                assert abs(rew - fx.reward[row]) <= 1e-5 * max(1.0, abs(fx.reward[row])), (dt, row, t)
                si = fx.samp_index[row]
                if si >= 0:
                    assert H.obs_close(obs[None], fx.obs[si][None].astype(np.float64), 1e-5, True), (dt, row, t, obs, fx.obs[si])
                    assert env._airplane.h == fx.state[si][2], (dt, row, t)
            n_eps += 1
        env.close()
    assert n_eps == 15


def test_unbounded_heading_single_env():
    """The same through the drop-in AtcGym (one env, packet polling): the four +-3 / +-2 episodes of G11 step by step."""
    from envs.atc import atc_gym
    fx = H.WideFixture("g11_unbounded.npz")
    eps = [ep for ep in fx.episodes if ep["scen"] == "LOWW" and not ep["discrete"] and ep["init_timesteps"] == 0
           and abs(float(fx.action[ep["start"], 2])) >= 2.0 and ep["init_state"][:2] == [10.0, 51.0]][:6]
    assert len(eps) >= 4
    env = atc_gym.AtcGym()
    for ep in eps:
        env.reset()
        for t in range(ep["steps"]):
            row = ep["start"] + t
            obs, rew, done, info = env.step(fx.action[row].astype(np.float32))
            assert bool(done) == bool(fx.done[row]) and env.actions_taken == fx.actions_taken[row], (row, t)
            assert abs(rew - fx.reward[row]) <= 1e-5 * max(1.0, abs(fx.reward[row])), (row, t)
            si = fx.samp_index[row]
            if si >= 0:
                assert np.all(np.abs(obs - fx.obs[si]) <= 1e-5), (row, t, obs, fx.obs[si])
                assert abs(env._airplane.phi - fx.state[si][3]) <= 1e-5
    env.close()


def test_heading_bound_flag_on_device():
    """ATC_F_PHI_LIMIT: the one bound left (heading targets beyond +-2^52 counts, |a_phi| > 2.98e6) is flagged on the step that
    clamps — same flags, counters and state as the fp32 oracle (tests/test_oracle_golden.py pins what the clamp changes
    against the reference)."""
    from atc_hip.vec_env import AtcVecEnv
    from oracle import oracle as O
    comp = H.compiled("LOWW", 0.5)
    env = AtcVecEnv(64, 1, scenario=H.make_scenario("LOWW"), auto_reset=False, keep_active=True, grid_cell=0.5)
    orc = O.OracleEnv(comp, 64, 1, O.make_params(keep_active=True), np.float32)
    a = np.zeros((64, 1, 3), np.float32)
    a[:, 0, 2] = np.concatenate([np.linspace(-1, 1, 16), [1.4222, 1.4223, -1.4223, 3.0, -3.0, 100.0, 2.9e6, 2.99e6, 4e6, -4e6,
                                                         1e30, -1e30, np.inf, -np.inf, np.nan, 5e6], np.linspace(-9e6, 9e6, 32)])
    for t in range(12):
        if t == 6:
            a[:, 0, 2] *= -0.5
        obs, rew, done, info = env.step(a)
        orc.step(a)
        fl = info["flags"].cpu().numpy().astype(np.uint16)
        assert np.array_equal(fl, orc.flags), t
        assert np.array_equal(env.actions_taken.cpu().numpy(), orc.actions_taken), t
        assert np.array_equal(env.phi_counts.cpu().numpy(), orc.phi_counts.astype(np.float64)), t
        on = obs.cpu().numpy().reshape(64, 1, 10)
        assert np.all(np.abs(on - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs))), t
    assert int((fl & 512 != 0).sum()) >= 10 and int((fl & 512 == 0).sum()) >= 20
    env.close()


def test_atcgym_keeps_flying_after_a_win():
    """The reference has no inactive state (atc_gym.py:128-192): stepping on after a win without reset keeps simulating
    the aircraft, which can win again (learning/atc-gym-compute-performance.py never resets).  Checked against the g9
    episodes that were stepped on past a win."""
    from envs.atc import atc_gym
    fx = H.WideFixture()
    eps = [ep for ep in fx.episodes if ep["scen"] == "LOWW" and ep["normalize"] and not ep["discrete"] and ep["dt"] == 1.0
           and ep["shaping"] and fx.done[ep["start"]:ep["start"] + ep["steps"] - 1].any()
           and (fx.flags[ep["start"]:ep["start"] + ep["steps"]] & H.F_WON).any()][:6]
    assert len(eps) >= 3
    env = atc_gym.AtcGym()
    for ep in eps:
        env.reset()
        ap = env._airplane
        ap.x, ap.y, ap.h, ap.phi, ap.v = ep["init_state"]
        env._vec.timesteps[0] = ep["init_timesteps"]
        env.timesteps = ep["init_timesteps"]
        # (the fixture's episodes start from their own recorded last_action, not from what the previous episode of this loop
        # left behind: without this line the float64 oracle itself disagrees with the fixture on the first step of episode 3,
        # where |target - leftover| = 4.9999999 but |target - recorded| = 13)
        env.last_action = ep["init_last_action"]
        wins = 0
        for t in range(ep["steps"]):
            row = ep["start"] + t
            obs, rew, done, info = env.step(fx.action[row])
            assert done == bool(fx.done[row]) and env.actions_taken == int(fx.actions_taken[row]), (t, done)
            gw = float(fx.reward[row])
            assert abs(rew - gw) <= (1e-5 + 1e-7) * max(1.0, abs(gw)), (t, rew, gw)   # 1e-5 + the fixture's float32 storage (6e-8)
            wins += int(done and rew > 9000)
        assert wins >= 2          # it won, kept flying inside the corridor and won again
    env.close()


# ------------------------------------------------------------------------------------------------ batched vs fp32 oracle
def _run_vs_oracle(scen_obj, comp, B, N, steps, seed, dt=1.0, discrete=False, spawn="lattice", hold=20, grid_cell=0.5,
                   use_rollout=0, timestep_limit=6000, full=True, shaping=True, normalize=True, sep_nm=3.0,
                   keep_active=False, held_hint=False, rollout_hold=1, wild=0.0):
    """full=False drives the fast kernel variant (obs / reward / done / flags only), full=True the one with every optional
    output; everything the variant produces is compared with the fp32 oracle.  held_hint: single steps that repeat the
    previous step's action array are launched with ATC_M_ACTIONS_HELD (must change nothing).
    rollout_hold > 1 (with use_rollout): the multi-step launches go through atc_rollout_hold — one action block per
    `rollout_hold` steps (frame skip, learning/atc-gym-demo.py:18-19), whose repeated steps skip the last-action bookkeeping
    inside the kernel; the oracle is stepped once per step with the block's actions.
    wild > 0: that fraction of the drawn action COMPONENTS lies outside the action space — U(-4, 4) (a tenth of those a further
    factor 50 out): the reference enforces no Box (atc_gym.py:128-141); speed / altitude targets beyond their limits are refused,
    heading targets are never validated and headings leave the state format's 32-bit range (include/atc_step.h, ABI 19)."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model
    from oracle import oracle as O
    sp = model.SimParameters(dt, discrete_action_space=discrete, reward_shaping=shaping, normalize_state=normalize)
    env = AtcVecEnv(B, N, sim_parameters=sp, scenario=scen_obj, auto_reset=True, spawn=spawn, seed=seed,
                    grid_cell=grid_cell, want_raw_obs=full, want_ac_reward=full, want_min_sep=full, want_term_obs=full,
                    timestep_limit=timestep_limit, sep_nm=sep_nm, keep_active=keep_active)
    p = O.make_params(dt=dt, discrete=discrete, auto_reset=True, random_entry=(spawn == "random"), seed=seed,
                      timestep_limit=timestep_limit, shaping=shaping, normalize=normalize, sep_nm=sep_nm,
                      keep_active=keep_active)
    orc = O.OracleEnv(comp, B, N, p, np.float32)
    o0 = env.obs.cpu().numpy().reshape(B, N, 10)
    assert np.all(np.abs(o0 - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs)))
    rng = np.random.default_rng(seed)
    half_range = 0.5 * comp.norm_max.astype(np.float64)
    n_done = 0
    seen = 0
    act = None
    if use_rollout:
        assert steps % use_rollout == 0
    if rollout_hold > 1:
        assert use_rollout and use_rollout % rollout_hold == 0 and hold % rollout_hold == 0
    t = 0
    while t < steps:
        chunk = use_rollout or 1
        acts = []
        repeated = act is not None and t % hold != 0
        for c in range(chunk):
            if (t + c) % hold == 0 or act is None:
                if discrete:
                    act = np.floor(rng.uniform(0, 1, (B, N, 3)) * np.array([20, 380, 360])).astype(np.float32)
                else:
                    act = rng.uniform(-1.05, 1.05, (B, N, 3)).astype(np.float32)
                if wild > 0.0:   # (drawn after the regular actions: a case without wild draws keeps its stream)
                    out_of_space = rng.uniform(-4.0, 4.0, (B, N, 3)) * np.where(rng.uniform(size=(B, N, 3)) < 0.1, 50.0, 1.0)
                    if discrete:
                        out_of_space = np.floor(out_of_space * np.array([20, 380, 360]))
                    act = np.where(rng.uniform(size=(B, N, 3)) < wild, out_of_space, act).astype(np.float32)
            acts.append(act)
        if use_rollout and rollout_hold > 1:
            assert all(acts[c] is acts[c - c % rollout_hold] for c in range(chunk))   # blocks are constant by construction
            out = env.rollout(torch.as_tensor(np.stack(acts[::rollout_hold])), hold=rollout_hold)
            res = [(out["obs"][c], out["reward"][c], out["done"][c], out["flags"][c]) for c in range(chunk)]
        elif use_rollout:
            out = env.rollout(torch.as_tensor(np.stack(acts)))
            res = [(out["obs"][c], out["reward"][c], out["done"][c], out["flags"][c]) for c in range(chunk)]
        else:
            o, r, d, info = env.step(acts[0], held=held_hint and repeated)
            res = [(o, r, d, info["flags"])]
        for c in range(chunk):
            orc.step(acts[c])
            o, r, d, fl = res[c]
            fl = fl.cpu().numpy().astype(np.uint32)
            assert np.array_equal(fl, orc.flags), ("flags", t + c, np.argwhere(fl != orc.flags)[:5])
            assert np.array_equal(d.cpu().numpy(), orc.done), ("done", t + c)
            on = o.cpu().numpy().reshape(B, N, 10)
            # envs that were auto-reset return RAW obs (large values): compare relative to magnitude; without
            # normalisation every obs is raw and 1e-5 in obs units is 1e-5 of the component's normalisation half-range
            scale = np.maximum(1.0, np.abs(orc.obs))
            if not normalize:
                scale = np.maximum(scale, half_range.astype(np.float32))
            assert np.all(np.abs(on - orc.obs) <= 1e-5 * scale), ("obs", t + c)
            rr = r.cpu().numpy()
            # env reward = sum over the env's aircraft of per-aircraft rewards that each meet 1e-5 (checked below when the
            # variant outputs them); the fp32 sum of N terms adds at most N/2 ulps of the running sum
            rtol = 1e-5 * np.maximum(1.0, np.abs(orc.reward)) + 6e-8 * N * np.abs(orc.ac_reward).sum(1)
            assert np.all(np.abs(rr - orc.reward) <= rtol), ("rew", t + c, np.abs(rr - orc.reward).max())
            n_done += int(orc.done.sum())
            seen |= int(np.bitwise_or.reduce(orc.flags.ravel()))
        if not use_rollout and full:
            # optional outputs and persistent state
            # heading arithmetic is exact in both implementations -> relative_angle (raw[9]) must be bit-identical
            # (checks the division-free Python-modulo of csrc/atc_device.h against the fmodf-based oracle)
            assert np.array_equal(info["original_state"].cpu().numpy().reshape(B, N, 10)[..., 9], orc.raw_obs[..., 9]), t
            assert np.array_equal(info["original_state"].cpu().numpy().reshape(B, N, 10)[..., 3], orc.raw_obs[..., 3]), t
            # raw (un-normalised) values: 1e-5 of each component's normalisation half-range (= 1e-5 in obs units)
            assert np.all(np.abs(info["original_state"].cpu().numpy().reshape(B, N, 10) - orc.raw_obs)
                          <= 1e-5 * half_range), t
            assert np.all(np.abs(info["aircraft_reward"].cpu().numpy() - orc.ac_reward)
                          <= 1e-5 * np.maximum(1.0, np.abs(orc.ac_reward))), t
            # positions are bit-identical and d^2 is the same fma on both sides: the minimum separation is too
            assert np.array_equal(info["min_separation"].cpu().numpy(), orc.min_sep), t
            dn = orc.done.astype(bool)
            if dn.any():
                tob = info["terminal_observation"].cpu().numpy().reshape(B, N, 10)
                tscale = np.maximum(1.0, np.abs(orc.term_obs[dn]))
                if not normalize:
                    tscale = np.maximum(tscale, half_range.astype(np.float32))
                assert np.all(np.abs(tob[dn] - orc.term_obs[dn]) <= 1e-5 * tscale), t
        t += chunk
    # persistent state after the run: integer state exact, float state within tolerance
    assert np.array_equal(env.timesteps.cpu().numpy(), orc.timesteps)
    assert np.array_equal(env.actions_taken.cpu().numpy(), orc.actions_taken)
    assert np.array_equal(env.episodes.cpu().numpy(), orc.episodes)
    assert np.array_equal(env.win_bits.cpu().numpy().astype(np.uint32), orc.win_bits)
    assert np.array_equal(env.active_mask.cpu().numpy().astype(np.uint64), orc.active_mask)
    assert np.array_equal(env.ep_length.cpu().numpy(), orc.ep_length)
    # the fp32 spec (include/atc_step.h: fixed-point position grid, shared heading kinematics, exact rate-limit
    # arithmetic) makes the whole aircraft state BIT-IDENTICAL to the fp32 oracle's
    assert np.array_equal(env.ac[:, 0].cpu().numpy(), orc.px) and np.array_equal(env.ac[:, 1].cpu().numpy(), orc.py)
    assert np.array_equal(env.h.cpu().numpy(), orc.h) and np.array_equal(env.phi_fix.cpu().numpy(), orc.phi_fix)
    # ... the exact counts of WIDE headings / last heading targets (beyond the 32-bit fields, ABI 19) included
    assert np.array_equal(env.phi_counts.cpu().numpy(), orc.phi_counts.astype(np.float64))
    la_wide = np.isin(orc.last_act[:, 1], (-2 ** 31, 2 ** 31 - 1))
    assert np.array_equal(env.phi_wide[:, 1].cpu().numpy()[la_wide], orc.phi_wide[la_wide, 1])
    assert np.array_equal(env.v_fix.cpu().numpy(), orc.v_fix)
    assert np.array_equal(env.last_act.cpu().numpy(), orc.last_act)
    assert np.array_equal(env.ep_actions.cpu().numpy(), orc.ep_actions)
    assert np.allclose(env.ep_return.cpu().numpy(), orc.ep_return, rtol=1e-5, atol=1e-3)
    env.close()
    return n_done, seen


def test_batched_n1_vs_oracle():
    from envs.atc import scenarios
    n_done, seen = _run_vs_oracle(scenarios.LOWW(), H.compiled("LOWW", 0.5), B=2048, N=1, steps=600, seed=5)
    assert n_done > 100 and (seen & H.F_OUTSIDE) and (seen & (H.F_INVALID_V | H.F_INVALID_H))


def test_batched_n1_random_entries_discrete_vs_oracle():
    from envs.atc import scenarios
    n_done, seen = _run_vs_oracle(scenarios.LOWW(random_entrypoints=True), H.compiled("LOWW_random", 0.5), B=1024, N=1,
                                  steps=400, seed=9, discrete=True, spawn="random", dt=2.0)
    assert n_done > 20


def test_batched_n16_vs_oracle():
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    n_done, seen = _run_vs_oracle(scn, scenarios.compile_scenario(scn, grid_cell=0.5), B=512, N=16, steps=400, seed=21, held_hint=True)
    assert n_done > 50 and (seen & H.F_CONFLICT)


def test_batched_n64_noise_vs_oracle():
    from envs.atc import scenarios
    scn = scenarios.LOWWDense()
    n_done, seen = _run_vs_oracle(scn, scenarios.compile_scenario(scn, grid_cell=0.5), B=128, N=64, steps=240, seed=33)
    assert n_done > 20 and (seen & H.F_CONFLICT)


def test_batched_odd_n_and_short_timeout_vs_oracle():
    """N that is not a power of two (idle lanes in the group), tiny timestep limit so that timeouts and win-less
    resets happen often, no lookup grid."""
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    n_done, seen = _run_vs_oracle(scn, scenarios.compile_scenario(scn), B=300, N=5, steps=200, seed=4, grid_cell=None,
                                  timestep_limit=37)
    assert seen & H.F_TIMEOUT


def test_rollout_equals_single_steps_vs_oracle():
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    _run_vs_oracle(scn, scenarios.compile_scenario(scn, grid_cell=0.5), B=256, N=16, steps=120, seed=8, use_rollout=24)
    _run_vs_oracle(scenarios.LOWW(), H.compiled("LOWW", 0.5), B=1000, N=1, steps=250, seed=2, use_rollout=50)


@pytest.mark.parametrize("N,B,T,rh,hold,full", [(16, 256, 20, 20, 20, False), (16, 256, 40, 20, 20, True), (16, 300, 20, 4, 20, False),
                                               (1, 1000, 40, 20, 40, False), (64, 64, 20, 20, 20, False), (5, 200, 24, 4, 8, True)])
def test_rollout_hold_matches_oracle(N, B, T, rh, hold, full):
    """atc_rollout_hold with hold > 1 — the entry behind bench.py's fused-rollout record and the protocol of
    learning/atc-gym-demo.py:18-19 — directly against the fp32 oracle (round-2 review: it was only compared with single
    steps of the same library): every step's flags / done exact, obs / rewards within 1e-5, and after the run the whole
    state incl. actions_taken and the last-action records bit-identical."""
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=True)
    n_done, seen = _run_vs_oracle(scn, scenarios.compile_scenario(scn, grid_cell=0.5), B=B, N=N, steps=12 * T, seed=100 + N + T,
                                  use_rollout=T, rollout_hold=rh, hold=hold, full=full, spawn="lattice")
    assert n_done > 0


@pytest.mark.parametrize("N,B,kw", [
    (1, 2048, dict()), (1, 2048, dict(use_rollout=20, rollout_hold=20, full=False)), (1, 1000, dict(discrete=True, dt=5.0, hold=5)),
    (16, 256, dict(held_hint=True)), (16, 256, dict(use_rollout=20, rollout_hold=20, full=False)),
    (16, 300, dict(use_rollout=20, rollout_hold=4, full=True)), (16, 256, dict(full=False, held_hint=True, timestep_limit=100000)),
    (5, 200, dict(use_rollout=8)), (64, 64, dict(full=False)), (64, 64, dict(use_rollout=20, rollout_hold=20, full=False)),
    (33, 64, dict(wild=1.0, dt=5.0))])
def test_actions_outside_the_action_space_vs_oracle(N, B, kw):
    """A third of all action components outside [-1, 1] (U(-4, 4), some 50 x further): refused speed / altitude targets, heading
    targets and headings beyond the 32-bit field (WIDE, include/atc_step.h ABI 19) in every kernel variant — single steps with
    and without the held hint, fused launches with held blocks, all-valid and general instantiations, DPP / xor / LDS scans.  Every
    step's flags / done exact, obs / rewards within 1e-5, the final state — exact 64-bit heading counts included — bit-identical."""
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=True)
    kw = dict(dict(wild=0.33, steps=400, seed=700 + N + B), **kw)
    if kw.get("use_rollout"):
        kw["steps"] = (kw["steps"] // kw["use_rollout"]) * kw["use_rollout"]
        kw.setdefault("hold", max(kw.get("rollout_hold", 1), 20 if kw.get("rollout_hold", 1) in (1, 20) else 8))
    n_done, seen = _run_vs_oracle(scn, scenarios.compile_scenario(scn, grid_cell=0.5), B=B, N=N, **kw)
    assert n_done > 0 and (seen & H.F_INVALID_V) and (seen & H.F_INVALID_H)


def test_flying_on_beyond_the_position_grid():
    """Documented limit of the fixed-point position grid (include/atc_step.h): an aircraft that is flown on without reset
    beyond the grid range is pinned at the range limit — it stays OUTSIDE the airspace like the reference's, its x / y
    observation stops growing, and HIP and the fp32 oracle still agree bit for bit."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWW()
    comp = H.compiled("LOWW", 0.5)
    env = AtcVecEnv(4, 1, scenario=scn, auto_reset=False, keep_active=True)
    orc = O.OracleEnv(comp, 4, 1, O.make_params(keep_active=True), np.float32)
    a = np.zeros((4, 1, 3), np.float32)
    a[:, 0, 0] = 1.0                                   # 300 kt
    a[:, 0, 2] = [-1.0, -0.5, 0.0, 0.5]                 # headings 0 / 90 / 180 / 270: one aircraft per compass direction
    lim_lo = np.array(comp.pos_origin) - 2.0 ** (31 - comp.pos_k)
    lim_hi = np.array(comp.pos_origin) + 2.0 ** (31 - comp.pos_k)
    for t in range(1400):                               # 1400 s x 300 kt = 117 nm: well past the +-64 nm grid
        obs, rew, done, info = env.step(a)
        orc.step(a)
        if t % 50 == 0 or t > 1350:
            assert np.array_equal(env.ac[:, 0].cpu().numpy(), orc.px) and np.array_equal(env.ac[:, 1].cpu().numpy(), orc.py)
            assert np.array_equal(info["flags"].cpu().numpy().astype(np.uint16), orc.flags)
    assert bool((info["flags"][:, 0] & H.F_OUTSIDE).all()) and bool(done.all())
    x, y = env.x.cpu().numpy(), env.y.cpu().numpy()
    assert y[0] == lim_hi[1] - 2.0 ** -comp.pos_k and x[1] == lim_hi[0] - 2.0 ** -comp.pos_k      # INT32_MAX counts
    assert y[2] == lim_lo[1] and x[3] == lim_lo[0]                                                  # INT32_MIN counts
    env.close()


# ------------------------------------------------------------------------------------------------ extension known answers
def test_separation_known_answers():
    """Two aircraft at 2.99 / 3.00 / 3.01 nm x 999 / 1000 / 1001 ft: conflict iff d < 3 nm and dh < 1000 ft."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    cases = [(d, dh) for d in (2.99, 3.0, 3.01) for dh in (999.0, 1000.0, 1001.0)]
    env = AtcVecEnv(len(cases), 2, scenario=scenarios.LOWW(random_entrypoints=True), auto_reset=False, want_min_sep=True)
    for b, (d, dh) in enumerate(cases):
        # both fly north at the same speed so the horizontal distance is unchanged by the step
        env.set_state(b, 0, 30.0, 60.0, 15000.0, 0.0, 250.0)
        env.set_state(b, 1, 30.0 + d, 60.0, 15000.0 + dh, 0.0, 250.0)
    hold = np.zeros((len(cases), 2, 3), np.float32)
    hold[:, :, 0] = 0.5      # v target 250
    hold[:, 0, 1] = 2 * 15000.0 / 38000.0 - 1
    for b, (d, dh) in enumerate(cases):
        hold[b, 1, 1] = 2 * (15000.0 + dh) / 38000.0 - 1
    hold[:, :, 2] = -1.0     # heading target 0
    obs, rew, done, info = env.step(hold)
    fl = info["flags"].cpu().numpy()
    for b, (d, dh) in enumerate(cases):
        expect = d < 3.0 and dh < 1000.0
        assert bool(fl[b, 0] & H.F_CONFLICT) == expect and bool(fl[b, 1] & H.F_CONFLICT) == expect, (d, dh)
        assert bool(done[b]) == expect
        assert abs(float(info["min_separation"][b]) - d) < 1e-4
        if expect:
            assert float(rew[b]) < -390.0  # 2 x (-200 + shaping)
    env.close()


def test_extension_known_answers():
    """The hand-computed one-step outcomes of helpers.extension_known_answers (override order, per-aircraft rewards and their env
    sum, termination, hand-over masks; nothing in them comes from the oracle) through the batched kernel — all cases as the
    envs of batches that share their parameters."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model, scenarios
    cases = H.extension_known_answers()
    groups = {}
    for c in cases:
        groups.setdefault(tuple(sorted(c[1].items())), []).append(c)
    for params, cs in groups.items():
        env = AtcVecEnv(len(cs), 3, sim_parameters=model.SimParameters(1, reward_shaping=False),
                        scenario=scenarios.LOWW(random_entrypoints=True), auto_reset=False, want_ac_reward=True, **dict(params))
        a = np.zeros((len(cs), 3, 3), np.float32)
        for b, (name, _, t0, aircraft, want) in enumerate(cs):
            for k, (st, act) in enumerate(aircraft):
                env.set_state(b, k, *st)
                a[b, k] = act
            env.timesteps[b] = t0
        obs, rew, done, info = env.step(a)
        fl = info["flags"].cpu().numpy()
        acr = info["aircraft_reward"].cpu().numpy()
        mask = env.active_mask.cpu().numpy()
        for b, (name, _, t0, aircraft, want) in enumerate(cs):
            assert [int(f) for f in fl[b]] == want["flags"], name
            assert np.allclose(acr[b], want["ac_reward"], rtol=0, atol=2e-3), (name, acr[b])
            assert abs(float(rew[b]) - sum(want["ac_reward"])) <= 4e-3, name
            assert bool(done[b]) == want["done"] and int(mask[b]) == want["mask_after"], name
        names = [c[0] for c in cs]
        if "win hands over, env continues" in names:
            b = names.index("win hands over, env continues")
            obs, rew, done, info = env.step(a)
            assert int(info["flags"][b, 0]) == H.F_INACTIVE and bool((obs[b, :10] == 0).all())
            assert float(info["aircraft_reward"][b, 0]) == 0.0 and abs(float(rew[b]) + 0.10) < 1e-6 and not bool(done[b])
        env.close()


def test_noise_abatement_area_penalty():
    """Aircraft inside a noise polygon below its ceiling pays the per-step penalty and is flagged; above the ceiling or
    outside the polygon it is not (extension; checked against the oracle's definition)."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWWDense()
    comp = scenarios.compile_scenario(scn, grid_cell=0.5)
    env = AtcVecEnv(3, 1, scenario=scn, auto_reset=False, spawn="lattice", want_ac_reward=True)
    orc = O.OracleEnv(comp, 3, 1, O.make_params(), np.float32)
    states = [(43.0, 38.0, 5000.0, 90.0, 200.0), (43.0, 38.0, 9000.0, 90.0, 200.0), (20.0, 60.0, 5000.0, 90.0, 200.0)]
    for b, st in enumerate(states):
        env.set_state(b, 0, *st)
        orc.set_state(b, 0, *st)
    a = np.zeros((3, 1, 3), np.float32)
    a[:, 0, 0] = 0.0
    a[:, 0, 1] = [2 * 5000 / 38000 - 1, 2 * 9000 / 38000 - 1, 2 * 5000 / 38000 - 1]
    a[:, 0, 2] = -0.5
    obs, rew, done, info = env.step(a)
    orc.step(a)
    fl = info["flags"].cpu().numpy()[:, 0]
    assert [bool(f & H.F_NOISE) for f in fl] == [True, False, False]
    assert np.array_equal(fl.astype(np.uint32), orc.flags[:, 0])
    assert np.allclose(rew.cpu().numpy(), orc.reward, atol=1e-5)
    env.close()


def test_win_hands_aircraft_over_and_env_continues():
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    env = AtcVecEnv(1, 2, scenario=scenarios.LOWW(random_entrypoints=True), auto_reset=False)
    env.set_state(0, 0, 48.9, 31.9, 3300.0, 345.0, 200.0)       # on the intercept (golden win case)
    env.set_state(0, 1, 20.0, 60.0, 15000.0, 90.0, 250.0)       # far away
    a = np.zeros((1, 2, 3), np.float32)
    a[0, 0] = [0.0, 2 * 2700.0 / 38000.0 - 1, 2 * 345.0 / 360.0 - 1]
    won_at = None
    for t in range(60):
        obs, rew, done, info = env.step(a)
        fl = info["flags"].cpu().numpy()[0]
        if fl[0] & H.F_WON:
            won_at = t
            assert not bool(done[0])            # the other aircraft is still under control
            assert float(rew[0]) > 10000
            break
    assert won_at is not None
    obs, rew, done, info = env.step(a)
    fl = info["flags"].cpu().numpy()[0]
    assert fl[0] == H.F_INACTIVE and int(env.active_mask[0]) == 2
    assert np.all(obs.cpu().numpy()[0, :10] == 0)
    env.close()


# ------------------------------------------------------------------------------------------------ metrics (G7)
def test_metrics_sequence_g7():
    from envs.atc import atc_gym
    g = H.golden_json("g7_metrics.json")
    env = atc_gym.AtcGym()
    for ep in g["episodes"]:
        env.reset()
        assert abs(env.winning_ratio - ep["after_reset"]["winning_ratio"]) < 1e-9
        assert env._win_buffer == ep["after_reset"]["win_buffer"]
        assert env._episodes_run == ep["after_reset"]["episodes_run"]
        ap = env._airplane
        ap.x, ap.y, ap.h, ap.phi, ap.v = ep["init_state"]
        env._vec.timesteps[0] = ep["init_timesteps"]
        env.timesteps = ep["init_timesteps"]
        for k, a in enumerate(ep["actions"]):
            _, r, done, _ = env.step(np.asarray(a))
            assert abs(env.actions_per_timestep - ep["actions_per_timestep"][k]) < 1e-12
        assert done
        f = ep["final"]
        assert env.timesteps == f["timesteps"] and env.actions_taken == f["actions_taken"]
        assert env._win_buffer == f["win_buffer"]
        assert abs(env.total_reward - f["total_reward"]) <= 1e-5 * max(1.0, abs(f["total_reward"]))
        assert abs(env.last_reward - f["last_reward"]) <= 1e-5 * max(1.0, abs(f["last_reward"]))
    env.reset()
    assert abs(env.winning_ratio - g["after_last_reset"]["winning_ratio"]) < 1e-9
    assert env._win_buffer == g["after_last_reset"]["win_buffer"]
    env.close()


def test_seeded_random_entry_draws_match_reference():
    """random.seed(7) + LOWW(random_entrypoints=True): the reference's first aircraft is (53, 60, 16000 ft, 260 deg, 250 kt),
    id 25875; AtcGym.reset consumes Python's RNG in the same order."""
    import random
    from envs.atc import atc_gym, scenarios
    random.seed(7)
    env = atc_gym.AtcGym(scenario=scenarios.LOWW(random_entrypoints=True))
    s = env.state
    assert list(s[:5]) == [53.0, 60.0, 16000.0, 260.0, 250.0] and env._airplane.id == 25875
    env.close()


# ------------------------------------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("B,N", [(65536, 1), (8192, 16), (65536, 16), (4096, 64)])
def test_full_size_properties(B, N):
    """At BASELINE.json's sizes the oracle is too slow for a full comparison; check size-independent properties:
    (1) determinism, (2) a prefix of the big batch equals the same envs run as a small batch (envs are independent),
    (3) done <=> a terminal flag or every aircraft handed over, (4) the first 256 envs match the oracle."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWWDense() if N == 64 else scenarios.LOWW(random_entrypoints=N > 1)
    steps, small = 60, 256
    g = torch.Generator(device="cpu").manual_seed(B + N)
    acts = [(torch.rand((B, N, 3), generator=g) * 2 - 1).cuda() for _ in range(3)]

    def run(nb):
        env = AtcVecEnv(nb, N, scenario=scn, auto_reset=True, seed=1)
        outs = []
        for t in range(steps):
            a = acts[(t // 20) % 3][:nb]
            o, r, d, info = env.step(a)
            outs.append((o.clone(), r.clone(), d.clone(), info["flags"].clone()))
        env.close()
        return outs

    big, big2, sm = run(B), run(B), run(small)
    comp = scenarios.compile_scenario(scn, grid_cell=0.5)
    orc = O.OracleEnv(comp, small, N, O.make_params(auto_reset=True, seed=1), np.float32)
    for t in range(steps):
        o, r, d, fl = big[t]
        assert torch.equal(o, big2[t][0]) and torch.equal(r, big2[t][1]) and torch.equal(fl, big2[t][3])
        assert torch.equal(o[:small], sm[t][0]) and torch.equal(fl[:small], sm[t][3]) and torch.equal(d[:small], sm[t][2])
        term = ((fl & TERMINAL) != 0).any(dim=1)
        assert bool((term <= (d != 0)).all())
        assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all())
        orc.step(acts[(t // 20) % 3][:small].cpu().numpy())
        assert np.array_equal(fl[:small].cpu().numpy().astype(np.uint32), orc.flags)
        assert np.array_equal(d[:small].cpu().numpy(), orc.done)
        on = o[:small].cpu().numpy().reshape(small, N, 10)
        assert np.all(np.abs(on - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs)))


@pytest.mark.parametrize("B,N", [(65536, 1), (8192, 16), (65536, 16), (4096, 64)])
def test_full_size_rollout_hold(B, N):
    """The fused entry (atc_rollout_hold, T = 20, hold = 20: the protocol of learning/atc-gym-demo.py:18-19 and the
    configuration behind every `fused_rollout` record of bench.py) at BASELINE.json's sizes, with the grid the library picks
    for the batch: (1) deterministic, (2) the first 256 envs of the big batch equal the same envs run as a 256-env batch (a
    different launch geometry: envs are independent), (3) those 256 envs match the fp32 oracle stepped once per step —
    flags / done exact, obs / reward within 1e-5 —, (4) the state after the launches is bit-identical to the oracle's."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWWDense() if N == 64 else scenarios.LOWW(random_entrypoints=N > 1)
    T, launches, small = 20, 3, 256
    g = torch.Generator(device="cpu").manual_seed(7 * B + N)
    blocks = [(torch.rand((1, B, N, 3), generator=g) * 2 - 1).cuda() for _ in range(launches)]

    def run(nb):
        env = AtcVecEnv(nb, N, scenario=scn, auto_reset=True, seed=3)
        outs = []
        for j in range(launches):
            o = env.rollout(blocks[j][:, :nb].contiguous(), hold=T)
            outs.append({k: v.clone() for k, v in o.items()})
        state = (env.ac.clone(), env.alt.clone(), env.last_act.clone(), env.env.clone())
        env.close()
        return outs, state

    (big, st_big), (big2, st_big2), (sm, st_sm) = run(B), run(B), run(small)
    for a, b in zip(st_big, st_big2):
        assert torch.equal(a, b)
    n_ac = small * N
    assert torch.equal(st_big[0][:n_ac], st_sm[0]) and torch.equal(st_big[1][:n_ac], st_sm[1])
    assert torch.equal(st_big[2][:n_ac], st_sm[2]) and torch.equal(st_big[3][:small], st_sm[3])
    orc = O.OracleEnv(scenarios.compile_scenario(scn, grid_cell=0.5), small, N, O.make_params(auto_reset=True, seed=3), np.float32)
    n_done = 0
    for j in range(launches):
        for k in ("obs", "reward", "done", "flags"):
            assert torch.equal(big[j][k], big2[j][k]), k
            assert torch.equal(big[j][k][:, :small], sm[j][k]), k
        assert bool(torch.isfinite(big[j]["obs"]).all()) and bool(torch.isfinite(big[j]["reward"]).all())
        a = blocks[j][0, :small].cpu().numpy()
        for t in range(T):
            orc.step(a)
            assert np.array_equal(sm[j]["flags"][t].cpu().numpy().astype(np.uint16), orc.flags), (j, t)
            assert np.array_equal(sm[j]["done"][t].cpu().numpy(), orc.done), (j, t)
            on = sm[j]["obs"][t].cpu().numpy().reshape(small, N, 10)
            assert np.all(np.abs(on - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs))), (j, t)
            rw = sm[j]["reward"][t].cpu().numpy()
            # (the fp32 sum of N per-aircraft terms adds at most N/2 ulps of the running sum: the bound of _run_vs_oracle)
            rtol = 1e-5 * np.maximum(1.0, np.abs(orc.reward)) + 6e-8 * N * np.abs(orc.ac_reward).sum(1)
            assert np.all(np.abs(rw - orc.reward) <= rtol), (j, t)
            n_done += int(orc.done.sum())
    assert np.array_equal(st_sm[0][:, 0].cpu().numpy(), orc.px) and np.array_equal(st_sm[0][:, 1].cpu().numpy(), orc.py)
    assert np.array_equal(st_sm[3][:, 1].cpu().numpy(), orc.actions_taken)
    assert np.array_equal(st_sm[2].cpu().numpy(), orc.last_act)
