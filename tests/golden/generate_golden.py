#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container; the outputs are committed).

Imports the read-only reference at /root/reference through the stand-in modules in
tests/oracle_shims/ (numba/gym/shapely/pyglet are not installed here; see that README) and records
inputs + expected outputs of the AtcGym.step() hot path as small fixtures:

  g1_constants.json   derived scenario constants (bbox, FAF/IAF/corners, normal, faf_mva, norm vectors, rings)
  g2_scripted.npz     scripted fixed-action trajectories + dt/discrete/shaping/normalise variants + invalid action
  g3_mva.npz          Airspace.get_mva_height over lattices, polygon vertices and edge midpoints (tie-breaks)
  g4_corridor.npz     Runway.inside_corridor truth tables
  g5_shaping.npz      shaping-reward functions, sigmoid_distance_func, relative_angle on grids
  g6_rollouts.npz     seeded random-action rollouts (continuous + discrete, random entry points, injected
                      win / timeout cases)
  g7_metrics.json     reset / metrics sequence over consecutive episodes
  g8_tiebreak.npz     Airspace.get_mva_height on a sector with integer / dyadic vertices (exactly representable in fp32):
                      vertices, edge midpoints, shared borders, overlapping polygons (list-order priority), points a few
                      2^-10 nm either side of every edge — the tie-break rules of model.py:282-289,318-337
  g10_render_geometry.json  the geometry the reference's render() builds (window size, MVA outlines, runway, FAF symbol,
                      approach dashes, aircraft symbol / label anchors / history dots after a scripted flight), captured with
                      the recording `rendering` stand-in of tests/oracle_shims
  g9_wide.npz         >= 500 000 reference steps in compact form (per step: flags, done, actions_taken, reward; observation
                      and state every 16th step and on the last step of every episode): LOWW / LOWW_random / Simple /
                      UnitTest, dt 1/2/5, continuous and discrete, shaping x normalisation off, >= 50 each of win /
                      below-MVA / timeout terminals, episodes stepped on past `done` (incl. past a win)
  g12_timesteps.npz   SimParameters.timestep in {0.05, 0.1, 0.15, 0.3, 0.7, 1.3, 3.7} s (round 6): sustained descents into the MVAs,
                      altitude TIES (n x 41 dt ft above an MVA, decided by the reference's own float64 rounding), landings on
                      targets, >= 3 000-step slow episodes (compact format of g9)
  g13_timestep_sweep.npz  ten MORE timesteps, 0.01 ... 47 s (0.01, 0.033, 0.25, 0.9, 1.7, 2.5, 7.3, 13, 29, 47), each with random held actions
                      (blocks of 1 / 5 / 20 / 100 steps) on LOWW_random / LOWW unshaped / Simple un-normalised / LOWW_random discrete / UnitTest,
                      every second episode a descent, the last of each configuration a time-out (compact format of g9)
  g11_unbounded.npz   actions outside the action space, replayed by the reference: sustained a_phi up to +-3 (heading to 720 / -360
                      deg), headings wound to +-5 500 deg and back, un-clipped random actions, discrete heading indices beyond 360,
                      G9's winning intercepts flown at heading + 360 k, altitude targets within an fp32 action step of the MVA; the compact form of g9
  model_test_known_answers.json  the 8 known answers of the reference's own envs/atc/model_test.py

Usage:
  PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/generate_golden.py
Nothing here is imported by the product; the GPU box never sees /root/reference.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIMS = os.path.join(os.path.dirname(HERE), "oracle_shims")
REF = os.environ.get("ATC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, SHIMS)

import numpy as np  # noqa: E402

import envs.atc.atc_gym as ref_gym  # noqa: E402
import envs.atc.model as ref_model  # noqa: E402
import envs.atc.scenarios as ref_scen  # noqa: E402
import shapely.geometry as shape  # noqa: E402

F_BELOW_MVA, F_OUTSIDE, F_WON, F_TIMEOUT, F_INVALID_V, F_INVALID_H = 1, 2, 4, 8, 16, 32


def unit_test_world():
    """The fixture world of envs/atc/model_test.py:94-113 (same data as SimpleScenario, runway phi=180)."""
    mvas = [
        ref_model.MinimumVectoringAltitude(shape.Polygon([(15, 0), (35, 0), (35, 26)]), 3500),
        ref_model.MinimumVectoringAltitude(shape.Polygon([(15, 0), (35, 26), (35, 30), (15, 30), (15, 27.8)]), 2400),
        ref_model.MinimumVectoringAltitude(shape.Polygon([(15, 30), (35, 30), (35, 40), (15, 40)]), 4000),
        ref_model.MinimumVectoringAltitude(shape.Polygon([(0, 10), (15, 0), (15, 28.7), (0, 17)]), 8000),
        ref_model.MinimumVectoringAltitude(shape.Polygon([(0, 17), (15, 28.7), (15, 40), (0, 32)]), 6500),
    ]
    runway = ref_model.Runway(20, 20, 0, 180)
    return mvas, runway, ref_model.Airspace(mvas, runway)


class UnitTestScenario(ref_scen.Scenario):
    def __init__(self):
        self.mvas, self.runway, self.airspace = unit_test_world()
        self.entrypoints = [ref_model.EntryPoint(5, 35, 90, [150])]


SCENARIOS = {
    "LOWW": lambda: ref_scen.LOWW(),
    "LOWW_random": lambda: ref_scen.LOWW(random_entrypoints=True),
    "Simple": lambda: ref_scen.SimpleScenario(),
    "UnitTest": lambda: UnitTestScenario(),
    "Dyadic": lambda: DyadicScenario(),
}
G1_SCENARIOS = ("LOWW", "LOWW_random", "Simple", "UnitTest")   # g1_constants.json (unchanged since round 1)


def make_env(scen="LOWW", dt=1, shaping=True, normalize=True, discrete=False):
    sp = ref_model.SimParameters(dt, reward_shaping=shaping, normalize_state=normalize, discrete_action_space=discrete)
    return ref_gym.AtcGym(sim_parameters=sp, scenario=SCENARIOS[scen]())


def mva_or_neg(airspace, x, y):
    try:
        return int(airspace.get_mva_height(x, y))
    except ValueError:
        return -1


# ----------------------------------------------------------------------------------------------- G1
def gen_g1():
    out = {}
    for name in G1_SCENARIOS:
        env = make_env(name)
        c = env._runway.corridor
        d = {
            "mva_rings": [np.asarray(m.area_as_list).tolist() for m in env._mvas],
            "mva_heights": [int(m.height) for m in env._mvas],
            "mva_bounds": [list(map(float, m.outer_bounds)) for m in env._mvas],
            "bbox": list(map(float, env._airspace.get_bounding_box())),
            "world_max_distance": float(env._world_max_distance),
            "runway": [float(env._runway.x), float(env._runway.y), float(env._runway.h),
                       float(env._runway.phi_from_runway), float(env._runway.phi_to_runway)],
            "faf": c.faf.ravel().tolist(), "iaf": c.iaf.ravel().tolist(),
            "corner1": c.corner1.ravel().tolist(), "corner2": c.corner2.ravel().tolist(),
            "faf_iaf_normal": c._faf_iaf_normal.ravel().tolist(),
            "corridor_horizontal": c.corridor_horizontal_list.tolist(),
            "corridor1": c.corridor1_list.tolist(), "corridor2": c.corridor2_list.tolist(),
            "faf_angle": float(c.faf_angle),
            "faf_mva": int(env._faf_mva),
            "norm_min": env.normalization_state_min.astype(np.float64).tolist(),
            "norm_max": env.normalization_state_max.astype(np.float64).tolist(),
            "action_factor_continuous": [float(v) for v in env.normalization_action_factor],
            "action_offset": [float(v) for v in env.normalization_action_offset],
            "entrypoints": [[float(e.x), float(e.y), float(e.phi)] + [int(l) for l in e.levels]
                            for e in env._scenario.entrypoints],
            "reset_state": env.reset().astype(np.float64).tolist() if name != "LOWW_random" else None,
        }
        envd = make_env(name, discrete=True)
        d["action_factor_discrete"] = [float(v) for v in envd.normalization_action_factor]
        d["action_space_discrete_nvec"] = [int(v) for v in envd.action_space.nvec]
        out[name] = d
    with open(os.path.join(HERE, "g1_constants.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


# ------------------------------------------------------------------------------------ trajectory recorder
class Recorder:
    """Collects per-step records for a list of episodes (each episode has its own config/initial state)."""

    def __init__(self):
        self.ep = []  # dicts
        self.rows = {k: [] for k in ("action", "state", "obs", "raw", "reward", "done", "flags", "mva",
                                     "timesteps", "actions_taken", "total_reward")}

    def run(self, env, actions, scen, dt, shaping, normalize, discrete, init_state=None, init_timesteps=None,
            stop_on_done=True, last_action=None):
        """Resets env, optionally injects a state (attribute pokes on the reference objects), steps through
        `actions` (array [T,3]) and records.  Returns number of steps taken."""
        reset_obs = env.reset()
        ap = env._airplane
        if init_state is not None:
            ap.x, ap.y, ap.h, ap.phi, ap.v = [float(v) for v in init_state]
        if init_timesteps is not None:
            env.timesteps = int(init_timesteps)
        if last_action is not None:
            env.last_action = [float(v) for v in last_action]
        start = len(self.rows["reward"])
        ep = dict(scen=scen, dt=float(dt), shaping=bool(shaping), normalize=bool(normalize), discrete=bool(discrete),
                  init_state=[float(ap.x), float(ap.y), float(ap.h), float(ap.phi), float(ap.v)],
                  init_timesteps=int(env.timesteps), init_last_action=[float(v) for v in env.last_action],
                  reset_obs=reset_obs.astype(np.float64).tolist(), start=start)
        n = 0
        for a in actions:
            a64 = np.asarray(a, dtype=np.float64)
            den = [env._denormalized_action(float(a64[i]), i) for i in range(3)]
            flags = 0
            if den[0] < ap.v_min or den[0] > ap.v_max:
                flags |= F_INVALID_V
            if den[1] < ap.h_min or den[1] > ap.h_max:
                flags |= F_INVALID_H
            obs, rew, done, info = env.step(a64)
            m = mva_or_neg(env._airspace, ap.x, ap.y)
            if m < 0:
                flags |= F_OUTSIDE
            elif ap.h < m:
                flags |= F_BELOW_MVA
            if env._runway.inside_corridor(ap.x, ap.y, ap.h, ap.phi):
                flags |= F_WON
            if env.timesteps > env.timestep_limit:
                flags |= F_TIMEOUT
            assert bool(flags & (F_OUTSIDE | F_BELOW_MVA | F_WON | F_TIMEOUT)) == bool(done)
            r = self.rows
            r["action"].append(a64.copy())
            r["state"].append([ap.x, ap.y, ap.h, ap.phi, ap.v])
            r["obs"].append(np.asarray(obs, dtype=np.float32))
            r["raw"].append(np.asarray(info["original_state"], dtype=np.float32))
            r["reward"].append(float(rew))
            r["done"].append(int(done))
            r["flags"].append(flags)
            r["mva"].append(m)
            r["timesteps"].append(int(env.timesteps))
            r["actions_taken"].append(int(env.actions_taken))
            r["total_reward"].append(float(env.total_reward))
            n += 1
            if done and stop_on_done:
                break
        ep["steps"] = n
        self.ep.append(ep)
        return n

    def save(self, path):
        r = self.rows
        np.savez_compressed(
            path,
            episodes=json.dumps(self.ep),
            action=np.asarray(r["action"], dtype=np.float64),
            state=np.asarray(r["state"], dtype=np.float64),
            obs=np.asarray(r["obs"], dtype=np.float32),
            raw=np.asarray(r["raw"], dtype=np.float32),
            reward=np.asarray(r["reward"], dtype=np.float64),
            done=np.asarray(r["done"], dtype=np.uint8),
            flags=np.asarray(r["flags"], dtype=np.uint32),
            mva=np.asarray(r["mva"], dtype=np.int32),
            timesteps=np.asarray(r["timesteps"], dtype=np.int32),
            actions_taken=np.asarray(r["actions_taken"], dtype=np.int32),
            total_reward=np.asarray(r["total_reward"], dtype=np.float64),
        )


def f32(a):
    """Round actions to float32-representable values so the fp32 device path sees identical inputs."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


# ----------------------------------------------------------------------------------------------- G2
def gen_g2():
    rec = Recorder()
    fixed = [[0, 0, 0], [0, -1, -0.5], [0, 0, 0.5], [0, 15000 / 19000 - 1, -0.5]]
    for a in fixed:
        env = make_env()
        rec.run(env, np.tile(f32(a), (7000, 1)), "LOWW", 1, True, True, False)
    # dt = 5, shaping off, normalise off
    rec.run(make_env(dt=5), np.tile(f32([0, 0, 0]), (2000, 1)), "LOWW", 5, True, True, False)
    rec.run(make_env(shaping=False), np.tile(f32([0, -1, -0.5]), (2000, 1)), "LOWW", 1, False, True, False)
    rec.run(make_env(normalize=False), np.tile(f32([0, 0, 0.5]), (2000, 1)), "LOWW", 1, True, False, False)
    # discrete action space: (v idx, h idx, phi idx)
    rec.run(make_env(discrete=True), np.tile(np.array([15.0, 30.0, 135.0]), (4000, 1)), "LOWW", 1, True, True, True)
    rec.run(make_env(discrete=True), np.tile(np.array([3.0, 120.0, 20.0]), (4000, 1)), "LOWW", 1, True, True, True)
    # invalid (out of range) action, then valid ones (quirk Q6)
    acts = np.concatenate([np.tile(f32([1.5, -1.2, 3.0]), (3, 1)), np.tile(f32([0.2, -0.3, 0.1]), (30, 1)),
                           np.tile(f32([-1.01, 1.01, -1.0]), (3, 1)), np.tile(f32([1.0, 1.0, 1.0]), (30, 1))])
    rec.run(make_env(), acts, "LOWW", 1, True, True, False)
    # SimpleScenario
    # SimpleScenario: its own entry point (5,35) lies outside the airspace (episode ends on step 1); the other
    # episodes start from injected states inside it
    rec.run(make_env("Simple"), np.tile(f32([0, -0.5, 0.0]), (3000, 1)), "Simple", 1, True, True, False)
    rec.run(make_env("Simple"), np.tile(f32([0, -0.5, 0.0]), (3000, 1)), "Simple", 1, True, True, False,
            init_state=(5.0, 30.0, 9000.0, 90.0, 250.0))
    rec.run(make_env("Simple"), np.tile(f32([-0.5, -0.8, -0.2]), (3000, 1)), "Simple", 1, True, True, False,
            init_state=(30.0, 35.0, 7000.0, 200.0, 220.0))
    rec.run(make_env("Simple"), np.tile(f32([-0.5, -0.9, -0.7]), (3000, 1)), "Simple", 1, True, True, False,
            init_state=(25.0, 5.0, 6000.0, 20.0, 220.0))
    # stepping past done without reset (compute-performance protocol, quirk Q9): 40 extra steps
    env = make_env()
    rec.run(env, np.tile(f32([0, 0, 0.5]), (146 + 40, 1)), "LOWW", 1, True, True, False, stop_on_done=False)
    rec.save(os.path.join(HERE, "g2_scripted.npz"))
    return rec


# ----------------------------------------------------------------------------------------------- G3
def gen_g3():
    out = {}
    for name in ("LOWW", "Simple"):
        env = make_env(name)
        asp = env._airspace
        x0, y0, x1, y1 = asp.get_bounding_box()
        step = 0.25
        # lattice slightly larger than the bbox so that "outside" is covered on every side
        xs = np.arange(np.floor(x0) - 1.0, np.ceil(x1) + 1.0 + 1e-9, step)
        ys = np.arange(np.floor(y0) - 1.0, np.ceil(y1) + 1.0 + 1e-9, step)
        # offset lattice (avoids exact integer coordinates) in float32-representable form
        xs = f32(xs + 0.0625)
        ys = f32(ys + 0.03125)
        lat = np.empty((len(ys), len(xs)), dtype=np.int32)
        for j, y in enumerate(ys):
            for i, x in enumerate(xs):
                lat[j, i] = mva_or_neg(asp, float(x), float(y))
        # exact polygon vertices and edge midpoints: tie-break cases (float64 inputs; see tests for fp32 handling)
        pts = []
        for m in env._mvas:
            ring = np.asarray(m.area_as_list)
            for k in range(len(ring) - 1):
                pts.append(ring[k])
                pts.append(0.5 * (ring[k] + ring[k + 1]))
        # a few exactly representable integer / half-integer points
        for x in np.arange(np.floor(x0), np.ceil(x1) + 1, 2.0):
            for y in np.arange(np.floor(y0), np.ceil(y1) + 1, 2.5):
                pts.append([x, y])
        pts = np.asarray(pts, dtype=np.float64)
        ph = np.array([mva_or_neg(asp, float(p[0]), float(p[1])) for p in pts], dtype=np.int32)
        # the same special points rounded to float32 first (what an fp32 path can represent)
        pts32 = f32(pts)
        ph32 = np.array([mva_or_neg(asp, float(p[0]), float(p[1])) for p in pts32], dtype=np.int32)
        out[name + "_xs"] = xs
        out[name + "_ys"] = ys
        out[name + "_lattice"] = lat
        out[name + "_pts"] = pts
        out[name + "_pts_h"] = ph
        out[name + "_pts32"] = pts32
        out[name + "_pts32_h"] = ph32
    np.savez_compressed(os.path.join(HERE, "g3_mva.npz"), **out)
    return out


# ----------------------------------------------------------------------------------------------- G4
def gen_g4():
    out = {}
    for name in ("LOWW", "UnitTest", "Simple"):
        env = make_env(name)
        rw = env._runway
        fx, fy = rw.corridor.faf.ravel()
        xs = f32(np.arange(-5.0, 5.0 + 1e-9, 0.25) + fx + 0.013)
        ys = f32(np.arange(-5.0, 5.0 + 1e-9, 0.25) + fy - 0.007)
        hs = np.array([2000.0, 3000.0, 3400.0, 3500.0, 4000.0]) if name == "LOWW" else \
            np.array([0.0, 1000.0, 2400.0, 2700.0, 4000.0])
        phis = np.arange(0.0, 360.0, 5.0)
        full = np.zeros((len(ys), len(xs), len(hs), len(phis)), dtype=np.uint8)
        ang = np.zeros((len(ys), len(xs), len(phis)), dtype=np.uint8)
        for j, y in enumerate(ys):
            for i, x in enumerate(xs):
                for p, phi in enumerate(phis):
                    ang[j, i, p] = rw.corridor._inside_corridor_angle(float(x), float(y), float(phi))
                    for k, h in enumerate(hs):
                        full[j, i, k, p] = rw.inside_corridor(float(x), float(y), float(h), float(phi))
        out[name + "_xs"], out[name + "_ys"], out[name + "_hs"], out[name + "_phis"] = xs, ys, hs, phis
        out[name + "_inside"] = np.packbits(full.ravel())
        out[name + "_inside_shape"] = np.array(full.shape)
        out[name + "_angle"] = np.packbits(ang.ravel())
        out[name + "_angle_shape"] = np.array(ang.shape)
        # unwrapped / negative headings also occur on the hot path (heading is never wrapped, quirk Q4)
        phis2 = np.array([-725.0, -380.0, -20.0, -0.5, 359.5, 700.0, 1060.0, 335.0 + 360.0, 340.0 - 720.0])
        ex = np.zeros((len(ys), len(xs), len(phis2)), dtype=np.uint8)
        for j, y in enumerate(ys):
            for i, x in enumerate(xs):
                for p, phi in enumerate(phis2):
                    ex[j, i, p] = rw.inside_corridor(float(x), float(y), float(hs[1]), float(phi))
        out[name + "_phis_unwrapped"] = phis2
        out[name + "_inside_unwrapped"] = ex
    np.savez_compressed(os.path.join(HERE, "g4_corridor.npz"), **out)
    return out


# ----------------------------------------------------------------------------------------------- G5
def gen_g5():
    G = ref_gym.AtcGym
    rng = np.random.default_rng(5)
    n = 4000
    d_faf = f32(rng.uniform(0, 110, n))
    phi_rel_faf = f32(rng.uniform(-180, 180, n))
    phi_plane = f32(rng.uniform(-400, 760, n))
    h = f32(rng.uniform(0, 38000, n))
    on_gp = f32(rng.uniform(2000, 36000, n))
    phi_to_rwy = 340.0
    diag = 102.46251265706887
    pos = np.array([G._reward_approach_position(float(a), phi_to_rwy, float(b), diag) for a, b in zip(d_faf, phi_rel_faf)])
    ang = np.array([G._reward_approach_angle(phi_to_rwy, float(b), float(c), float(p))
                    for b, c, p in zip(phi_rel_faf, phi_plane, pos)])
    gs = np.array([G._reward_glideslope(float(a), float(b), float(p)) for a, b, p in zip(h, on_gp, pos)])
    sig = np.array([ref_gym.sigmoid_distance_func(float(a), diag) for a in d_faf])
    a1 = f32(rng.uniform(-720, 720, n))
    a2 = f32(rng.uniform(-720, 720, n))
    rel = np.array([ref_model.relative_angle(float(a), float(b)) for a, b in zip(a1, a2)])
    # exact multiples (modulo edge cases)
    e1 = np.array([0.0, 340.0, 340.0, 340.0, 160.0, 0.0, 360.0, -360.0, 180.0, 90.0])
    e2 = np.array([0.0, 160.0, 520.0, -20.0, 340.0, 180.0, 0.0, 0.0, 0.0, -90.0])
    erel = np.array([ref_model.relative_angle(float(a), float(b)) for a, b in zip(e1, e2)])
    np.savez_compressed(os.path.join(HERE, "g5_shaping.npz"), d_faf=d_faf, phi_rel_faf=phi_rel_faf,
                        phi_plane=phi_plane, h=h, on_gp=on_gp, phi_to_rwy=phi_to_rwy, diag=diag, pos=pos, ang=ang,
                        gs=gs, sig=sig, a1=a1, a2=a2, rel=rel, e1=e1, e2=e2, erel=erel)


# ----------------------------------------------------------------------------------------------- G6
def hold_actions(rng, n_steps, hold, discrete, nvec=None):
    acts = []
    cur = None
    for t in range(n_steps):
        if t % hold == 0:
            if discrete:
                cur = np.floor(rng.uniform(0, 1, 3) * nvec).astype(np.float64)
            else:
                cur = f32(rng.uniform(-1, 1, 3))
        acts.append(cur)
    return np.asarray(acts)


def gen_g6():
    rec = Recorder()
    # continuous, default LOWW, actions resampled every 20 steps (atc-gym-demo.py:18-19 protocol)
    env = make_env()
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        rec.run(env, hold_actions(rng, 6100, 20, False), "LOWW", 1, True, True, False)
    # per-step resampled actions (short horizon) — exercises the rate limiters and actions_taken
    for seed in range(4):
        rng = np.random.default_rng(2000 + seed)
        rec.run(env, hold_actions(rng, 400, 1, False), "LOWW", 1, True, True, False)
    # low-altitude biased targets -> below-MVA terminals; slightly out-of-range components -> invalid actions
    for seed in range(16):
        rng = np.random.default_rng(2500 + seed)
        acts = hold_actions(rng, 6100, 20, False)
        acts[:, 1] = f32(-1.0 + 0.12 * (acts[:, 1] + 1.0) * 0.5)
        rec.run(env, acts, "LOWW", 1, True, True, False)
    for seed in range(8):
        rng = np.random.default_rng(2600 + seed)
        acts = f32(hold_actions(rng, 6100, 20, False) * 1.08)
        rec.run(env, acts, "LOWW", 1, True, True, False)
    # discrete
    envd = make_env(discrete=True)
    nvec = np.array([20, 380, 360])
    for seed in range(12):
        rng = np.random.default_rng(3000 + seed)
        rec.run(envd, hold_actions(rng, 6100, 20, True, nvec), "LOWW", 1, True, True, True)
    # random entry points (the drawn entry/level is recoverable from init_state)
    random.seed(7)
    envr = make_env("LOWW_random")
    for seed in range(24):
        rng = np.random.default_rng(4000 + seed)
        rec.run(envr, hold_actions(rng, 6100, 20, False), "LOWW_random", 1, True, True, False)
    # dt = 2 with random entries
    envr2 = make_env("LOWW_random", dt=2)
    for seed in range(6):
        rng = np.random.default_rng(5000 + seed)
        rec.run(envr2, hold_actions(rng, 3100, 10, False), "LOWW_random", 2, True, True, False)
    # injected wins: aircraft placed on the intercept, descending through the glide path (LOWW FAF 47.69,36.31)
    env = make_env()
    win_inits = [
        # x, y, h, phi, v
        (48.9, 31.9, 3300.0, 345.0, 200.0),
        (47.0, 32.0, 3400.0, 10.0, 220.0),
        (50.6, 33.2, 3200.0, 310.0, 180.0),
        (49.2, 30.5, 5000.0, 340.0, 250.0),
        (46.2, 31.8, 3600.0, 20.0, 250.0),
        (51.3, 33.0, 2900.0, 300.0, 160.0),
    ]
    for k, st in enumerate(win_inits):
        # hold heading, descend to 2700 ft, slow down
        a_h = 2.0 * 2700.0 / 38000.0 - 1.0
        a_phi = 2.0 * st[3] / 360.0 - 1.0
        a_v = 2.0 * (st[4] - 100.0) / 200.0 - 1.0
        acts = np.tile(f32([a_v, a_h, a_phi]), (600, 1))
        rec.run(env, acts, "LOWW", 1, True, True, False, init_state=st, init_timesteps=100 * k)
    # win with the time bonus exhausted (timesteps near the limit) and win on the timeout step (timeout overrides)
    a = f32([0.0, 2.0 * 2700.0 / 38000.0 - 1.0, 2.0 * 345.0 / 360.0 - 1.0])
    rec.run(env, np.tile(a, (600, 1)), "LOWW", 1, True, True, False, init_state=win_inits[0], init_timesteps=5900)
    rec.run(env, np.tile(a, (600, 1)), "LOWW", 1, True, True, False, init_state=win_inits[0], init_timesteps=5990)
    # corridor entry exactly on step 6001: timeout overrides the win (both flags set)
    rec.run(env, np.tile(a, (600, 1)), "LOWW", 1, True, True, False, init_state=win_inits[0], init_timesteps=5972)
    rec.run(env, np.tile(a, (600, 1)), "LOWW", 1, True, True, False, init_state=win_inits[0], init_timesteps=5971)
    # timeouts: start at 5990 far from everything
    for seed in range(3):
        rng = np.random.default_rng(6000 + seed)
        rec.run(env, hold_actions(rng, 40, 5, False), "LOWW", 1, True, True, False, init_timesteps=5990 - seed)
    rec.save(os.path.join(HERE, "g6_rollouts.npz"))
    return rec


# ----------------------------------------------------------------------------------------------- G7
def gen_g7():
    """Reset/metrics sequence: winning_ratio, _win_buffer, actions_per_timestep over consecutive episodes."""
    env = make_env()
    win_state = (48.9, 31.9, 3300.0, 345.0, 200.0)
    a_win = f32([0.0, 2.0 * 2700.0 / 38000.0 - 1.0, 2.0 * 345.0 / 360.0 - 1.0])
    plan = ["lose", "win", "win", "lose", "win", "timeout", "win", "lose", "lose", "win", "win", "win", "lose", "win",
            "lose", "lose"]
    episodes = []
    rng = np.random.default_rng(77)
    for kind in plan:
        env.reset()
        ep = {"kind": kind, "after_reset": {"winning_ratio": float(env.winning_ratio),
                                            "win_buffer": [int(v) for v in env._win_buffer],
                                            "episodes_run": int(env._episodes_run)}}
        if kind == "win":
            ap = env._airplane
            ap.x, ap.y, ap.h, ap.phi, ap.v = win_state
            acts = np.tile(a_win, (600, 1))
        elif kind == "timeout":
            env.timesteps = 5980
            acts = hold_actions(rng, 60, 7, False)
        else:
            acts = hold_actions(rng, 6100, 20, False)
        ep["init_state"] = [float(v) for v in (env._airplane.x, env._airplane.y, env._airplane.h, env._airplane.phi,
                                                env._airplane.v)]
        ep["init_timesteps"] = int(env.timesteps)
        used = []
        apt = []
        for a in acts:
            _, r, done, _ = env.step(np.asarray(a, dtype=np.float64))
            used.append([float(v) for v in a])
            apt.append(float(env.actions_per_timestep))
            if done:
                break
        ep["actions"] = used
        ep["actions_per_timestep"] = apt
        ep["final"] = {"total_reward": float(env.total_reward), "last_reward": float(env.last_reward),
                       "timesteps": int(env.timesteps), "actions_taken": int(env.actions_taken),
                       "win_buffer": [int(v) for v in env._win_buffer], "winning_ratio": float(env.winning_ratio)}
        episodes.append(ep)
    env.reset()
    tail = {"winning_ratio": float(env.winning_ratio), "win_buffer": [int(v) for v in env._win_buffer],
            "episodes_run": int(env._episodes_run)}
    with open(os.path.join(HERE, "g7_metrics.json"), "w") as f:
        json.dump({"episodes": episodes, "after_last_reset": tail}, f)


# ----------------------------------------------------------------------------------------------- G8
DYADIC_MVAS = [
    # list order = lookup priority (model.py:283).  Polygon 0 overlaps the partition below: it must win inside its ring.
    ([(16, 4), (28, 16), (16, 28), (4, 16)], 2500),
    ([(0, 0), (16, 0), (16, 16), (0, 16)], 3000),
    ([(16, 0), (32, 0), (32, 16), (16, 16)], 4000),
    ([(0, 16), (16, 16), (24, 32), (0, 32)], 5000),
    ([(16, 16), (32, 16), (32, 32), (24, 32)], 6000),
    ([(32, 8), (40, 8), (40, 24.5), (32, 24.5)], 7000),   # shares part of the x = 32 border of two polygons
]


def dyadic_world():
    mvas = [ref_model.MinimumVectoringAltitude(shape.Polygon(p), h) for p, h in DYADIC_MVAS]
    runway = ref_model.Runway(16, 10, 0, 180)
    return mvas, runway, ref_model.Airspace(mvas, runway)


def gen_g8():
    mvas, runway, asp = dyadic_world()
    pts = []
    eps = [0.0, 2.0 ** -10, -2.0 ** -10, 3 * 2.0 ** -10, -3 * 2.0 ** -10]
    for m in mvas:
        ring = np.asarray(m.area_as_list)
        for k in range(len(ring) - 1):
            a, b = ring[k], ring[k + 1]
            for t in (0.0, 0.25, 0.5, 0.75):
                p = a * (1 - t) + b * t
                for dx in eps:
                    for dy in eps:
                        pts.append([p[0] + dx, p[1] + dy])
    for x in np.arange(-1.0, 41.5, 0.5):
        for y in np.arange(-1.0, 33.5, 0.5):
            pts.append([x, y])
    pts = np.asarray(pts, dtype=np.float64)
    assert np.array_equal(pts, pts.astype(np.float32).astype(np.float64))  # every input is exactly representable in fp32
    h = np.array([mva_or_neg(asp, float(p[0]), float(p[1])) for p in pts], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "g8_tiebreak.npz"), pts=pts, h=h,
                        rings=json.dumps([[list(map(float, q)) for q in p] for p, _ in DYADIC_MVAS]),
                        heights=np.array([hh for _, hh in DYADIC_MVAS], dtype=np.int32),
                        runway=np.array([16.0, 10.0, 0.0, 180.0]))
    return pts, h


# ----------------------------------------------------------------------------------------------- G9
class DyadicScenario(ref_scen.Scenario):
    def __init__(self):
        self.mvas, self.runway, self.airspace = dyadic_world()
        self.entrypoints = [ref_model.EntryPoint(2, 30, 90, [150])]


class WideRecorder:
    """Compact recorder: every step keeps flags / done / actions_taken / reward, every `stride`-th step and the last step of
    an episode also keep the observation and the float64 state."""

    def __init__(self, stride=16):
        self.stride = stride
        self.ep = []
        self.flags, self.done, self.acts, self.reward = [], [], [], []
        self.samp_rows, self.obs, self.state = [], [], []
        self.act_rows, self.act_vals = [], []

    def run(self, env, actions, scen, dt, shaping, normalize, discrete, init_state=None, init_timesteps=None,
            extra_after_done=0):
        env.reset()
        ap = env._airplane
        if init_state is not None:
            ap.x, ap.y, ap.h, ap.phi, ap.v = [float(v) for v in init_state]
        if init_timesteps is not None:
            env.timesteps = int(init_timesteps)
        start = len(self.flags)
        ep = dict(scen=scen, dt=float(dt), shaping=bool(shaping), normalize=bool(normalize), discrete=bool(discrete),
                  init_state=[float(ap.x), float(ap.y), float(ap.h), float(ap.phi), float(ap.v)],
                  init_timesteps=int(env.timesteps), init_last_action=[float(v) for v in env.last_action], start=start)
        n, after, prev = 0, -1, None
        for a in actions:
            a64 = np.asarray(a, dtype=np.float64)
            den = [env._denormalized_action(float(a64[i]), i) for i in range(3)]
            flags = 0
            if den[0] < ap.v_min or den[0] > ap.v_max:
                flags |= F_INVALID_V
            if den[1] < ap.h_min or den[1] > ap.h_max:
                flags |= F_INVALID_H
            obs, rew, done, info = env.step(a64)
            m = mva_or_neg(env._airspace, ap.x, ap.y)
            if m < 0:
                flags |= F_OUTSIDE
            elif ap.h < m:
                flags |= F_BELOW_MVA
            if env._runway.inside_corridor(ap.x, ap.y, ap.h, ap.phi):
                flags |= F_WON
            if env.timesteps > env.timestep_limit:
                flags |= F_TIMEOUT
            assert bool(flags & (F_OUTSIDE | F_BELOW_MVA | F_WON | F_TIMEOUT)) == bool(done)
            row = start + n
            if prev is None or not np.array_equal(prev, a64):
                self.act_rows.append(row)
                self.act_vals.append(a64.copy())
                prev = a64
            self.flags.append(flags)
            self.done.append(int(done))
            self.acts.append(int(env.actions_taken))
            self.reward.append(float(rew))
            n += 1
            if done and after < 0:
                after = extra_after_done
            last = (after == 0)
            if n % self.stride == 0 or done or last:
                self.samp_rows.append(row)
                self.obs.append(np.asarray(obs, dtype=np.float32))
                self.state.append([ap.x, ap.y, ap.h, ap.phi, ap.v])
            if after == 0:
                break
            if after > 0:
                after -= 1
        ep["steps"] = n
        ep["total_reward"] = float(env.total_reward)
        self.ep.append(ep)
        return n

    def save(self, path):
        np.savez_compressed(
            path, episodes=json.dumps(self.ep), flags=np.asarray(self.flags, dtype=np.uint8),
            done=np.asarray(self.done, dtype=np.uint8), actions_taken=np.asarray(self.acts, dtype=np.int32),
            reward=np.asarray(self.reward, dtype=np.float64).astype(np.float32),
            samp_rows=np.asarray(self.samp_rows, dtype=np.int64), obs=np.asarray(self.obs, dtype=np.float32),
            state=np.asarray(self.state, dtype=np.float64), act_rows=np.asarray(self.act_rows, dtype=np.int64),
            act_vals=np.asarray(self.act_vals, dtype=np.float64).astype(np.float32))


def interior_states(rng, scen_env, n, h_lo, h_hi):
    """Random aircraft states strictly inside the airspace (rejection sampling against the reference's own lookup)."""
    x0, y0, x1, y1 = scen_env._airspace.get_bounding_box()
    out = []
    while len(out) < n:
        x, y = float(np.float32(rng.uniform(x0, x1))), float(np.float32(rng.uniform(y0, y1)))
        m = mva_or_neg(scen_env._airspace, x, y)
        if m < 0:
            continue
        out.append((x, y, float(np.float32(max(m + 500.0, rng.uniform(h_lo, h_hi)))), float(rng.integers(0, 360)),
                    float(rng.integers(150, 280))))
    return out


def gen_g9():
    rec = WideRecorder()
    nvec = np.array([20, 380, 360])

    def batch(scen, n_eps, seed0, horizon, hold, dt=1, shaping=True, normalize=True, discrete=False, tweak=None,
              inits=None, init_t=None, extra=0):
        env = make_env(scen, dt=dt, shaping=shaping, normalize=normalize, discrete=discrete)
        for k in range(n_eps):
            rng = np.random.default_rng(seed0 + k)
            acts = hold_actions(rng, horizon, hold, discrete, nvec)
            if tweak is not None:
                acts = tweak(acts, rng)
            rec.run(env, acts, scen, dt, shaping, normalize, discrete,
                    init_state=None if inits is None else inits[k], init_timesteps=None if init_t is None else init_t(k),
                    extra_after_done=extra)

    random.seed(11)
    batch("LOWW", 260, 10000, 6100, 20)
    batch("LOWW_random", 260, 20000, 6100, 20)
    batch("LOWW_random", 60, 21000, 3100, 10, dt=2, discrete=True)
    batch("LOWW_random", 60, 22000, 1300, 4, dt=5, discrete=True)
    batch("LOWW", 80, 23000, 6100, 20, discrete=True)
    batch("LOWW_random", 80, 24000, 6100, 20, shaping=False, normalize=False)
    batch("LOWW_random", 40, 24500, 6100, 20, shaping=False)
    batch("LOWW_random", 40, 24600, 6100, 20, normalize=False)

    def low(acts, rng):   # low altitude targets -> below-MVA terminals
        acts = acts.copy()
        acts[:, 1] = f32(-1.0 + 0.14 * (acts[:, 1] + 1.0) * 0.5)
        return acts
    batch("LOWW_random", 90, 25000, 6100, 20, tweak=low)
    batch("LOWW", 40, 26000, 6100, 20, tweak=lambda a, r: f32(a * 1.06))   # invalid targets now and then

    # wins: randomised starts on the LOWW intercept, heading held, descending through the glide path; some of them are
    # stepped on past `done` (the aircraft keeps flying and can win again, quirk Q9)
    base = [(48.9, 31.9, 3300.0, 345.0, 200.0), (47.0, 32.0, 3400.0, 10.0, 220.0), (50.6, 33.2, 3200.0, 310.0, 180.0),
            (49.2, 30.5, 5000.0, 340.0, 250.0), (46.2, 31.8, 3600.0, 20.0, 250.0), (51.3, 33.0, 2900.0, 300.0, 160.0)]
    env = make_env()
    rng = np.random.default_rng(27000)
    for k in range(120):
        b = base[k % len(base)]
        st = (float(np.float32(b[0] + rng.uniform(-0.25, 0.25))), float(np.float32(b[1] + rng.uniform(-0.25, 0.25))),
              float(np.float32(b[2] + rng.uniform(-150, 150))), float(np.round(b[3] + rng.uniform(-4, 4))),
              float(np.round(b[4] + rng.uniform(-20, 20))))
        a = f32([2.0 * (st[4] - 100.0) / 200.0 - 1.0, 2.0 * 2700.0 / 38000.0 - 1.0, 2.0 * st[3] / 360.0 - 1.0])
        rec.run(env, np.tile(a, (700, 1)), "LOWW", 1, True, True, False, init_state=st,
                init_timesteps=int(rng.integers(0, 5800)), extra_after_done=(25 if k % 4 == 0 else 0))
    # timeouts (timestep counter close to the limit)
    batch("LOWW_random", 70, 28000, 60, 5, init_t=lambda k: 5975 + (k % 25))
    # stepping on past other terminals
    batch("LOWW", 12, 28500, 6100, 20, extra=40)
    # SimpleScenario / the reference's unit-test world / the dyadic tie-break sector, from interior states
    for scen, seed0 in (("Simple", 29000), ("UnitTest", 30000), ("Dyadic", 31000)):
        env0 = make_env(scen)
        inits = interior_states(np.random.default_rng(seed0), env0, 90, 8500.0, 12000.0)
        batch(scen, 90, seed0 + 100, 3000, 20, inits=inits)
    rec.save(os.path.join(HERE, "g9_wide.npz"))
    return rec


# ----------------------------------------------------------------------------------------------- G10
def gen_g10():
    """What AtcGym.render() draws (atc_gym.py:367-552), as plain geometry in the reference's screen coordinates."""
    def pts(v):
        return [[float(np.asarray(p).ravel()[0]), float(np.asarray(p).ravel()[1])] for p in v]

    def geom(g):
        d = {"kind": type(g).__name__, "color": [float(c) for c in g.color] if g.color else None,
             "linewidth": float(g.linewidth)}
        if hasattr(g, "v"):
            d["v"] = pts(g.v)
        if hasattr(g, "close"):
            d["close"] = bool(g.close)
        if hasattr(g, "radius"):
            d["radius"] = float(g.radius)
            d["translation"] = [float(t) for t in g.attrs[0].translation]
        if hasattr(g, "text"):
            d["text"], d["x"], d["y"] = g.text, float(g.x), float(g.y)
        return d

    out = {}
    for name in ("LOWW", "Simple"):
        env = make_env(name)
        frames = []
        acts = [f32([0.0, -0.2, 0.5])] * 60
        if name == "Simple":
            ap = env._airplane
            ap.x, ap.y, ap.h, ap.phi, ap.v = 5.0, 30.0, 9000.0, 90.0, 250.0
        env.render(mode='rgb_array')
        static = [geom(g) for g in env.viewer.geoms]
        frames.append({"step": 0, "state": [float(v) for v in (env._airplane.x, env._airplane.y, env._airplane.h,
                                                                 env._airplane.phi, env._airplane.v)],
                       "geoms": [geom(g) for g in env.viewer.last_frame]})
        for t, a in enumerate(acts):
            env.step(np.asarray(a, dtype=np.float64))
            if (t + 1) % 20 == 0:
                env.render(mode='rgb_array')
                frames.append({"step": t + 1, "state": [float(v) for v in (env._airplane.x, env._airplane.y, env._airplane.h,
                                                                             env._airplane.phi, env._airplane.v)],
                               "total_reward": float(env.total_reward), "last_reward": float(env.last_reward),
                               "geoms": [geom(g) for g in env.viewer.last_frame]})
        out[name] = {"width": int(env.viewer.width), "height": int(env.viewer.height), "scale": float(env._scale),
                     "padding": int(env._padding), "static": static, "frames": frames,
                     "actions": [[float(v) for v in a] for a in acts],
                     "init_state": frames[0]["state"]}
    with open(os.path.join(HERE, "g10_render_geometry.json"), "w") as f:
        json.dump(out, f)
    return out


# ------------------------------------------------------------------------- reference unit-test known answers
def gen_model_test():
    """Inputs and expected outputs of envs/atc/model_test.py:10-92, re-evaluated here against the reference."""
    mvas, rw, asp = unit_test_world()
    faf = rw.corridor.faf
    mva_faf = asp.get_mva_height(faf[0][0], faf[1][0])
    cases = {
        "world": "UnitTest",
        "mva_at_faf": int(mva_faf),
        "get_mva_height": [{"x": 34, "y": 1, "expect": int(asp.get_mva_height(34, 1))}],
        "inside_corridor": [
            {"x": 19, "y": 10, "h": mva_faf + 300, "phi": 30, "expect": bool(rw.inside_corridor(19, 10, mva_faf + 300, 30))},
            {"x": 19, "y": 10, "h": mva_faf, "phi": 330, "expect": bool(rw.inside_corridor(19, 10, mva_faf, 330))},
        ],
        "inside_corridor_angle": [
            {"x": 21, "y": 10, "phi": 30, "expect": bool(rw.corridor._inside_corridor_angle(21, 10, 30))},
            {"x": 19, "y": 10, "phi": 340, "expect": bool(rw.corridor._inside_corridor_angle(19, 10, 340))},
            {"x": 19, "y": 10, "phi": 190, "expect": bool(rw.corridor._inside_corridor_angle(19, 10, 190))},
            {"x": 21, "y": 10, "phi": 340, "expect": bool(rw.corridor._inside_corridor_angle(21, 10, 340))},
        ],
        "bounding_box": [float(v) for v in asp.get_bounding_box()],
    }
    # expected values as written in the reference test file (model_test.py:16,27,38,49,59,69,79,89-92)
    assert cases["get_mva_height"][0]["expect"] == 3500
    assert [c["expect"] for c in cases["inside_corridor"]] == [True, False]
    assert [c["expect"] for c in cases["inside_corridor_angle"]] == [False, False, False, True]
    assert cases["bounding_box"] == [0.0, 0.0, 35.0, 40.0]
    with open(os.path.join(HERE, "model_test_known_answers.json"), "w") as f:
        json.dump(cases, f, indent=1)


# ----------------------------------------------------------------------------------------------- G11
def gen_g11():
    """Actions OUTSIDE the action space (the reference enforces nothing: Box(-1, 1) is not applied by step(), atc_gym.py:128-141;
    Airplane.action_phi validates nothing and never wraps, model.py:104-120): sustained a_phi in {+-1.2, +-1.43, +-2, +-3} with
    a_v / a_h in and out of range alongside, headings wound up to several thousand degrees and back, un-clipped
    random continuous actions, discrete heading indices beyond MultiDiscrete's 360 — every episode stepped on past `done` so
    that the heading passes -76 / 436 deg (the 32-bit range of the fp32 state format) and reaches +-360 / 720 and beyond."""
    rec = WideRecorder(stride=8)
    env = make_env()
    # Every episode flies on for at most 240 s after its first `done` (an aircraft that leaves the sector stays within the 24 nm
    # the fp32 position grid reaches beyond the bounding box, include/atc_step.h "Aircraft positions").
    # 1. sustained out-of-range heading actions from the reset state
    for k, ap in enumerate((1.2, -1.2, 1.43, -1.43, 2.0, -2.0, 3.0, -3.0)):
        for av, ah in ((0.0, 0.0), (1.5, -1.2), (-0.3, 0.4)):
            acts = np.tile(f32([av, ah, ap]), (1400, 1))
            rec.run(env, acts, "LOWW", 1, True, True, False, extra_after_done=240)
    # 2. wind the heading up in circles (radius 1.3 nm: the position stays put) and unwind it again through zero
    inits = interior_states(np.random.default_rng(41000), env, 6, 9000.0, 14000.0)
    for k, st in enumerate(inits):
        up = (30.0, 12.5, 7.25, -31.0, 11.37, -6.6)[k]
        t1, t2 = 180.0 + 180.0 * up, 180.0 - 180.0 * up            # the two heading targets (atc_gym.py:333-335)
        n1, n2 = int(abs(t1 - st[3]) / 3) + 8, int(abs(t2 - t1) / 3) + 8   # turning all the way at 3 deg / s, then 8 s straight
        acts = np.concatenate([np.tile(f32([0.0, 0.1, up]), (n1, 1)), np.tile(f32([0.2, 0.1, -up]), (n2, 1)),
                               np.tile(f32([0.2, 0.1, 0.25]), (140, 1))])
        rec.run(env, acts, "LOWW", 1, True, True, False, init_state=st, extra_after_done=len(acts))
    # 3. un-clipped random continuous actions (what an un-squashed Gaussian policy emits), held 20 steps
    for scen, dt, n_eps, seed0, horizon in (("LOWW_random", 1, 40, 42000, 1500), ("LOWW_random", 5, 16, 43000, 600),
                                            ("Simple", 2, 12, 44000, 800)):
        e2 = make_env(scen, dt=dt)
        its = interior_states(np.random.default_rng(seed0 + 500), e2, n_eps, 8000.0, 15000.0)
        for k in range(n_eps):
            rng = np.random.default_rng(seed0 + k)
            nb = horizon // 20
            a = np.stack([rng.uniform(-1.2, 1.2, nb), rng.uniform(-0.4, 1.1, nb), rng.uniform(-4.0, 4.0, nb)], 1)
            a[rng.uniform(size=nb) < 0.15, 2] *= 6.0   # now and then far out
            rec.run(e2, np.repeat(f32(a), 20, 0), scen, dt, True, True, False, init_state=its[k], extra_after_done=240 // dt)
    # 4. discrete action space, heading indices outside [0, 360) (atc_gym.py:329-330: index * 1 + 0, never validated)
    e3 = make_env("LOWW", discrete=True)
    for k, idx in enumerate((500, -90, 1000, -700, 436, 437, -76, -77)):
        acts = np.tile(np.array([15.0, 120.0, float(idx)]), (1400, 1))
        rec.run(e3, acts, "LOWW", 1, True, True, True, extra_after_done=240)
    # 5. the corridor with an un-wrapped heading: the winning intercepts of G9 flown at heading + 360 k (placed so, and held by an
    #    action outside the action space): Runway.inside_corridor and the angle window are periodic in the heading (model.py:212-231)
    #    — every one of these must win like its wrapped twin, some stepped on past the win
    base = [(48.9, 31.9, 3300.0, 345.0, 200.0), (47.0, 32.0, 3400.0, 10.0, 220.0), (50.6, 33.2, 3200.0, 310.0, 180.0),
            (49.2, 30.5, 5000.0, 340.0, 250.0), (46.2, 31.8, 3600.0, 20.0, 250.0), (51.3, 33.0, 2900.0, 300.0, 160.0)]
    rng = np.random.default_rng(45000)
    for k in range(36):
        b = base[k % len(base)]
        turns = (1, -1, 2, -2, 5, -6)[(k // len(base)) % 6]
        st = (float(np.float32(b[0] + rng.uniform(-0.25, 0.25))), float(np.float32(b[1] + rng.uniform(-0.25, 0.25))),
              float(np.float32(b[2] + rng.uniform(-150, 150))), float(np.round(b[3] + rng.uniform(-4, 4))) + 360.0 * turns,
              float(np.round(b[4] + rng.uniform(-20, 20))))
        a = f32([2.0 * (st[4] - 100.0) / 200.0 - 1.0, 2.0 * 2700.0 / 38000.0 - 1.0, 2.0 * st[3] / 360.0 - 1.0])
        rec.run(env, np.tile(a, (700, 1)), "LOWW", 1, True, True, False, init_state=st,
                init_timesteps=int(rng.integers(0, 5800)), extra_after_done=(25 if k % 4 == 0 else 0))
    # 6. "descend to the MVA": an altitude target within one fp32 step of the action of the MVA height below the aircraft, on either side
    #    of it and exactly on it (T = a * 19000 + 19000 in float64, atc_gym.py:333-335: 1.1e-3 ft per fp32 step of a).  The aircraft is
    #    30 ft above, so it lands ON the target in its first step: below-MVA (atc_gym.py:149-153) iff the target is below — to the
    #    last bit of a float64
    its = interior_states(np.random.default_rng(46000), env, 24, 2000.0, 2001.0)
    for k, st in enumerate(its):
        m = mva_or_neg(env._airspace, st[0], st[1])
        a0 = np.float32(m / 19000.0 - 1.0)
        a = [np.nextafter(a0, np.float32(-2)), a0, np.nextafter(a0, np.float32(2)), np.nextafter(np.nextafter(a0, np.float32(-2)), np.float32(-2))][k % 4]
        acts = np.tile(f32([2.0 * (st[4] - 100.0) / 200.0 - 1.0, a, 2.0 * st[3] / 360.0 - 1.0]), (6, 1))
        rec.run(env, acts, "LOWW", 1, True, True, False, init_state=(st[0], st[1], float(m + 30), st[3], st[4]), extra_after_done=5)
    rec.save(os.path.join(HERE, "g11_unbounded.npz"))
    return rec


# ----------------------------------------------------------------------------------------------- G12
G12_DTS = (0.05, 0.1, 0.15, 0.3, 0.7, 1.3, 3.7)


def disc_inside(airspace, x, y, radius, m):
    """every point of the disc around (x, y) — sampled on three rings — lies in the MVA of height m"""
    for rr in (radius, 0.66 * radius, 0.33 * radius):
        for k in range(16):
            a = 2.0 * np.pi * k / 16.0
            if mva_or_neg(airspace, x + rr * np.cos(a), y + rr * np.sin(a)) != m:
                return False
    return True


def gen_g12():
    """SimParameters.timestep values OTHER than 1 / 2 / 5 s (model.py:132-145: any float; it scales h_dot_min / h_dot_max / a / phi_dot
    in Airplane.action_* (model.py:75-78,97-100,117-120), the displacement (model.py:126) and the base reward (atc_gym.py:137)).
    41 * 0.1 and 15 * 0.15 are not small multiples of an fp32 ulp of an altitude: an fp32 altitude accumulator drifts by 0.4 ulp
    per step and the below-MVA flag (atc_gym.py:149-153) comes one step off (round-5 review: 19 of 120 descents).
      1. sustained descents from the entry points at the maximum rate into whatever MVA lies below: fixed action
         [U(-1, 1), U(-1, -0.7), U(-1, 1)] until done (the review's probe), LOWW and LOWW_random, every timestep above
      2. TIES: an aircraft circling (radius 0.53 nm) inside one MVA polygon, n x 41 dt ft above its height with n x 41 dt an
         integer: after exactly n steps the reference's altitude is the MVA height up to ITS OWN accumulated float64 rounding,
         and `h < mva` is decided by that rounding — every polygon of LOWW that holds the circle, several (dt, n)
      3. climbs and descents that LAND on their targets (h == target to the last bit from then on), then leave them again;
         "descend to the MVA" (the G11 case) at dt = 0.1 / 0.15 / 0.3: the aircraft reaches a target one fp32 action step above /
         on / below the MVA height after several rate-limited steps
      4. >= 3 000-step episodes of slow random-held actions at dt 0.05 / 0.1 / 0.15 (the review's 2.9e-5 altitude observation
         at step 3 698 of dt = 0.1), stepped on whatever happens; discrete actions and shaping / normalisation off at dt = 0.1"""
    rec = WideRecorder(stride=8)
    # 1. sustained descents
    rng = np.random.default_rng(5)
    for dt in G12_DTS:
        for scen in ("LOWW", "LOWW_random"):
            env = make_env(scen, dt=dt)
            random.seed(int(dt * 1000) + len(scen))
            for k in range(4 if dt < 0.1 else 8):
                a = f32([rng.uniform(-1, 1), rng.uniform(-1, -0.7), rng.uniform(-1, 1)])
                rec.run(env, np.tile(a, (int(16000 / (41 * dt)) + 50, 1)), scen, dt, True, True, False, extra_after_done=3)
    # 2. ties
    probe = make_env("LOWW")
    asp = probe._airspace
    x0, y0, x1, y1 = asp.get_bounding_box()
    heights = sorted(set(int(m.height) for m in probe._mvas))
    rng = np.random.default_rng(12000)
    spots = {}
    for _ in range(60000):
        x, y = float(np.float32(rng.uniform(x0, x1))), float(np.float32(rng.uniform(y0, y1)))
        m = mva_or_neg(asp, x, y)
        if m < 0 or len(spots.get(m, [])) >= 3:
            continue
        if disc_inside(asp, x, y, 1.25, m):
            spots.setdefault(m, []).append((x, y))
    ties = 0
    for dt, n in ((0.05, 2000), (0.1, 1000), (0.1, 3000), (0.15, 400), (0.3, 100), (0.3, 1000), (0.7, 100), (1.3, 100), (3.7, 100),
                  (0.05, 400), (0.15, 2000)):
        drop = round(n * 41 * dt * 1000) / 1000.0
        assert drop == int(drop), (dt, n, drop)
        env = make_env("LOWW", dt=dt)
        for m, pts in sorted(spots.items()):
            if m + drop > 38000:
                continue
            for j, (x, y) in enumerate(pts[:2 if n <= 1000 else 1]):
                st = (x, y, float(m + int(drop)), float((37 * j + m // 100) % 360), 100.0)
                a = f32([-1.0, -1.0, 3.0])   # 100 kt, descend at the limit, turn right for ever (target heading 720 deg)
                rec.run(env, np.tile(a, (n + 6, 1)), "LOWW", dt, True, True, False, init_state=st, extra_after_done=4)
                ties += 1
    # 3. landing on targets
    for dt in (0.05, 0.1, 0.15, 0.3, 0.7, 1.3, 3.7):
        env = make_env("LOWW_random", dt=dt)
        its = interior_states(np.random.default_rng(12500 + int(dt * 100)), env, 6, 9000.0, 14000.0)
        for k, st in enumerate(its):
            rng = np.random.default_rng(12600 + 10 * int(dt * 100) + k)
            blocks = []
            h = st[2]
            for b in range(7):
                step_ft = (15 if b % 2 == 0 else -41) * dt
                n_land = int(rng.integers(8, 30))
                tgt = h + step_ft * (n_land - rng.uniform(0.05, 0.95))   # reached inside step n_land, not on a multiple of the rate
                a_h = float(np.float32(tgt / 19000.0 - 1.0))
                blocks.append(np.tile(f32([-0.8, a_h, rng.uniform(-1, 1)]), (n_land + int(rng.integers(3, 25)), 1)))
                h = float(np.float64(np.float32(a_h)) * 19000.0 + 19000.0)
            rec.run(env, np.concatenate(blocks), "LOWW_random", dt, True, True, False, init_state=st, extra_after_done=400)
    env1 = make_env("LOWW")
    for dt in (0.1, 0.15, 0.3):
        env = make_env("LOWW", dt=dt)
        its = interior_states(np.random.default_rng(12900 + int(dt * 100)), env1, 12, 2000.0, 2001.0)
        for k, st in enumerate(its):
            m = mva_or_neg(env._airspace, st[0], st[1])
            a0 = np.float32(m / 19000.0 - 1.0)
            a = [np.nextafter(a0, np.float32(-2)), a0, np.nextafter(a0, np.float32(2)),
                 np.nextafter(np.nextafter(a0, np.float32(-2)), np.float32(-2))][k % 4]
            n_land = int(np.ceil(30.0 / (41 * dt))) + 1
            acts = np.tile(f32([2.0 * (st[4] - 100.0) / 200.0 - 1.0, a, 2.0 * st[3] / 360.0 - 1.0]), (n_land + 6, 1))
            rec.run(env, acts, "LOWW", dt, True, True, False, init_state=(st[0], st[1], float(m + 30), st[3], st[4]),
                    extra_after_done=5)
    # 4. long episodes of slow random-held actions (100 .. 150 kt: they stay inside the fixed-point position range whatever they do)
    def slow(rng, horizon, lo, hi, discrete=False):
        out = []
        while sum(len(b) for b in out) < horizon:
            hold = int(rng.integers(lo, hi))
            if discrete:
                a = np.array([float(rng.integers(0, 6)), float(rng.integers(70, 230)), float(rng.integers(0, 360))])
            else:
                a = f32([rng.uniform(-1, -0.5), rng.uniform(-0.62, 0.2), rng.uniform(-1, 1)])
            out.append(np.tile(a, (hold, 1)))
        return np.concatenate(out)[:horizon]
    for dt, horizon, n_eps in ((0.05, 3600, 4), (0.1, 3800, 8), (0.15, 2600, 6), (0.3, 1300, 4)):
        env = make_env("LOWW_random", dt=dt)
        its = interior_states(np.random.default_rng(13000 + int(dt * 100)), env, n_eps, 9000.0, 16000.0)
        for k in range(n_eps):
            rng = np.random.default_rng(13100 + 10 * int(dt * 100) + k)
            rec.run(env, slow(rng, horizon, 40, 160), "LOWW_random", dt, True, True, False, init_state=its[k],
                    extra_after_done=horizon)
    for (shaping, normalize, discrete) in ((True, True, True), (False, False, False)):
        env = make_env("LOWW_random", dt=0.1, shaping=shaping, normalize=normalize, discrete=discrete)
        its = interior_states(np.random.default_rng(13500), env, 4, 9000.0, 16000.0)
        for k in range(4):
            rng = np.random.default_rng(13600 + k + 10 * int(discrete))
            rec.run(env, slow(rng, 3200, 40, 160, discrete), "LOWW_random", 0.1, shaping, normalize, discrete, init_state=its[k],
                    extra_after_done=3200)
    rec.save(os.path.join(HERE, "g12_timesteps.npz"))
    return rec, ties, sorted(spots)


G13_DTS = (0.01, 0.033, 0.25, 0.9, 1.7, 2.5, 7.3, 13.0, 29.0, 47.0)


def gen_g13():
    """A sweep over SimParameters.timestep far from G12's values — 0.01 s (rate limits of a few hundredths of a unit per step) to 47 s
    (the largest the fixed-point position format takes is 51 s: a step of 4 nm) — with ordinary inputs: random actions held for 1 /
    5 / 20 / 100 steps, every configuration switch once (shaping off, normalisation off, discrete actions), three sectors.  The
    probe of the round-5 review ran six timesteps against the build; this is the same question asked of ten others, kept."""
    rec = WideRecorder(stride=8)
    rng = np.random.default_rng(777)
    for dt in G13_DTS:
        for scen, shaping, normalize, discrete in (("LOWW_random", True, True, False), ("LOWW", False, True, False),
                                                   ("Simple", True, False, False), ("LOWW_random", True, True, True),
                                                   ("UnitTest", True, True, False)):
            env = make_env(scen, dt=dt, shaping=shaping, normalize=normalize, discrete=discrete)
            random.seed(int(dt * 1000) + len(scen) + 7)
            n_ep = 3 if dt < 0.1 else 6
            for k in range(n_ep):
                max_steps = int(min(6500, 2500 / dt)) if dt < 1 else int(min(3000, 6000 / dt + 50))
                hold = int(rng.choice([1, 5, 20, 100]))
                acts = []
                for b in range(max_steps // hold + 1):
                    if discrete:
                        a = np.array([rng.integers(0, 20), rng.integers(0, 380), rng.integers(0, 360)], dtype=np.float64)
                    else:
                        a = f32([rng.uniform(-1, 1), rng.uniform(-1, 1) if k % 2 else rng.uniform(-1, -0.6), rng.uniform(-1, 1)])
                    acts.append(np.tile(a, (hold, 1)))
                acts = np.concatenate(acts)[:max_steps]
                it = (6000 - int(rng.integers(1, 40))) if k == n_ep - 1 else None      # the last episode runs into the time limit
                rec.run(env, acts, scen, dt, shaping, normalize, discrete, init_timesteps=it, extra_after_done=3)
    rec.save(os.path.join(HERE, "g13_timestep_sweep.npz"))
    return rec


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "mt", "g8", "g9", "g10", "g11", "g12", "g13"]
    if "g1" in which:
        gen_g1()
    if "mt" in which:
        gen_model_test()
    if "g5" in which:
        gen_g5()
    if "g2" in which:
        r = gen_g2()
        print("g2 episodes", [(e["steps"]) for e in r.ep])
    if "g3" in which:
        gen_g3()
    if "g4" in which:
        gen_g4()
    if "g6" in which:
        r = gen_g6()
        print("g6 episodes", len(r.ep), "steps", len(r.rows["reward"]))
        fl = np.asarray(r.rows["flags"])
        dn = np.asarray(r.rows["done"])
        for name, bit in (("below", 1), ("outside", 2), ("won", 4), ("timeout", 8), ("inv_v", 16), ("inv_h", 32)):
            print(name, int(((fl & bit) != 0).sum()), "terminal:", int((((fl & bit) != 0) & (dn != 0)).sum()))
    if "g7" in which:
        gen_g7()
    if "g10" in which:
        o = gen_g10()
        print("g10", {k: (v["width"], v["height"], len(v["static"]), [len(f["geoms"]) for f in v["frames"]]) for k, v in o.items()})
    if "g8" in which:
        pts, h = gen_g8()
        print("g8 points", len(pts), "heights", sorted(set(h.tolist())))
    if "g9" in which:
        r = gen_g9()
        fl, dn = np.asarray(r.flags), np.asarray(r.done)
        print("g9 episodes", len(r.ep), "steps", len(fl), "sampled rows", len(r.samp_rows))
        for name, bit in (("below", 1), ("outside", 2), ("won", 4), ("timeout", 8), ("inv_v", 16), ("inv_h", 32)):
            print(name, int(((fl & bit) != 0).sum()), "terminal:", int((((fl & bit) != 0) & (dn != 0)).sum()))
    if "g11" in which:
        r = gen_g11()
        fl, dn = np.asarray(r.flags), np.asarray(r.done)
        phi = np.asarray(r.state)[:, 3]
        print("g11 episodes", len(r.ep), "steps", len(fl), "sampled rows", len(r.samp_rows), "heading range", phi.min(), phi.max(),
              "rows beyond [-76, 436):", int(((phi < -76) | (phi >= 436)).sum()))
        for name, bit in (("below", 1), ("outside", 2), ("won", 4), ("timeout", 8), ("inv_v", 16), ("inv_h", 32)):
            print(name, int(((fl & bit) != 0).sum()), "terminal:", int((((fl & bit) != 0) & (dn != 0)).sum()))
    if "g12" in which:
        r, ties, hts = gen_g12()
        fl, dn = np.asarray(r.flags), np.asarray(r.done)
        print("g12 episodes", len(r.ep), "steps", len(fl), "sampled rows", len(r.samp_rows), "tie episodes", ties, "MVA heights with a circle", hts,
              "timesteps", sorted(set(e["dt"] for e in r.ep)))
        for name, bit in (("below", 1), ("outside", 2), ("won", 4), ("timeout", 8), ("inv_v", 16), ("inv_h", 32)):
            print(name, int(((fl & bit) != 0).sum()), "terminal:", int((((fl & bit) != 0) & (dn != 0)).sum()))
    if "g13" in which:
        r = gen_g13()
        fl, dn = np.asarray(r.flags), np.asarray(r.done)
        print("g13 episodes", len(r.ep), "steps", len(fl), "sampled rows", len(r.samp_rows), "timesteps", sorted(set(e["dt"] for e in r.ep)))
        for name, bit in (("below", 1), ("outside", 2), ("won", 4), ("timeout", 8), ("inv_v", 16), ("inv_h", 32)):
            print(name, int(((fl & bit) != 0).sum()), "terminal:", int((((fl & bit) != 0) & (dn != 0)).sum()))
    print("done")
