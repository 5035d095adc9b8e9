"""GPU test of the stable-baselines-shaped adapter: AtcSBVecEnv must behave like a list of the reference-surface AtcGym
envs driven the way SubprocVecEnv + Monitor drive them (step, auto-reset returning the reset observation, episode
statistics)."""
import numpy as np
import pytest

import helpers as H  # noqa: F401

pytestmark = pytest.mark.gpu


def test_adapter_equals_loop_of_single_envs():
    import random
    from atc_hip.sb_adapter import AtcSBVecEnv
    from envs.atc import atc_gym
    B = 6
    venv = AtcSBVecEnv(B)                      # default LOWW, one entry point -> resets are deterministic
    singles = [atc_gym.AtcGym() for _ in range(B)]
    obs = venv.reset()
    sobs = np.stack([e.reset() for e in singles])
    assert np.allclose(obs, sobs, atol=1e-5)
    assert venv.observation_space.shape == (10,) and venv.action_space.shape == (3,)
    rng = np.random.default_rng(0)
    ep_returns = [0.0] * B
    ep_lens = [0] * B
    finished = 0
    for t in range(700):
        if t % 25 == 0:
            a = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
            a[:, 1] = -1.0 + 0.1 * a[:, 1]      # low altitude targets: episodes end within a few hundred steps
        o, r, d, infos = venv.step(a)
        for b, e in enumerate(singles):
            so, sr, sd, si = e.step(a[b])
            ep_returns[b] += sr
            ep_lens[b] += 1
            assert bool(d[b]) == sd
            assert abs(r[b] - sr) <= 1e-5 * max(1.0, abs(sr))
            assert np.allclose(infos[b]["original_state"], si["original_state"], rtol=1e-6, atol=1e-4)
            if sd:
                finished += 1
                assert np.allclose(infos[b]["terminal_observation"], so, atol=1e-5)
                assert infos[b]["episode"]["l"] == ep_lens[b]
                assert abs(infos[b]["episode"]["r"] - ep_returns[b]) <= 1e-5 * max(1.0, abs(ep_returns[b]))
                so = e.reset()                  # what a SubprocVecEnv worker does
                ep_returns[b], ep_lens[b] = 0.0, 0
            else:
                assert "episode" not in infos[b]
            assert np.allclose(o[b], so, atol=1e-5)
    assert finished >= B
    assert np.allclose(venv.get_attr("actions_per_timestep"), [e.actions_per_timestep for e in singles], atol=1e-12)
    assert venv.get_attr("timesteps") == [e.timesteps for e in singles]
    # winning_ratio: the batched ring counts won episodes among the last 10; no wins happen in this scenario
    assert venv.get_attr("winning_ratio") == [0.0] * B == [e.winning_ratio for e in singles]
    venv.set_attr("timesteps", 5995, indices=[2])
    assert venv.get_attr("timesteps", indices=2) == [5995]
    first = venv.env_method("reset", indices=[1, 4])
    assert len(first) == 2 and first[0][2] == 15000.0
    venv.close()
    for e in singles:
        e.close()


def test_adapter_against_reference_episodes_with_wins():
    """AtcSBVecEnv against the REFERENCE (g9 fixture), not against another HIP env: 24 recorded episodes — 16 that end in the
    approach corridor, 8 that end outside / below the MVA — run side by side through the adapter's numpy surface.  Every
    step's observation / reward / done, Monitor's episode record on done, the terminal observation, the raw reset
    observation returned in its place (SB semantics + quirk Q1), and winning_ratio afterwards (atc_gym.py:359-363)."""
    from atc_hip.sb_adapter import AtcSBVecEnv
    fx = H.WideFixture()
    std = [ep for ep in fx.episodes if ep["scen"] == "LOWW" and ep["dt"] == 1.0 and ep["shaping"] and ep["normalize"]
           and not ep["discrete"] and int(fx.done[ep["start"]:ep["start"] + ep["steps"]].sum()) == 1 and 2 <= ep["steps"] <= 700]
    last = lambda ep: ep["start"] + ep["steps"] - 1   # noqa: E731
    wins = [ep for ep in std if fx.flags[last(ep)] & H.F_WON and not fx.flags[last(ep)] & H.F_TIMEOUT][:16]
    lost = [ep for ep in std if not fx.flags[last(ep)] & H.F_WON][:8]
    assert len(wins) == 16 and len(lost) == 8
    eps = wins + lost
    B = len(eps)
    venv = AtcSBVecEnv(B)
    venv.reset()
    for b, ep in enumerate(eps):
        venv.vec.set_state(b, 0, *ep["init_state"])
        venv.set_attr("timesteps", ep["init_timesteps"], indices=[b])
        venv.vec.set_last_action(b, 0, ep["init_last_action"])
    steps = np.array([ep["steps"] for ep in eps])
    starts = np.array([ep["start"] for ep in eps])
    finished = np.zeros(B, bool)
    for t in range(int(steps.max())):
        live = (t < steps) & ~finished
        rows = np.where(live, starts + t, starts)
        obs, rew, done, infos = venv.step(fx.action[rows].astype(np.float32))
        for b in np.nonzero(live)[0]:
            row = rows[b]
            gw = fx.reward[row]
            assert bool(done[b]) == bool(fx.done[row]), (b, t)
            assert abs(rew[b] - gw) <= (1e-5 + 1e-7) * max(1.0, abs(gw)), (b, t, rew[b], gw)      # 1e-5 + float32 storage of the fixture (6e-8)
            si = fx.samp_index[row]
            if done[b]:
                ep = eps[b]
                assert si >= 0 and np.all(np.abs(infos[b]["terminal_observation"] - fx.obs[si]) <= 1e-5), (b, t)
                assert infos[b]["episode"]["l"] == ep["init_timesteps"] + ep["steps"]     # Monitor 'l' = env.timesteps
                gt = ep["total_reward"]                                                  # Monitor 'r' = sum of rewards
                assert abs(infos[b]["episode"]["r"] - gt) <= (1e-5 + 6e-8 * ep["steps"]) * max(1.0, abs(gt)), (b, gt)
                assert list(obs[b][:5]) == [10.0, 51.0, 15000.0, 90.0, 250.0]            # raw reset observation
                finished[b] = True
            elif si >= 0:
                assert np.all(np.abs(obs[b] - fx.obs[si]) <= 1e-5), (b, t)
    assert finished.all()
    wr = venv.get_attr("winning_ratio")
    assert wr[:16] == [0.1] * 16 and wr[16:] == [0.0] * 8
    assert min(venv.get_attr("episodes")) >= 2
    venv.close()


def test_sparse_infos_for_large_batches():
    from atc_hip.sb_adapter import AtcSBVecEnv
    venv = AtcSBVecEnv(1024)
    assert venv.sparse_infos
    venv.reset()
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, (1024, 3)).astype(np.float32)
    a[:, 1] = -0.98
    seen = 0
    for t in range(320):
        o, r, d, infos = venv.step(a)
        assert len(infos) == 1024 and venv.original_state.shape == (1024, 10)
        for b in np.nonzero(d)[0]:
            assert infos[b]["episode"]["l"] > 0 and infos[b]["terminal_observation"].shape == (10,)
            seen += 1
        assert all(("episode" in infos[b]) == bool(d[b]) for b in range(0, 1024, 37))
    assert seen > 100
    venv.close()


def test_render_frames_equal_reference_geometry():
    """G10 on the GPU: AtcGym flies the scripted episode the reference flew; at steps 0 / 20 / 40 / 60 the frame geometry the
    renderer builds from the DEVICE state (aircraft symbol, label anchors and texts, history dots, reward labels) equals what
    the reference drew, and `render('rgb_array')` returns a frame of the reference's window size."""
    from test_sector_io_render import _same_geoms
    from atc_hip import render
    from envs.atc import atc_gym
    for scen in ("LOWW", "Simple"):
        g = H.golden_json("g10_render_geometry.json")[scen]
        env = atc_gym.AtcGym(scenario=H.make_scenario(scen))
        env.reset()
        ap = env._airplane
        ap.x, ap.y, ap.h, ap.phi, ap.v = g["init_state"]
        frames = {f["step"]: f for f in g["frames"]}

        def check(step):
            f = frames[step]
            got = render.frame_scene(env._vec.compiled, [{"x": ap.x, "y": ap.y, "h": ap.h, "v": ap.v, "name": ap.name,
                                                          "history": ap.position_history}], env.total_reward, env.last_reward)
            _same_geoms(got, f["geoms"], 2e-3)            # fp32 positions: 1e-4 nm x 9.4 px / nm
            img = env.render(mode='rgb_array')
            assert img.shape == (g["height"], g["width"], 3) and img.dtype == np.uint8
            return img

        img0 = check(0)
        for t, a in enumerate(g["actions"]):
            env.step(np.asarray(a))
            if (t + 1) in frames:
                img = check(t + 1)
        assert (img != img0).any() and env.render(mode='human') is None
        env.close()


def test_adapter_is_a_vecenv_and_runs_the_runner_loop():
    """Row f1's purpose: the reference's trainer hands its vector env straight to PPO2 (learning/atc-gym-stable-baselines.py:
    76-90), and stable-baselines 2.8 wraps anything that is not a `VecEnv` INSTANCE in DummyVecEnv.  With the library importable
    (here: the interface-only stand-in of tests/sb_shim) AtcSBVecEnv must be a VecEnv subclass with the base's three attributes,
    every abstract method implemented with SB's signatures, and survive the exact call sequence of PPO2's Runner.run plus the
    reference's TensorBoard callback (learning/atc-gym-stable-baselines.py:31-49)."""
    import importlib
    import inspect
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    added = [os.path.join(here, "sb_shim"), os.path.join(here, "oracle_shims")]
    had_gym = "gym" in sys.modules
    sys.path[:0] = added
    try:
        vmod = importlib.import_module("stable_baselines.common.vec_env")
        import atc_hip.sb_adapter as sba
        sba = importlib.reload(sba)
        assert issubclass(sba.AtcSBVecEnv, vmod.VecEnv)
        assert not getattr(sba.AtcSBVecEnv, "__abstractmethods__", None)        # every abstract method is implemented
        for name in ("env_method", "get_attr", "set_attr", "step_async", "reset"):
            want = list(inspect.signature(getattr(vmod.VecEnv, name)).parameters)
            got = list(inspect.signature(getattr(sba.AtcSBVecEnv, name)).parameters)
            assert got == want, (name, got, want)
        n_envs, n_steps = 8, 128                                                  # the reference: 8 workers, PPO2's default n_steps
        env = sba.AtcSBVecEnv(n_envs)
        assert vmod.wrap_like_base_rl_model(env) is env                           # BaseRLModel keeps it as it is
        assert isinstance(env, vmod.VecEnv) and env.num_envs == n_envs and env.unwrapped is env
        assert type(env.observation_space).__module__.startswith("gym") and env.observation_space.shape == (10,)
        assert env.seed(7) == [[7 + i] for i in range(n_envs)]
        assert env.env_is_wrapped(object) == [False] * n_envs
        # AbstractEnvRunner.__init__ + Runner.run (ppo2.py): obs buffer from reset(), then n_steps x (policy step, clip to the
        # Box, env.step, collect info.get('episode'))
        obs = np.zeros((n_envs,) + env.observation_space.shape, dtype=np.float32)
        obs[:] = env.reset()
        assert obs[0, 2] == 15000.0                                               # the raw reset observation (quirk Q1)
        rng = np.random.default_rng(0)
        dones = [False] * n_envs
        ep_infos, mb = [], []
        for update in range(3):
            for _ in range(n_steps):
                actions = rng.normal(0.0, 0.8, (n_envs, 3)).astype(np.float32)    # an un-squashed Gaussian policy
                actions[:, 1] = np.minimum(actions[:, 1], -0.8)                   # descend: episodes end inside the loop
                clipped = np.clip(actions, env.action_space.low, env.action_space.high)
                obs[:], rewards, dones, infos = env.step(clipped)
                assert rewards.shape == (n_envs,) and len(infos) == n_envs and np.asarray(dones).shape == (n_envs,)
                for info in infos:
                    maybe = info.get('episode')
                    if maybe:
                        ep_infos.append(maybe)
                mb.append(rewards)
            # the reference's callback once per update (learning/atc-gym-stable-baselines.py:31-49)
            apt = env.get_attr("actions_per_timestep")
            wr = env.get_attr("winning_ratio")
            assert len(apt) == len(wr) == n_envs and all(0.0 <= w <= 1.0 for w in wr) and all(a >= 0.0 for a in apt)
        assert len(ep_infos) >= n_envs and all(set(e) == {"r", "l", "t"} and e["l"] > 0 for e in ep_infos)
        assert np.isfinite(np.asarray(mb)).all()
        assert env.env_method("seed", 3, indices=[0, 1]) == [3, 3]
        assert len(env.get_images()) == n_envs
        env.close()
    finally:
        for p in added:
            sys.path.remove(p)
        for m in [m for m in sys.modules if m == "stable_baselines" or m.startswith("stable_baselines.")]:
            del sys.modules[m]
        if not had_gym:
            for m in [m for m in sys.modules if m == "gym" or m.startswith("gym.")]:
                del sys.modules[m]
        import atc_hip.sb_adapter as sba2
        importlib.reload(sba2)
