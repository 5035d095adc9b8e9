"""Oracle-INDEPENDENT property of the multi-aircraft extension (SURVEY 8a-ext has no reference code; every other N > 1 check is
HIP-vs-own-oracle): the aircraft of an env are a SET.  Permuting the slots of every env — state, last-action records and actions
alike — must permute the per-aircraft outputs and leave done and the minimum separation bit-identical, the env reward equal up
to the rounding of an fp32 sum taken in another order.  It checks the DPP hand-back / xor pairing / LDS rotation of the
separation scan and the group reductions against THEMSELVES under a symmetry the shared oracle cannot vouch for."""
import numpy as np
import pytest

import helpers as H  # noqa: F401

pytestmark = pytest.mark.gpu


def _pair(B, N, seed, **kw):
    import torch
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=True)
    mk = lambda: AtcVecEnv(B, N, scenario=scn, auto_reset=False, spawn="lattice", grid_cell=0.5, **kw)   # noqa: E731
    a, b = mk(), mk()
    g = torch.Generator(device="cpu").manual_seed(seed)
    # aircraft scattered over a 16 x 16 nm patch inside the sector at 8 flight levels 500 ft apart: a few pairs per env lose
    # their separation at once or during the run, most do not
    x = 30.0 + 16.0 * torch.rand(B * N, generator=g)
    y = 40.0 + 16.0 * torch.rand(B * N, generator=g)
    lvl = torch.randint(0, 8, (B * N,), generator=g)
    for i in range(B * N):
        a.set_state(i // N, i % N, float(x[i]), float(y[i]), 9000.0 + 500.0 * float(lvl[i]), float(torch.randint(0, 360, (1,), generator=g)),
                    float(torch.randint(150, 290, (1,), generator=g)))
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])        # b's slot j holds a's aircraft perm[e, j]
    idx = (torch.arange(B)[:, None] * N + perm).reshape(-1).to(a.device)
    for name in ("ac", "alt", "last_act", "phi_wide"):
        getattr(b, name).copy_(getattr(a, name)[idx])
    return a, b, perm.to(a.device), idx, g


@pytest.mark.parametrize("N", [2, 5, 8, 16, 33, 64])
def test_slot_permutation_single_steps(N):
    import torch
    B = {2: 96, 5: 64, 8: 64, 16: 48, 33: 24, 64: 16}[N]
    a, b, perm, idx, g = _pair(B, N, 100 + N, want_ac_reward=True, want_min_sep=True, want_raw_obs=True)
    n_conf = 0
    for t in range(40):
        if t % 5 == 0:
            act = (torch.rand((B, N, 3), generator=g) * 2.2 - 1.1).to(a.device)   # now and then a refused target as well
            act_b = act.reshape(B * N, 3)[idx].reshape(B, N, 3).contiguous()
        oa, ra, da, ia = a.step(act)
        ob, rb, db, ib = b.step(act_b)
        assert torch.equal(ob.reshape(B * N, 10), oa.reshape(B * N, 10)[idx]), t
        assert torch.equal(ib["flags"].reshape(-1), ia["flags"].reshape(-1)[idx]), t
        assert torch.equal(ib["aircraft_reward"].reshape(-1), ia["aircraft_reward"].reshape(-1)[idx]), t
        assert torch.equal(ib["original_state"].reshape(B * N, 10), ia["original_state"].reshape(B * N, 10)[idx]), t
        assert torch.equal(db, da) and torch.equal(ib["min_separation"], ia["min_separation"]), t
        tol = 1e-6 * ia["aircraft_reward"].abs().sum(1) + 1e-6
        assert bool(((rb - ra).abs() <= tol).all()), t
        n_conf += int((ia["flags"].to(torch.int32) & H.F_CONFLICT).ne(0).any(1).sum())
    for name in ("ac", "alt", "last_act"):
        assert torch.equal(getattr(b, name), getattr(a, name)[idx]), name
    assert torch.equal(a.actions_taken, b.actions_taken) and torch.equal(a.timesteps, b.timesteps)
    assert n_conf > 0
    a.close()
    b.close()


@pytest.mark.parametrize("N,full", [(5, False), (16, False), (16, True), (33, False), (64, False), (64, True)])
def test_slot_permutation_fused(N, full):
    import torch
    B = {5: 64, 16: 48, 33: 24, 64: 16}[N]
    a, b, perm, idx, g = _pair(B, N, 200 + N)
    T, hold = 20, 5
    for launch in range(3):
        act = (torch.rand((T // hold, B, N, 3), generator=g) * 2.2 - 1.1).to(a.device)
        act_b = act.reshape(T // hold, B * N, 3)[:, idx].reshape(T // hold, B, N, 3).contiguous()
        outs = []
        for env, ac in ((a, act), (b, act_b)):
            out = None
            if full:
                out = {"obs": torch.empty((T, B, N * 10), device=env.device), "reward": torch.empty((T, B), device=env.device),
                       "done": torch.empty((T, B), dtype=torch.uint8, device=env.device),
                       "flags": torch.empty((T, B, N), dtype=torch.int16, device=env.device),
                       "ac_reward": torch.empty((T, B, N), device=env.device), "min_sep": torch.empty((T, B), device=env.device)}
            outs.append(env.rollout(ac, out=out, hold=hold))
        ua, ub = outs
        assert torch.equal(ub["obs"].reshape(T, B * N, 10), ua["obs"].reshape(T, B * N, 10)[:, idx]), launch
        assert torch.equal(ub["flags"].reshape(T, -1), ua["flags"].reshape(T, -1)[:, idx]), launch
        assert torch.equal(ub["done"], ua["done"]), launch
        assert bool(((ub["reward"] - ua["reward"]).abs() <= 1e-5 * ua["reward"].abs().clamp(min=1.0)).all()), launch
        if full:
            assert torch.equal(ub["ac_reward"].reshape(T, -1), ua["ac_reward"].reshape(T, -1)[:, idx]), launch
            assert torch.equal(ub["min_sep"], ua["min_sep"]), launch
    for name in ("ac", "alt", "last_act"):
        assert torch.equal(getattr(b, name), getattr(a, name)[idx]), name
    a.close()
    b.close()


@pytest.mark.parametrize("N", [8, 16, 32, 64])
def test_conflict_flags_against_a_brute_force_over_the_state(N):
    """A second oracle-independent check of the separation scans — and of the scan HORIZON of the multi-step launches, which leaves
    scans out: the rule itself (README.md:51: 3 nm AND 1 000 ft between two aircraft under control) evaluated in numpy over the state
    a launch leaves behind.  Envs of scattered aircraft (nobody is handed over, episodes do not restart: the traffic keeps crossing)
    fly launches of 1 .. 15 steps, so that the last step of a launch — the one whose positions can be read back — falls on every
    phase of a horizon; its CONFLICT flags must be exactly the brute force's, for every aircraft whose closest call is not within
    rounding of a minimum."""
    import torch
    from atc_hip import layout as L
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    B = {8: 64, 16: 48, 32: 32, 64: 24}[N]
    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=True)
    a = AtcVecEnv(B, N, scenario=scn, auto_reset=False, spawn="lattice", keep_active=True)
    g = torch.Generator(device="cpu").manual_seed(77 + N)
    # scattered over 50 x 65 nm and twelve flight levels 1 000 ft apart: a pair in a hundred is near at any time — most partner batches
    # of a 64-aircraft env are clear when a horizon begins, a few are not
    rnd = lambda lo, hi: lo + (hi - lo) * torch.rand(B * N, generator=g)   # noqa: E731
    x, y, lvl, phi, v = rnd(10, 60), rnd(10, 75), torch.randint(0, 12, (B * N,), generator=g), rnd(0, 360), rnd(150, 290)
    for i in range(B * N):
        a.set_state(i // N, i % N, float(x[i]), float(y[i]), 9000.0 + 1000.0 * float(lvl[i]), float(phi[i]), float(v[i]))
    seen = {"conflict": 0, "clear": 0, "launches": 0}
    for launch in range(45):
        T = 1 + launch % 15
        acts = (torch.rand((1, B, N, 3), generator=g) * 2 - 1).to(a.device).expand(T, B, N, 3).contiguous()
        out = a.rollout(acts, hold=1)
        fl = out["flags"][T - 1].cpu().numpy().reshape(B, N)
        x, y, h = (t.cpu().numpy().astype(np.float64).reshape(B, N) for t in (a.x, a.y, a.h))
        d = np.hypot(x[:, :, None] - x[:, None, :], y[:, :, None] - y[:, None, :])
        dh = np.abs(h[:, :, None] - h[:, None, :])
        eye = np.eye(N, dtype=bool)[None]
        conflict = ((d < 3.0) & (dh < 1000.0) & ~eye).any(axis=2)
        # aircraft with a partner within rounding of a minimum (fp32 positions / squared distance against float64 here) stay out
        edge = ((np.abs(d - 3.0) < 1e-3) & (dh < 1000.5) | (np.abs(dh - 1000.0) < 0.05) & (d < 3.001)) & ~eye
        sure = ~edge.any(axis=2)
        got = (fl & L.F_CONFLICT) != 0
        assert np.array_equal(got[sure], conflict[sure]), (launch, T, np.argwhere((got != conflict) & sure)[:5])
        seen["conflict"] += int(conflict[sure].sum())
        seen["clear"] += int((~conflict[sure]).sum())
        seen["launches"] += 1
    assert seen["conflict"] > 200 and seen["clear"] > 5 * seen["conflict"] and seen["launches"] == 45, seen
    a.close()
