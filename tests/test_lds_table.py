"""LDS-resident lookup table (include/atc_step.h ABI 21, atc_hip/scenario.py:build_lds_table, csrc/atc_device.h:lds_resolve): a
second, compact form of Airspace.find_mva (model.py:282-289) that the multi-step launches of one-aircraft envs stage in LDS.  It
must give the answers of the ordered polygon scan wherever it answers at all, and send everything else to the lookup grid.
CPU: the numpy restatement of the kernel's lookup (scenario.lds_table_lookup) against the fp32 oracle.  GPU: the kernels."""
import numpy as np
import pytest

import helpers as H
from atc_hip import layout as L
from atc_hip import scenario as S
from test_random_sectors import random_sector


def _points(comp, rng, n_random, per_edge=400):
    b = comp.bbox
    pts = [np.stack([rng.uniform(b[0] - 3, b[2] + 3, n_random), rng.uniform(b[1] - 3, b[3] + 3, n_random)], 1)]
    for ring in comp.mva_rings:
        for k in range(len(ring) - 1):
            t = rng.uniform(0, 1, per_edge)[:, None]
            p = ring[k][None, :] * (1 - t) + ring[k + 1][None, :] * t
            for scale in (1, 300, 20000):   # on the border, inside a LINE record's margin band, just beyond it
                pts.append(p + rng.integers(-3, 4, p.shape) * scale * np.spacing(np.abs(p).astype(np.float32)).astype(np.float64))
            pts.append(np.repeat(ring[k][None, :], 49, 0) + np.stack(np.meshgrid(np.arange(-3, 4), np.arange(-3, 4)), -1)
                       .reshape(-1, 2) * np.spacing(np.abs(ring[k]).astype(np.float32)).astype(np.float64))
    # beyond the table, non-finite: the clamped indices name the (clean, outside) border ring
    pts.append(np.array([[np.nan, 1.0], [1e30, 5.0], [-1e30, -1e30], [np.inf, np.nan], [0.0, 0.0], [-np.inf, 40.0], [3e9, 3e9]]))
    return np.concatenate(pts).astype(np.float32)


def _heights(code, h):
    return np.where(code > 0, h, -1).astype(np.int32)


@pytest.mark.parametrize("scen", ["LOWW", "Simple", "UnitTest"])
def test_table_lookup_equals_fp32_oracle(scen):
    from oracle import oracle as O
    comp = H.compiled(scen)
    tab = comp.lds_table()
    assert tab is not None and tab.dtype == np.uint8 and len(tab) % 16 == 0
    pts = _points(comp, np.random.default_rng(3), 1500000)
    exp = O.OracleQueries(comp, np.float32).mva(pts[:, 0], pts[:, 1])
    code, h, cand, walked = S.lds_table_lookup(tab, pts[:, 0], pts[:, 1])
    ok = code >= 0
    assert np.array_equal(_heights(code, h)[ok], exp[ok])
    assert walked.sum() > 1000 and np.array_equal(_heights(code, h)[walked], exp[walked])   # (the residual sub-cells' record walks)
    # ... and it answers nearly everything: a uniformly placed aircraft inside LOWW walks records in 0.2 % of the cases (the sub-cells
    # around the vertices) and is inside a line's margin band — the wavefront asks the grid — in 0.04 %; the test sectors'
    # HORIZONTAL borders have no LINE form (the crossing test never counts such an edge, model.py:328-329: the border is the y tests
    # of its neighbours): sub-cells along them are residual over their whole length
    n = 1500000
    inside = exp[:n] >= 0
    assert inside.sum() > 100000
    assert (~ok[:n] & inside).sum() <= 0.001 * inside.sum()
    assert (walked[:n] & inside).sum() <= (0.004 if scen == "LOWW" else 0.02) * inside.sum()
    # the corridor candidate bit covers the bounds of the corridor's horizontal triangle (model.py:198)
    th = comp.corridor["tri_h"]
    inb = (pts[:, 0] >= th[:, 0].min()) & (pts[:, 0] <= th[:, 0].max()) & (pts[:, 1] >= th[:, 1].min()) & (pts[:, 1] <= th[:, 1].max())
    assert inb.sum() > 100 and np.all(cand[inb] == 1)


@pytest.mark.parametrize("seed,n_poly", [(1, 3), (2, 8), (3, 14), (4, 20), (6, 11)])
def test_table_lookup_on_random_sectors(seed, n_poly):
    """Arbitrary polygon sets — overlapping, concave, shared and axis-aligned edges — against the float64 ordered scan at points
    that are exactly representable in fp32 (so both see the same point) and away from the borders by more than the fp32 margin,
    and against the fp32 oracle everywhere."""
    from oracle import oracle as O
    mvas, runway, entries = random_sector(seed, n_poly)
    comp = S.compile_sector(mvas, runway, entries, grid_cell=None)
    tab = comp.lds_table()
    assert tab is not None
    pts = _points(comp, np.random.default_rng(100 + seed), 300000, per_edge=60)
    exp = O.OracleQueries(comp, np.float32).mva(pts[:, 0], pts[:, 1])
    code, h, _, walked = S.lds_table_lookup(tab, pts[:, 0], pts[:, 1])
    ok = code >= 0
    assert ok.mean() > 0.9 and walked.sum() > 100
    assert np.array_equal(_heights(code, h)[ok], exp[ok])


def test_table_structure():
    comp = H.compiled("LOWW")
    tab = comp.lds_table()
    assert tab is comp.lds_table()      # built once
    hdr = tab[:4 * L.LDS_HDR_WORDS].view(np.uint32)
    assert hdr[L.LDS_H_MAGIC] == L.LDS_MAGIC and hdr[L.LDS_H_BYTES] == len(tab) and hdr[L.LDS_H_SUB] == 8
    nx, ny, off_l1, off_sub, n_sub, off_line, n_line = (int(hdr[k]) for k in (L.LDS_H_NX, L.LDS_H_NY, L.LDS_H_OFF_L1, L.LDS_H_OFF_SUB,
                                                                               L.LDS_H_N_SUB, L.LDS_H_OFF_LINE, L.LDS_H_N_LINE))
    off_resid, n_resid, lds_part, off_pool, n_rec = (int(hdr[k]) for k in (L.LDS_H_OFF_RESID, L.LDS_H_N_RESID, L.LDS_H_LDS_BYTES,
                                                                           L.LDS_H_OFF_POOL, L.LDS_H_N_REC))
    assert lds_part + 10240 <= 160 * 1024 and off_pool >= lds_part and off_pool + 32 * n_rec <= len(tab)
    ww = tab[off_resid:off_resid + 4 * n_resid].view(np.uint32)
    assert np.all((ww >> 24) >= 1) and np.all((ww & 0xffffff) + (ww >> 24) <= n_rec)
    l1 = tab[off_l1:off_l1 + 2 * nx * ny].view(np.uint16).reshape(ny, nx)
    assert not l1[0].any() and not l1[-1].any() and not l1[:, 0].any() and not l1[:, -1].any()
    k1, p1 = (l1 >> 13) & 3, l1 & 0x1fff
    assert not (k1 == L.LDS_RESID).any() and np.all(p1[k1 == L.LDS_SUB] < n_sub) and np.all(p1[k1 == L.LDS_LINE] < n_line)
    assert sorted(p1[k1 == L.LDS_SUB].tolist()) == list(range(n_sub))
    l2 = tab[off_sub:off_sub + 2 * 64 * n_sub].view(np.uint16)
    k2, p2 = (l2 >> 13) & 3, l2 & 0x1fff
    assert not (k2 == L.LDS_SUB).any() and not (l2 & 0x8000).any() and np.all(p2[k2 == L.LDS_LINE] < n_line)
    assert sorted(p2[k2 == L.LDS_RESID].tolist()) == list(range(n_resid))
    assert np.all(p1[k1 == L.LDS_CLEAN] <= comp.n_mva) and np.all(p2[k2 == L.LDS_CLEAN] <= comp.n_mva)
    # sectors the codes cannot describe get no table: noise-abatement areas (no candidate masks)
    from envs.atc import scenarios
    assert scenarios.compile_scenario(scenarios.LOWWDense()).lds_table() is None


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("scen", ["LOWW", "Simple", "UnitTest"])
def test_device_table_lookup_equals_oracle(scen):
    from atc_hip import lib
    from oracle import oracle as O
    comp = H.compiled(scen, grid_cell=0.125)
    sec = lib.Scenario(comp, lds_table=True)
    assert sec.has_lds_table
    pts = _points(comp, np.random.default_rng(5), 1500000)
    exp = O.OracleQueries(comp, np.float32).mva(pts[:, 0], pts[:, 1])
    got, src = sec.query_mva_lds(pts[:, 0], pts[:, 1])
    assert np.array_equal(got, exp)
    # the table itself answered (a wavefront of 64 consecutive points goes to the grid as a whole when one of them is residual)
    code, _, _, _ = S.lds_table_lookup(comp.lds_table(), pts[:, 0], pts[:, 1])
    n = 1500000 // 64 * 64
    assert src[:n].mean() > 0.9
    assert np.array_equal(src[:n].reshape(-1, 64).all(axis=1), (code[:n] >= 0).reshape(-1, 64).all(axis=1)), \
        "the kernel and its numpy restatement agree on which wavefronts the table answers"
    sec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_poly", [(11, 5), (12, 16), (13, 22)])
def test_device_table_lookup_on_random_sectors(seed, n_poly):
    from atc_hip import lib
    from oracle import oracle as O
    mvas, runway, entries = random_sector(seed, n_poly)
    comp = S.compile_sector(mvas, runway, entries, grid_cell=0.5)
    sec = lib.Scenario(comp, lds_table=True)
    staged = int(comp.lds_table()[:4 * L.LDS_HDR_WORDS].view(np.uint32)[L.LDS_H_LDS_BYTES])
    if staged + 10240 > 160 * 1024:   # (22 overlapping polygons: more refined cells than one CU's LDS holds — such a sector steps from the grid)
        assert not sec.has_lds_table
        sec.close()
        return
    assert sec.has_lds_table
    pts = _points(comp, np.random.default_rng(200 + seed), 400000, per_edge=60)
    exp = O.OracleQueries(comp, np.float32).mva(pts[:, 0], pts[:, 1])
    got, src = sec.query_mva_lds(pts[:, 0], pts[:, 1])
    assert np.array_equal(got, exp) and src.mean() > 0.5
    sec.close()


@pytest.mark.gpu
def test_attach_refuses_malformed_tables():
    import ctypes as C
    from atc_hip import lib
    from envs.atc import scenarios
    comp = H.compiled("LOWW", grid_cell=0.25)
    sec = lib.Scenario(comp)
    good = np.array(comp.lds_table(), copy=True)
    raw = lib.load()

    def attach(t):
        t = np.ascontiguousarray(t)
        return raw.atc_scenario_attach_lds_table(sec.handle, t.ctypes.data_as(C.c_void_p), t.nbytes)
    hdr = lambda t: t[:4 * L.LDS_HDR_WORDS].view(np.uint32)   # noqa: E731
    for mutate in (lambda t: hdr(t).__setitem__(L.LDS_H_MAGIC, 1), lambda t: hdr(t).__setitem__(L.LDS_H_BYTES, len(t) - 16),
                   lambda t: hdr(t).__setitem__(L.LDS_H_NX, 4000), lambda t: hdr(t).__setitem__(L.LDS_H_N_SUB, 1),
                   lambda t: hdr(t).__setitem__(L.LDS_H_OFF_LINE, int(hdr(t)[L.LDS_H_OFF_LINE]) + 8),
                   lambda t: hdr(t).__setitem__(L.LDS_H_SUB, 4)):
        bad = good.copy()
        mutate(bad)
        assert attach(bad) == -1
    off_l1, nx = int(hdr(good)[L.LDS_H_OFF_L1]), int(hdr(good)[L.LDS_H_NX])
    for code in ((L.LDS_LINE << 13) | 0x1fff, (L.LDS_SUB << 13) | 0x1fff, (L.LDS_RESID << 13), 64):   # (RESID: sub-cells only)
        bad = good.copy()
        bad[off_l1 + 2 * (5 * nx + 5):off_l1 + 2 * (5 * nx + 5) + 2].view(np.uint16)[0] = code
        assert attach(bad) == -1
    off_resid, n_rec = int(hdr(good)[L.LDS_H_OFF_RESID]), int(hdr(good)[L.LDS_H_N_REC])
    for word in (0, n_rec | (1 << 24), (n_rec - 1) | (2 << 24), 64 << 24):       # walk words: 1 <= n < 64 records inside the pool
        bad = good.copy()
        bad[off_resid:off_resid + 4].view(np.uint32)[0] = word
        assert attach(bad) == -1
    bad = good.copy()
    bad[off_l1:off_l1 + 2].view(np.uint16)[0] = 1      # the border ring must stay clean and outside
    assert attach(bad) == -1
    assert attach(good[:-16]) == -1
    assert attach(good) == 0
    assert raw.atc_scenario_attach_lds_table(sec.handle, None, 0) == 0      # detach
    assert raw.atc_query_mva_lds(sec.handle, 1, None, None, None, None, None) == -1
    sec.close()
    # no lookup grid behind it / noise-abatement areas: refused
    nogrid = lib.Scenario(H.compiled("LOWW"))
    t = np.ascontiguousarray(good)
    assert raw.atc_scenario_attach_lds_table(nogrid.handle, t.ctypes.data_as(C.c_void_p), t.nbytes) == -1
    nogrid.close()
    dense = lib.Scenario(scenarios.compile_scenario(scenarios.LOWWDense(), grid_cell=0.25))
    assert raw.atc_scenario_attach_lds_table(dense.handle, t.ctypes.data_as(C.c_void_p), t.nbytes) == -1
    assert not dense.attach_lds_table()
    dense.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,scen,dt", [(256, "LOWW", 1.0), (4096, "LOWW", 5.0), (65536, "LOWW", 1.0), (1024, "Simple", 2.0)])
def test_rollouts_identical_with_and_without_the_table(B, scen, dt):
    """The multi-step launch of one-aircraft envs with the table staged in LDS against the same launch from the lookup grid: every
    output and the whole state bit for bit, over launches that cross MVA floors, leave the airspace, win and restart."""
    import torch
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model
    scn = H.make_scenario(scen)
    T, launches = 20, 12
    g = torch.Generator(device="cpu").manual_seed(B + 17)
    blocks = [(torch.rand((2, B, 1, 3), generator=g) * 2 - 1) for _ in range(launches)]
    for blk in blocks:
        blk[:, ::3, :, 1] = -0.9          # a third of the envs descends towards the MVAs

    def run(lds):
        env = AtcVecEnv(B, 1, sim_parameters=model.SimParameters(dt), scenario=scn, auto_reset=True, seed=5, lds_table=lds)
        assert env.sector.has_lds_table == lds
        outs = []
        for blk in blocks:
            o = env.rollout(blk.cuda(), hold=T // 2)
            outs.append({k: v.clone() for k, v in o.items()})
        state = (env.ac.clone(), env.alt.clone(), env.last_act.clone(), env.env.clone(), env.stats.clone())
        env.close()
        return outs, state
    (a, sa), (b, sb) = run(True), run(False)
    for x, y in zip(sa, sb):
        assert torch.equal(x, y)
    for oa, ob in zip(a, b):
        for k in ("obs", "reward", "done", "flags"):
            assert torch.equal(oa[k], ob[k]), k
    flags = torch.cat([o["flags"].reshape(-1) for o in a]).to(torch.int32) & 0xffff
    assert int(((flags & L.F_OUTSIDE) != 0).sum()) > 0
    if scen == "LOWW":   # (the test sector's aircraft have left it before a descent reaches a floor)
        assert int(((flags & L.F_BELOW_MVA) != 0).sum()) > 0
    assert int(sa[4][:, L.STAT_EPISODES].sum()) > B // 8


def _scenario_of(mvas, runway, entries):
    from envs.atc import model, scenarios
    s = scenarios.Scenario()
    s.mvas = [model.MinimumVectoringAltitude([tuple(map(float, p)) for p in ring], int(h)) for ring, h in mvas]
    s.runway = model.Runway(*runway)
    s.airspace = model.Airspace(s.mvas, s.runway)
    s.entrypoints = [model.EntryPoint(x, y, phi, list(levels)) for x, y, phi, levels in entries]
    s.noise_areas = []
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_poly,dt", [(21, 6, 1.0), (22, 12, 2.0), (23, 9, 0.7)])
def test_table_rollouts_on_random_sectors_against_the_oracle(seed, n_poly, dt):
    """The LDSG launch on arbitrary polygon sets (overlaps, shared and axis-aligned borders, horizontal edges: residual walks and
    margin-band fall-backs in most wavefronts) against the fp32 oracle stepped once per step: flags / done exact, obs / reward
    within 1e-5, state and counters identical afterwards."""
    import torch
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model, scenarios
    from oracle import oracle as O
    mvas, runway, _ = random_sector(seed, n_poly)
    rng = np.random.default_rng(seed)
    entries = [(float(x), float(y), float(rng.integers(0, 360)), [60, 90, 120]) for x, y in rng.uniform(15, 45, (6, 2))]
    scn = _scenario_of(mvas, runway, entries)
    B, T, hold = 512, 20, 10
    env = AtcVecEnv(B, 1, sim_parameters=model.SimParameters(dt), scenario=scn, auto_reset=True, seed=3, grid_cell=0.25, spawn="random")
    if not env.sector.has_lds_table:
        pytest.skip("this sector's table does not fit one CU's LDS")
    orc = O.OracleEnv(scenarios.compile_scenario(scn, grid_cell=0.25), B, 1,
                      O.make_params(dt=dt, auto_reset=True, seed=3, random_entry=True), np.float32)
    g = torch.Generator(device="cpu").manual_seed(seed)
    events = 0
    for j in range(10):
        blocks = torch.rand((T // hold, B, 1, 3), generator=g) * 2 - 1
        blocks[:, ::2, :, 1] = -0.95      # half of the envs descend into the floors
        out = env.rollout(blocks, hold=hold)
        for t in range(T):
            orc.step(blocks[t // hold].numpy())
            fl = out["flags"][t].cpu().numpy().astype(np.uint16).reshape(B, 1)
            assert np.array_equal(fl, orc.flags), (j, t)
            assert np.array_equal(out["done"][t].cpu().numpy(), orc.done), (j, t)
            o = out["obs"][t].cpu().numpy().reshape(B, 1, 10)
            assert np.all(np.abs(o - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs))), (j, t)
            r = out["reward"][t].cpu().numpy()
            assert np.all(np.abs(r - orc.reward) <= 1e-5 * np.maximum(1.0, np.abs(orc.reward))), (j, t)
            events += int((fl & (L.F_BELOW_MVA | L.F_OUTSIDE)).astype(bool).sum())
    assert events > B // 4
    assert np.array_equal(env.ac[:, 0].cpu().numpy(), orc.px) and np.array_equal(env.ac[:, 1].cpu().numpy(), orc.py)
    assert np.array_equal(env.alt.cpu().numpy(), np.asarray(orc.h, dtype=np.float64).ravel())
    assert np.array_equal(env.actions_taken.cpu().numpy(), orc.actions_taken)
    env.close()
