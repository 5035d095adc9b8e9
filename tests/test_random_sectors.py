"""Randomised sectors: the edge-list lookup grid must equal the reference's ordered polygon scan for ARBITRARY polygon
sets — overlapping, concave, sharing edges, with axis-aligned edges, tiny and huge cells.  Host emulation here; the device
walk is compared with the fp32 oracle on the same sectors in test_hip_parity-style GPU tests below."""
import numpy as np
import pytest

import helpers as H  # noqa: F401
from atc_hip import layout as L
from atc_hip import scenario as S
from test_scenario_compile import _walk_grid


def random_sector(seed, n_poly):
    rng = np.random.default_rng(seed)
    mvas = []
    for p in range(n_poly):
        cx, cy = rng.uniform(10, 50, 2)
        n = int(rng.integers(3, 12))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(3, 18, n)               # star-shaped (generally concave) ring
        pts = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
        if rng.random() < 0.4:                    # snap some vertices to a coarse lattice: shared / axis-aligned edges
            pts = np.round(pts / 2.0) * 2.0
            keep = [0]
            for k in range(1, len(pts)):
                if not np.array_equal(pts[k], pts[keep[-1]]):
                    keep.append(k)
            pts = pts[keep]
            if len(pts) >= 2 and np.array_equal(pts[0], pts[-1]):
                pts = pts[:-1]
            if len(pts) < 3:
                pts = np.array([[cx, cy], [cx + 6, cy], [cx + 6, cy + 6]])
        mvas.append((pts, float(rng.integers(20, 90)) * 100.0))
    # two exact rectangles sharing an edge (the classic tie-break case)
    mvas.append(([(20, 20), (30, 20), (30, 30), (20, 30)], 2500.0))
    mvas.append(([(30, 20), (40, 20), (40, 30), (30, 30)], 3500.0))
    return mvas, (30.0, 30.0, 500.0, float(rng.integers(0, 360))), [(12.0, 12.0, 45.0, [150])]


def probe_points(comp, rng, n_random=2500):
    bb = comp.bbox
    pts = [np.stack([rng.uniform(bb[0] - 1, bb[2] + 1, n_random), rng.uniform(bb[1] - 1, bb[3] + 1, n_random)], 1)]
    for ring in comp.mva_rings:
        for k in range(len(ring) - 1):
            t = rng.uniform(0, 1, 6)[:, None]
            pts.append(ring[k][None, :] * (1 - t) + ring[k + 1][None, :] * t + rng.normal(0, 3e-4, (6, 2)))
            pts.append(ring[k][None, :] + rng.normal(0, 1e-5, (2, 2)))
            pts.append(ring[k][None, :])          # exact vertices
    return np.concatenate(pts)


@pytest.mark.parametrize("seed,n_poly,cell", [(1, 3, 0.5), (2, 8, 0.5), (3, 14, 1.0), (4, 20, 0.25), (5, 6, 4.0), (6, 11, 0.5)])
def test_grid_equals_ordered_scan_on_random_sectors(seed, n_poly, cell):
    mvas, runway, entries = random_sector(seed, n_poly)
    comp = S.compile_sector(mvas, runway, entries, grid_cell=cell)
    b = comp.blob64
    g = int(b[L.H_OFF_GRID])
    rng = np.random.default_rng(100 + seed)
    for x, y in probe_points(comp, rng):
        truth = S._first_polygon(x, y, comp.mva_rings, comp.mva_bounds)
        assert _walk_grid(b, g, comp.mva_bounds, comp.mva_heights, x, y) == truth, (seed, x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_poly,cell", [(11, 5, 0.5), (12, 16, 0.5), (13, 22, 1.0), (14, 9, 0.25)])
def test_device_lookup_equals_oracle_on_random_sectors(seed, n_poly, cell):
    from atc_hip import lib
    from oracle import oracle as O
    mvas, runway, entries = random_sector(seed, n_poly)
    comp = S.compile_sector(mvas, runway, entries, grid_cell=cell)
    rng = np.random.default_rng(200 + seed)
    pts = probe_points(comp, rng, n_random=400000).astype(np.float32)
    exp = O.OracleQueries(comp, np.float32).mva(pts[:, 0], pts[:, 1])
    sec = lib.Scenario(comp)
    assert np.array_equal(sec.query_mva(pts[:, 0], pts[:, 1], use_grid=True), exp)
    assert np.array_equal(sec.query_mva(pts[:, 0], pts[:, 1], use_grid=False), exp)
    assert (exp >= 0).mean() > 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("seed,N", [(21, 1), (22, 4), (23, 16)])
def test_full_step_on_random_sector_vs_oracle(seed, N):
    """Whole step (not only the lookup) on a random sector described in the JSON sector format: HIP vs fp32 oracle."""
    from atc_hip import sector_io
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    from oracle import oracle as O
    mvas, runway, entries = random_sector(seed, 9)
    doc = {"format": "atc-sector/1", "name": "random%d" % seed,
           "runway": {"x": runway[0], "y": runway[1], "h": runway[2], "phi_from_runway": runway[3]},
           "mvas": [{"height": h, "ring": [[float(a), float(b)] for a, b in ring]} for ring, h in mvas],
           "entrypoints": [{"x": 22.0 + 3 * i, "y": 24.0 + 2 * i, "phi": 40 * i, "levels": [110 + 20 * j for j in range(4)]}
                           for i in range(5)]}
    scn = sector_io.from_dict(doc)
    comp = scenarios.compile_scenario(scn, grid_cell=0.5)
    B = 256
    env = AtcVecEnv(B, N, scenario=scn, auto_reset=True, spawn="random", seed=seed)
    orc = O.OracleEnv(comp, B, N, O.make_params(auto_reset=True, random_entry=True, seed=seed), np.float32)
    rng = np.random.default_rng(seed)
    n_done = 0
    for t in range(150):
        if t % 10 == 0:
            a = rng.uniform(-1.02, 1.02, (B, N, 3)).astype(np.float32)
        obs, rew, done, info = env.step(a)
        orc.step(a)
        assert np.array_equal(info["flags"].cpu().numpy().astype(np.uint32), orc.flags), t
        assert np.array_equal(done.cpu().numpy(), orc.done), t
        o = obs.cpu().numpy().reshape(B, N, 10)
        assert np.all(np.abs(o - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs))), t
        assert np.all(np.abs(rew.cpu().numpy() - orc.reward) <= 1e-5 * np.maximum(1.0, np.abs(orc.reward)) + 6e-8 * N * np.abs(orc.ac_reward).sum(1)), t
        n_done += int(orc.done.sum())
    assert n_done > 5
    env.close()
