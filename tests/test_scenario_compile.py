"""G1: the product's sector compiler (host, construction-time) against constants captured from the reference."""
import numpy as np
import pytest

import helpers as H
from atc_hip import layout as L


@pytest.mark.parametrize("name", ["LOWW", "LOWW_random", "Simple", "UnitTest"])
def test_derived_constants(name):
    g = H.golden_json("g1_constants.json")[name]
    c = H.compiled(name)
    assert len(c.mva_rings) == len(g["mva_rings"])
    for ring, gr in zip(c.mva_rings, g["mva_rings"]):
        assert np.array_equal(ring, np.asarray(gr))           # closed rings, input order
    assert c.mva_heights == [float(h) for h in g["mva_heights"]]
    assert np.array_equal(np.asarray(c.mva_bounds), np.asarray(g["mva_bounds"]))
    assert list(c.bbox) == g["bbox"]
    assert c.world_diag == g["world_max_distance"]
    cg = c.corridor
    for key, gk in (("faf", "faf"), ("iaf", "iaf"), ("corner1", "corner1"), ("corner2", "corner2"),
                    ("normal", "faf_iaf_normal")):
        assert np.array_equal(cg[key], np.asarray(g[gk])), key
    assert np.array_equal(cg["tri_h"], np.asarray(g["corridor_horizontal"]))
    assert np.array_equal(cg["tri_1"], np.asarray(g["corridor1"]))
    assert np.array_equal(cg["tri_2"], np.asarray(g["corridor2"]))
    assert cg["phi_to_runway"] == g["runway"][4]
    assert c.faf_mva == g["faf_mva"]
    assert np.array_equal(c.norm_min.astype(np.float64), np.asarray(g["norm_min"]))
    assert np.array_equal(c.norm_max.astype(np.float64), np.asarray(g["norm_max"]))
    ents = [[e[0], e[1], e[2]] + e[3] for e in c.entrypoints]
    assert ents == g["entrypoints"]
    assert c.blob64[L.C_ALIGNED_OK] == 1.0


def test_blob_layout_roundtrip():
    c = H.compiled("LOWW")
    b = c.blob64
    assert b[L.H_VERSION] == L.BLOB_VERSION and b[L.H_NWORDS] == len(b)
    assert int(b[L.H_N_MVA]) == 12 and int(b[L.H_N_ENTRY]) == 1 and int(b[L.H_N_NOISE]) == 0
    nverts = 0
    for p in range(12):
        rec = int(b[L.H_OFF_POLY]) + p * L.P_WORDS
        n, off = int(b[rec + L.P_NVERT]), int(b[rec + L.P_VOFF])
        ring = b[off:off + 2 * n].reshape(n, 2)
        assert np.array_equal(ring, c.mva_rings[p])
        assert np.array_equal(ring[0], ring[-1])
        nverts += n
    assert nverts == 123  # closed-ring vertex counts of LOWW sum to 123
    assert np.array_equal(c.blob32, b.astype(np.float32))
    # every integer-valued field survives the float32 device copy
    for idx in (L.H_NWORDS, L.H_OFF_POLY, L.H_OFF_VERT, L.H_OFF_ENTRY):
        assert float(c.blob32[idx]) == b[idx]


def test_dense_scenario_slots_are_conflict_free():
    from envs.atc import scenarios
    c = scenarios.compile_scenario(scenarios.LOWWDense())
    ents = c.entrypoints
    assert len(ents) == 9 and all(len(e[3]) == 8 for e in ents)
    for i in range(9):
        for j in range(i + 1, 9):
            assert np.hypot(ents[i][0] - ents[j][0], ents[i][1] - ents[j][1]) > 3.0
    assert all(np.all(np.diff(e[3]) * 100 >= 2000) for e in ents)
    assert c.n_noise == 4


def test_grid_matches_ordered_scan_on_host():
    """The lookup grid's clean cells agree with the ordered scan at random points (host check of the compiler)."""
    from atc_hip.scenario import _first_polygon
    c = H.compiled("LOWW", grid_cell=0.5)
    b = c.blob64
    g = int(b[L.H_OFF_GRID])
    x0, y0, inv, nx, ny = b[g + L.G_X0], b[g + L.G_Y0], b[g + L.G_INV], int(b[g + L.G_NX]), int(b[g + L.G_NY])
    cells = b[g + L.G_HDR:g + L.G_HDR + nx * ny].reshape(ny, nx)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(c.bbox[0], c.bbox[2], 20000), rng.uniform(c.bbox[1], c.bbox[3], 20000)], 1)
    n_clean = 0
    for x, y in pts:
        i, j = int((x - x0) * inv), int((y - y0) * inv)
        v = cells[j, i]
        truth = _first_polygon(x, y, c.mva_rings, c.mva_bounds)
        if v < L.GRID_MASK_BASE:
            assert int(v) - 1 == truth
            n_clean += 1
        else:
            mask = int(v - L.GRID_MASK_BASE)
            assert truth < 0 or (mask >> truth) & 1
    assert n_clean > 10000
