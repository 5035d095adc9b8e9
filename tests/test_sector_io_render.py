"""CPU tests of the on-disk sector format (round trip compiles to the identical blob) and of the headless renderer's
static layer (no GPU needed for the background)."""
import json
import os

import numpy as np
import pytest

import helpers as H  # noqa: F401
from atc_hip import render, sector_io
from envs.atc import scenarios


@pytest.mark.parametrize("name,make", [("LOWW", lambda: scenarios.LOWW()),
                                       ("LOWW_random_entrypoints", lambda: scenarios.LOWW(random_entrypoints=True)),
                                       ("SimpleScenario", lambda: scenarios.SimpleScenario()),
                                       ("LOWWDense", lambda: scenarios.LOWWDense())])
def test_bundled_sector_files_round_trip(name, make, tmp_path):
    code = scenarios.compile_scenario(make(), grid_cell=1.0)
    disk = scenarios.compile_scenario(sector_io.load(name), grid_cell=1.0)
    assert np.array_equal(code.blob64, disk.blob64)
    p = sector_io.dump(make(), str(tmp_path / "s.json"))
    again = scenarios.compile_scenario(sector_io.load(p), grid_cell=1.0)
    assert np.array_equal(code.blob64, again.blob64)
    d = json.load(open(p))
    assert d["format"] == "atc-sector/1" and len(d["mvas"]) == len(code.mva_rings)


def test_malformed_documents_are_rejected(tmp_path):
    d = sector_io.to_dict(scenarios.SimpleScenario())
    for broken in ({**d, "format": "other"}, {k: v for k, v in d.items() if k != "runway"}, {**d, "entrypoints": []},
                   {**d, "mvas": [{"height": 1000, "ring": [[0, 0], [1, 1]]}]}):
        with pytest.raises(ValueError):
            sector_io.from_dict(broken)
    with pytest.raises(FileNotFoundError):
        sector_io.load("no_such_sector")


def test_custom_sector_compiles_and_matches_oracle_lookup():
    """A sector written by hand in the JSON schema (two squares + a runway) goes through the whole host pipeline."""
    from oracle import oracle as O
    doc = {"format": "atc-sector/1", "name": "toy",
           "runway": {"x": 10, "y": 10, "h": 100, "phi_from_runway": 90},
           "mvas": [{"height": 2000, "ring": [[0, 0], [20, 0], [20, 20], [0, 20]]},
                    {"height": 5000, "ring": [[20, 0], [40, 0], [40, 20], [20, 20]]}],
           "entrypoints": [{"x": 35, "y": 10, "phi": 270, "levels": [80, 100]}]}
    comp = scenarios.compile_scenario(sector_io.from_dict(doc), grid_cell=2.0)
    q = O.OracleQueries(comp, np.float64)
    assert list(q.mva([5, 30, 20, 50], [5, 5, 5, 5])) == [2000, 5000, 2000, -1]   # shared border: first polygon wins
    assert comp.faf_mva == 2000 and comp.n_entry == 1


def _same_geoms(got, exp, tol):
    assert len(got) == len(exp), (len(got), len(exp))
    for g, e in zip(got, exp):
        kind = "FilledPolygon" if (g["kind"] == "Circle") else g["kind"]
        assert kind == e["kind"], (g["kind"], e["kind"])
        if g["kind"] == "Label":
            assert g["text"] == e["text"] and abs(g["x"] - e["x"]) <= tol and abs(g["y"] - e["y"]) <= tol, (g, e)
            continue
        if g["kind"] == "Circle":
            assert g["radius"] == e["radius"] and np.allclose(g["translation"], e["translation"], rtol=0, atol=tol)
        else:
            assert np.allclose(np.asarray(g["v"], float), np.asarray(e["v"], float), rtol=0, atol=tol), (g["kind"], g["v"], e["v"])
            assert g.get("close", None) == e.get("close", None)
            assert float(g["linewidth"]) == e["linewidth"]
        assert np.allclose(g["color"], e["color"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("scen", ["LOWW", "Simple"])
def test_render_static_geometry_equals_reference(scen):
    """G10: what the reference's render() draws once (atc_gym.py:384-398) — window size, background, filled MVA polygons,
    their outlines, runway line, FAF triangle, approach dashes — captured from the imported reference with a recording
    `rendering` stand-in; the headless renderer builds the same primitives in the same screen coordinates, in the same order,
    with the same colours and line widths."""
    g = H.golden_json("g10_render_geometry.json")[scen]
    comp = H.compiled(scen)
    sc = render.Screen(comp.bbox)
    assert (sc.width, sc.height) == (g["width"], g["height"]) and abs(sc.scale - g["scale"]) < 1e-12
    assert render.PADDING == g["padding"]
    _same_geoms(render.static_scene(comp), g["static"], 1e-9)
    # per-frame geometry from the recorded aircraft states (the GPU test produces the states itself)
    for f in g["frames"]:
        x, y, h, phi, v = f["state"]
        got = render.frame_scene(comp, [{"x": x, "y": y, "h": h, "v": v}], f.get("total_reward", 0.0), f.get("last_reward", 0.0))
        exp = [e for e in f["geoms"] if not (e["kind"] == "FilledPolygon" and "radius" in e)]   # history dots: GPU test
        _same_geoms(got, exp, 1e-9)


def test_render_rasterises_the_scene():
    comp = H.compiled("LOWW")
    img, sc = render.background(comp)
    assert img.shape == (sc.height, sc.width, 3) == (768, 620, 3) and img.dtype == np.uint8
    line = render._u8(render.LINES_INFO)
    lines = np.all(img == line, axis=2)
    assert 2000 < lines.sum() < 40000                     # outlines, runway, FAF symbol, approach dashes
    for ring in comp.mva_rings:                           # every polygon vertex lands on an outline pixel
        for u, v in sc.padded(ring):
            assert lines[sc.height - 1 - int(round(v)), int(round(u))]
    inside = np.all(img == render._u8(render.BACKGROUND_ACTIVE), axis=2)
    outside = np.all(img == render._u8(render.BACKGROUND_INACTIVE), axis=2)
    assert inside.sum() > 0.4 * img.shape[0] * img.shape[1] and outside.sum() > 0.1 * img.shape[0] * img.shape[1]
    frame = render.rasterise(sc.width, sc.height, render.frame_scene(comp, [{"x": 30.0, "y": 40.0, "h": 9000.0, "v": 250.0,
                                                                             "history": [(29.0 + 0.1 * i, 40.0) for i in range(30)]}]), img.copy())
    plane = np.all(frame == render._u8(render.AIRPLANE), axis=2)
    assert 20 < plane.sum() < 400


def test_render_draws_label_text():
    """Label text (rendering.py:7-23: left / top anchored) in the built-in 5 x 7 font: callsign, altitude / speed and the two
    reward lines appear at the anchors the scene names, in the label colour, and different texts give different pixels."""
    comp = H.compiled("LOWW")
    bg, sc = render.background(comp)
    ac = [{"x": 30.0, "y": 40.0, "h": 9000.0, "v": 250.0, "name": "FLT07"}]
    scene = render.frame_scene(comp, ac, 12.5, -0.05)
    labels = [g for g in scene if g["kind"] == "Label"]
    assert [g["text"] for g in labels] == ["FLT07", "90  25", "Total reward: 12.50", "Last reward: -0.05"]
    frame = render.rasterise(sc.width, sc.height, scene, bg.copy())
    ink = np.all(frame == render._u8(render.LABEL), axis=2)
    assert not np.all(bg == render._u8(render.LABEL), axis=2).any()
    for g in labels:                                       # all the ink of a label lies in its left / top anchored box
        r0, c0 = sc.height - 1 - int(round(g["y"])), int(round(g["x"]))
        box = ink[r0:r0 + render.GLYPH_H, c0:c0 + render.GLYPH_ADVANCE * len(g["text"])]
        assert box.sum() >= 5 * len(g["text"].replace(" ", "")), g["text"]
    boxes = np.zeros_like(ink)
    for g in labels:
        r0, c0 = sc.height - 1 - int(round(g["y"])), int(round(g["x"]))
        boxes[r0:r0 + render.GLYPH_H, c0:c0 + render.GLYPH_ADVANCE * len(g["text"])] = True
    assert not (ink & ~boxes).any()
    other = render.rasterise(sc.width, sc.height, render.frame_scene(comp, ac, 13.5, -0.05), bg.copy())
    assert (other != frame).any()
    masks = [render._GLYPH_MASKS[c] for c in "0123456789"]
    assert all((masks[i] != masks[j]).any() for i in range(10) for j in range(i))
    clipped = render.rasterise(60, 30, [{"kind": "Label", "text": "CLIPPED AT THE EDGE", "x": 40, "y": 3}])   # no exception
    assert clipped.shape == (30, 60, 3)
