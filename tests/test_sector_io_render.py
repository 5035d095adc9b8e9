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


def test_render_background_layer():
    comp = H.compiled("LOWW")
    img, view = render.background(comp, size=400)
    assert img.shape == (400, 400, 3) and img.dtype == np.uint8
    lines = np.all(img == np.array(render.LINES, np.uint8), axis=2)
    assert 1500 < lines.sum() < 20000                     # polygon outlines were drawn
    # every polygon vertex lands on an outline pixel
    for ring in comp.mva_rings:
        for u, v in view.px(ring):
            assert lines[int(round(v)), int(round(u))]
    corr = np.all(img == np.array(render.CORRIDOR, np.uint8), axis=2)
    u, v = view.px([comp.corridor["faf"]])[0]
    assert corr[int(round(v)), int(round(u))]
