"""On-disk sector format: a sector is DATA — MVA rings, runway, entry points, noise-abatement areas —
stored as JSON and compiled to the device blob by atc_hip.scenario.compile_sector.  The reference keeps its sectors as
Python code (envs/atc/scenarios.py:14-207); `dump`/`load` round-trip those classes' data exactly.

Schema (all coordinates in nautical miles, altitudes in feet, headings in degrees, flight levels in hundreds of feet):
{
  "format": "atc-sector/1",
  "name": "LOWW",
  "runway": {"x": 45.16, "y": 43.26, "h": 586, "phi_from_runway": 160},
  "mvas": [{"height": 4800, "ring": [[48.43, 2.09], ...]}, ...],          # list order = lookup priority (model.py:283)
  "entrypoints": [{"x": 10, "y": 51, "phi": 90, "levels": [150]}, ...],
  "noise_areas": [{"ceiling": 7000, "penalty": 0.02, "ring": [[...], ...]}, ...]   # optional extension
}
"""
import json
import os

FORMAT = "atc-sector/1"
SECTOR_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sectors")


def to_dict(scn, name=None):
    rw = scn.runway
    return {
        "format": FORMAT,
        "name": name or type(scn).__name__,
        "runway": {"x": rw.x, "y": rw.y, "h": rw.h, "phi_from_runway": rw.phi_from_runway},
        "mvas": [{"height": m.height, "ring": [[float(x), float(y)] for x, y in m.area_as_list]} for m in scn.mvas],
        "entrypoints": [{"x": e.x, "y": e.y, "phi": e.phi, "levels": list(e.levels)} for e in scn.entrypoints],
        "noise_areas": [{"ceiling": a.ceiling, "penalty": a.penalty, "ring": [[float(x), float(y)] for x, y in a.area_as_list]}
                        for a in getattr(scn, "noise_areas", [])],
    }


def dump(scn, path, name=None):
    with open(path, "w") as f:
        json.dump(to_dict(scn, name), f, indent=1)
    return path


def from_dict(d):
    """dict -> scenario object with the reference's attribute names (mvas, runway, airspace, entrypoints)."""
    from envs.atc import model, scenarios
    if d.get("format") != FORMAT:
        raise ValueError("not an %s document" % FORMAT)
    for key in ("runway", "mvas", "entrypoints"):
        if key not in d:
            raise ValueError("sector document lacks %r" % key)
    scn = scenarios.Scenario()
    scn.name = d.get("name", "sector")
    scn.mvas = []
    for m in d["mvas"]:
        if len(m["ring"]) < 3:
            raise ValueError("an MVA ring needs at least 3 vertices")
        scn.mvas.append(model.MinimumVectoringAltitude([tuple(p) for p in m["ring"]], m["height"]))
    r = d["runway"]
    scn.runway = model.Runway(r["x"], r["y"], r["h"], r["phi_from_runway"])
    scn.airspace = model.Airspace(scn.mvas, scn.runway)
    scn.entrypoints = [model.EntryPoint(e["x"], e["y"], e["phi"], list(e["levels"])) for e in d["entrypoints"]]
    if not scn.entrypoints:
        raise ValueError("a sector needs at least one entry point")
    scn.noise_areas = [model.NoiseAbatementArea([tuple(p) for p in a["ring"]], a["ceiling"], a["penalty"])
                       for a in d.get("noise_areas", [])]
    return scn


def load(path_or_name):
    """Loads a sector from a JSON file, or by name from the bundled `sectors/` directory (e.g. load("LOWW"))."""
    path = path_or_name
    if not os.path.exists(path):
        cand = os.path.join(SECTOR_DIR, path_or_name + ".json")
        if os.path.exists(cand):
            path = cand
        else:
            raise FileNotFoundError(path_or_name)
    with open(path) as f:
        return from_dict(json.load(f))
