"""Sector data (reference: envs/atc/scenarios.py:14-207) restated as plain vertex tables.

Same class names / attributes as the reference (`mvas`, `runway`, `airspace`, `entrypoints`).  Extensions that have no
reference counterpart: `noise_areas` (README.md:62 prose) and the `LOWWDense` variant used by the 64-aircraft config.
"""
from typing import List

from . import model

# (height_ft, closed ring) in lookup-priority order — Vienna approach sector (scenarios.py:38-187)
_LOWW_MVA = (
    (4800, ((48.43, 2.09), (39.36, 4.22), (27.26, 20.01), (54.03, 12.95), (48.43, 2.09))),
    (3700, ((27.26, 20.01), (26.37, 21.35), (29.73, 26.39), (28.83, 31.09), (34.32, 25.55), (46.08, 22.36), (42.47, 16),
            (27.26, 20.01))),
    (5700, ((26.37, 21.35), (13.15, 38.60), (22.0, 36.13), (22.0, 30.65), (29.73, 26.39), (26.37, 21.35))),
    (4600, ((29.73, 26.39), (22.0, 30.65), (22.0, 36.13), (13.15, 38.60), (8, 45.68), (18.75, 44.98), (28.83, 31.09),
            (29.73, 26.39))),
    (4100, ((28.83, 31.09), (18.75, 44.98), (22.0, 45.68), (26.37, 43.08), (28.83, 31.09))),
    (4000, ((28.83, 31.09), (28.83, 33.45), (31.29, 34.12), (29.73, 41.29), (26.9, 40.47), (28.83, 31.09))),
    (3500, ((22.0, 45.68), (18.75, 44.98), (8, 45.68), (4.08, 50.25), (15.73, 76.12), (29.73, 80.71), (56.16, 82.05),
            (58.51, 69.84), (42.94, 71.97), (22.56, 65.36), (16.17, 50.25), (23.23, 49.01), (22.0, 45.68))),
    (3000, ((46.08, 22.36), (34.32, 25.55), (31.5, 28.4), (36.22, 35.46), (44.46, 31.76), (46.08, 22.36))),
    (3500, ((31.5, 28.4), (28.83, 31.09), (28.83, 33.45), (31.29, 34.12), (29.73, 41.29), (26.9, 40.47), (26.37, 43.08),
            (22.0, 45.68), (23.23, 49.01), (31.29, 48.01), (30.17, 45.71), (32.19, 44.98), (35.14, 41.62),
            (36.22, 42.29), (37.56, 36.69), (36.22, 35.46), (31.5, 28.4))),
    (3200, ((35.14, 41.62), (32.19, 44.98), (30.17, 45.71), (31.29, 48.01), (23.23, 49.01), (16.17, 50.25),
            (22.56, 65.36), (36.58, 69.91), (39.47, 60.55), (35.73, 59.13), (36.22, 56.18), (38.46, 53.72),
            (34.32, 45.68), (35.14, 41.62))),
    (2700, ((46.08, 22.36), (44.95, 28.91), (53.5, 31.43), (57.97, 41.89), (47.17, 55.97), (40.75, 53.72),
            (38.46, 53.72), (36.22, 56.18), (35.73, 59.13), (39.47, 60.55), (36.58, 69.91), (42.94, 71.97),
            (58.51, 69.84), (54.78, 60.01), (68.15, 38.6), (66.34, 36.85), (65.53, 30.62), (62.92, 29.97),
            (66.58, 20.58), (52.88, 18.68), (51.64, 21.35), (46.08, 22.36))),
    (2600, ((44.95, 28.91), (44.46, 31.76), (36.22, 35.46), (37.56, 36.69), (36.22, 42.29), (35.14, 41.62),
            (34.32, 45.68), (38.46, 53.72), (40.75, 53.72), (47.17, 55.97), (57.97, 41.89), (53.5, 31.43),
            (44.95, 28.91))),
)
_LOWW_RUNWAY = (45.16, 43.26, 586, 160)  # scenarios.py:188
# x, y, heading, flight levels (scenarios.py:192-207)
_LOWW_ENTRIES_RANDOM = (
    (10, 51, 90, (130, 150, 170, 190, 210, 230)),
    (17, 74.6, 120, (130, 150, 170, 190, 210, 230)),
    (19.0, 34.0, 45, (130, 150, 170, 190, 210, 230)),
    (29.8, 79.4, 170, (130, 150, 170, 190, 210, 230)),
    (54.0, 80.5, 230, (140, 160, 180, 200, 220, 240)),
    (53.0, 60.0, 260, (140, 160, 180, 200, 220, 240)),
    (66.0, 39.0, 290, (140, 160, 180, 200, 220)),
    (64.4, 22.0, 320, (140, 160, 180, 200, 220)),
    (46.0, 7.0, 320, (140, 160, 180, 200, 220, 240, 260)),
)
_LOWW_ENTRIES_FIXED = ((10, 51, 90, (150,)),)

# scenarios.py:17-21 (rings closed by the polygon constructor)
_SIMPLE_MVA = (
    (3500, ((15, 0), (35, 0), (35, 26))),
    (2400, ((15, 0), (35, 26), (35, 30), (15, 30), (15, 27.8))),
    (4000, ((15, 30), (35, 30), (35, 40), (15, 40))),
    (8000, ((0, 10), (15, 0), (15, 28.7), (0, 17))),
    (6500, ((0, 17), (15, 28.7), (15, 40), (0, 32))),
)


class Scenario:
    runway: model.Runway
    mvas: List[model.MinimumVectoringAltitude]
    airspace: model.Airspace
    entrypoints: List[model.EntryPoint]
    noise_areas: List[model.NoiseAbatementArea] = []


def _assemble(scn, mva_table, runway, entries, noise=()):
    scn.mvas = [model.MinimumVectoringAltitude(list(ring), h) for h, ring in mva_table]
    scn.runway = model.Runway(*runway)
    scn.airspace = model.Airspace(scn.mvas, scn.runway)
    scn.entrypoints = [model.EntryPoint(x, y, phi, list(levels)) for x, y, phi, levels in entries]
    scn.noise_areas = [model.NoiseAbatementArea(list(ring), c, p) for ring, c, p in noise]


class SimpleScenario(Scenario):
    """scenarios.py:14-32"""

    def __init__(self, random_entrypoints=False):
        _assemble(self, _SIMPLE_MVA, (20, 20, 0, 130), ((5, 35, 90, (150,)),))


class LOWW(Scenario):
    """scenarios.py:35-207"""

    def __init__(self, random_entrypoints=False):
        super().__init__()
        _assemble(self, _LOWW_MVA, _LOWW_RUNWAY, _LOWW_ENTRIES_RANDOM if random_entrypoints else _LOWW_ENTRIES_FIXED)


class LOWWDense(Scenario):
    """Build-defined variant for many-aircraft configs (no reference counterpart): the 9 LOWW entry points, each with 8
    flight levels 2000 ft apart (9 x 8 = 72 conflict-free spawn slots >= 64 aircraft), plus 4 synthetic
    noise-abatement areas below the approach paths."""

    NOISE = (
        (((40.0, 36.0), (44.0, 34.5), (47.5, 37.0), (46.5, 41.0), (43.0, 42.5), (40.5, 40.0)), 7000.0, 0.02),
        (((50.0, 40.0), (54.5, 38.5), (57.0, 42.0), (55.0, 46.5), (51.0, 47.0), (49.0, 43.5)), 6000.0, 0.02),
        (((30.0, 50.0), (35.5, 49.0), (38.0, 52.5), (37.0, 57.0), (33.0, 59.0), (30.5, 57.5), (29.0, 53.5)), 8000.0, 0.01),
        (((44.0, 24.0), (49.0, 23.0), (52.5, 25.5), (53.0, 29.0), (50.0, 31.5), (46.0, 31.0), (43.5, 28.0), (43.0, 25.5)),
         6500.0, 0.015),
    )

    def __init__(self, random_entrypoints=True, noise=True):
        levels = (130, 150, 170, 190, 210, 230, 250, 270)
        entries = tuple((x, y, phi, levels) for x, y, phi, _ in _LOWW_ENTRIES_RANDOM)
        _assemble(self, _LOWW_MVA, _LOWW_RUNWAY, entries, self.NOISE if noise else ())


_compiled = {}   # (sector description, grid cell) -> CompiledSector: compiling is pure, and the lookup grid takes seconds


def compile_scenario(scn, grid_cell=None):
    """Scenario object (this module's or a duck-typed one) -> atc_hip.scenario.CompiledSector (cached per process by the
    sector's content: the result is read-only data)."""
    from atc_hip import scenario as _scn
    rw = scn.runway
    mvas = [(m.area_as_list, m.height) for m in scn.mvas]
    runway = (rw.x, rw.y, rw.h, rw.phi_from_runway)
    entries = [(e.x, e.y, e.phi, list(e.levels)) for e in scn.entrypoints]
    noise = [(a.area_as_list, a.ceiling, a.penalty) for a in getattr(scn, "noise_areas", [])]
    key = repr(([([tuple(map(float, p)) for p in ring], float(h)) for ring, h in mvas], tuple(map(float, runway)),
                [(float(x), float(y), float(phi), [float(v) for v in lev]) for x, y, phi, lev in entries],
                [([tuple(map(float, p)) for p in ring], float(c), float(pen)) for ring, c, pen in noise], grid_cell))
    if key not in _compiled:
        _compiled[key] = _scn.compile_sector(mvas, runway, entries, noise=noise, grid_cell=grid_cell)
    return _compiled[key]
