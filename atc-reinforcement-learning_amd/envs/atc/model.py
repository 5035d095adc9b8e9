"""Host mirror of the reference's `envs.atc.model` interface (reference: envs/atc/model.py).

Same class names, constructor arguments and method meaning as the reference so that its callers and its tests
(envs/atc/model_test.py) read unchanged — but the geometry predicates on the step path
(`Airspace.get_mva_height`, `Runway.inside_corridor`, `Corridor._inside_corridor_angle`, `ray_tracing`) are evaluated by
the HIP kernels in libatcstep.so on the GPU (they raise if the library / a GPU is missing; there is no CPU fallback).
Constructors only assemble data (like the reference's, which use shapely at construction time only).
"""
import math  # noqa: F401  (kept for API familiarity)
from typing import List

import numpy as np

from atc_hip import scenario as _scn

nautical_miles_to_feet = 6076  # ft/nm (model.py:10)


def _ring_of(area):
    """Accepts a list of (x, y) points or any object exposing `.exterior.coords` (e.g. a shapely Polygon)."""
    ext = getattr(area, "exterior", None)
    pts = list(ext.coords) if ext is not None else list(area)
    return _scn.close_ring(pts)


class SimParameters:
    """model.py:132-145"""

    def __init__(self, timestep: float, precision: float = 0.5, reward_shaping: bool = True,
                 normalize_state: bool = True, discrete_action_space: bool = False):
        self.timestep = timestep
        self.precision = precision  # never read by the reference either (model.py:142)
        self.reward_shaping = reward_shaping
        self.normalize_state = normalize_state
        self.discrete_action_space = discrete_action_space


class EntryPoint:
    """model.py:309-315"""

    def __init__(self, x: float, y: float, phi: int, levels: List[int]):
        self.x = x
        self.y = y
        self.phi = phi
        self.levels = levels


class MinimumVectoringAltitude:
    """model.py:260-268.  `area`: list of (x, y) vertices or a shapely-like polygon."""

    def __init__(self, area, height: int):
        self.area = area
        self.height = height
        self.area_as_list = _ring_of(area)
        r = self.area_as_list
        self.outer_bounds = (float(r[:, 0].min()), float(r[:, 1].min()), float(r[:, 0].max()), float(r[:, 1].max()))


class NoiseAbatementArea:
    """Extension (README.md:62 of the reference is prose only): polygon with a ceiling [ft] and a per-step penalty."""

    def __init__(self, area, ceiling: float, penalty: float):
        self.area = area
        self.area_as_list = _ring_of(area)
        self.ceiling = ceiling
        self.penalty = penalty


class _DeviceSector:
    """Lazily compiled + uploaded scenario used by the query methods below."""

    def __init__(self, build):
        self._build = build
        self._handle = None
        self.compiled = None

    def handle(self):
        if self._handle is None:
            from atc_hip import lib
            self.compiled = self._build()
            self._handle = lib.Scenario(self.compiled)
        return self._handle


class Corridor:
    """model.py:148-231"""

    def __init__(self, x: int, y: int, h: int, phi_from_runway: int):
        self.x = x
        self.y = y
        self.h = h
        self.phi_from_runway = phi_from_runway
        self.phi_to_runway = (phi_from_runway + 180) % 360
        g = _scn.corridor_geometry(x, y, h, phi_from_runway)
        col = lambda v: np.asarray(v, dtype=np.float64).reshape(2, 1)  # noqa: E731  (reference keeps column vectors)
        self._faf_iaf_normal = col(g["normal"])
        self.faf_angle = 45
        self.faf = col(g["faf"])
        self.corner1 = col(g["corner1"])
        self.corner2 = col(g["corner2"])
        self.iaf = col(g["iaf"])
        self.corridor_horizontal_list = g["tri_h"]
        self.corridor1_list = g["tri_1"]
        self.corridor2_list = g["tri_2"]
        self._dev = _DeviceSector(lambda: _scn.compile_sector([], (x, y, h, phi_from_runway), []))

    def inside_corridor(self, x, y, h, phi):
        return bool(self._dev.handle().query_corridor([x], [y], [h], [phi])[0])

    def _inside_corridor_angle(self, x, y, phi):
        return bool(self._dev.handle().query_corridor([x], [y], [0.0], [phi], angle_only=True)[0])


class Runway:
    """model.py:234-257"""

    def __init__(self, x, y, h, phi):
        self.x = x
        self.y = y
        self.h = h
        self.phi_from_runway = phi
        self.phi_to_runway = (phi + 180) % 360
        self.corridor = Corridor(x, y, h, phi)

    def inside_corridor(self, x: int, y: int, h: int, phi: int):
        return self.corridor.inside_corridor(x, y, h, phi)


class Airspace:
    """model.py:271-306"""

    def __init__(self, mvas: List[MinimumVectoringAltitude], runway: Runway):
        self.mvas = mvas
        self.runway = runway
        self._dev = _DeviceSector(lambda: _scn.compile_sector(
            [(m.area_as_list, m.height) for m in self.mvas],
            (runway.x, runway.y, runway.h, runway.phi_from_runway), []))

    def find_mva(self, x, y):
        """First MVA in list order containing (x, y) (model.py:282-289); raises ValueError outside the airspace."""
        idx = int(self._dev.handle().query_mva_index([x], [y])[0])
        if idx < 0:
            raise ValueError('Outside of airspace')
        return self.mvas[idx]

    def get_mva_heights(self, xs, ys):
        """Batched get_mva_height: heights [ft] for arrays of points in ONE launch, -1 outside the airspace (build-own
        convenience; the reference evaluates one point per call)."""
        return self._dev.handle().query_mva(xs, ys)

    def get_mva_height(self, x, y):
        return self.find_mva(x, y).height

    def get_bounding_box(self):
        """(minx, miny, maxx, maxy) over all MVA polygons (model.py:294-301)."""
        b = [m.outer_bounds for m in self.mvas]
        return (min(v[0] for v in b), min(v[1] for v in b), max(v[2] for v in b), max(v[3] for v in b))


_ray_sectors = {}   # one-polygon device sectors of recent ray_tracing() calls, keyed by the ring's bytes


def ray_tracing(x, y, poly):
    """model.py:318-337 — evaluated on the device against a one-polygon sector (compiled and uploaded once per polygon;
    `x`, `y` may be arrays: one launch for all points)."""
    ring = _scn.close_ring(poly)
    key = ring.tobytes()
    sector = _ray_sectors.get(key)
    if sector is None:
        from atc_hip import lib
        if len(_ray_sectors) >= 64:
            _ray_sectors.pop(next(iter(_ray_sectors))).close()
        sector = _ray_sectors[key] = lib.Scenario(_scn.compile_sector([(ring, 1)], (0.0, 0.0, 0.0, 0.0), []))
    xs, ys = np.atleast_1d(x), np.atleast_1d(y)
    hit = sector.query_mva_index(xs, ys) == 0
    return bool(hit[0]) if np.ndim(x) == 0 and np.ndim(y) == 0 else hit


def relative_angle(angle1, angle2):
    """model.py:340-342 (host helper; the step kernels carry their own copy)."""
    return (angle2 - angle1 + 180) % 360 - 180


def rot_matrix(phi):
    """model.py:345-348"""
    return _scn.rot_matrix(phi)
