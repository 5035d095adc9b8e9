import numpy as np


class _Ring:
    def __init__(self, coords):
        self.coords = coords


class Polygon:
    """Keeps vertices in input order and closes the ring by repeating the first point (Shapely 1.6)."""

    def __init__(self, shell):
        pts = [tuple(float(c) for c in np.asarray(p, dtype=float).reshape(-1)[:2]) for p in shell]
        if pts[0] != pts[-1]:
            pts.append(pts[0])
        self.exterior = _Ring(pts)

    @property
    def bounds(self):
        xs = [p[0] for p in self.exterior.coords]
        ys = [p[1] for p in self.exterior.coords]
        return (min(xs), min(ys), max(xs), max(ys))


class Point:
    def __init__(self, x, y):
        self.x, self.y = x, y
