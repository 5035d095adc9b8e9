"""Headless `rgb_array` renderer: replaces the reference's pyglet viewer (atc_gym.py:367-552) for
recorders such as VecVideoRecorder (learning/atc-gym-stable-baselines.py:82-83).  Pure numpy rasterisation on the host of
one env's current state (copied from the device): MVA polygon outlines, runway, approach corridor, FAF, aircraft with a
heading tick.  Not on the step path."""
import numpy as np

BACKGROUND = (24, 60, 70)
LINES = (70, 170, 165)
CORRIDOR = (230, 200, 90)
AIRCRAFT = (160, 225, 175)
INACTIVE = (110, 120, 120)


def _line(img, x0, y0, x1, y1, color):
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs = np.clip(np.rint(np.linspace(x0, x1, n)).astype(int), 0, img.shape[1] - 1)
    ys = np.clip(np.rint(np.linspace(y0, y1, n)).astype(int), 0, img.shape[0] - 1)
    img[ys, xs] = color


def _poly(img, pts, color):
    for a, b in zip(pts[:-1], pts[1:]):
        _line(img, a[0], a[1], b[0], b[1], color)


class View:
    """World (nm, y up) -> pixel (row 0 on top) transform that fits the sector with a margin."""

    def __init__(self, bbox, size=800, padding=10):
        x0, y0, x1, y1 = bbox
        self.size = size
        self.scale = (size - 2 * padding) / max(x1 - x0, y1 - y0)
        self.x0, self.y0, self.pad = x0, y0, padding

    def px(self, pts):
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        u = self.pad + (pts[:, 0] - self.x0) * self.scale
        v = self.size - 1 - (self.pad + (pts[:, 1] - self.y0) * self.scale)
        return np.stack([u, v], 1)


def background(compiled, size=800):
    """Static layer: sector outline, runway, corridor."""
    img = np.empty((size, size, 3), np.uint8)
    img[:] = BACKGROUND
    view = View(compiled.bbox, size)
    for ring in compiled.mva_rings:
        _poly(img, view.px(ring), LINES)
    for ring in compiled.noise_rings:
        _poly(img, view.px(ring), INACTIVE)
    cg = compiled.corridor
    _poly(img, view.px(cg["tri_h"]), CORRIDOR)
    rw = view.px([[cg["x"], cg["y"]], cg["faf"], cg["iaf"]])
    _line(img, rw[0, 0], rw[0, 1], rw[2, 0], rw[2, 1], CORRIDOR)
    return img, view


def rgb_array(vec, env=0, size=800):
    """RGB frame [size, size, 3] uint8 of env `env` of an AtcVecEnv."""
    key = ("_render_bg", size)
    cache = vec.__dict__.setdefault("_render_cache", {})
    if key not in cache:
        cache[key] = background(vec.compiled, size)
    bg, view = cache[key]
    img = bg.copy()
    n = vec.N
    lo, hi = env * n, (env + 1) * n
    pos = np.stack([vec.x[lo:hi].cpu().numpy(), vec.y[lo:hi].cpu().numpy()], 1)
    heading = vec.phi[lo:hi].cpu().numpy()
    mask = int(vec.active_mask[env])
    p = view.px(pos)
    for k in range(n):
        color = AIRCRAFT if (mask >> k) & 1 else INACTIVE
        u, v = p[k]
        if not (0 <= u < size and 0 <= v < size):
            continue
        u0, v0 = int(round(u)), int(round(v))
        img[max(v0 - 2, 0):v0 + 3, max(u0 - 2, 0):u0 + 3] = color
        phi = np.radians(heading[k])  # compass heading: 0 = +y, clockwise (model.py:345-348)
        _line(img, u, v, u + 12 * np.sin(phi), v - 12 * np.cos(phi), color)
    return img
