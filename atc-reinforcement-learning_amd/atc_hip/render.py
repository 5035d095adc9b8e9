"""Headless `rgb_array` renderer: replaces the reference's pyglet viewer (atc_gym.py:367-552) for
recorders such as VecVideoRecorder (learning/atc-gym-stable-baselines.py:82-83).  Not on the step path.

Two layers:
  * `static_scene` / `frame_scene` build WHAT the reference draws — the same primitives in the same screen coordinates
    (window 600 px + 2 x 10 px padding wide, scale = 600 / sector width; atc_gym.py:372-382): sector background, filled MVA
    polygons and their outlines (:502-524), runway line (:526-540), FAF triangle (:475-497), approach dashes (:454-473),
    aircraft symbol, label anchors and history dots (:411-452), reward labels (:402-409).  Including its quirk: MVA polygons
    are shifted by the padding, everything placed with `_screen_vector` (:542-552) is not.  This layer is pinned against the
    geometry captured from the reference (tests/golden/g10_render_geometry.json).
  * `rasterise` turns a scene into an RGB array with a small numpy rasteriser (even-odd polygon fill, thick lines, dots, and
    label text in a built-in 5 x 7 bitmap font at the reference's anchors: left / top, rendering.py:7-23 — the reference
    asks pyglet for 9 pt Arial, a font renderer that does not exist here).
"""
import math

import numpy as np

# the reference's colour scheme (envs/atc/themes.py), as data
BACKGROUND_INACTIVE = (29 / 256, 69 / 256, 76 / 256)
BACKGROUND_ACTIVE = (84 / 256, 121 / 256, 128 / 256)
LINES_INFO = (69 / 256, 173 / 256, 168 / 256)
AIRPLANE = (157 / 256, 224 / 256, 173 / 256)
LABEL = (157 / 255, 224 / 255, 173 / 255)         # ColorScheme.label (157, 224, 173, 255)
INACTIVE = (110 / 256, 120 / 256, 120 / 256)      # handed-over aircraft (extension; no reference counterpart)
SCREEN_WIDTH, PADDING = 600, 10                   # atc_gym.py:373-374


def _rot(phi_deg, vec):
    """compass rotation of a screen vector (model.py:345-348 applied to [[x], [y]])"""
    p = math.radians(phi_deg)
    return (math.cos(p) * vec[0] + math.sin(p) * vec[1], -math.sin(p) * vec[0] + math.cos(p) * vec[1])


class Screen:
    """World (nm) -> screen transform of atc_gym.py:376-382,542-552."""

    def __init__(self, bbox):
        self.x_min, self.y_min, x_max, y_max = bbox
        self.scale = SCREEN_WIDTH / (x_max - self.x_min)
        self.width = SCREEN_WIDTH + 2 * PADDING
        self.height = int((y_max - self.y_min) * self.scale) + 2 * PADDING

    def vector(self, x, y):                      # _screen_vector: no padding (reference quirk)
        return ((x - self.x_min) * self.scale, (y - self.y_min) * self.scale)

    def padded(self, pts):                       # transform_world_to_screen of _render_mvas: with padding
        return [((p[0] - self.x_min) * self.scale + PADDING, (p[1] - self.y_min) * self.scale + PADDING) for p in pts]


def static_scene(compiled):
    """Geometry drawn once (atc_gym.py:384-398): list of dicts {kind, v, color, linewidth[, close]} in drawing order."""
    sc = Screen(compiled.bbox)
    w, h = sc.width, sc.height
    out = [{"kind": "FilledPolygon", "v": [(0, 0), (0, h), (w, h), (w, 0)], "color": BACKGROUND_INACTIVE, "linewidth": 1}]
    rings = [sc.padded(r) for r in compiled.mva_rings]
    out += [{"kind": "FilledPolygon", "v": r, "color": BACKGROUND_ACTIVE, "linewidth": 1} for r in rings]
    out += [{"kind": "PolyLine", "v": r, "close": True, "color": LINES_INFO, "linewidth": 1} for r in rings]
    cg = compiled.corridor
    rv = sc.vector(cg["x"], cg["y"])
    half = _rot(cg["phi_from_runway"], (0.0, 1.7 * sc.scale / 2))
    out.append({"kind": "PolyLine", "v": [(rv[0] - half[0], rv[1] - half[1]), (rv[0] + half[0], rv[1] + half[1])],
                "close": False, "color": LINES_INFO, "linewidth": 5})
    fv = sc.vector(cg["faf"][0], cg["faf"][1])
    c = (0.0, 6.0)
    tri = [(fv[0] + c[0], fv[1] + c[1])] + [(fv[0] + r[0], fv[1] + r[1]) for r in (_rot(121, c), _rot(242, c))]
    out.append({"kind": "PolyLine", "v": tri, "close": True, "color": LINES_INFO, "linewidth": 2})
    dashes = 48
    d = ((cg["iaf"][0] - cg["x"]) * sc.scale, (cg["iaf"][1] - cg["y"]) * sc.scale)
    for i in range(int(dashes / 2 + 1)):
        out.append({"kind": "PolyLine", "v": [(rv[0] + d[0] / dashes * 2 * i, rv[1] + d[1] / dashes * 2 * i),
                                              (rv[0] + d[0] / dashes * (2 * i + 1), rv[1] + d[1] / dashes * (2 * i + 1))],
                    "close": False, "color": LINES_INFO, "linewidth": 1})
    return out


def aircraft_geoms(sc, x, y, h, v, name="FLT01", history=(), color=AIRPLANE):
    """One aircraft (atc_gym.py:411-452): symbol, two label anchors, history dots."""
    vec = sc.vector(x, y)
    corner = (0.0, 4.0)
    sym = [(vec[0] + r[0], vec[1] + r[1]) for r in (_rot(a, corner) for a in (45, 135, 225, 315))]
    out = [{"kind": "PolyLine", "v": sym, "close": True, "color": color, "linewidth": 2}]
    lp = _rot(135, (0.0, 8.0))
    lx, ly = lp[0] + vec[0], lp[1] + vec[1]
    out.append({"kind": "Label", "text": name, "x": lx, "y": ly})
    out.append({"kind": "Label", "text": "%d  %d" % (round(h / 100), round(v / 10)), "x": lx, "y": ly - 15})
    n = len(history)
    for i in range(n - 5, max(0, n - 25), -1):
        if i % 5 == 0:
            out.append({"kind": "Circle", "radius": 2.0, "translation": sc.vector(history[i][0], history[i][1]),
                        "color": color})
    return out


def frame_scene(compiled, aircraft, total_reward=None, last_reward=None):
    """Per-frame geometry.  aircraft: iterable of dicts {x, y, h, v[, name, history, active]}."""
    sc = Screen(compiled.bbox)
    out = []
    for a in aircraft:
        out += aircraft_geoms(sc, a["x"], a["y"], a["h"], a["v"], a.get("name", "FLT01"), a.get("history", ()),
                              AIRPLANE if a.get("active", True) else INACTIVE)
    if total_reward is not None:             # _render_reward, atc_gym.py:402-409
        out.append({"kind": "Label", "text": "Total reward: %.2f" % total_reward, "x": 10, "y": 40})
        out.append({"kind": "Label", "text": "Last reward: %.2f" % last_reward, "x": 10, "y": 25})
    return out


# ---------------------------------------------------------------------------------------------------- rasteriser
def _u8(color):
    return np.array([int(round(c * 255)) for c in color], np.uint8)


def _fill(img, pts, color):
    """even-odd fill of a polygon given in screen coordinates (y up)"""
    H, W = img.shape[:2]
    p = np.asarray(pts, np.float64)
    x0, x1 = max(int(np.floor(p[:, 0].min())), 0), min(int(np.ceil(p[:, 0].max())), W - 1)
    y0, y1 = max(int(np.floor(p[:, 1].min())), 0), min(int(np.ceil(p[:, 1].max())), H - 1)
    if x1 < x0 or y1 < y0:
        return
    xs, ys = np.meshgrid(np.arange(x0, x1 + 1) + 0.5, np.arange(y0, y1 + 1) + 0.5)
    inside = np.zeros(xs.shape, bool)
    q = np.roll(p, -1, axis=0)
    for (ax, ay), (bx, by) in zip(p, q):
        if ay == by:
            continue
        cond = (ys > min(ay, by)) & (ys <= max(ay, by))
        xi = (ys - ay) * (bx - ax) / (by - ay) + ax
        inside ^= cond & (xs <= xi)
    rows = (H - 1 - np.arange(y0, y1 + 1))[:, None]
    cols = np.arange(x0, x1 + 1)[None, :]
    block = img[rows, cols]
    block[inside] = color
    img[rows, cols] = block


def _line(img, a, b, color, width=1):
    H, W = img.shape[:2]
    n = int(max(abs(b[0] - a[0]), abs(b[1] - a[1])) * 2) + 2
    xs, ys = np.linspace(a[0], b[0], n), np.linspace(a[1], b[1], n)
    r = max(0, (int(round(width)) - 1) // 2)
    for dx in range(-r, r + 1):
        for dy in range(-r, r + 1):
            u = np.rint(xs + dx).astype(int)
            v = H - 1 - np.rint(ys + dy).astype(int)
            ok = (u >= 0) & (u < W) & (v >= 0) & (v < H)
            img[v[ok], u[ok]] = color


# 5 x 7 bitmap font: one string of 7 rows x 5 columns per glyph ('#' = ink); lower case is drawn as upper case
_GLYPHS = {
    "0": ".###. #...# #..## #.#.# ##..# #...# .###.", "1": "..#.. .##.. ..#.. ..#.. ..#.. ..#.. .###.",
    "2": ".###. #...# ....# ...#. ..#.. .#... #####", "3": ".###. #...# ....# ..##. ....# #...# .###.",
    "4": "...#. ..##. .#.#. #..#. ##### ...#. ...#.", "5": "##### #.... ####. ....# ....# #...# .###.",
    "6": "..##. .#... #.... ####. #...# #...# .###.", "7": "##### ....# ...#. ..#.. .#... .#... .#...",
    "8": ".###. #...# #...# .###. #...# #...# .###.", "9": ".###. #...# #...# .#### ....# ...#. .##..",
    "A": ".###. #...# #...# ##### #...# #...# #...#", "B": "####. #...# #...# ####. #...# #...# ####.",
    "C": ".###. #...# #.... #.... #.... #...# .###.", "D": "####. #...# #...# #...# #...# #...# ####.",
    "E": "##### #.... #.... ####. #.... #.... #####", "F": "##### #.... #.... ####. #.... #.... #....",
    "G": ".###. #...# #.... #.### #...# #...# .###.", "H": "#...# #...# #...# ##### #...# #...# #...#",
    "I": ".###. ..#.. ..#.. ..#.. ..#.. ..#.. .###.", "J": "..### ...#. ...#. ...#. ...#. #..#. .##..",
    "K": "#...# #..#. #.#.. ##... #.#.. #..#. #...#", "L": "#.... #.... #.... #.... #.... #.... #####",
    "M": "#...# ##.## #.#.# #.#.# #...# #...# #...#", "N": "#...# ##..# #.#.# #..## #...# #...# #...#",
    "O": ".###. #...# #...# #...# #...# #...# .###.", "P": "####. #...# #...# ####. #.... #.... #....",
    "Q": ".###. #...# #...# #...# #.#.# #..#. .##.#", "R": "####. #...# #...# ####. #.#.. #..#. #...#",
    "S": ".#### #.... #.... .###. ....# ....# ####.", "T": "##### ..#.. ..#.. ..#.. ..#.. ..#.. ..#..",
    "U": "#...# #...# #...# #...# #...# #...# .###.", "V": "#...# #...# #...# #...# #...# .#.#. ..#..",
    "W": "#...# #...# #...# #.#.# #.#.# ##.## #...#", "X": "#...# #...# .#.#. ..#.. .#.#. #...# #...#",
    "Y": "#...# #...# .#.#. ..#.. ..#.. ..#.. ..#..", "Z": "##### ....# ...#. ..#.. .#... #.... #####",
    ":": "..... ..#.. ..#.. ..... ..#.. ..#.. .....", ".": "..... ..... ..... ..... ..... .##.. .##..",
    "-": "..... ..... ..... ##### ..... ..... .....", "+": "..... ..#.. ..#.. ##### ..#.. ..#.. .....",
    " ": "..... ..... ..... ..... ..... ..... .....", "?": ".###. #...# ....# ...#. ..#.. ..... ..#..",
}
_GLYPH_MASKS = {c: np.array([[ch == "#" for ch in row] for row in g.split()], bool) for c, g in _GLYPHS.items()}
GLYPH_W, GLYPH_H, GLYPH_ADVANCE = 5, 7, 6


def _text(img, text, x, y, color):
    """label text with its LEFT / TOP corner at screen position (x, y) (y up), rendering.py:18-23"""
    H, W = img.shape[:2]
    col0, row0 = int(round(x)), H - 1 - int(round(y))
    for n, ch in enumerate(str(text)):
        m = _GLYPH_MASKS.get(ch.upper(), _GLYPH_MASKS["?"])
        c0 = col0 + n * GLYPH_ADVANCE
        r_lo, r_hi = max(row0, 0), min(row0 + GLYPH_H, H)
        c_lo, c_hi = max(c0, 0), min(c0 + GLYPH_W, W)
        if r_lo >= r_hi or c_lo >= c_hi:
            continue
        sub = m[r_lo - row0:r_hi - row0, c_lo - c0:c_hi - c0]
        block = img[r_lo:r_hi, c_lo:c_hi]
        block[sub] = color
        img[r_lo:r_hi, c_lo:c_hi] = block


def rasterise(width, height, geoms, img=None):
    """RGB uint8 array [height, width, 3] of a list of scene primitives (row 0 = top of the window)."""
    if img is None:
        img = np.zeros((height, width, 3), np.uint8)
    for g in geoms:
        kind = g["kind"]
        if kind == "FilledPolygon":
            _fill(img, g["v"], _u8(g["color"]))
        elif kind == "PolyLine":
            v = list(g["v"]) + ([g["v"][0]] if g.get("close") else [])
            for a, b in zip(v[:-1], v[1:]):
                _line(img, a, b, _u8(g["color"]), g.get("linewidth", 1))
        elif kind == "Circle":
            t, r = g["translation"], g["radius"]
            ang = np.linspace(0, 2 * np.pi, 12, endpoint=False)
            _fill(img, np.stack([t[0] + r * np.cos(ang), t[1] + r * np.sin(ang)], 1), _u8(g["color"]))
        elif kind == "Label":
            _text(img, g["text"], g["x"], g["y"], _u8(g.get("color", LABEL)))
    return img


def background(compiled):
    sc = Screen(compiled.bbox)
    return rasterise(sc.width, sc.height, static_scene(compiled)), sc


def rgb_array(vec, env=0, history=None, total_reward=None, last_reward=None):
    """RGB frame of env `env` of an AtcVecEnv (window size as in the reference: 620 x (sector height x scale + 20))."""
    cache = vec.__dict__.setdefault("_render_cache", {})
    if "bg" not in cache:
        cache["bg"] = background(vec.compiled)
    bg, sc = cache["bg"]
    n = vec.N
    lo, hi = env * n, (env + 1) * n
    xs, ys = vec.x[lo:hi].cpu().numpy(), vec.y[lo:hi].cpu().numpy()
    hs, vs = vec.h[lo:hi].cpu().numpy(), vec.v[lo:hi].cpu().numpy()
    mask = int(vec.active_mask[env])
    aircraft = [{"x": float(xs[k]), "y": float(ys[k]), "h": float(hs[k]), "v": float(vs[k]), "name": "FLT%02d" % (k + 1),
                 "history": history if (history is not None and k == 0) else (), "active": bool((mask >> k) & 1)}
                for k in range(n)]
    return rasterise(sc.width, sc.height, frame_scene(vec.compiled, aircraft, total_reward, last_reward), bg.copy())
