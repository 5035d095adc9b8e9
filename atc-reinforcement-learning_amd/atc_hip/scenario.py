"""Sector compiler: sector description -> flat scenario blob (float64 master + float32 device copy).

Construction-time only (runs once per scenario on the host, like the reference's constructors);
nothing here is on the step path.  Follows the reference's derived quantities:

  * closed rings in input order                      — Shapely 1.6 behaviour behind model.py:266
  * per-polygon bounds, world bbox                   — model.py:267,294-306
  * corridor geometry (FAF/IAF/corners/normal)       — model.py:155-186
  * faf_mva, world diagonal, normalisation vectors   — atc_gym.py:49-58,88-110
  * aircraft limits and rates                        — model.py:13,45-50

The optional lookup grid is an acceleration structure for Airspace.find_mva (model.py:282-289); cells
are classified conservatively so results are identical to the ordered polygon scan (see build_grid).
"""
import math

import numpy as np

from . import layout as L


def close_ring(points):
    """Vertices in input order, first point repeated at the end (what shapely.geometry.Polygon(...).exterior.coords
    yields for the reference's inputs, model.py:266)."""
    pts = [tuple(float(c) for c in np.asarray(p, dtype=np.float64).reshape(-1)[:2]) for p in points]
    if pts[0] != pts[-1]:
        pts.append(pts[0])
    return np.asarray(pts, dtype=np.float64)


def rot_matrix(phi_deg):
    """Compass rotation used by the reference's geometry (model.py:345-348)."""
    phi = math.radians(phi_deg)
    return np.array([[math.cos(phi), math.sin(phi)], [-math.sin(phi), math.cos(phi)]])


def corridor_geometry(x, y, h, phi_from_runway):
    """Approach-corridor geometry (model.py:155-186): returns dict of float64 values."""
    faf_threshold_distance = 7.4
    faf_angle = 45
    faf_iaf_distance = 3
    corner_dist = faf_iaf_distance / math.cos(math.radians(faf_angle))
    r_from = rot_matrix(phi_from_runway)
    origin = np.array([[x], [y]], dtype=np.float64)
    normal = np.dot(r_from, np.array([[0], [1]]))
    faf = origin + np.dot(r_from, np.array([[0], [faf_threshold_distance]]))
    along = np.dot(r_from, [[0], [corner_dist]])
    corner1 = np.dot(rot_matrix(faf_angle), along) + faf
    corner2 = np.dot(rot_matrix(-faf_angle), along) + faf
    iaf = origin + np.dot(r_from, np.array([[0], [faf_threshold_distance + faf_iaf_distance]]))
    phi_to = (phi_from_runway + 180) % 360
    dir_rwy = np.dot(rot_matrix(phi_to), np.array([[0], [1]]))
    return {
        "x": float(x), "y": float(y), "h": float(h),
        "phi_from_runway": float(phi_from_runway), "phi_to_runway": float(phi_to),
        "faf": faf.ravel(), "iaf": iaf.ravel(), "corner1": corner1.ravel(), "corner2": corner2.ravel(),
        "normal": normal.ravel(), "dir_rwy": dir_rwy.ravel(), "faf_angle": float(faf_angle),
        "tri_h": close_ring([faf, corner1, corner2]),
        "tri_1": close_ring([faf, corner1, iaf]),
        "tri_2": close_ring([faf, corner2, iaf]),
    }


def _crossing_inside(x, y, ring):
    """Host crossing-number test with the reference's inequality set (model.py:318-337); construction-time use
    only (faf_mva, grid cell classification)."""
    n = len(ring)
    inside = False
    p1x, p1y = ring[0]
    for i in range(n + 1):
        p2x, p2y = ring[i % n]
        if y > min(p1y, p2y) and y <= max(p1y, p2y) and x <= max(p1x, p2x):
            if p1y != p2y:
                xints = (y - p1y) * (p2x - p1x) / (p2y - p1y) + p1x
            if p1x == p2x or x <= xints:
                inside = not inside
        p1x, p1y = p2x, p2y
    return inside


def _first_polygon(x, y, rings, bounds):
    for i, (ring, b) in enumerate(zip(rings, bounds)):
        if b[0] <= x <= b[2] and b[1] <= y <= b[3] and _crossing_inside(x, y, ring):
            return i
    return -1


def _first_polygon_many(X, Y, rings, bounds):
    """_first_polygon for arrays of points (the same float64 operations, vectorised over the points): the clean cells of a fine
    lookup grid are hundreds of thousands of centre points."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    res = np.full(X.shape, -1, dtype=np.int64)
    for i, (ring, b) in enumerate(zip(rings, bounds)):
        todo = np.nonzero((res < 0) & (b[0] <= X) & (X <= b[2]) & (b[1] <= Y) & (Y <= b[3]))[0]
        if len(todo) == 0:
            continue
        x, y = X[todo], Y[todo]
        inside = np.zeros(len(todo), dtype=bool)
        n = len(ring)
        p1x, p1y = ring[0]
        for k in range(n + 1):
            p2x, p2y = ring[k % n]
            if p1y != p2y:   # (an edge with p1y == p2y fails y > min or y <= max for every y)
                cond = (y > min(p1y, p2y)) & (y <= max(p1y, p2y)) & (x <= max(p1x, p2x))
                xints = (y - p1y) * (p2x - p1x) / (p2y - p1y) + p1x
                inside ^= cond & ((p1x == p2x) | (x <= xints))
            p1x, p1y = p2x, p2y
        res[todo[inside]] = i
    return res


def _seg_near_box(p, q, bx0, by0, bx1, by1):
    """True if segment p-q may touch the (already inflated) box — conservative: segment bbox overlap + separating axis
    along the segment normal."""
    if max(p[0], q[0]) < bx0 or min(p[0], q[0]) > bx1 or max(p[1], q[1]) < by0 or min(p[1], q[1]) > by1:
        return False
    dx, dy = q[0] - p[0], q[1] - p[1]
    if dx == 0.0 and dy == 0.0:
        return True
    s = [(cx - p[0]) * dy - (cy - p[1]) * dx for cx in (bx0, bx1) for cy in (by0, by1)]
    return not (min(s) > 0.0 or max(s) < 0.0)


def _ring_edges(ring):
    """Edges the reference's ray_tracing visits (model.py:324-335): consecutive vertex pairs plus the wrap edge
    ring[n-1] -> ring[0]; zero-height edges can never be counted (y > min and y <= max) and are dropped."""
    n = len(ring)
    e = [(ring[k - 1], ring[k]) for k in range(1, n)] + [(ring[n - 1], ring[0])]
    return np.array([[a[0], a[1], b[0], b[1]] for a, b in e if a[1] != b[1]], dtype=np.float64).reshape(-1, 4)


class SectorTooLarge(ValueError):
    """The compiled sector does not fit the blob's formats (offsets are exactly representable fp32 numbers: < 2^24 words; the
    lookup grid indexes cells with 24-bit multiplies) — choose a coarser lookup grid.  AtcVecEnv(grid_cell='auto') does."""


GRID_EDGE_WORDS = 8      # edge: p1x, p1y, p2x, p2y | min(p1y,p2y), max(p1y,p2y), polygon height, 16 * polygon index + flags
#                          terminator: polygon bounds x0, y0, x1, y1 | 0, 0, polygon height, 16 * polygon index + flags
GRID_F_TERM = 1.0        # terminator record: parity (+ base) and bounds test decide the polygon now
GRID_F_CERTAIN = 2.0     # every point of the cell lies left of this edge: crossing needs no intersection test
GRID_F_LAST = 4.0        # last edge of its polygon, whose bounds contain the whole cell: parity (+ base) decides now
GRID_F_BASE = 8.0        # the polygon has an ODD number of edges that every point of the cell crosses (folded away)
_BIG = 3.0e38


def _box_records(cx0s, cx1s, cy0s, cy1s, edges, bounds, heights):
    """Edge-list of one (slack-inflated) box, polygon by polygon in priority order — see build_grid.  [] = no point of the box
    is inside any polygon; a single record with -_BIG bounds = every point is inside that polygon."""
    recs = []
    for pi, (e, b) in enumerate(zip(edges, bounds)):
        if b[2] < cx0s or b[0] > cx1s or b[3] < cy0s or b[1] > cy1s:
            continue  # the (inclusive) bounds test of model.py:286 rejects every point of the cell
        ymin = np.minimum(e[:, 1], e[:, 3])
        ymax = np.maximum(e[:, 1], e[:, 3])
        xmin = np.minimum(e[:, 0], e[:, 2])
        xmax = np.maximum(e[:, 0], e[:, 2])
        rel = (cy1s > ymin) & (cy0s <= ymax) & (cx0s <= xmax) & (ymax > ymin)   # (a horizontal edge is never counted)
        # Where does the box lie relative to the edge's LINE over the part of the edge's y-range it can reach?  (Round 3; the
        # edge's bounding box was used before: every cell inside the bounding box of a long diagonal edge was dirty.)  The
        # x-intersection of model.py:331 is linear in y: its extremes over the clipped y-interval are at the interval's ends.
        with np.errstate(divide="ignore", invalid="ignore"):
            slope = np.where(ymax > ymin, (e[:, 2] - e[:, 0]) / (e[:, 3] - e[:, 1]), 0.0)
        ya = np.clip(cy0s, ymin, ymax)
        yb = np.clip(cy1s, ymin, ymax)
        xa = e[:, 0] + (ya - e[:, 1]) * slope
        xb = e[:, 0] + (yb - e[:, 1]) * slope
        # Margin per edge: the kernel (and the fp32 oracle) evaluate the x-intersection in fp32 from fp32-rounded vertices —
        # (y - p1y) carries half an ulp of y and of p1y, and the division by (p2y - p1y) multiplies that by |dx / dy|: for a
        # near-horizontal long edge (a sliver: dy = 0.03, dx = 50 at y ~ 50 nm) 2.5e-3 nm, measured.  1e-3 nm covers every
        # ordinary edge (LOWW's worst: 2e-4) and the rounding of the position; the slope term covers the slivers.
        ulp32 = np.spacing(np.abs(e).max(axis=1).astype(np.float32)).astype(np.float64)
        margin = 1e-3 + 4.0 * ulp32 * (1.0 + np.abs(slope))
        right = cx1s < np.minimum(xa, xb) - margin      # whole box left of the line: x <= xints holds whatever the rounding
        rel &= ~(cx0s > np.maximum(xa, xb) + margin)    # whole box right of the line: x <= xints fails for every point
        const = rel & right & (cy0s > ymin) & (cy1s <= ymax)
        base = GRID_F_BASE if int(const.sum()) & 1 else 0.0
        idx = np.nonzero(rel & ~const)[0]
        inside_bounds = b[0] <= cx0s and cx1s <= b[2] and b[1] <= cy0s and cy1s <= b[3]
        if len(idx) == 0:
            if not base:
                continue              # no point of the cell is inside this polygon
            if inside_bounds:         # every point of the cell is inside it: nothing below it can be reached
                recs.append([-_BIG, -_BIG, _BIG, _BIG, 0.0, 0.0, heights[pi], 16.0 * pi + GRID_F_TERM + base])
                break
            recs.append([b[0], b[1], b[2], b[3], 0.0, 0.0, heights[pi], 16.0 * pi + GRID_F_TERM + base])
            continue
        certain = right[idx]
        order = np.argsort(certain, kind="stable")  # intersection-test edges first: lanes diverge less
        for pos, k in enumerate(order):
            ek = e[idx[k]]
            fl = GRID_F_CERTAIN if certain[k] else 0.0
            if inside_bounds and pos == len(order) - 1:
                fl += GRID_F_LAST + base
            recs.append([ek[0], ek[1], ek[2], ek[3], min(ek[1], ek[3]), max(ek[1], ek[3]), heights[pi], 16.0 * pi + fl])
        if not inside_bounds:
            recs.append([b[0], b[1], b[2], b[3], 0.0, 0.0, heights[pi], 16.0 * pi + GRID_F_TERM + base])
    return recs


def _seg_near_boxes(p, q, bx0, bx1, by0, by1):
    """_seg_near_box for a block of boxes: bx0 / bx1 are the (inflated) bounds of the block's columns, by0 / by1 of its rows.
    Returns a [rows, columns] boolean array — the same float64 operations as the scalar form, element by element."""
    bx0, bx1, by0, by1 = bx0[None, :], bx1[None, :], by0[:, None], by1[:, None]
    miss = (max(p[0], q[0]) < bx0) | (min(p[0], q[0]) > bx1) | (max(p[1], q[1]) < by0) | (min(p[1], q[1]) > by1)
    dx, dy = q[0] - p[0], q[1] - p[1]
    if dx == 0.0 and dy == 0.0:
        return ~miss
    s = [(cx - p[0]) * dy - (cy - p[1]) * dx for cx in (bx0, bx1) for cy in (by0, by1)]
    lo = np.minimum(np.minimum(s[0], s[1]), np.minimum(s[2], s[3]))
    hi = np.maximum(np.maximum(s[0], s[1]), np.maximum(s[2], s[3]))
    return ~miss & ~((lo > 0.0) | (hi < 0.0))


def _box_records_many(cx0s, cx1s, cy0s, cy1s, edges, bounds, heights):
    """_box_records for MANY (slack-inflated) boxes at once: the per-edge classification — the expensive part — is evaluated for
    all boxes x all edges of a polygon in array operations (the same float64 arithmetic, element by element), and only the
    assembly of each box's record list stays a Python loop.  Returns one record list per box, identical to _box_records'."""
    n = len(cx0s)
    X0, X1, Y0, Y1 = cx0s[:, None], cx1s[:, None], cy0s[:, None], cy1s[:, None]
    per_poly = []
    for e, b in zip(edges, bounds):
        skip = (b[2] < cx0s) | (b[0] > cx1s) | (b[3] < cy0s) | (b[1] > cy1s)
        if len(e) == 0 or skip.all():
            per_poly.append((skip, None, None, None, None))
            continue
        ymin = np.minimum(e[:, 1], e[:, 3])
        ymax = np.maximum(e[:, 1], e[:, 3])
        xmax = np.maximum(e[:, 0], e[:, 2])
        rel = (Y1 > ymin) & (Y0 <= ymax) & (X0 <= xmax) & (ymax > ymin)
        with np.errstate(divide="ignore", invalid="ignore"):
            slope = np.where(ymax > ymin, (e[:, 2] - e[:, 0]) / (e[:, 3] - e[:, 1]), 0.0)
        ya = np.clip(Y0, ymin, ymax)
        yb = np.clip(Y1, ymin, ymax)
        xa = e[:, 0] + (ya - e[:, 1]) * slope
        xb = e[:, 0] + (yb - e[:, 1]) * slope
        ulp32 = np.spacing(np.abs(e).max(axis=1).astype(np.float32)).astype(np.float64)
        margin = 1e-3 + 4.0 * ulp32 * (1.0 + np.abs(slope))
        right = X1 < np.minimum(xa, xb) - margin
        rel &= ~(X0 > np.maximum(xa, xb) + margin)
        const = rel & right & (Y0 > ymin) & (Y1 <= ymax)
        odd = (const.sum(axis=1) & 1).astype(bool)
        listed = rel & ~const
        inside_bounds = (b[0] <= cx0s) & (cx1s <= b[2]) & (b[1] <= cy0s) & (cy1s <= b[3])
        per_poly.append((skip, listed, right, odd, inside_bounds))
    out = []
    for d in range(n):
        recs = []
        for pi, (skip, listed, right, odd, inside_b) in enumerate(per_poly):
            if skip[d]:
                continue
            e, b = edges[pi], bounds[pi]
            base = GRID_F_BASE if (listed is not None and odd[d]) else 0.0
            idx = np.nonzero(listed[d])[0] if listed is not None else ()
            inside_bounds = bool(inside_b[d]) if listed is not None else \
                (b[0] <= cx0s[d] and cx1s[d] <= b[2] and b[1] <= cy0s[d] and cy1s[d] <= b[3])
            if len(idx) == 0:
                if not base:
                    continue
                if inside_bounds:
                    recs.append([-_BIG, -_BIG, _BIG, _BIG, 0.0, 0.0, heights[pi], 16.0 * pi + GRID_F_TERM + base])
                    break
                recs.append([b[0], b[1], b[2], b[3], 0.0, 0.0, heights[pi], 16.0 * pi + GRID_F_TERM + base])
                continue
            certain = right[d][idx]
            order = np.argsort(certain, kind="stable")
            last = len(order) - 1
            for pos, k in enumerate(order):
                ek = e[idx[k]]
                fl = GRID_F_CERTAIN if certain[k] else 0.0
                if inside_bounds and pos == last:
                    fl += GRID_F_LAST + base
                recs.append([ek[0], ek[1], ek[2], ek[3], min(ek[1], ek[3]), max(ek[1], ek[3]), heights[pi], 16.0 * pi + fl])
            if not inside_bounds:
                recs.append([b[0], b[1], b[2], b[3], 0.0, 0.0, heights[pi], 16.0 * pi + GRID_F_TERM + base])
        out.append(recs)
    return out


GRID_CELL_LINE = 2.0 ** 23   # cell flag: the cell's first record is a LINE record (see _line_split)


def _line_split(recs, cy0s, cy1s, heights):
    """SPLIT cell: every edge record of the cell whose crossing depends on the point lies on ONE line that runs through the whole
    cell (no vertex, no second border inside it) — by far the commonest dirty cell.  Then a point more than a margin m to the
    LEFT of the line crosses all of those edges and a point more than m to the RIGHT crosses none, whatever the rounding of the
    reference's x-intersection (model.py:331-333): each side has ONE answer, found here by running the cell's record list with
    the crossing forced.  Returns the LINE record [p1x, p1y, dx/dy, m | left polygon + 1, left height, right polygon + 1,
    right height] or None.  The kernel evaluates  xl = p1x + (y - p1y) * dx/dy  and answers  x < xl - m -> left,
    x > xl + m -> right;  only points inside the band walk the ordinary records (which follow the LINE record unchanged)."""
    line = None
    m = 0.0
    for r in recs:
        fl = int(r[7]) % 16
        if fl & int(GRID_F_TERM):
            if r[0] != -_BIG:
                return None            # polygon bounds cut the cell: the answer depends on more than the side of the line
            continue
        if not (r[4] < cy0s and r[5] >= cy1s):
            return None                # the edge's y tests (model.py:328-329) do not hold for every point of the cell
        if fl & int(GRID_F_CERTAIN):
            continue                   # crossed by every point of the cell
        p1x, p1y, p2x, p2y = r[0:4]
        slope = (p2x - p1x) / (p2y - p1y)
        if line is None:
            line = (p1x, p1y, slope)
        else:                          # the same geometric line (a border shared by two polygons)?
            for qx, qy in ((p1x, p1y), (p2x, p2y)):
                if abs(line[0] + (qy - line[1]) * line[2] - qx) > 1e-9 * max(1.0, abs(qx)):
                    return None
        ulp32 = float(np.spacing(np.float32(max(abs(p1x), abs(p1y), abs(p2x), abs(p2y)))))
        # fp32 x-intersection of the reference's formula (1.5 ulp(y) |dx/dy| + ulp(x)), the kernel's own line evaluation with
        # an fp32 slope (6e-8 |y - p1y| |dx/dy|, |y - p1y| < 170 nm) and the collinearity tolerance, with room to spare
        m = max(m, 2e-3 + 8.0 * ulp32 * (1.0 + abs(slope)) + 2e-5 * abs(slope))
    if line is None:
        return None

    def side(crossing):
        inside = False
        for r in recs:
            pi, fl = int(r[7]) // 16, int(r[7]) % 16
            if fl & int(GRID_F_TERM):
                decide = True
            else:
                inside ^= True if (fl & int(GRID_F_CERTAIN)) else crossing
                decide = bool(fl & int(GRID_F_LAST))
            if decide:
                if inside != bool(fl & int(GRID_F_BASE)):
                    return pi + 1.0, heights[pi]
                inside = False
        return 0.0, 0.0
    left, right = side(True), side(False)
    return [line[0], line[1], line[2], m, left[0], left[1], right[0], right[1]]


def build_grid(rings, bounds, heights, bbox, cell, guard=None, noise_bounds=(), corridor_bounds=None):
    """Uniform lookup grid for Airspace.find_mva (model.py:282-289) with IDENTICAL results to the ordered polygon scan.

    * CLEAN cell: every point of the cell has the same answer; stored directly.
    * DIRTY cell: an edge list, polygon by polygon in priority order.  For one polygon and one cell an edge of the ring is
        - IRRELEVANT if one of the reference's three cheap tests (y > min(p1y,p2y), y <= max(p1y,p2y), x <= max(p1x,p2x),
          model.py:328-330) fails for EVERY point of the cell: it can never be counted and is dropped;
        - CONSTANT if all three hold for every point of the cell and the edge lies more than 1e-3 nm to the right of it
          (then x <= x-intersection holds whatever the rounding): every point crosses it exactly once — only the PARITY of
          the number of such edges matters and is folded into the polygon's BASE flag;
        - CERTAIN if it lies more than 1e-3 nm to the right of the cell but its y-range covers the cell only partly: listed,
          evaluated with the two y tests only;
        - otherwise listed and evaluated with the reference's own formula (model.py:331-333).
      Crossing parity of the listed edges xor BASE = result of ray_tracing over the full ring, for every point of the cell.
      A polygon whose bounds (the inclusive test of model.py:286) contain the whole cell ends with its last edge (LAST);
      otherwise a terminator record carries the bounds.  A polygon with no listed edge is dropped when BASE is even (no
      point of the cell is inside it), and ends the list when BASE is odd and its bounds contain the cell (every point of
      the cell is inside it: lower-priority polygons can never be reached) — if it is the first entry the cell is clean.
    Cell bounds are inflated by `slack` (guard + fp32 indexing error) so that a point the device bins into a neighbouring
    cell because of fp32 rounding is still covered.

    Noise-abatement areas (extension): every cell also carries the bit mask of the noise polygons whose bounds meet the
    (inflated) cell — the step kernel tests an aircraft only against those (usually none) instead of every noise polygon.

    Corridor candidate: bit 22 of |c| marks the cells that meet the bounds of the corridor's horizontal triangle (the exact
    early-out of Runway.inside_corridor, model.py:198) — the step kernel asks the cell it has already fetched instead of
    comparing four bounds per aircraft.

    Layout (words): header[8] = x0, y0, inv_cell, nx, ny, offset of the edge pool (from grid start), n_records, 0;
    cells[ny*nx][2] = (c, first_record | MVA height) with |c| = code + 64 * noise mask + 2^22 * corridor candidate
    + 2^23 * split: c > 0 dirty cell, code = n_records (< 64; a split cell's LINE record comes first and is not counted);
    c <= 0 clean cell, code = polygon + 1 (0 = outside the airspace); pool of 8-word records."""
    if guard is None:
        guard = 1e-3
    x0, y0, x1, y1 = bbox
    # the grid covers the noise-abatement areas and the corridor triangle too: beyond it neither has to be tested
    for b in list(noise_bounds) + ([corridor_bounds] if corridor_bounds is not None else []):
        x0, y0, x1, y1 = min(x0, b[0]), min(y0, b[1]), max(x1, b[2]), max(y1, b[3])
    # Two cells of padding all round: the OUTERMOST ring of cells is then clean, outside the airspace and free of noise-area
    # candidates (checked below) — the kernel clamps the cell index of a point beyond the grid into that ring instead of
    # testing the range (csrc/atc_device.h:mva_cell_load), so the ring must give the answer of "beyond the grid".
    gx0 = x0 - 2 * cell
    gy0 = y0 - 2 * cell
    nx = int(math.ceil((x1 - gx0) / cell)) + 2
    ny = int(math.ceil((y1 - gy0) / cell)) + 2
    if nx >= 1 << 20 or ny >= 1 << 20 or L.G_HDR + 2 * nx * ny >= 2 ** 24:
        raise SectorTooLarge("a %g nm lookup grid over this sector has %d x %d cells: beyond the blob's 2^24 words" % (cell, nx, ny))
    inv = 1.0 / cell
    slack = guard + 1e-4 * max(1.0, abs(x1), abs(y1)) * 2.0 ** -10
    assert slack < 0.5 * cell
    edges = [_ring_edges(r) for r in rings]
    near = np.zeros((ny, nx), dtype=bool)
    for ring in rings:
        for k in range(len(ring) - 1):
            p, q = ring[k], ring[k + 1]
            i0 = max(0, int(math.floor((min(p[0], q[0]) - slack - gx0) * inv)) - 1)
            i1 = min(nx - 1, int(math.floor((max(p[0], q[0]) + slack - gx0) * inv)) + 1)
            j0 = max(0, int(math.floor((min(p[1], q[1]) - slack - gy0) * inv)) - 1)
            j1 = min(ny - 1, int(math.floor((max(p[1], q[1]) + slack - gy0) * inv)) + 1)
            if i1 < i0 or j1 < j0:
                continue
            ii, jj = np.arange(i0, i1 + 1, dtype=np.float64), np.arange(j0, j1 + 1, dtype=np.float64)
            near[j0:j1 + 1, i0:i1 + 1] |= _seg_near_boxes(p, q, gx0 + ii * cell - slack, gx0 + (ii + 1) * cell + slack,
                                                          gy0 + jj * cell - slack, gy0 + (jj + 1) * cell + slack)
    cells = np.zeros((ny, nx, 2), dtype=np.float64)
    pool = []
    # clean cells (no edge near): the answer of the centre point, all of them at once
    jj, ii = np.nonzero(~near)
    pis = _first_polygon_many(gx0 + (ii + 0.5) * cell, gy0 + (jj + 0.5) * cell, rings, bounds)
    hit = pis >= 0
    cells[jj[hit], ii[hit], 0] = -(pis[hit] + 1.0)
    cells[jj[hit], ii[hit], 1] = np.asarray(heights, dtype=np.float64)[pis[hit]]
    # cells an edge passes near (row-major order = the order of their records in the pool): classified all at once
    dj, di = np.nonzero(near)
    dcy0, dcy1 = gy0 + dj * cell - slack, gy0 + (dj + 1) * cell + slack
    all_recs = _box_records_many(gx0 + di * cell - slack, gx0 + (di + 1) * cell + slack, dcy0, dcy1, edges, bounds, heights)
    for j, i, cy0s, cy1s, recs in zip(dj, di, dcy0, dcy1, all_recs):
        if len(recs) == 1 and recs[0][0] == -_BIG:   # a single unconditional answer: the cell is clean after all
            cells[j, i, 0] = -(recs[0][7] // 16 + 1.0)
            cells[j, i, 1] = recs[0][6]
            continue
        cells[j, i, 0] = len(recs)
        cells[j, i, 1] = len(pool) if recs else 0.0   # (0, 0): nothing can match = clean cell outside the airspace
        split = _line_split(recs, cy0s, cy1s, heights) if recs else None
        if split is not None:      # LINE record first, the ordinary records behind it
            cells[j, i, 0] += GRID_CELL_LINE
            pool.append(split)
        pool.extend(recs)
    assert len(noise_bounds) <= 16, "at most 16 noise-abatement areas"
    assert (np.maximum(cells[:, :, 0], 0.0) % GRID_CELL_LINE).max() < 64
    # candidate masks: which noise-area bounds / the corridor triangle's bounds meet each (inflated) cell — rows x columns
    ai, aj = np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64)
    acx0, acx1 = gx0 + ai * cell - slack, gx0 + (ai + 1) * cell + slack
    acy0, acy1 = gy0 + aj * cell - slack, gy0 + (aj + 1) * cell + slack
    mask = np.zeros((ny, nx), dtype=np.float64)
    for bit, b in [(q, b) for q, b in enumerate(noise_bounds)] + ([(16, corridor_bounds)] if corridor_bounds is not None else []):
        meets = ~((b[3] < acy0) | (b[1] > acy1))[:, None] & ~((b[2] < acx0) | (b[0] > acx1))[None, :]   # (16: bit 22 of |c|)
        mask += np.where(meets, float(1 << bit), 0.0)
    cells[:, :, 0] = np.where(cells[:, :, 0] > 0, cells[:, :, 0] + 64.0 * mask, cells[:, :, 0] - 64.0 * mask)
    border = np.concatenate([cells[0, :, :].ravel(), cells[-1, :, :].ravel(), cells[:, 0, :].ravel(), cells[:, -1, :].ravel()])
    assert not border.any(), "the outermost ring of lookup cells must be clean, outside the airspace, without noise candidates"
    n_rec = len(pool)
    hdr = np.zeros(L.G_HDR, dtype=np.float64)
    hdr[L.G_X0], hdr[L.G_Y0], hdr[L.G_INV], hdr[L.G_NX], hdr[L.G_NY] = gx0, gy0, inv, nx, ny
    off_pool = (L.G_HDR + cells.size + 3) & ~3  # edge records are read as 16-byte vectors
    hdr[L.G_OFF_POOL] = off_pool
    hdr[L.G_NREC] = n_rec
    if not (n_rec < 2 ** 21 and off_pool + n_rec * GRID_EDGE_WORDS < 2 ** 24):
        raise SectorTooLarge("a %g nm lookup grid over this sector needs %d edge records: beyond the blob's 2^24 words" % (cell, n_rec))
    pool_arr = np.asarray(pool, dtype=np.float64).reshape(-1, GRID_EDGE_WORDS)
    pad = np.zeros(off_pool - L.G_HDR - cells.size)
    return np.concatenate([hdr, cells.ravel(), pad, pool_arr.ravel()])


# ---------------------------------------------------------------------------------------------------------------------
# LDS-resident lookup table (ABI 21; include/atc_step.h: atc_scenario_attach_lds_table) for the latency-bound multi-step launches
# of ONE-aircraft envs — 64 envs per wavefront, one wavefront per SIMD: the lookup grid's gather is 0.8 us of exposed wait per
# step there and its dirty cells a second / third dependent trip in 40 % of the wavefront-steps (profiles/experiments/README.md,
# round 6).  Same answers as the ordered polygon scan (model.py:282-289), by the same construction as build_grid:
#   level 1: cells of `cell` nm (0.5) over the padded sector, one 16-bit code each;
#   level 2: every level-1 cell that is neither clean nor split by ONE border line is refined into sub x sub (8 x 8) sub-cells,
#            one 16-bit code each;
#   code   : bit 15 = corridor candidate (level 1 only), bits 14..13 = kind, bits 12..0 = payload
#            CLEAN (0): payload = polygon + 1 (0: outside the airspace);  LINE (1): payload = index of a LINE record (_line_split:
#            p1x, p1y, dx/dy, margin | left polygon + 1, height, right polygon + 1, height);  SUB (2, level 1 only): payload = index
#            of the cell's sub-cell block;  RESIDUAL (3): a vertex or a second border inside the sub-cell — the kernel answers the
#            WHOLE wavefront from the global lookup grid (0.2 % of the sub-cells' area: one wavefront-step in eight).
# A point inside a LINE record's margin band is residual too.  Boxes are inflated by the same slack as build_grid's.
LDS_KIND_CLEAN, LDS_KIND_LINE, LDS_KIND_SUB, LDS_KIND_RESID = L.LDS_CLEAN, L.LDS_LINE, L.LDS_SUB, L.LDS_RESID
LDS_MAGIC, LDS_HDR_WORDS = L.LDS_MAGIC, L.LDS_HDR_WORDS


def _near_any_edge(rings, x0s, x1s, y0s, y1s):
    """Boxes (already inflated) some ring edge may touch — _seg_near_box, vectorised over a flat list of boxes."""
    near = np.zeros(len(x0s), dtype=bool)
    for ring in rings:
        for k in range(len(ring) - 1):
            p, q = ring[k], ring[k + 1]
            miss = (max(p[0], q[0]) < x0s) | (min(p[0], q[0]) > x1s) | (max(p[1], q[1]) < y0s) | (min(p[1], q[1]) > y1s)
            dx, dy = q[0] - p[0], q[1] - p[1]
            if dx == 0.0 and dy == 0.0:
                near |= ~miss
                continue
            s = [(cx - p[0]) * dy - (cy - p[1]) * dx for cx in (x0s, x1s) for cy in (y0s, y1s)]
            lo = np.minimum(np.minimum(s[0], s[1]), np.minimum(s[2], s[3]))
            hi = np.maximum(np.maximum(s[0], s[1]), np.maximum(s[2], s[3]))
            near |= ~miss & ~((lo > 0.0) | (hi < 0.0))
    return near


def _classify_boxes(x0, x1, y0, y1, slack, rings, edges, bounds, heights, lines):
    """Kind and payload of each box [x0, x1] x [y0, y1] (inflated by `slack` here): CLEAN / LINE / RESID (a level-1 caller turns
    RESID into SUB).  `lines`: dict LINE record (8 float32 words as bytes) -> index, extended in place.  Also returns the record
    list (build_grid's edge records, _box_records) of every RESID box."""
    x0s, x1s, y0s, y1s = x0 - slack, x1 + slack, y0 - slack, y1 + slack
    n = len(x0)
    kind = np.zeros(n, dtype=np.int64)
    payload = np.zeros(n, dtype=np.int64)
    walks = {}
    near = _near_any_edge(rings, x0s, x1s, y0s, y1s)
    far = np.nonzero(~near)[0]
    pis = _first_polygon_many(0.5 * (x0[far] + x1[far]), 0.5 * (y0[far] + y1[far]), rings, bounds)
    payload[far] = pis + 1
    idx = np.nonzero(near)[0]
    recs_all = _box_records_many(x0s[idx], x1s[idx], y0s[idx], y1s[idx], edges, bounds, heights)
    for d, recs in zip(idx, recs_all):
        if not recs:
            continue                                     # no point of the box is inside any polygon: CLEAN, outside
        if len(recs) == 1 and recs[0][0] == -_BIG:       # a single unconditional answer
            payload[d] = int(recs[0][7]) // 16 + 1
            continue
        split = _line_split(recs, y0s[d], y1s[d], heights)
        if split is None:
            kind[d] = LDS_KIND_RESID
            walks[int(d)] = recs
            continue
        key = np.asarray(split, dtype=np.float32).tobytes()
        if key not in lines:
            lines[key] = len(lines)
        kind[d] = LDS_KIND_LINE
        payload[d] = lines[key]
    return kind, payload, walks


def build_lds_table(rings, bounds, heights, bbox, corridor_bounds=None, cell=0.5, sub=8, guard=None):
    """The LDS-resident lookup table (see above) as a uint8 array, or None when the sector does not fit its 13-bit payloads.
    Layout (include/atc_step.h: ATC_LDS_H_*): 24 header words (offsets in bytes), then — the part a workgroup stages in LDS —
    level-1 codes u16[ny * nx], sub-cell codes u16[n_sub][sub * sub], LINE records f32[n_line][8], heights f32[64] (index
    polygon + 1; [0] = 0), the RESIDUAL sub-cells' walk words u32[n_resid] (first record | n_records << 24); behind it, read from
    global memory: the pool of those sub-cells' edge records f32[n_rec][8] (build_grid's format, walked by mva_walk)."""
    if guard is None:
        guard = 1e-3
    assert sub == 8, "the kernel's sub-cell index is three bits per axis"
    x0, y0, x1, y1 = bbox
    if corridor_bounds is not None:
        b = corridor_bounds
        x0, y0, x1, y1 = min(x0, b[0]), min(y0, b[1]), max(x1, b[2]), max(y1, b[3])
    gx0, gy0 = x0 - 2 * cell, y0 - 2 * cell
    nx = int(math.ceil((x1 - gx0) / cell)) + 2
    ny = int(math.ceil((y1 - gy0) / cell)) + 2
    if nx >= 1 << 11 or ny >= 1 << 11 or len(rings) > 62:
        return None
    slack = guard + 1e-4 * max(1.0, abs(x1), abs(y1)) * 2.0 ** -10
    assert slack < 0.5 * cell / sub
    edges = [_ring_edges(r) for r in rings]
    lines = {}
    jj, ii = np.divmod(np.arange(nx * ny), nx)
    k1, p1, _ = _classify_boxes(gx0 + ii * cell, gx0 + (ii + 1) * cell, gy0 + jj * cell, gy0 + (jj + 1) * cell, slack,
                                rings, edges, bounds, heights, lines)
    resid = np.nonzero(k1 == LDS_KIND_RESID)[0]
    n_sub = len(resid)
    sj, si = np.divmod(np.arange(sub * sub), sub)
    sc = cell / sub
    bx = (gx0 + ii[resid] * cell)[:, None] + si[None, :] * sc     # sub-cell s of block r: [bx, bx + sc] x [by, by + sc]
    by = (gy0 + jj[resid] * cell)[:, None] + sj[None, :] * sc
    k2, p2, walks = _classify_boxes(bx.ravel(), bx.ravel() + sc, by.ravel(), by.ravel() + sc, slack, rings, edges, bounds,
                                    heights, lines)
    k1[resid] = LDS_KIND_SUB
    p1[resid] = np.arange(n_sub)
    # RESIDUAL sub-cells: their edge records go to a pool in global memory, one walk word each stays in LDS
    pool, walk_words = [], []
    for d in sorted(walks):
        recs = walks[d]
        if len(recs) >= 64:
            return None
        p2[d] = len(walk_words)
        walk_words.append(len(pool) | (len(recs) << 24))
        pool.extend(recs)
    if n_sub >= 1 << 13 or len(lines) >= 1 << 13 or len(walk_words) >= 1 << 13 or len(pool) >= 1 << 24:
        return None
    code1 = (k1 << 13) | p1
    if corridor_bounds is not None:   # the (inflated) cell meets the bounds of the corridor's horizontal triangle (model.py:198)
        b = corridor_bounds
        cx0, cx1 = gx0 + ii * cell - slack, gx0 + (ii + 1) * cell + slack
        cy0, cy1 = gy0 + jj * cell - slack, gy0 + (jj + 1) * cell + slack
        code1 |= np.where(~((b[3] < cy0) | (b[1] > cy1) | (b[2] < cx0) | (b[0] > cx1)), 1 << 15, 0)
    code2 = (k2 << 13) | p2
    c1 = code1.reshape(ny, nx)
    border = np.concatenate([c1[0, :], c1[-1, :], c1[:, 0], c1[:, -1]])
    assert not border.any(), "the outermost ring of level-1 cells must be clean and outside the airspace"
    line_arr = np.zeros((max(1, len(lines)), 8), dtype=np.float32)
    for key, i in lines.items():
        line_arr[i] = np.frombuffer(key, dtype=np.float32)
    hts = np.zeros(64, dtype=np.float32)
    hts[1:1 + len(heights)] = np.asarray(heights, dtype=np.float32)
    walk_arr = np.asarray(walk_words, dtype=np.uint32)
    pool_arr = np.asarray(pool, dtype=np.float32).reshape(-1, GRID_EDGE_WORDS)

    def pad16(n):
        return (n + 15) & ~15
    off_l1 = 4 * LDS_HDR_WORDS
    off_sub = pad16(off_l1 + 2 * nx * ny)
    off_line = pad16(off_sub + 2 * len(code2))
    off_hts = off_line + line_arr.nbytes
    off_resid = off_hts + hts.nbytes
    lds_bytes = pad16(off_resid + walk_arr.nbytes)
    off_pool = lds_bytes
    total = pad16(off_pool + pool_arr.nbytes)
    t = np.zeros(total, dtype=np.uint8)
    hdr = np.zeros(LDS_HDR_WORDS, dtype=np.uint32)
    hdr[L.LDS_H_MAGIC], hdr[L.LDS_H_BYTES] = LDS_MAGIC, total
    hdr[L.LDS_H_X0:L.LDS_H_X0 + 3] = np.array([gx0, gy0, 1.0 / cell], dtype=np.float32).view(np.uint32)
    for k, v in ((L.LDS_H_NX, nx), (L.LDS_H_NY, ny), (L.LDS_H_OFF_L1, off_l1), (L.LDS_H_OFF_SUB, off_sub), (L.LDS_H_N_SUB, n_sub),
                 (L.LDS_H_OFF_LINE, off_line), (L.LDS_H_N_LINE, len(line_arr)), (L.LDS_H_OFF_HTS, off_hts), (L.LDS_H_SUB, sub),
                 (L.LDS_H_OFF_RESID, off_resid), (L.LDS_H_N_RESID, len(walk_arr)), (L.LDS_H_LDS_BYTES, lds_bytes),
                 (L.LDS_H_OFF_POOL, off_pool), (L.LDS_H_N_REC, len(pool_arr))):
        hdr[k] = v
    t[:off_l1] = hdr.view(np.uint8)
    t[off_l1:off_l1 + 2 * nx * ny] = code1.astype(np.uint16).view(np.uint8)
    t[off_sub:off_sub + 2 * len(code2)] = code2.astype(np.uint16).view(np.uint8)
    t[off_line:off_line + line_arr.nbytes] = line_arr.ravel().view(np.uint8)
    t[off_hts:off_hts + hts.nbytes] = hts.view(np.uint8)
    t[off_resid:off_resid + walk_arr.nbytes] = walk_arr.view(np.uint8)
    t[off_pool:off_pool + pool_arr.nbytes] = pool_arr.ravel().view(np.uint8)
    return t


def _walk_records_f32(recs, x, y):
    """mva_walk (csrc/atc_device.h) for ONE point, the same fp32 operations: (polygon index or -1, height)."""
    f32 = np.float32
    inside = False
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        for r in recs:
            code = int(r[7])
            ok = True
            if code & 1:     # terminator: polygon bounds
                decide = True
                ok = bool(r[0] <= x <= r[2] and r[1] <= y <= r[3])
            else:
                if y > r[4] and y <= r[5] and x <= max(r[0], r[2]):
                    cross = bool(code & 2)
                    if not cross:
                        xints = f32(f32(f32(f32(y - r[1]) * f32(r[2] - r[0])) / f32(r[3] - r[1])) + r[0])
                        cross = bool(r[0] == r[2] or x <= xints)
                    inside = inside != cross
                decide = bool(code & 4)
            if decide:
                if inside != bool(code & 8) and ok:
                    return code >> 4, r[6]
                inside = False
    return -1, f32(0)


def lds_table_lookup(table, x, y):
    """numpy restatement of the kernel's LDS lookup (csrc/atc_device.h: lds_cell_load / lds_resolve + the walk of a RESIDUAL
    sub-cell's records), the same fp32 operations: returns (polygon + 1 — or -1 where the kernel sends the wavefront to the
    lookup grid: a point inside a LINE record's margin band —, height, corridor candidate, answered-by-a-walk)."""
    f32 = np.float32
    t = np.asarray(table, dtype=np.uint8)
    hdr = t[:4 * LDS_HDR_WORDS].view(np.uint32)
    assert hdr[L.LDS_H_MAGIC] == LDS_MAGIC and hdr[L.LDS_H_BYTES] == len(t)
    gx0, gy0, inv = hdr[L.LDS_H_X0:L.LDS_H_X0 + 3].view(np.float32)
    nx, ny, off_l1, off_sub, n_sub, off_line, n_line, off_hts, sub, off_resid, n_resid, off_pool, n_rec = (
        int(hdr[k]) for k in (L.LDS_H_NX, L.LDS_H_NY, L.LDS_H_OFF_L1, L.LDS_H_OFF_SUB, L.LDS_H_N_SUB, L.LDS_H_OFF_LINE, L.LDS_H_N_LINE,
                              L.LDS_H_OFF_HTS, L.LDS_H_SUB, L.LDS_H_OFF_RESID, L.LDS_H_N_RESID, L.LDS_H_OFF_POOL, L.LDS_H_N_REC))
    l1 = t[off_l1:off_l1 + 2 * nx * ny].view(np.uint16)
    l2 = t[off_sub:off_sub + 2 * n_sub * sub * sub].view(np.uint16) if n_sub else np.zeros(64, dtype=np.uint16)
    lines = t[off_line:off_line + 32 * n_line].view(np.float32).reshape(-1, 8)
    hts = t[off_hts:off_hts + 256].view(np.float32)
    walk_words = t[off_resid:off_resid + 4 * n_resid].view(np.uint32)
    pool = t[off_pool:off_pool + 32 * n_rec].view(np.float32).reshape(-1, 8)
    x = np.asarray(x, dtype=f32)
    y = np.asarray(y, dtype=f32)

    def cvt_i32_sat(v):   # v_cvt_i32_f32: truncation, saturation, NaN -> 0
        v = np.nan_to_num(v.astype(np.float64), nan=0.0, posinf=2147483647.0, neginf=-2147483648.0)
        return np.clip(np.trunc(v), -2147483648.0, 2147483647.0).astype(np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        fx = (x - gx0) * inv
        fy = (y - gy0) * inv
        ix = np.minimum(cvt_i32_sat(fx) & 0xffffffff, nx - 1)
        iy = np.minimum(cvt_i32_sat(fy) & 0xffffffff, ny - 1)
        sx = np.minimum(cvt_i32_sat((fx - ix.astype(f32)) * f32(sub)) & 0xffffffff, sub - 1)
        sy = np.minimum(cvt_i32_sat((fy - iy.astype(f32)) * f32(sub)) & 0xffffffff, sub - 1)
    c1 = l1[iy * nx + ix].astype(np.int64)
    cand = (c1 >> 15) & 1
    is_sub = ((c1 >> 13) & 3) == LDS_KIND_SUB
    c2 = l2[np.where(is_sub, (c1 & 0x1fff) * sub * sub + sy * sub + sx, 0)].astype(np.int64)
    c = np.where(is_sub, c2, c1)
    kind, pay = (c >> 13) & 3, c & 0x1fff
    rec = lines[np.where(kind == LDS_KIND_LINE, pay, 0)]
    with np.errstate(invalid="ignore", over="ignore"):
        # the kernel's fmaf: the product of two fp32 numbers is exact in float64, the sum is rounded once more — to fp32
        xl = ((y - rec[:, 1]).astype(np.float64) * rec[:, 2].astype(np.float64) + rec[:, 0].astype(np.float64)).astype(f32)
        left, right = x < xl - rec[:, 3], x > xl + rec[:, 3]
    decided = (kind == LDS_KIND_LINE) & (left | right)
    clean = kind == LDS_KIND_CLEAN
    code = np.where(clean, pay, np.where(left, rec[:, 4], rec[:, 6]).astype(np.int64))
    h = np.where(clean, hts[np.where(clean, pay, 0)], np.where(left, rec[:, 5], rec[:, 7])).astype(f32)
    ok = clean | decided
    walked = kind == LDS_KIND_RESID
    for i in np.nonzero(walked)[0]:
        w = int(walk_words[pay[i]])
        pi, hh = _walk_records_f32(pool[(w & 0xffffff):(w & 0xffffff) + (w >> 24)], x[i], y[i])
        code[i], h[i] = pi + 1, hh
    ok = ok | walked
    return np.where(ok, code, -1), np.where(ok, h, f32(0)), cand, walked


_SOURCE_TAG = None


def _grid_cache_path(args, kind="grid"):
    """File of the on-disk copy of one lookup grid (kind "grid") or of one whole device blob ("blob32"), or None when the cache is off.  The key is a sha256 over every input of
    build_grid (floats by their exact hex form), the blob / ABI versions and THIS FILE's source text — any change to the compiler
    retires every cached grid.  Directory: $ATC_HIP_CACHE ('' or '0': no cache), else $XDG_CACHE_HOME/atc_hip, else
    ~/.cache/atc_hip."""
    import hashlib
    import os
    global _SOURCE_TAG
    d = os.environ.get("ATC_HIP_CACHE")
    if d is None:
        d = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "atc_hip")
    if d in ("", "0"):
        return None
    if _SOURCE_TAG is None:
        with open(__file__, "rb") as f:
            _SOURCE_TAG = hashlib.sha256(f.read()).hexdigest()

    def canon(v):
        if isinstance(v, (list, tuple)):
            return "[" + ",".join(canon(x) for x in v) + "]"
        if isinstance(v, np.ndarray):
            return canon(v.tolist())
        return "None" if v is None else float(v).hex()
    key = hashlib.sha256(("%s|%s|%s|%s" % (_SOURCE_TAG, L.ABI_VERSION, L.BLOB_VERSION, canon(args))).encode()).hexdigest()
    return os.path.join(d, "%s_%s.npy" % (kind, key[:40]))


def _cache_save(path, arr):
    import os
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = "%s.%d.tmp.npy" % (path, os.getpid())
        np.save(tmp, arr)
        os.replace(tmp, path)   # atomic: concurrent workers (8 x SubprocVecEnv) never see a partial file
    except OSError:
        pass


def build_grid_cached(rings, bounds, heights, bbox, cell, guard=None, noise_bounds=(), corridor_bounds=None):
    """build_grid behind an on-disk cache (see _grid_cache_path): LOWW at 0.0625 nm — the grid `auto` selects for 65 536 x 1 and
    8 192 x 16 — takes 0.6 s to classify and 10 ms to load.  A cache that cannot be read or written is ignored."""
    import os
    path = _grid_cache_path([rings, bounds, heights, bbox, cell, guard, list(noise_bounds), corridor_bounds])
    if path is not None and os.path.exists(path):
        try:
            g = np.load(path, allow_pickle=False)
            if g.dtype == np.float64 and g.ndim == 1 and len(g) > L.G_HDR and g[L.G_INV] == 1.0 / cell:
                return g
        except (OSError, ValueError):
            pass
    g = build_grid(rings, bounds, heights, bbox, cell, guard, noise_bounds=noise_bounds, corridor_bounds=corridor_bounds)
    if path is not None:
        _cache_save(path, g)
    return g


def position_grid(points):
    """Fixed-point position grid of the fp32 path (include/atc_step.h, "Aircraft positions"): integer-valued origin at the
    centre of `points` and the largest k <= POS_MAX_K with 2^(31-k) nm >= 1.5 x the half extent.  Returns (x0, y0, k)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    x0 = float(np.rint(0.5 * (pts[:, 0].min() + pts[:, 0].max())))
    y0 = float(np.rint(0.5 * (pts[:, 1].min() + pts[:, 1].max())))
    half = max(1e-6, float(np.abs(pts - np.array([x0, y0])).max()))
    k = int(min(L.POS_MAX_K, math.floor(31 - math.log2(1.5 * half))))
    return x0, y0, k


def to_fix(value, origin, k):
    """nm -> grid counts (host placement of an aircraft: float64 arithmetic, round half even, saturating)."""
    c = np.rint((np.asarray(value, dtype=np.float64) - origin) * 2.0 ** k)
    return np.clip(c, -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64).astype(np.int32)


def from_fix(fix, origin, k):
    """grid counts -> nm (float64, exact)."""
    return np.asarray(fix, dtype=np.float64) * 2.0 ** -k + origin


class CompiledSector:
    """Result of compile_sector: `.blob32` (device copy), `.blob64` (float64 master: what the float64 test oracle reads) and the
    derived constants."""

    def __init__(self, blob64, meta, spawn_words=None, grid_fn=None, blob32=None):
        """spawn_words: the int32 image [n_records, SPAWN_WORDS] of the spawn records (include/atc_step.h: ATC_H_OFF_SPAWN) — their
        integer words are 32-bit PATTERNS in the device blob, not float values, so the device blob is built HERE, in the one
        place that can build it: a CompiledSector re-made from a float64 master alone (no spawn_words) refuses a master that
        has spawn records instead of silently shipping float-valued counts.
        grid_fn: the lookup grid's words are filled into the master on first use (`blob64`) — a process that only needs the
        device blob, and finds it in the on-disk cache (`blob32`), never touches the 2 x larger float64 image."""
        self._b64 = np.ascontiguousarray(blob64, dtype=np.float64)
        self._grid_fn = grid_fn
        self.meta = meta
        self.spawn_words = None if spawn_words is None else np.array(spawn_words, dtype=np.int32, copy=True)
        n_entry, off_spawn = int(self._b64[L.H_N_ENTRY]), int(self._b64[L.H_OFF_SPAWN])
        if n_entry and off_spawn and spawn_words is None:
            raise ValueError("a sector with entry points needs the spawn records' integer image (spawn_words): the float64 "
                             "master holds their counts as values, the device blob as 32-bit patterns")
        if blob32 is not None:
            self.blob32 = blob32
            return
        self.blob32 = np.ascontiguousarray(self.blob64.astype(np.float32))
        if n_entry and off_spawn:
            n_spawn = L.MAX_AIRCRAFT + n_entry
            sw = np.ascontiguousarray(spawn_words, dtype=np.int32).reshape(n_spawn, L.SPAWN_WORDS)
            self.blob32[off_spawn:off_spawn + n_spawn * L.SPAWN_WORDS].view(np.int32)[:] = sw.ravel()

    @property
    def blob64(self):
        if self._grid_fn is not None:
            og = int(self._b64[L.H_OFF_GRID])
            self._b64[og:] = self._grid_fn()
            self._grid_fn = None
        return self._b64

    def lds_table(self):
        """The LDS-resident lookup table of this sector (build_lds_table; built on first use, kept), or None: no MVA polygon, a
        noise-abatement area (the table's codes carry no candidate masks), or payloads beyond 13 bits."""
        if "_lds_table" not in self.__dict__:
            m = self.meta
            t = None
            if m["n_mva"] and not m["n_noise"]:
                th = m["corridor"]["tri_h"]
                t = build_lds_table(m["mva_rings"], m["mva_bounds"], m["mva_heights"], m["bbox"],
                                    (th[:, 0].min(), th[:, 1].min(), th[:, 0].max(), th[:, 1].max()))
            self.__dict__["_lds_table"] = t
        return self.__dict__["_lds_table"]

    def __getattr__(self, name):
        try:
            return self.__dict__["meta"][name]
        except KeyError:
            raise AttributeError(name)


def compile_sector(mvas, runway, entrypoints, noise=(), grid_cell=None, grid_guard=None):
    """mvas: [(points, height_ft)], runway: (x, y, h, phi_from_runway), entrypoints: [(x, y, phi, [levels])],
    noise: [(points, ceiling_ft, penalty_per_step)].  Returns CompiledSector."""
    assert len(noise) <= 16, "at most 16 noise-abatement areas"
    mva_rings = [close_ring(p) for p, _ in mvas]
    mva_heights = [float(hh) for _, hh in mvas]
    noise_rings = [close_ring(p) for p, _, _ in noise]
    rings = mva_rings + noise_rings
    heights = mva_heights + [float(c) for _, c, _ in noise]
    penalties = [0.0] * len(mva_rings) + [float(pen) for _, _, pen in noise]
    bounds = [(r[:, 0].min(), r[:, 1].min(), r[:, 0].max(), r[:, 1].max()) for r in rings]
    cg = corridor_geometry(*runway)

    if mva_rings:
        mb = bounds[:len(mva_rings)]
        bbox = (min(b[0] for b in mb), min(b[1] for b in mb), max(b[2] for b in mb), max(b[3] for b in mb))
    else:
        bbox = (0.0, 0.0, 1.0, 1.0)
    x_len = bbox[2] - bbox[0]
    y_len = bbox[3] - bbox[1]
    world_diag = float(np.hypot(x_len, y_len))
    fi = _first_polygon(cg["faf"][0], cg["faf"][1], mva_rings, bounds[:len(mva_rings)]) if mva_rings else -1
    faf_mva = mva_heights[fi] if fi >= 0 else 0.0

    v_min, v_max, h_min, h_max = 100.0, 300.0, 0.0, 38000.0
    norm_min = np.array([bbox[0], bbox[1], 0, 0, v_min, 0, 0, 0, -180, -180], dtype=np.float32)
    norm_max = np.array([x_len, y_len, h_max, 360, v_max - v_min, h_max, h_max, world_diag, 360, 360], dtype=np.float32)

    n_poly = len(rings)
    off_poly = L.C_END
    off_vert = off_poly + n_poly * L.P_WORDS
    n_vertw = 2 * sum(len(r) for r in rings)
    # alignment the device code relies on: polygon records are read as 16-byte vectors, ring vertices as 8-byte pairs
    assert off_poly % 4 == 0 and L.P_WORDS % 4 == 0 and off_vert % 2 == 0
    off_entry = off_vert + n_vertw
    n_entry = len(entrypoints)
    for ex, ey, ephi, _ in entrypoints:
        # the device's 32-bit heading field holds (-76, 436) deg; beyond it an aircraft is WIDE with its exact counts in a side word
        # that a reset does not write (include/atc_step.h, ABI 19).  The reference never validates an entry heading; every sector it
        # ships uses [0, 360).  Kinematics, corridor and relative angles are periodic, only the raw heading observation is not.
        if not -76.0 < float(ephi) < 436.0:
            raise ValueError("entry point heading %r outside (-76, 436) deg: wrap it into [0, 360)" % (ephi,))
    for _, _, _, levels in entrypoints:
        # model.py:35-36: Airplane.__init__ refuses an altitude outside [h_min, h_max] — the reference raises it from the reset() that
        # draws such a level; the spawn records are evaluated here, once, so the same error comes at construction
        if any((100.0 * float(lv) < 0.0) or (100.0 * float(lv) > 38000.0) for lv in levels):
            raise ValueError("invalid altitude")
    off_slot = (off_entry + n_entry * L.E_WORDS + 3) & ~3  # 16-byte aligned float4 records
    end = off_slot + (4 * L.MAX_AIRCRAFT if n_entry else 0)
    off_spawn = (end + 15) & ~15   # 64-byte aligned spawn records: 64 lattice slots, then one per entry point
    n_spawn = (L.MAX_AIRCRAFT + n_entry) if n_entry else 0
    end = off_spawn + n_spawn * L.SPAWN_WORDS
    grid = grid_fn = cached32 = cache32_path = None
    off_grid = 0
    if grid_cell is not None and mva_rings:
        th = cg["tri_h"]
        grid_args = (mva_rings, bounds[:len(mva_rings)], mva_heights, bbox, float(grid_cell), grid_guard)
        grid_kw = dict(noise_bounds=bounds[len(mva_rings):],
                       corridor_bounds=(th[:, 0].min(), th[:, 1].min(), th[:, 0].max(), th[:, 1].max()))
        off_grid = (end + 3) & ~3  # 16-byte aligned: cells are read as 8-byte pairs, edge records as 16-byte vectors
        # The whole device blob of a sector with a lookup grid is kept on disk too (see _grid_cache_path): a process that finds it
        # maps the file and never builds — or even touches — the grid's float64 image (23 MB for LOWW at 0.0625 nm); the
        # float64 master fills its grid words in on first use (only the float64 test oracle reads them).
        cache32_path = _grid_cache_path([[m[0] for m in mvas], [m[1] for m in mvas], list(runway),
                                         [[e[0], e[1], e[2], list(e[3])] for e in entrypoints],
                                         [[n[0], n[1], n[2]] for n in noise], grid_cell, grid_guard], kind="blob32")
        if cache32_path is not None:
            import os
            if os.path.exists(cache32_path):
                try:
                    c32 = np.load(cache32_path, mmap_mode="r", allow_pickle=False)
                    if c32.dtype == np.float32 and c32.ndim == 1 and len(c32) > off_grid + L.G_HDR and \
                            c32[L.H_VERSION] == L.BLOB_VERSION and int(c32[L.H_NWORDS]) == len(c32) and int(c32[L.H_OFF_GRID]) == off_grid:
                        cached32 = c32
                except (OSError, ValueError):
                    pass
        if cached32 is not None:
            end = len(cached32)
            grid_fn = lambda: build_grid_cached(*grid_args, **grid_kw)   # noqa: E731
        else:
            grid = build_grid_cached(*grid_args, **grid_kw)
            end = off_grid + len(grid)

    b = np.zeros(end, dtype=np.float64)
    b[L.H_VERSION] = L.BLOB_VERSION
    b[L.H_NWORDS] = end
    b[L.H_N_MVA] = len(mva_rings)
    b[L.H_N_NOISE] = len(noise_rings)
    b[L.H_N_ENTRY] = n_entry
    b[L.H_OFF_POLY], b[L.H_OFF_VERT], b[L.H_OFF_ENTRY], b[L.H_OFF_GRID] = off_poly, off_vert, off_entry, off_grid
    b[L.H_N_VERTW] = n_vertw
    b[L.H_OFF_SLOT] = off_slot if n_entry else 0
    b[L.H_OFF_SPAWN] = off_spawn if n_entry else 0
    b[L.C_RWY_X], b[L.C_RWY_Y], b[L.C_RWY_H] = cg["x"], cg["y"], cg["h"]
    b[L.C_PHI_TO_RWY] = cg["phi_to_runway"]
    b[L.C_FAF_X], b[L.C_FAF_Y] = cg["faf"]
    b[L.C_NRM_X], b[L.C_NRM_Y] = cg["normal"]
    b[L.C_FAF_ANGLE] = cg["faf_angle"]
    b[L.C_GS_TAN] = math.tan(3 * math.pi / 180)
    b[L.C_FAF_MVA] = faf_mva
    b[L.C_WORLD_DIAG] = world_diag
    b[L.C_NM_TO_FT] = 6076
    b[L.C_V_MIN], b[L.C_V_MAX], b[L.C_H_MIN], b[L.C_H_MAX] = v_min, v_max, h_min, h_max
    b[L.C_A_MIN], b[L.C_A_MAX], b[L.C_HDOT_MIN], b[L.C_HDOT_MAX] = -5, 5, -41, 15
    b[L.C_PHIDOT_MIN], b[L.C_PHIDOT_MAX], b[L.C_V_INIT] = -3, 3, 250
    b[L.C_TRI_H:L.C_TRI_H + 8] = cg["tri_h"].ravel()
    b[L.C_TRI_1:L.C_TRI_1 + 8] = cg["tri_1"].ravel()
    b[L.C_TRI_2:L.C_TRI_2 + 8] = cg["tri_2"].ravel()
    b[L.C_NORM_MIN:L.C_NORM_MIN + 10] = norm_min.astype(np.float64)
    b[L.C_NORM_MAX:L.C_NORM_MAX + 10] = norm_max.astype(np.float64)
    half = np.float32(0.5) * norm_max                       # float32 arithmetic, like the reference's numpy vectors
    b[L.C_NORM_A:L.C_NORM_A + 10] = (np.float32(1.0) / half).astype(np.float64)
    b[L.C_NORM_B:L.C_NORM_B + 10] = (-(norm_min + half) / half).astype(np.float64)
    b[L.C_ACT_DISCR:L.C_ACT_DISCR + 3] = (5, 50, 0.5)
    b[L.C_BBOX:L.C_BBOX + 4] = bbox
    b[L.C_DIR_RWY_X], b[L.C_DIR_RWY_Y] = cg["dir_rwy"]
    th = cg["tri_h"]
    b[L.C_TRI_BBOX:L.C_TRI_BBOX + 4] = (th[:, 0].min(), th[:, 1].min(), th[:, 0].max(), th[:, 1].max())
    # knife-edge of the reference's angle window for a heading exactly equal to the runway heading (model.py:216-229):
    # min_angle = 45 - (45 - arccos(dir . dir)) must be <= relative_angle == 0.  Evaluated with numpy like the reference.
    d = np.dot(rot_matrix(cg["phi_to_runway"]), np.array([[0], [1]]))
    min_angle = cg["faf_angle"] - (cg["faf_angle"] - np.arccos(np.dot(np.transpose(d), d))[0][0])
    b[L.C_ALIGNED_OK] = 1.0 if min_angle <= 0.0 else 0.0
    # fixed-point position grid: covers the airspace, the corridor and every entry point
    gpts = [(bbox[0], bbox[1]), (bbox[2], bbox[3]), cg["faf"], cg["iaf"], cg["corner1"], cg["corner2"], (cg["x"], cg["y"])]
    gpts += [(e[0], e[1]) for e in entrypoints]
    px0, py0, pk = position_grid(gpts)
    b[L.C_POS_X0], b[L.C_POS_Y0], b[L.C_POS_SCALE], b[L.C_POS_INV] = px0, py0, 2.0 ** pk, 2.0 ** -pk
    for axis, (val, org) in enumerate(((cg["faf"][0], px0), (cg["faf"][1], py0))):
        fix = int(to_fix(val, org, pk))
        b[L.C_FAF_FIX + 2 * axis], b[L.C_FAF_FIX + 2 * axis + 1] = fix >> 16, fix & 0xffff
    voff = off_vert
    for i, ring in enumerate(rings):
        rec = off_poly + i * L.P_WORDS
        b[rec + L.P_MINX:rec + L.P_MINX + 4] = bounds[i]
        b[rec + L.P_HEIGHT] = heights[i]
        b[rec + L.P_VOFF] = voff
        b[rec + L.P_NVERT] = len(ring)
        b[rec + L.P_PENALTY] = penalties[i]
        b[voff:voff + 2 * len(ring)] = ring.ravel()
        voff += 2 * len(ring)
    for i, (ex, ey, ephi, levels) in enumerate(entrypoints):
        assert 1 <= len(levels) <= L.E_MAXLEV, "1..8 levels per entry point"
        rec = off_entry + i * L.E_WORDS
        b[rec + L.E_X], b[rec + L.E_Y], b[rec + L.E_PHI], b[rec + L.E_NLEV] = ex, ey, ephi, len(levels)
        b[rec + L.E_LEV0:rec + L.E_LEV0 + len(levels)] = levels
    for k in range(L.MAX_AIRCRAFT if n_entry else 0):  # slot lattice: entry k mod E, level (k div E) mod n_levels
        ex, ey, ephi, levels = entrypoints[k % n_entry]
        b[off_slot + 4 * k:off_slot + 4 * k + 4] = (ex, ey, ephi, levels[(k // n_entry) % len(levels)] * 100)
    if grid is not None:
        b[off_grid:off_grid + len(grid)] = grid
    elif grid_fn is not None:   # (filled in by CompiledSector.blob64 on first use; the header words the head check needs are there)
        pass
    if end >= 2 ** 24:
        raise SectorTooLarge("the compiled sector has %d words: blob offsets must stay exactly representable in fp32 (< 2^24)" % end)
    # spawn records (include/atc_step.h: ATC_H_OFF_SPAWN): what AtcGym.reset computes for an aircraft placed at a lattice slot /
    # an entry point — fixed-point state + the raw reset observation — evaluated here once instead of in every reset of
    # every env (under the measurement protocol an env of 16 aircraft resets every ~25 steps: the reset is a hot path)
    spawn_i = np.zeros((n_spawn, L.SPAWN_WORDS), dtype=np.int32)
    if n_spawn:
        f32 = np.float32
        faf_fix = [int(to_fix(cg["faf"][a], (px0, py0)[a], pk)) for a in range(2)]
        pos_inv = f32(2.0 ** -pk)
        to_rwy = f32(cg["phi_to_runway"])
        places = [(entrypoints[k % n_entry], entrypoints[k % n_entry][3][(k // n_entry) % len(entrypoints[k % n_entry][3])] * 100.0)
                  for k in range(L.MAX_AIRCRAFT)] + [(e, 0.0) for e in entrypoints]
        for r, ((ex, ey, ephi, _), h) in enumerate(places):
            fix = []
            for val, org in ((ex, px0), (ey, py0)):   # csrc/atc_device.h: pos_spawn (fp32 like the device)
                c = f32(f32(f32(val) - f32(org)) * f32(2.0 ** pk))
                fix.append(int(np.rint(min(max(float(c), -2147483648.0), 2147483520.0))))
            # (an entry heading outside (-76, 436) deg would saturate the 32-bit field — the WIDE sentinels of ABI 19, with no side
            # word written at reset: compile_sector refuses such entries above, so this stays inside the range)
            phi_fix = int(np.rint((float(f32(ephi)) - L.PHI_FIX_OFFSET) * 2.0 ** L.PHI_FIX_SHIFT))
            assert -2 ** 31 < phi_fix < 2 ** 31 - 1
            phi32 = f32(f32(phi_fix) * f32(2.0 ** -L.PHI_FIX_SHIFT) + f32(L.PHI_FIX_OFFSET))   # phi_real (exact for these values)
            x32, y32 = f32(px0 + fix[0] * 2.0 ** -pk), f32(py0 + fix[1] * 2.0 ** -pk)       # pos_to_real: one rounding
            tfx = f32(f32(min(max(faf_fix[0] - fix[0], -2 ** 31), 2 ** 31 - 1)) * pos_inv)
            tfy = f32(f32(min(max(faf_fix[1] - fix[1], -2 ** 31), 2 ** 31 - 1)) * pos_inv)
            d_faf = f32(math.hypot(float(tfx), float(tfy)))
            phi_rel_faf = f32(math.degrees(math.atan2(float(tfy), float(tfx))))
            on_gp = f32(float(f32(318.4)) * float(d_faf) + float(f32(f32(faf_mva) - f32(200.0))))
            a = f32(f32(phi32 - to_rwy) + f32(180.0))                       # relative_angle (model.py:340-342) in fp32 steps
            m = f32(math.fmod(float(a), 360.0))
            if m != 0 and (m < 0):
                m = f32(m + f32(360.0))
            rel_rwy = f32(m - f32(180.0))
            obs = np.array([x32, y32, h, phi32, 250.0, h, on_gp, d_faf, phi_rel_faf, rel_rwy], dtype=np.float32)
            spawn_i[r, 0], spawn_i[r, 1], spawn_i[r, 3] = fix[0], fix[1], phi_fix
            spawn_i[r, 2] = np.float32(h).view(np.int32)
            spawn_i[r, 4:14] = obs.view(np.int32)
            b[off_spawn + r * L.SPAWN_WORDS:off_spawn + r * L.SPAWN_WORDS + 4] = (fix[0], fix[1], h, phi_fix)
            b[off_spawn + r * L.SPAWN_WORDS + 4:off_spawn + r * L.SPAWN_WORDS + 14] = obs.astype(np.float64)

    meta = dict(
        mva_rings=mva_rings, mva_heights=mva_heights, mva_bounds=bounds[:len(mva_rings)], noise_rings=noise_rings,
        bbox=bbox, world_diag=world_diag, faf_mva=faf_mva, corridor=cg, norm_min=norm_min, norm_max=norm_max,
        entrypoints=[(float(a), float(b_), float(c), [int(l) for l in lv]) for a, b_, c, lv in entrypoints],
        n_mva=len(mva_rings), n_noise=len(noise_rings), n_entry=n_entry, has_grid=(grid is not None or grid_fn is not None),
        v_min=v_min, v_max=v_max, h_min=h_min, h_max=h_max, pos_origin=(px0, py0), pos_k=pk,
    )
    cs = CompiledSector(b, meta, spawn_words=spawn_i if n_spawn else None, grid_fn=grid_fn, blob32=cached32)
    if cached32 is not None:   # the cached device blob must be THIS sector's: its head (everything but the grid) is rebuilt here
        head = b[:off_grid].astype(np.float32)
        if n_spawn:
            head[off_spawn:off_spawn + n_spawn * L.SPAWN_WORDS].view(np.int32)[:] = spawn_i.ravel()
        if not np.array_equal(head.view(np.int32), np.asarray(cached32[:off_grid]).view(np.int32)):
            cs = CompiledSector(b, meta, spawn_words=spawn_i if n_spawn else None, grid_fn=grid_fn)   # stale file: rebuild
            cached32 = None
    if cached32 is None and cache32_path is not None:
        _cache_save(cache32_path, cs.blob32)
    return cs
