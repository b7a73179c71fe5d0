"""CPU ORACLE (test infrastructure): an independent pure-Python restatement of the Segmenting phase,
treee/src/segmenting.rs:110-235 (segmentation), :242-396 (Tree), :419-524 (TreeSet), :526-545 (centroid), in the
deterministic variant SURVEY.md section 8 F1 prescribes.  Written from the cited Rust lines, not from the C++ in
treee_amd/csrc/segmenting.cpp; it shares only the four documented rules that replace the reference's randomness:
  D1 shuffle = Fisher-Yates from the end with SplitMix64(seed ^ (0xD1B54A32D192ED03 * (key - min_key + 1)))
  D2 HashMap<u32, Centroid> iterated in ascending key order
  D3 new ids 1, 2, 3 ... in creation order (0 = DELETED_INDEX)
  D4 Voronoi cells = box clipped by the bisector half-planes of all other sites + 4 helper sites (voronator 0.2.1 is absent
     from the snapshot; parity with the crate unpinned), f64, vertices stored as f32
f32 arithmetic is done with numpy.float32 scalars in the reference's operation order.  Slow: small clouds only.
PARITY STATUS: unpinned (the reference is random and has no tests)."""
import math

import numpy as np

F = np.float32
F32_MAX = F(np.finfo(np.float32).max)
MASK = (1 << 64) - 1


def edge_dist(a, b, p):
    """segmenting.rs:347-353 / :305-311 / :337-342"""
    dx, dy = b[0] - a[0], b[1] - a[1]
    ox, oy = dy, -dx
    with np.errstate(all="ignore"):
        norm = np.sqrt(ox * ox + oy * oy)
        nx, ny = ox / norm, oy / norm
    fx, fy = p[0] - a[0], p[1] - a[1]
    return nx * fx + ny * fy


def centroid(points):
    """segmenting.rs:526-545 -> ((cx, cy), area)"""
    cx = cy = area = F(0)
    a = points[0]
    for i in range(1, len(points) - 1):
        bx, by = points[i][0] - a[0], points[i][1] - a[1]
        ccx, ccy = points[i + 1][0] - a[0], points[i + 1][1] - a[1]
        tx, ty = (bx + ccx) / F(3), (by + ccy) / F(3)
        t_area = bx * ccy - by * ccx
        cx, cy, area = cx + tx * t_area, cy + ty * t_area, area + t_area
    with np.errstate(all="ignore"):
        return (a[0] + cx / area, a[1] + cy / area), area / F(2)


def fmin(a, b):
    return b if a != a else (a if b != b or a < b else b)


def fmax(a, b):
    return b if a != a else (a if b != b or a > b else b)


class Tree:
    def __init__(self, points, mn, mx):
        self.points, self.min, self.max = points, mn, mx

    @classmethod
    def new(cls, p):  # :258-269
        return cls([p, (p[0] + F(0.1), p[1]), (p[0], p[1] + F(0.1))], p, (p[0] + F(0.1), p[1] + F(0.1)))

    @classmethod
    def from_points(cls, points):  # :271-296
        points = list(points)
        if not points:
            return cls([], (F32_MAX, F32_MAX), (-F32_MAX, -F32_MAX))
        if len(points) == 1:
            points.append((points[0][0] + F(0.1), points[0][1] + F(0.0)))
            points.append((points[0][0] + F(0.0), points[0][1] + F(0.1)))
        elif len(points) == 2:
            dx, dy = points[1][0] - points[0][0], points[1][1] - points[0][1]
            vx, vy = -dy, dx
            with np.errstate(all="ignore"):
                norm = np.sqrt(vx * vx + vy * vy)
                points.append((points[0][0] + (vx / norm) * F(0.1), points[0][1] + (vy / norm) * F(0.1)))
        mn = mx = points[0]
        for q in points[1:]:
            mn = (fmin(mn[0], q[0]), fmin(mn[1], q[1]))
            mx = (fmax(mx[0], q[0]), fmax(mx[1], q[1]))
        return cls(points, mn, mx)

    def outside_bounds(self, p, md):  # :318-323
        return p[0] + md < self.min[0] or self.max[0] + md <= p[0] or p[1] + md < self.min[1] or self.max[1] + md <= p[1]

    def distance(self, p, md):  # :298-316
        if self.outside_bounds(p, md):
            return F32_MAX
        best = -F32_MAX
        n = len(self.points)
        for i in range(n):
            d = edge_dist(self.points[i], self.points[(i + 1) % n], p)
            if d > md:
                return F32_MAX
            best = fmax(best, d)
        return best

    def contains(self, p, md):  # :330-345
        if self.outside_bounds(p, md):
            return False
        n = len(self.points)
        return not any(edge_dist(self.points[i], self.points[(i + 1) % n], p) > md for i in range(n))

    def insert(self, p):  # :346-396
        tree, n = self.points, len(self.points)
        last = edge_dist(tree[n - 1], tree[0], p) > 0
        start = end = None
        for i in range(n):
            out = edge_dist(tree[i], tree[(i + 1) % n], p) > 0
            if not last and out:
                start = i
            elif last and not out:
                end = i
            last = out
        if start is None or end is None:
            return
        if end < start:
            tree[start + 1 :] = [p]
            del tree[:end]
        else:
            tree[start + 1 : end] = [p]
        self.min = (fmin(self.min[0], p[0]), fmin(self.min[1], p[1]))
        self.max = (fmax(self.max[0], p[0]), fmax(self.max[1], p[1]))


def tree_set(points, md):
    """TreeSet::new without the shuffle (:423-435): add_point (:437-467) for every point, then filter_trees (:469-486)"""
    trees = []
    for p in points:
        near, inside = [], False
        for i, t in enumerate(trees):
            d = t.distance(p, md)
            if d <= 0:
                inside = True
                break
            if d <= md:
                near.append(i)
        if inside:
            continue
        if not near:
            trees.append(Tree.new(p))
        elif len(near) == 1:
            trees[near[0]].insert(p)
        else:
            target = near[0]
            for other in reversed(near[1:]):
                o = trees.pop(other)
                for q in o.points:
                    trees[target].insert(q)
            trees[target].insert(p)
    for i in range(len(trees) - 1, -1, -1):
        c, area = centroid(trees[i].points)
        if area < (md * md) / F(4):
            del trees[i]
            continue
        if any(trees[j].contains(c, F(0.1)) for j in range(i)):
            del trees[i]
    return trees


def tree_positions(trees, prev, md, next_id):
    """:488-524 with D2 / D3 -> (dict id -> centroid, next_id)"""
    res = {}
    cents = [centroid(t.points)[0] for t in trees]
    for key in sorted(prev):
        center = prev[key]
        nearest, nearest_dist = None, md * F(2)
        for i, c in enumerate(cents):
            dx, dy = center[0] - c[0], center[1] - c[1]
            d = np.sqrt(dx * dx + dy * dy)
            if d < nearest_dist:
                nearest, nearest_dist = i, d
        if nearest is not None:
            res[key] = cents[nearest]
            cents[nearest] = cents[-1]
            cents.pop()
        else:
            res[key] = center
    for c in cents:
        while next_id == 0 or next_id in res:
            next_id += 1
        res[next_id] = c
        next_id += 1
    return res, next_id


def clip(poly, a, b, c):
    if not any(a * x + b * y > c for x, y in poly):
        return poly
    out, n = [], len(poly)
    for i in range(n):
        (px, py), (qx, qy) = poly[i], poly[(i + 1) % n]
        sp, sq = a * px + b * py - c, a * qx + b * qy - c
        if sp <= 0.0:
            out.append((px, py))
        if (sp <= 0.0) != (sq <= 0.0):
            t = sp / (sp - sq)
            out.append((px + t * (qx - px), py + t * (qy - py)))
    return out


def voronoi_cells(bmin, bmax, sites):
    x0, y0, x1, y1 = float(bmin[0]), float(bmin[1]), float(bmax[0]), float(bmax[1])
    w, h = x1 - x0, y1 - y0
    pts = [(float(s[0]), float(s[1])) for s in sites]
    real = len(pts)
    pts += [(x0 - w, y0 + h / 2.0), (x1 + w, y0 + h / 2.0), (x0 + w / 2.0, y0 - h), (x0 + w / 2.0, y1 + h)]
    cells = []
    for i in range(real):
        poly = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
        sx, sy = pts[i]
        for j, (tx, ty) in enumerate(pts):
            if j == i or not poly:
                continue
            a, b = tx - sx, ty - sy
            if a == 0.0 and b == 0.0:
                continue
            c = ((tx * tx + ty * ty) - (sx * sx + sy * sy)) / 2.0
            poly = clip(poly, a, b, c)
        cells.append([(F(x), F(y)) for x, y in poly])
    return cells


def splitmix(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return state, z ^ (z >> 31)


def segment_slices(points, bbox_min_xz, bbox_max_xz, max_distance=0.75, seed=0):
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    md = F(max_distance)
    seg = np.zeros(len(pts), np.uint32)
    slices = {}
    for i, y in enumerate(pts[:, 1]):
        slices.setdefault(int(math.floor(float(y))), []).append(i)  # loading.rs:99
    next_id, cents = 1, {}
    if slices:
        lo, hi = min(slices), max(slices)
        for key in range(hi, lo - 1, -1):
            order = list(slices.get(key, []))
            state = (int(seed) ^ ((0xD1B54A32D192ED03 * (key - lo + 1)) & MASK)) & MASK
            for i in range(len(order), 1, -1):  # D1
                state, r = splitmix(state)
                j = r % i
                order[i - 1], order[j] = order[j], order[i - 1]
            xz = [(pts[p, 0], pts[p, 2]) for p in order]
            trees = tree_set(xz, md)
            cents, next_id = tree_positions(trees, cents, md, next_id)
            ids = sorted(cents)  # D2
            cells = voronoi_cells(bbox_min_xz, bbox_max_xz, [cents[k] for k in ids])
            deque = [(Tree.from_points(c), k) for c, k in zip(cells, ids)]
            for p, q in zip(order, xz):  # :189-203
                for at, (t, k) in enumerate(deque):
                    if t.contains(q, F(0.1)):
                        seg[p] = k
                        deque.insert(0, deque.pop(at))
                        break
                else:
                    seg[p] = 0
    return seg, next_id
