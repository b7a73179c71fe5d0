"""CPU-only: the Segmenting phase -- host code in libtreee_hip.so, as in the reference.
The library's C++ (treee_amd/csrc/segmenting.cpp) against the independent pure-Python restatement of the cited Rust lines
(oracle/segmenting_restatement.py), hand-traced cases, and properties the algorithm must have whatever the shuffle."""
import numpy as np
import pytest


def _cloud(rng, centers, n_per, height=6.0, radius=0.8, y0=0.0):
    parts, truth = [], []
    for t, (cx, cz) in enumerate(centers):
        a = rng.uniform(0, 2 * np.pi, n_per)
        r = radius * np.sqrt(rng.uniform(0, 1, n_per))
        y = rng.uniform(y0, y0 + height, n_per)
        parts.append(np.stack([cx + r * np.cos(a), y, cz + r * np.sin(a)], 1))
        truth.append(np.full(n_per, t))
    return np.concatenate(parts).astype(np.float32), np.concatenate(truth)


def _bbox(pts):
    lo, hi = pts.min(0), pts.max(0)
    return (lo[0], lo[2]), (hi[0], hi[2])


@pytest.mark.parametrize("seed,max_distance", [(0, 0.75), (7, 0.75), (3, 0.3), (11, 1.5)])
def test_library_matches_the_python_restatement(seed, max_distance):
    """every point's id identical: same hulls, same centroid hand-off, same cells, same move-to-front assignment"""
    import treee_amd as T
    from oracle import segmenting_restatement as R
    from treee_amd import synth

    pts, _ = synth.forest_cloud(3000, plot=8.0)
    rng = np.random.default_rng(seed)
    extra, _ = _cloud(rng, [(-6.0, 5.0), (7.5, -6.5)], 400, height=9.0, y0=3.0)  # two loners, and slices with few points
    pts = np.concatenate([pts, extra, pts[:50]])  # duplicates too
    bmin, bmax = _bbox(pts)
    seg, S = T.segment_slices(pts, max_distance=max_distance, seed=seed)
    want, S2 = R.segment_slices(pts, bmin, bmax, max_distance, seed)
    assert S == S2 and np.array_equal(seg, want)
    assert seg.max() < S and set(np.unique(seg)) - {0} == set(range(1, S))  # ids 1, 2, 3 ... in creation order, none skipped


def test_separate_trees_get_one_id_each_whatever_the_shuffle():
    """columns much farther apart than 2 * max_distance: the centroid hand-off (segmenting.rs:497-511) keeps one id per
    column through all slices, the Voronoi cells split the box between them, and the seed only permutes the ids"""
    import treee_amd as T

    rng = np.random.default_rng(1)
    centers = [(-8.0, -8.0), (8.0, -7.0), (0.0, 9.0), (7.0, 8.0), (-9.0, 6.0)]
    pts, truth = _cloud(rng, centers, 900, height=7.5)
    results = []
    for seed in (0, 1, 2):
        seg, S = T.segment_slices(pts, seed=seed)
        assert S == len(centers) + 1 and (seg > 0).all()
        for t in range(len(centers)):
            assert len(np.unique(seg[truth == t])) == 1
        assert len(np.unique(seg)) == len(centers)
        results.append(seg)
    again, _ = T.segment_slices(pts, seed=2)
    assert np.array_equal(again, results[2])  # same seed, same ids


def test_hand_traced_two_slices():
    """traced by hand from the cited lines.  Slice y in [1, 2): two clusters 10 m apart -> trees 1 and 2 (creation order =
    order of first appearance after the shuffle; the cluster holding the first point gets id 1).  Slice y in [0, 1): one
    cluster under the first (its centroid within 2 * max_distance of the handed-down one -> keeps that id,
    segmenting.rs:497-508), one new cluster 10 m away from both -> the next id; the second tree of the upper slice keeps
    its centroid (:509-511) and its cell, which holds no points here."""
    import treee_amd as T

    rng = np.random.default_rng(5)

    def disc(cx, cz, y0, n=300):
        a, r = rng.uniform(0, 2 * np.pi, n), 0.6 * np.sqrt(rng.uniform(0, 1, n))
        return np.stack([cx + r * np.cos(a), rng.uniform(y0 + 0.05, y0 + 0.95, n), cz + r * np.sin(a)], 1)

    top_a, top_b = disc(0.0, 0.0, 1.0), disc(10.0, 0.0, 1.0)
    low_a, low_c = disc(0.2, 0.1, 0.0), disc(5.0, 9.0, 0.0)
    pts = np.concatenate([top_a, top_b, low_a, low_c]).astype(np.float32)
    seg, S = T.segment_slices(pts, seed=4)
    ia, ib = seg[0], seg[300]
    assert S == 4 and {ia, ib} == {1, 2}
    assert (seg[:300] == ia).all() and (seg[300:600] == ib).all()
    assert (seg[600:900] == ia).all()  # under tree a: same id
    assert (seg[900:] == 3).all()  # the newcomer: next id


def test_voronoi_cells_partition_the_box():
    """D4 (the stand-in for voronator 0.2.1): with sites spread over the box, the cell that `contains` a point with zero
    slack is the cell of its nearest site, and every point of the box lies in exactly one cell (up to edges)"""
    from oracle import segmenting_restatement as R

    rng = np.random.default_rng(9)
    sites = [(np.float32(x), np.float32(y)) for x, y in rng.uniform(-10, 10, (12, 2))]
    cells = R.voronoi_cells((-10.0, -10.0), (10.0, 10.0), sites)
    trees = [R.Tree.from_points(c) for c in cells]
    for q in rng.uniform(-10, 10, (300, 2)).astype(np.float32):
        d = [(float(q[0]) - float(s[0])) ** 2 + (float(q[1]) - float(s[1])) ** 2 for s in sites]
        order = np.argsort(d)
        if d[order[1]] - d[order[0]] < 1e-3:
            continue  # on an edge
        inside = [i for i, t in enumerate(trees) if t.contains((q[0], q[1]), np.float32(0.0))]
        assert inside == [order[0]]


def test_errors_and_empty_input():
    import treee_amd as T

    seg, S = T.segment_slices(np.zeros((0, 3), np.float32))
    assert len(seg) == 0 and S == 1
    with pytest.raises(T.TreeeError):
        T.segment_slices(np.zeros((4, 3), np.float32), max_distance=0.0)
    one, S = T.segment_slices(np.array([[1.0, 2.0, 3.0]], np.float32))
    # a lone point: its seed triangle (area 0.005) is below max_distance^2 / 4 and is filtered (segmenting.rs:472-475):
    # no tree, no cell, DELETED_INDEX
    assert one.tolist() == [0] and S == 1


def test_nan_points_and_clipped_away_cells_follow_the_reference_rule():
    """a point whose x or z is NaN fails every comparison, so `Tree::contains` (segmenting.rs:331-349) holds for
    EVERY cell -- also for one clipped to nothing (`from_points` of no vertices, :272-278: min = MAX, max = MIN, no edge to
    fail) -- and `position()` (:195-197) takes the first cell of the move-to-front order.  That is the reference's rule, and
    the library keeps it (the advisor's "skip empty cells" would change results): ids identical to the restatement on a
    cloud with NaN points in several slices and with coincident sites (centroids handed down onto the same spot)."""
    import treee_amd as T
    from oracle import segmenting_restatement as R

    rng = np.random.default_rng(21)
    pts, _ = _cloud(rng, [(-4.0, -4.0), (4.0, -3.0), (0.0, 5.0)], 700, height=5.0)
    twin, _ = _cloud(np.random.default_rng(21), [(-4.0, -4.0)], 700, height=5.0)  # the first column again: coincident hull areas
    pts = np.concatenate([pts, twin[:700]])
    nan_rows = rng.choice(len(pts), 25, replace=False)
    bmin, bmax = _bbox(pts)  # (the box of the finite cloud: the reference takes it from the LAS header, not from the points)
    pts[nan_rows[:10], 0] = np.nan
    pts[nan_rows[10:20], 2] = np.nan
    pts[nan_rows[20:], 0] = np.nan
    pts[nan_rows[20:], 2] = np.nan
    for seed in (0, 5):
        seg, S = T.segment_slices(pts, bbox_min=(bmin[0], 0.0, bmin[1]), bbox_max=(bmax[0], 0.0, bmax[1]), max_distance=0.75, seed=seed)
        want, S2 = R.segment_slices(pts, bmin, bmax, 0.75, seed)
        assert S == S2 and np.array_equal(seg, want)
