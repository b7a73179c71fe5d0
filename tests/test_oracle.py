"""Pins the CPU oracle (CPU-only tests).

The reference holds no tests, fixtures or golden vectors for this path,
so the oracle is pinned by: hand-traced KATs derived from the cited Rust lines, the prose
invariants of k-nearest/readme.md:26-34,90-92, an independent pure-Python restatement, brute
force, closed forms, and the committed golden vectors."""
import math
import os

import numpy as np
import pytest

from conftest import check_knn_contract, f32_dist_sq

F32_MAX = float(np.finfo(np.float32).max)


# ---- k-nearest -------------------------------------------------------------------------------
def test_kdtree_hand_traced_kat(oracle):
    # 4 collinear points; middle = len/2 (upper median), Lomuto partition
    pts = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]], np.float32)
    t = oracle.KDTree(pts)
    xyz, idx = t.dump()
    assert idx.tolist() == [0, 1, 2, 3] and np.array_equal(xyz, pts)
    out, c = t.k_nearest([0, 0, 0], 2, F32_MAX)
    assert c == 2 and out["distance"].tolist() == [0.0, 1.0] and out["index"].tolist() == [0, 1]
    out, c = t.k_nearest([0, 0, 0], 2, 1.0)
    assert c == 1 and out["index"][0] == 0 and out["index"][1] == oracle.INVALID


def test_kdtree_partition_hand_trace_reversed_input(oracle):
    # 5 points x = 4,3,2,1,0.  create_tree(dim 0): middle=2.  partition(target 2):
    #  lower=0 upper=4 pivot_index=2 pivot=2 -> swap(2,4): [4,3,0,1,2]; scan i=0..3 with `< 2`:
    #  i=2 (0): swap(0,2) -> [0,3,4,1,2] store=1; i=3 (1): swap(1,3) -> [0,1,4,3,2] store=2;
    #  swap(store=2, upper=4) -> [0,1,2,3,4]; target==store -> done.
    #  left [0,1] (dim 1, all y equal): middle=1, pivot_index=0 pivot=0 -> swap(0,1): [1,0];
    #  nothing `< 0`; swap(store=0, upper=1) -> [0,1]; target 1 > store 0 -> lower=1; then
    #  pivot_index=1, swap(1,1), no scan, swap(1,1), store=1 == target.  Left stays [0,1].
    #  right [3,4] likewise stays [3,4].  Original indices: x=k has index 4-k.
    pts = np.array([[4, 0, 0], [3, 0, 0], [2, 0, 0], [1, 0, 0], [0, 0, 0]], np.float32)
    xyz, idx = oracle.KDTree(pts).dump()
    assert xyz[:, 0].tolist() == [0, 1, 2, 3, 4]
    assert idx.tolist() == [4, 3, 2, 1, 0]


def test_fixedset_result_counts_and_order(oracle):
    # fewer points than k: invalid entries carry max_distance and pop first (best_set.rs:56-66)
    pts = np.array([[0, 0, 0], [0, 3, 0], [0, 0, 2]], np.float32)
    out, c = oracle.KDTree(pts).k_nearest([0, 0, 0], 5, F32_MAX)
    assert c == 3 and out["distance"][:3].tolist() == [0.0, 4.0, 9.0] and out["index"][:3].tolist() == [0, 2, 1]
    assert np.all(out["index"][3:] == oracle.INVALID) and np.all(out["distance"][3:] == np.float32(F32_MAX))


def test_kdtree_matches_independent_python_restatement(oracle):
    from oracle import py_restatement as R

    rng = np.random.default_rng(11)
    pts = (np.round(rng.random((300, 3)) * 8 * 4) / 4).astype(np.float32)  # coarse lattice: many ties
    tree_c = oracle.KDTree(pts)
    tree_py = R.kdtree_new(pts)
    xyz, idx = tree_c.dump()
    assert idx.tolist() == [i for _, i in tree_py]  # identical quickselect permutation
    for qi in range(0, 300, 13):
        for k, maxd in ((31, F32_MAX), (4, 2.0), (40, 0.5)):
            vals, count = R.k_nearest(tree_py, pts[qi], k, maxd)
            out, c = tree_c.k_nearest(pts[qi], k, maxd)
            assert c == count
            assert [float(v[0]) for v in vals] == out["distance"].tolist()
            assert [v[1] for v in vals] == out["index"].tolist()  # same ties, same heap mechanics


@pytest.mark.parametrize("n,k,maxd", [(1, 3, F32_MAX), (31, 31, F32_MAX), (32, 31, F32_MAX), (33, 31, F32_MAX), (2000, 31, F32_MAX), (2000, 31, 0.3), (5000, 7, 1.0)])
def test_kdtree_vs_brute_force(oracle, n, k, maxd):
    rng = np.random.default_rng(n + k)
    pts = (np.round(rng.random((n, 3)) * 10 * 1000) / 1000).astype(np.float32)
    t = oracle.KDTree(pts)
    ent, cnt = t.k_nearest_batch(pts, k, maxd)
    brute = np.zeros((n, k), oracle.ENTRY_DTYPE)
    bcnt = np.zeros(n, np.uint32)
    for i in range(0, n, max(1, n // 200)):
        brute[i], bcnt[i] = oracle.brute_k_nearest(pts, pts[i], k, maxd)
        ent_i = ent[i : i + 1]
        check_knn_contract(pts, pts[i : i + 1], ent_i, cnt[i : i + 1], brute[i : i + 1], bcnt[i : i + 1])
        d = ent[i]["distance"][: cnt[i]]
        assert np.all(d[1:] >= d[:-1]) and np.all(d < maxd)  # ascending, strict radius
        if maxd == F32_MAX:
            assert d[0] == 0.0  # readme.md:31: offset 0 is the original point
        assert np.array_equal(f32_dist_sq(pts[ent[i]["index"][: cnt[i]].astype(np.int64)], pts[i]), d)  # squared (readme.md:90-92)


def test_radius_and_empty(oracle):
    rng = np.random.default_rng(2)
    pts = rng.random((3000, 3)).astype(np.float32) * 5
    t = oracle.KDTree(pts)
    for q in rng.random((10, 3)).astype(np.float32) * 5:
        r = t.nearest(q, 0.25)
        d = f32_dist_sq(pts, q)
        assert sorted(r["index"].tolist()) == np.nonzero(d < np.float32(0.25))[0].tolist()
        assert np.all(r["distance"][1:] >= r["distance"][:-1])
        assert t.empty(q, 0.25) == (len(r) == 0)
    assert t.empty([100, 100, 100], 1.0) and not t.empty(pts[0], 1e-6)


# ---- curvature ---------------------------------------------------------------------------------
def test_fast_eigenvalues(oracle):
    assert oracle.fast_eigenvalues([[3, 0, 0], [0, 1, 0], [0, 0, 2]]).tolist() == [3.0, 1.0, 2.0]  # p1 == 0: diagonal, unsorted
    rng = np.random.default_rng(4)
    for _ in range(50):
        a = rng.random((3, 3))
        m = (a @ a.T).astype(np.float32)
        e = oracle.fast_eigenvalues(m)
        ref = np.linalg.eigvalsh(m.astype(np.float64))[::-1]
        assert np.allclose(e, ref, rtol=2e-4, atol=2e-5) and e[0] >= e[1] >= e[2]


def test_curve_planar_and_isotropic(oracle):
    rng = np.random.default_rng(8)
    plane = rng.random((2000, 3)).astype(np.float32)
    plane[:, 1] = 0.0
    c = oracle.curvature(plane)
    assert np.nanmax(np.abs(c)) < 1e-3  # flat neighbourhood: smallest eigenvalue 0
    ball = rng.random((4000, 3)).astype(np.float32)
    c = oracle.curvature(ball)
    assert 0.3 < np.nanmean(c) < 1.0 and np.all((c[~np.isnan(c)] > -1e-3) & (c[~np.isnan(c)] <= 1.0 + 1e-3))
    assert oracle.map_to_u32(0.0) == 0 and oracle.map_to_u32(1.0) == 0xFFFFFFFF and oracle.map_to_u32(float("nan")) == 0
    assert oracle.map_to_u32(0.5) == 0x80000000 and oracle.map_to_u32(-1.0) == 0


# ---- ingest --------------------------------------------------------------------------------------
def test_las_decode_hand_computed(oracle):
    from treee_amd import synth

    ints = np.array([[0, 0, 0], [1000, 2000, 3000]], np.int32)
    img = synth.las_bytes(ints, scale=(0.001, 0.001, 0.001), offset=(100.0, 200.0, 50.0))
    h = oracle.las_header_parse(img)
    # world points (100,200,50) and (101,202,53); swizzle (x, z, -y); centre of the swizzled bbox
    assert list(h.min) == [100.0, 50.0, -202.0] and list(h.max) == [101.0, 53.0, -200.0]
    assert list(h.center) == [100.5, 51.5, -201.0]
    dec = oracle.decode_las_records(img[h.offset_to_point_data:], 2, 20, list(h.scale), list(h.offset), list(h.center))
    assert dec.tolist() == [[-0.5, -1.5, 1.0], [0.5, 1.5, -1.0]]
    with pytest.raises(ValueError):
        oracle.las_header_parse(b"XXXX" + img[4:])
    with pytest.raises(ValueError):
        oracle.las_header_parse(img[:100])


def test_slice_keys_floor_semantics(oracle):
    y = np.array([0.0, -0.0, 0.999, 1.0, -0.001, -1.0, -1.5, 29.9995, np.nan, 1e30, -1e30], np.float32)
    pts = np.zeros((len(y), 3), np.float32)
    pts[:, 1] = y
    k = oracle.slice_keys(pts)
    assert k[:8].tolist() == [0, 0, 0, 1, -1, -1, -2, 29] and k[8] == 0
    assert k[9] == np.iinfo(np.int64).max and k[10] == np.iinfo(np.int64).min
    mn, order, offsets = oracle.slice_bin(pts[:8])
    assert mn == -2 and offsets.tolist() == [0, 1, 3, 6, 7] + [7] * 27 + [8]
    assert order.tolist() == [6, 4, 5, 0, 1, 2, 3, 7]  # arrival order inside a slice


# ---- splat -----------------------------------------------------------------------------------------
def _cam(oracle, w, h, dist):
    view = np.eye(4, dtype=np.float32)
    view[2, 3] = -dist
    return np.ascontiguousarray(view.T).reshape(16), oracle.perspective(w / h, math.pi / 4, 0.1, 10_000.0)


def test_perspective_closed_form(oracle):
    p = oracle.perspective(16 / 9, math.pi / 4, 0.1, 10_000.0).reshape(4, 4).T  # row-major view
    t = 1.0 / math.tan(math.pi / 8)
    assert np.isclose(p[1, 1], t) and np.isclose(p[0, 0], t / (16 / 9))
    assert np.isclose(p[2, 2], -(10_000.1) / 9999.9) and np.isclose(p[2, 3], -2 * 10_000 * 0.1 / 9999.9) and p[3, 2] == -1.0


def test_splat_single_point_closed_form(oracle):
    w, h = 200, 100
    view, proj = _cam(oracle, w, h, 10.0)
    depth, ids, index = oracle.splat(np.zeros((1, 3), np.float32), np.array([5], np.uint32), view, proj, 0.5, 0, 0xFFFFFFFF, w, h)
    hit = depth < 1.0
    m22, m23 = proj[10], proj[14]
    expect = np.float32((m22 * np.float32(-10.0) + m23) / np.float32(10.0))
    assert np.all(depth[hit] == expect) and np.all(ids[hit] == 5) and np.all(index[hit] == 0)
    ry = 0.5 * proj[5] / 10.0 * 0.5 * h  # disc radius in pixels (vertical)
    rx = 0.5 * proj[0] / 10.0 * 0.5 * w
    assert abs(hit.sum() - math.pi * rx * ry) < 0.15 * math.pi * rx * ry
    ys, xs = np.nonzero(hit)
    assert abs(xs.mean() - (w - 1) / 2) < 0.51 and abs(ys.mean() - (h - 1) / 2) < 0.51
    assert np.all(index[~hit] == 0xFFFFFFFF) and np.all(ids[~hit] == 0)


def test_splat_occlusion_ties_cull_and_clip(oracle):
    w, h = 64, 64
    view, proj = _cam(oracle, w, h, 10.0)
    pts = np.array([[0, 0, 0], [0, 0, 2], [0, 0, 2], [0, 0, 9.95], [0, 0, 20]], np.float32)
    val = np.array([1, 2, 3, 4, 5], np.uint32)
    depth, ids, index = oracle.splat(pts, val, view, proj, 0.3, 0, 0xFFFFFFFF, w, h)
    c = (h // 2, w // 2)
    assert ids[c] == 2 and index[c] == 1  # nearer splat wins; equal depth: first drawn wins
    depth, ids, index = oracle.splat(pts, val, view, proj, 0.3, 3, 5, w, h)  # value-range cull
    assert ids[c] == 3
    # point 3 sits 0.05 in front of the eye: nalgebra's GL projection under wgpu's 0..1 depth
    # clips ndc z < 0 (effective near plane ~0.2); point 4 is behind the camera
    assert not np.any(index == 3) and not np.any(index == 4)


# ---- octree (build-defined) ----------------------------------------------------------------------
def test_morton_and_octree_tiny(oracle):
    pts = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.999, 0.999, 0.999]], np.float32)
    keys = oracle.morton_keys(pts, [0, 0, 0], 1.0)
    full = (1 << 30) - 1
    assert keys.tolist()[:5] == [0, full, 0x09249249, 0x12492492, 0x24924924]
    assert keys[5] == full - 7  # 0.999*1024 = 1022.97 -> 1022 on every axis: lowest bit clear
    sk, perm = oracle.sort_keys(keys)
    assert perm.tolist() == [0, 2, 3, 4, 5, 1]
    nodes, leaf, lod = oracle.octree_build(sk, 1, 4)
    assert nodes[0]["count"] == 6 and nodes[0]["level"] == 0 and nodes[0]["first_child"] == 1
    lv1 = nodes[nodes["level"] == 1]
    assert lv1["key_prefix"].tolist() == [0, 1, 2, 4, 7] and lv1["count"].tolist() == [1, 1, 1, 1, 2]
    assert np.all(nodes["count"][nodes["first_child"] == 0xFFFFFFFF][:-1] >= 1)
    assert nodes[leaf]["first_child"].tolist() == [0xFFFFFFFF] * 6
    assert lod[:4].tolist() == [0, 1, 3, 4]  # root: floor(j*6/4)


# ---- committed golden vectors ------------------------------------------------------------------------
def test_golden_vectors_reproduce(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "forest3k.npz"))
    pts = g["points"]
    t = oracle.KDTree(pts)
    assert np.array_equal(t.dump()[1].astype(np.uint32), g["tree_order"])
    ent, cnt = t.k_nearest_batch(pts, 31, F32_MAX)
    assert np.array_equal(ent["distance"], g["knn_dist"]) and np.array_equal(ent["index"].astype(np.uint32), g["knn_index"])
    assert np.array_equal(cnt, g["knn_count"])
    ent, cnt = t.k_nearest_batch(pts, 31, 0.0225)
    assert np.array_equal(cnt, g["knn_r_count"]) and np.array_equal(ent["distance"], g["knn_r_dist"])
    assert np.array_equal(oracle.curvature(pts), g["curve"], equal_nan=True)
    keys = oracle.morton_keys(pts, g["corner"], float(g["size"]))
    sk, perm = oracle.sort_keys(keys)
    nodes, leaf, lod = oracle.octree_build(sk, 64, 32)
    assert np.array_equal(keys, g["keys"]) and np.array_equal(perm, g["perm"])
    assert np.array_equal(nodes.view(np.uint32).reshape(-1, 8), g["nodes"]) and np.array_equal(leaf, g["leaf"]) and np.array_equal(lod, g["lod"])
    assert np.array_equal(oracle.slice_keys(pts), g["slice_keys"])
    s = np.load(os.path.join(golden_dir, "splat3k.npz"))
    depth, ids, index = oracle.splat(pts, g["segment"], s["view"], s["proj"], 0.1, 0, 0xFFFFFFFF, 96, 54)
    assert np.array_equal(depth, s["depth"]) and np.array_equal(ids, s["ids"]) and np.array_equal(index, s["index"])
    assert np.array_equal(oracle.eye_dome(depth, 0.3 ** 6), s["eye_dome"])
    l = np.load(os.path.join(golden_dir, "las4.npz"))
    img = l["image"].tobytes()
    h = oracle.las_header_parse(img)
    dec = oracle.decode_las_records(img[h.offset_to_point_data:], 4, 20, list(h.scale), list(h.offset), list(h.center))
    assert np.array_equal(dec, l["decoded"])


# ---- Calculations phase (oracle/calc_oracle.cpp) --------------------------------------------------------------------
def test_hull_insert_hand_traced(oracle):
    """Tree::new + Tree::insert (segmenting.rs:258-269,346-396), traced by hand from the cited lines: the seed triangle is
    (p, p + (0.1, 0), p + (0, 0.1)); a point outside exactly one run of edges replaces the vertices strictly between the
    run's ends; a point inside changes nothing."""
    h, a = oracle.hull_insert_all(np.array([[0, 0]], np.float32))
    assert np.array_equal(h, np.array([[0, 0], [0.1, 0], [0, 0.1]], np.float32)) and a == np.float32(0.1) * np.float32(0.1) / 2
    h, a = oracle.hull_insert_all(np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]], np.float32))
    # (1,0): outside the hypotenuse only -> inserted between (0.1,0) and (0,0.1); (1,1): between (1,0) and (0,0.1);
    # (0,1): replaces (0,0.1)'s neighbourhood run; the seed's collinear vertices (0.1,0) and (0,0.1) stay (strict > 0)
    assert np.array_equal(h, np.array([[0, 0], [0.1, 0], [1, 0], [1, 1], [0, 1], [0, 0.1]], np.float32)) and a == 1.0
    h2, a2 = oracle.hull_insert_all(np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5], [0.25, 0.75], [1, 0.5]], np.float32))
    assert np.array_equal(h2, h) and a2 == a  # interior and on-edge points are no-ops
    # duplicates of the seed: zero-length edges never appear (the seed triangle has none), the hull is unchanged
    h3, a3 = oracle.hull_insert_all(np.zeros((5, 2), np.float32))
    assert len(h3) == 3 and a3 == np.float32(0.1) * np.float32(0.1) / 2


def test_hull_matches_scipy_convex_hull_area(oracle):
    """order-independent property: away from degeneracies the incremental hull is THE convex hull (of the points plus the
    seed triangle's two helper vertices)"""
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(2)
    for n in (10, 200, 5000):
        xz = rng.normal(0, 2, (n, 2)).astype(np.float32)
        h, a = oracle.hull_insert_all(xz)
        ref = ConvexHull(np.concatenate([xz, xz[:1] + np.array([[0.1, 0], [0, 0.1]], np.float32)]).astype(np.float64))
        assert abs(a - ref.volume) < 1e-3 * ref.volume and len(h) == len(ref.vertices)


def test_segment_information_hand_case(oracle):
    """SegmentInformation::new (calculations.rs:210-276) on a synthetic 'tree': a thin trunk from y = 0 to 5 under a
    wide crown from 5 to 9 -> the first slice wider than a 2 m disc starts the crown; nothing near the ground is wider
    than 1.5 x the thinnest slice of the first 10 m, so there is no ground"""
    rng = np.random.default_rng(3)
    trunk = np.stack([rng.uniform(-0.1, 0.1, 4000), rng.uniform(0, 5, 4000), rng.uniform(-0.1, 0.1, 4000)], 1)
    crown = np.stack([rng.uniform(-2, 2, 40000), rng.uniform(5, 9, 40000), rng.uniform(-2, 2, 40000)], 1)
    pts = np.concatenate([trunk, crown]).astype(np.float32)
    mn, mx, info, cls = oracle.segment_data(pts)
    trunk_height, crown_height, ground_sep, crown_sep = info
    assert ground_sep == mn[1] and abs(crown_sep - 5.0) < 0.11 and abs(trunk_height - 5.0) < 0.11 and abs(crown_height - 4.0) < 0.11
    # (SegmentData::new recomputes the separators as max.y - crown_height - trunk_height (calculations.rs:149-150): the rounding can
    # put the very lowest point below `ground_sep`)
    assert (cls[:4000] == 1).mean() > 0.97 and (cls[4000:] == 2).mean() > 0.97 and (cls == 0).sum() <= 2
    e, h = oracle.segment_update(pts, cls, mn[1], mx[1], info)
    assert h.min() == 0.0 and h.max() == 1.0 and np.nanmax(e) == 1.0
    assert e[:4000][cls[:4000] == 1].max() == 0.0  # trunk slices hold no crown points: area 0 -> diameter 0
    areas = oracle.size_areas(pts, mn[1], mx[1] - mn[1])
    assert len(areas) == int(np.float32(mx[1] - mn[1]) / np.float32(0.1)) + 1 and areas[:45].max() < 0.1 and areas[55:85].min() > 10


def test_display_values_constants(oracle):
    out = oracle.display_values(np.array([0, 1, 2], np.uint8), np.array([0.0, 0.5, 2.0], np.float32), np.array([1.0, 2.0, 4.0], np.float32),
                           np.array([-1.0, np.nan, 1.0], np.float32))
    assert out[0].tolist() == [0xFFFFFFFF // 8, 0xFFFFFFFF // 8 * 3, 0xFFFFFFFF // 8 * 6]
    assert out[1].tolist() == [0, 1 << 31, 0xFFFFFFFF] and out[2].tolist() == [1 << 30, 1 << 31, 0xFFFFFFFF] and out[3].tolist() == [0, 0, 0xFFFFFFFF]


def test_hilbert_keys_are_a_curve_through_the_cells(oracle):
    """the Hilbert index used by the kNN structure (Skilling 2004): a bijection of the 2^30 cells that, unlike the Morton key,
    always steps to a face neighbour; and a prefix code of the octree -- the points of a level-l cell share the top 3 l bits"""
    g = np.arange(8, dtype=np.float32)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + np.float32(0.5)  # the 512 cells of an 8^3 corner ...
    keys = oracle.hilbert_keys(pts * np.float32(128.0), [0, 0, 0], 1024.0)  # ... scaled to 128-cell blocks: top 9 bits
    top = keys >> 21
    assert len(np.unique(top)) == 512
    order = np.argsort(top)
    steps = np.abs(np.diff(pts[order], axis=0)).sum(1)
    assert np.all(steps == 1.0)  # consecutive cells share a face
    mk = oracle.morton_keys(pts * np.float32(128.0), [0, 0, 0], 1024.0) >> 21
    assert np.abs(np.diff(pts[np.argsort(mk)], axis=0)).sum(1).max() > 1.0  # the Z curve jumps
    rng = np.random.default_rng(0)
    p = (rng.random((20000, 3)) * 1024).astype(np.float32)
    h, m = oracle.hilbert_keys(p, [0, 0, 0], 1024.0), oracle.morton_keys(p, [0, 0, 0], 1024.0)
    for level in (1, 2, 3, 5):
        sh = 30 - 3 * level
        # same partition into level cells
        assert len(np.unique(np.stack([h >> sh, m >> sh], 1), axis=0)) == len(np.unique(h >> sh)) == len(np.unique(m >> sh))


def test_library_hilbert_index_equals_the_oracles(oracle):
    """the library's own hilbert30 / hilbert_of_morton (treee_amd/csrc/common.h -- one source for host and device), compiled
    for the host, against the oracle's independent statement of Skilling's algorithm: 4 096 grid cells + 20 000 random cells"""
    import subprocess
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as G

    exe = G.build_hilbert_host_test()
    rows = np.array([[int(v) for v in line.split()] for line in subprocess.check_output([exe], text=True).splitlines()], dtype=np.int64)
    assert rows.shape == (4096 + 20000, 5)
    pts = rows[:, :3].astype(np.float32) + np.float32(0.5)  # the centre of cell (x, y, z) of the 1024^3 grid
    want = oracle.hilbert_keys(pts, [0, 0, 0], 1024.0).astype(np.int64)
    assert np.array_equal(rows[:, 3], want)  # from the three cell coordinates
    assert np.array_equal(rows[:, 4], want)  # from the 30-bit Morton key
    assert len(np.unique(rows[:4096, 3] >> 18)) == 4096  # (the 16^3 grid: one index prefix per level-4 cell)


def test_shared_acos_cos_against_this_hosts_libm(oracle):
    """include/treee_f32_trig.h (one definition of f32::acos / cos for the kernels and the oracle, f64 evaluation rounded
    once) against the platform libm Rust would call: never more than 1 ulp apart (glibc 2.35's acosf is not correctly
    rounded: 92 % of uniform arguments agree, cosf 98.7 %), exact at the special values; and equal to the correctly rounded
    value of numpy's f64 functions for every argument tried."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1, 1, 400_000), np.linspace(-1, 1, 100_001), [0.0, 1.0, -1.0, 0.5, -0.5, 1e-30, -1e-30]]).astype(np.float32)
    ac, _, lac, _ = oracle.trig_f32(x)
    d = np.abs(ac.view(np.int32).astype(np.int64) - lac.view(np.int32).astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() > 0.9
    assert np.array_equal(ac.view(np.uint32), np.arccos(x.astype(np.float64)).astype(np.float32).view(np.uint32))
    assert ac[-6] == 0.0 and ac[-5] == np.float32(np.pi) and ac[-7] == np.float32(np.pi / 2)
    y = np.concatenate([rng.uniform(0, np.pi + 1e-6, 400_000), np.linspace(0, 3.2, 100_001), rng.uniform(-50, 50, 50_000), [0.0, np.pi / 3]]).astype(np.float32)
    _, co, _, lco = oracle.trig_f32(y)
    d = np.abs(co.view(np.int32).astype(np.int64) - lco.view(np.int32).astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() > 0.97
    assert np.array_equal(co.view(np.uint32), np.cos(y.astype(np.float64)).astype(np.float32).view(np.uint32))
    bad = np.array([np.nan, np.inf, -np.inf, 2.0, -1.5], np.float32)
    ac, co, _, _ = oracle.trig_f32(bad)
    assert np.isnan(ac).all() and np.isnan(co[:3]).all()
