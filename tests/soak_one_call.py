"""Soak of treee_sort_build_knn_dev (not collected by pytest: run on a GPU box):
    python tests/soak_one_call.py [seed] [seconds]
Random clouds -- 1 .. 3 M points; uniform, clustered, forest, a blob inside a few Morton cells, lattices with duplicates; with
and without a caller's cube; leaf / LOD capacities from 1 to 512 -- through Octree.with_knn (one call, the octree's levels and
the kNN order's tables on the context's auxiliary stream) against Octree(...) + KDTree.from_octree (round 1-4's calls): every
octree array bit for bit, the 31-NN of every point (distance bits, indices, counts) and `curve` bit for bit.  The handles of
one iteration are freed in a random order while the next iteration's build is being enqueued (pool reuse across the streams)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import treee_amd as T  # noqa: E402
from treee_amd import synth  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time()
cases = 0
keep = []
while time.time() - t0 < budget:
    n = int(rng.choice([1, 2, 63, 64, 65, 257, 1000, 5000, 70_000, 300_000, 1_100_000, 3_000_000]))
    shape = rng.choice(["uniform", "clusters", "forest", "blob", "lattice"])
    if shape == "uniform":
        p = rng.random((n, 3)) * rng.choice([1e-2, 1.0, 300.0])
    elif shape == "clusters":
        p = rng.normal(0, 0.05, (n, 3)) + rng.integers(0, 7, (n, 1)) * [9.0, 0.5, 4.0]
    elif shape == "forest":
        p, _ = synth.forest_cloud(n, plot=max(1.0, (n / 375.0) ** 0.5), seed=int(rng.integers(1 << 30)))
    elif shape == "blob":
        p = rng.standard_normal((n, 3)) * [5.0, 2.0, 5.0]
        p[: n // 2] *= 0.0004
    else:
        p = rng.integers(0, 40, (n, 3)) * 0.125
    p = (np.asarray(p) + rng.choice([0.0, 500.0, -2e4])).astype(np.float32)
    corner, size = None, 0.0
    if rng.random() < 0.3:
        mn = p.min(0)
        corner, size = (mn - np.float32(1.0)).astype(np.float32), float((p.max(0) - mn).max()) + 3.0
    leaf_cap, lod_cap = int(rng.choice([1, 8, 64, 256, 512])), int(rng.choice([1, 32, 256]))
    d = torch.from_numpy(p).cuda()
    ref = T.Octree(d, corner=corner, size=size, leaf_cap=leaf_cap, lod_cap=lod_cap)
    ref_tree = T.KDTree.from_octree(ref)
    oc, tree = T.Octree.with_knn(d, corner=corner, size=size, leaf_cap=leaf_cap, lod_cap=lod_cap)
    tag = (n, shape, leaf_cap, lod_cap, corner is not None)
    assert (oc.n_points, oc.n_nodes, oc.n_lod) == (ref.n_points, ref.n_nodes, ref.n_lod), tag
    assert np.array_equal(oc.corner, ref.corner) and oc.size == ref.size, tag
    for arr in ("keys", "perm", "nodes", "leaf", "lod", "sorted_xyz"):
        assert np.array_equal(oc.array(arr).view(np.uint32), ref.array(arr).view(np.uint32)), (tag, arr)
    k = 31
    a = tree.self_neighbors_dev(k)
    b = ref_tree.self_neighbors_dev(k)
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), tag
    assert torch.equal(tree.curvature_dev(k).view(torch.int32), ref_tree.curvature_dev(k).view(torch.int32)), tag
    # frees of this iteration's handles happen while the next build is enqueued
    for h in keep:
        h.close()
    keep = [oc, tree, ref, ref_tree]
    rng.shuffle(keep)
    cases += 1
for h in keep:
    h.close()
torch.cuda.synchronize()
print("cases", cases, "seconds", round(time.time() - t0, 1))
