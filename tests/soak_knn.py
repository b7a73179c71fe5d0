"""Randomised soak of the kNN paths against the oracle (not collected by pytest: run by hand on the GPU box).

    python tests/soak_knn.py <seed> <seconds>

Clouds of 1 .. 70 000 points (cubes, planes, lines, clusters, lattices, duplicates; offsets up to 3e5), k from 1 to 100, four
radii, queries on and off the points; every case through the SURVEY H2 contract (counts and distance bits identical, indices
inside their tie groups), and the structure built over an octree against the one built from the points (identical answers).
Round 2: 31 000 cases, clean.  Round 5: the tie flags of the self query too."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import treee_amd as T
from oracle import oracle as O
from conftest import check_knn_contract
O.build()
F32_MAX = float(np.finfo(np.float32).max)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); cases = 0; ties = 0; flagged = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 90:
    n = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 200, 257, 1000, 4097, 20000, 70000]))
    shape = rng.choice(['cube', 'plane', 'line', 'clusters', 'lattice', 'dups'])
    if shape == 'cube': p = rng.random((n, 3)) * rng.choice([1e-3, 1.0, 1e4])
    elif shape == 'plane': p = rng.random((n, 3)) * [100, 1e-4, 100]
    elif shape == 'line': p = np.outer(rng.random(n), [3.0, 1.0, -2.0]) + rng.normal(0, 1e-5, (n, 3))
    elif shape == 'clusters': p = rng.normal(0, 0.01, (n, 3)) + rng.integers(0, 5, (n, 1)) * [10.0, 0, 3.0]
    elif shape == 'lattice': p = rng.integers(0, 12, (n, 3)) * 0.25
    else: p = np.repeat(rng.random((max(1, n // 40), 3)), 40, 0)[:n]; p = np.concatenate([p, rng.random((n - len(p), 3))]) if len(p) < n else p
    p = (p + rng.choice([0.0, 1000.0, -3e5])).astype(np.float32)
    k = int(rng.choice([1, 2, 5, 16, 17, 31, 32, 33, 64, 65, 100]))
    maxd = float(rng.choice([F32_MAX, 1.0, 1e-6, 0.07]))
    nq = min(n, 400)
    qsel = rng.choice(n, nq, replace=False)
    q = p[qsel] if rng.random() < 0.7 else (p[qsel] + rng.normal(0, 0.05, (nq, 3))).astype(np.float32)
    r_ent, r_cnt = O.KDTree(p).k_nearest_batch(q, k, maxd)
    tree = T.KDTree.new(p)
    g_ent, g_cnt = tree.k_nearest_batch(q, k, maxd)
    try:
        ties += check_knn_contract(p, q, g_ent, g_cnt, r_ent, r_cnt)
    except AssertionError as e:
        print('FAIL', n, shape, k, maxd, e); np.save(os.path.join(ROOT, 'gpurun_out', 'soak_fail.npy'), p); raise
    # the structure over an octree's Morton-sorted cloud (its Hilbert order by cells where cells hold several points, by points
    # otherwise: knn.hip hilbert_order_by_cells) must answer exactly like the one built from the raw points -- indices included
    import torch
    oc = T.Octree(torch.from_numpy(p).cuda())
    t2 = T.KDTree.from_octree(oc)
    h_ent, h_cnt = t2.k_nearest_batch(q, k, maxd)
    if not (np.array_equal(h_cnt, g_cnt) and np.array_equal(h_ent['distance'].view(np.uint32), g_ent['distance'].view(np.uint32)) and np.array_equal(h_ent['index'], g_ent['index'])):
        print('FAIL (octree structure differs from the point structure)', n, shape, k, maxd); np.save(os.path.join(ROOT, 'gpurun_out', 'soak_fail.npy'), p); raise SystemExit(1)
    # round 5: the per-row tie flags of the self query against their definition on the oracle's (k + 1)-lists
    if k <= 63 and n > 1:
        from test_gpu_parity import _flags_by_definition
        dtree = T.KDTree.new(torch.from_numpy(p).cuda())
        dd, di, dc, fl = dtree.self_neighbors_dev(k, maxd, ties=True)
        r1_ent, r1_cnt = O.KDTree(p).k_nearest_batch(p, k + 1, maxd)
        want = _flags_by_definition(r1_ent['distance'], k, r1_cnt)
        if not np.array_equal(fl.cpu().numpy(), want):
            print('FAIL (tie flags)', n, shape, k, maxd); np.save(os.path.join(ROOT, 'gpurun_out', 'soak_fail.npy'), p); raise SystemExit(1)
        rs_ent, rs_cnt = O.KDTree(p).k_nearest_batch(p, k, maxd)
        differs = (di.cpu().numpy().view(np.uint32) != rs_ent['index'].astype(np.uint32)).any(1)
        if (differs & (want == 0)).any():
            print('FAIL (a differing row without a flag)', n, shape, k, maxd); np.save(os.path.join(ROOT, 'gpurun_out', 'soak_fail.npy'), p); raise SystemExit(1)
        dtree.close(); flagged += int((want != 0).sum())
    t2.close(); oc.close()
    tree.close(); cases += 1
print('cases', cases, 'tie-resolved indices', ties, 'flagged rows', flagged)
