"""How many candidate points does a query NEED at bucket sizes 64 / 32 / 16 (final radii, exact box tests), what does the
neediest lane of a wave need, and what is the wave's union?  CPU analysis (scipy) on a 0.94 M-point patch at the 15 M
configuration's density; Results: profiles/README.md, "Would finer candidate
granularity pay?"."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
from treee_amd import synth
def hilbert_keys(points, corner, size):
    """30-bit Hilbert index of the 2^-10 cells (Skilling 2004), vectorised -- the order the kNN structure cuts its buckets in"""
    q = np.clip(np.floor((points - corner) / size * 1024.0), 0, 1023).astype(np.int64)
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    Q = 512
    while Q > 1:
        P = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = 512
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]
    h = np.zeros_like(X[0])
    for b in range(9, -1, -1):
        h = (h << 3) | (((X[0] >> b) & 1) << 2) | (((X[1] >> b) & 1) << 1) | ((X[2] >> b) & 1)
    return h


n = 940_000
pts, _ = synth.forest_cloud(n, plot=50.0)
mn = pts.min(0); size = float((pts.max(0) - mn).max())
h = hilbert_keys(pts.astype(np.float64), mn.astype(np.float64), size)
order = np.argsort(h, kind='stable')
P = pts[order].astype(np.float64)
tree = cKDTree(P)
rng = np.random.default_rng(0)
nb64 = n // 64
waves = rng.choice(nb64 - 2, 400, replace=False) + 1
d, _ = tree.query(P, k=32)   # includes self -> 31 neighbours + self
rk = d[:, 31]
for B in (64, 32, 16):
    nb = n // B
    lo = P[: nb * B].reshape(nb, B, 3).min(1); hi = P[: nb * B].reshape(nb, B, 3).max(1)
    centers = (lo + hi) / 2; half = (hi - lo) / 2
    btree = cKDTree(centers)
    maxhalf = np.linalg.norm(half, axis=1).max()
    need_avg, need_max, union = [], [], []
    for w in waves:
        q = P[w * 64:(w + 1) * 64]; r = rk[w * 64:(w + 1) * 64]
        per = []
        allb = set()
        for qi in range(64):
            cand = btree.query_ball_point(q[qi], r[qi] + maxhalf)
            cand = np.asarray(cand)
            dd = np.maximum(np.maximum(lo[cand] - q[qi], q[qi] - hi[cand]), 0)
            ok = cand[(dd * dd).sum(1) <= r[qi] ** 2]
            per.append(len(ok)); allb.update(ok.tolist())
        need_avg.append(np.mean(per)); need_max.append(np.max(per)); union.append(len(allb))
    print(f'bucket {B}: needed per query avg {np.mean(need_avg):.1f} buckets = {np.mean(need_avg)*B:.0f} pts; max over a wave {np.mean(need_max):.1f} = {np.mean(need_max)*B:.0f} pts; union per wave {np.mean(union):.1f} buckets = {np.mean(union)*B:.0f} pts')
