"""Randomised soak of the Segmenting phase (library C++, csrc/segmenting.cpp -- grid over the areas, tombstones, inscribed-disc accept,
chunked assignment with time stamps) against the independent pure-Python restatement of the reference's lines
(oracle/segmenting_restatement.py: plain lists, every area scanned, the VecDeque moved to front).  CPU only; not collected by pytest.

    python tests/soak_segmenting.py <seed> <seconds>

Forest patches, columns, rings (every point a hull vertex), grids with equal coordinates, duplicates, sparse slices; max_distance from
0.2 to 2.0; given and computed boxes.  Every point's id must be identical."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import treee_amd as T
from oracle import segmenting_restatement as R
from treee_amd import synth

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    kind = rng.choice(["forest", "columns", "rings", "grid", "mix"])
    n = int(rng.choice([1, 7, 300, 2500, 6000]))
    if kind == "forest":
        pts, _ = synth.forest_cloud(max(n, 50), plot=float(rng.choice([4.0, 9.0, 20.0])), seed=int(rng.integers(1, 1 << 30)))
    elif kind == "columns":
        c = rng.uniform(-8, 8, (int(rng.integers(1, 9)), 2))
        k = rng.integers(0, len(c), n)
        pts = np.stack([c[k, 0] + rng.normal(0, 0.3, n), rng.uniform(0, float(rng.choice([0.5, 7.0])), n), c[k, 1] + rng.normal(0, 0.3, n)], 1)
    elif kind == "rings":
        a = rng.uniform(0, 2 * np.pi, n)
        r = rng.choice([0.5, 2.0, 6.0], n)
        pts = np.stack([r * np.cos(a), rng.uniform(0, 2.0, n), r * np.sin(a)], 1)
    elif kind == "grid":
        g = rng.integers(0, 9, (n, 3)).astype(np.float64) * float(rng.choice([0.1, 0.5, 1.0]))
        pts = g
    else:
        a, _ = synth.forest_cloud(max(n, 50), plot=6.0, seed=int(rng.integers(1, 1 << 30)))
        pts = np.concatenate([a, a[: len(a) // 3], a[:5] + np.array([30.0, 0.0, -20.0])])
    pts = np.ascontiguousarray(pts, np.float32)
    md = float(rng.choice([0.2, 0.75, 1.3, 2.0]))
    seed = int(rng.integers(0, 1 << 40))
    lo, hi = pts.min(0), pts.max(0)
    if rng.random() < 0.5:
        pad = float(rng.choice([0.0, 3.0]))
        bmin, bmax = (lo[0] - pad, lo[2] - pad), (hi[0] + pad, hi[2] + pad)
        got, S = T.segment_slices(pts, np.array([bmin[0], 0, bmin[1]], np.float32), np.array([bmax[0], 0, bmax[1]], np.float32), max_distance=md, seed=seed)
    else:
        bmin, bmax = (lo[0], lo[2]), (hi[0], hi[2])
        got, S = T.segment_slices(pts, max_distance=md, seed=seed)
    want, S2 = R.segment_slices(pts, bmin, bmax, md, seed)
    if S != S2 or not np.array_equal(got, want):
        print("FAIL", kind, len(pts), md, seed, "ids differ at", int((got != want).sum()), "points; segments", S, S2)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "soak_seg_fail.npz"), pts=pts, md=md, seed=seed, bmin=bmin, bmax=bmax)
        raise SystemExit(1)
    cases += 1
print("cases", cases)
