"""Randomised soak of the Calculations phase (csrc/calculations.hip: one convex hull per 0.1 m slice built by the reference's sequential
insertion, wave- and workgroup-cooperatively, with the LDS copies of 128 / 2 048 vertices spilling to HBM) against oracle/calc_oracle.cpp,
bit for bit.  Run by hand on the GPU box; not collected by pytest.

    python tests/soak_calculations.py <seed> <seconds>

Segments are rings and ellipses of 50 .. 6 000 points in one slice (hulls that reach and cross both LDS capacities through either splice
branch, in insertion orders from sorted-by-angle to shuffled), blobs, lines, duplicates, spirals; 1 .. 700 segments per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import treee_amd as T
from oracle import oracle as O
from test_gpu_parity import _check_segments_against_oracle

O.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    nseg = int(rng.choice([1, 3, 40, 700]))
    parts = []
    for s in range(nseg):
        kind = rng.choice(["ring", "arc", "blob", "line", "dups", "spiral"])
        m = int(rng.choice([1, 3, 50, 127, 128, 129, 131, 300, 2047, 2048, 2049, 2100, 6000])) if nseg <= 40 else int(rng.choice([1, 60, 127, 129, 140, 300]))
        cx, cz, y = float(rng.uniform(-50, 50)), float(rng.uniform(-50, 50)), float(rng.uniform(0, 3))
        if kind in ("ring", "arc"):
            a = rng.uniform(0, 2 * np.pi if kind == "ring" else 5.0, m)
            order = rng.choice(["sorted", "reversed", "shuffled"])
            a = np.sort(a) if order != "shuffled" else a
            a = a[::-1] if order == "reversed" else a
            rx, rz = float(rng.uniform(0.5, 40)), float(rng.uniform(0.5, 40))
            p = np.stack([cx + rx * np.cos(a), np.full(m, y + 0.02), cz + rz * np.sin(a)], 1)
        elif kind == "blob":
            p = np.stack([cx + rng.normal(0, 2, m), y + np.abs(rng.normal(0, 2, m)), cz + rng.normal(0, 2, m)], 1)
        elif kind == "line":
            t = rng.uniform(0, 1, m)
            p = np.stack([cx + 5 * t, np.full(m, y), cz - 3 * t], 1)
        elif kind == "dups":
            p = np.repeat(np.array([[cx, y, cz]]), m, 0)
        else:
            t = np.linspace(0, 12 * np.pi, m)
            p = np.stack([cx + 0.05 * t * np.cos(t), np.full(m, y + 0.03), cz + 0.05 * t * np.sin(t)], 1)
        parts.append(p)
    pts = np.concatenate(parts).astype(np.float32)
    seg = np.concatenate([np.full(len(p), s) for s, p in enumerate(parts)]).astype(np.uint32)
    if rng.random() < 0.4:
        perm = rng.permutation(len(pts))
        pts, seg = pts[perm], seg[perm]
    try:
        _check_segments_against_oracle(T, O, pts, seg, nseg + int(rng.integers(0, 2)), properties=bool(rng.random() < 0.5), device=bool(rng.random() < 0.5))
    except AssertionError as e:
        print("FAIL", nseg, len(pts), e)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "soak_calc_fail.npz"), pts=pts, seg=seg)
        raise SystemExit(1)
    cases += 1
print("cases", cases)
