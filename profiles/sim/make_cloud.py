"""Input of profiles/sim/knn_sim.cpp: a forest patch at the 15 M configuration's density (375 points / m^2), in the kNN
structure's order (Hilbert index of the 2^-10 cells of the cloud's cube, then original index), as raw float32 [n][3]."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from treee_amd import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/knn_sim_cloud.f32"
O.build()
pts, _ = synth.forest_cloud(n, plot=200.0 * (n / 15_000_000) ** 0.5)
mn = pts.min(0)
size = float((pts.max(0) - mn).max())
h = O.hilbert_keys(pts, mn, size)
order = np.argsort(h, kind="stable")
pts[order].astype(np.float32).tofile(out)
print(out, n)
