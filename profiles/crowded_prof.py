"""Why is kNN per point slower on the 1.005 G-point whole cloud (cube 1800 m: 2^-10 cells of 1.76 m, every cell crowded)?
A few 15 M-point tiles placed at the corners of the 1800 m x 1600 m region reproduce the cell size with 60 M points.
Run with TREEE_KNN_STATS=1 for the per-wave counters.  usage: python profiles/crowded_prof.py [spread|adjacent]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import treee_amd as T
from treee_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else "spread"
dev = torch.device("cuda", 0)
n = 15_000_000
origins = [(0.0, 0.0), (1600.0, 0.0), (0.0, 1400.0), (1600.0, 1400.0)] if mode == "spread" else [(0.0, 0.0), (200.0, 0.0), (0.0, 200.0), (200.0, 200.0)]
parts = [synth.forest_cloud_torch(n, dev, plot=200.0, seed=synth.SEED_FOREST + 100 + i, origin=o)[0] for i, o in enumerate(origins)]
pts = torch.cat(parts)
del parts
ctx = T.default_context(dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    oc = T.Octree(pts, ctx=ctx)
    tree = T.KDTree.from_octree(oc, ctx)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    curve = tree.curvature_dev(31, ctx=ctx)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{mode}: n={pts.shape[0]} cube={oc.size:.0f} m structure {1e3*(t1-t0):.1f} ms, knn+curve {1e3*(t2-t1):.1f} ms = {1e3*(t2-t1)/(pts.shape[0]/15e6):.2f} ms per 15 M", flush=True)
    tree.close(); oc.close()
