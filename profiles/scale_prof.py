"""kNN + curvature time per 15 M points as the cloud grows on the 9 x 8 tile grid of configs[4] (same density, same local
geometry; only n, the cube and the cell size change).  usage: python profiles/scale_prof.py 4 16 32 67"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import treee_amd as T
from treee_amd import synth

dev = torch.device("cuda", 0)
n = 15_000_000
base = [synth.forest_cloud_torch(n, dev, plot=200.0, seed=synth.SEED_FOREST + 100 + i)[0] for i in range(3)]
ctx = T.default_context(dev)
for tiles in [int(a) for a in sys.argv[1:]]:
    pts = torch.empty((tiles * n, 3), dtype=torch.float32, device=dev)
    for i in range(tiles):
        pts[i * n:(i + 1) * n] = base[i % 3]
        pts[i * n:(i + 1) * n, 0] += (i % 9) * 200.0 - 800.0
        pts[i * n:(i + 1) * n, 2] += (i // 9) * 200.0 - 700.0
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oc = T.Octree(pts, ctx=ctx)
        tree = T.KDTree.from_octree(oc, ctx)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        curve = tree.curvature_dev(31, ctx=ctx)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"tiles={tiles} n={pts.shape[0]} cube={oc.size:.0f} m: structure {1e3*(t1-t0):.1f} ms, knn+curve {1e3*(t2-t1):.1f} ms = {1e3*(t2-t1)/tiles:.2f} ms per 15 M", flush=True)
        del curve
        tree.close(); oc.close()
    del pts
    torch.cuda.empty_cache()
