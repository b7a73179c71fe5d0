"""The per-segment kNN structure + curvature alone, for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/segknn -- python profiles/segknn_prof.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import treee_amd as T
from treee_amd import synth

n = 15_000_000
dev = torch.device("cuda", 0)
ctx = T.default_context(dev)
pts, seg = synth.forest_cloud_torch(n, dev)
nseg = int(seg.max().item()) + 1
out = torch.empty((n,), dtype=torch.float32, device=dev)
for _ in range(3):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    tree = T.KDTree.new_segments(pts, seg, nseg, ctx)
    e1.record()
    tree.curvature_dev(31, 3.4028234663852886e38, ctx, out=out)
    e2.record()
    torch.cuda.synchronize()
    print(f"build {e0.elapsed_time(e1):.3f} ms, kNN + curvature {e1.elapsed_time(e2):.3f} ms")
    tree.close()
