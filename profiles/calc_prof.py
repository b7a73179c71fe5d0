"""The Calculations phase alone, for rocprofv3 (kernel trace of treee_segments_calculate_dev + treee_display_values_dev on the
15 M-point cloud):  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/calc -- python profiles/calc_prof.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import treee_amd as T
from treee_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000_000
dev = torch.device("cuda", 0)
ctx = T.default_context(dev)
pts, seg = synth.forest_cloud_torch(n, dev, plot=200.0 * (n / 15e6) ** 0.5)
which = sys.argv[2] if len(sys.argv) > 2 else "all"  # "ground": segment 0 alone (a few slices of ~300 k points); "trees": the rest
if which == "ground":
    pts, seg = pts[seg == 0].contiguous(), seg[seg == 0].contiguous()
elif which == "trees":
    pts, seg = pts[seg != 0].contiguous(), (seg[seg != 0] - 1).contiguous()
n = pts.shape[0]
nseg = int(seg.max().item()) + 1
curve = torch.rand((n,), dtype=torch.float32, device=dev)
for _ in range(3):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    r = T.segments_calculate(pts, seg, nseg, ctx=ctx)
    e1.record()
    T.display_values_dev(seg, nseg, r.classification, curve, r.expansion, r.height, ctx=ctx)
    e2.record()
    torch.cuda.synchronize()
    print(f"{which} {n}: calculate {e0.elapsed_time(e1):.3f} ms, display values {e1.elapsed_time(e2):.3f} ms, {nseg} segments")
