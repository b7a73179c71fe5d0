"""Randomised soak of the sorted-record splat (treee_splat_octree_dev: LDS tile per workgroup, fallback for large boxes, deferred
large splats) against the fragment-by-fragment draw of the caller's cloud (treee_splat_dev), which the parity tests hold bit-exact to
the software rasteriser.  Not collected by pytest: run by hand on the GPU box.

    python tests/soak_splat.py <seed> <seconds>

Random clouds (forest patches, blobs, coincident points), cameras anywhere (inside the cloud, far away, looking away), splat scales from
a hundredth of a pixel to half the frame, frames from 3 x 2 to 4096 x 2160, value ranges that cull.  Depth and id must be identical."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import treee_amd as T
from treee_amd import synth

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    kind = rng.choice(["forest", "blob", "dups", "tiny"])
    if kind == "forest":
        pts, _ = synth.forest_cloud(int(rng.choice([5_000, 200_000, 1_500_000])), plot=float(rng.choice([5.0, 40.0, 120.0])), seed=int(rng.integers(1, 1 << 30)))
    elif kind == "blob":
        pts = (rng.normal(0, float(rng.choice([0.01, 1.0, 30.0])), (int(rng.choice([1, 64, 1025, 300_000])), 3))).astype(np.float32)
    elif kind == "dups":
        base = rng.normal(0, 3.0, (int(rng.choice([3, 500])), 3)).astype(np.float32)
        pts = np.repeat(base, int(rng.choice([2, 700])), 0)
    else:
        pts = rng.normal(0, 1.0, (int(rng.integers(1, 5)), 3)).astype(np.float32)
    val = rng.integers(0, 2**32, len(pts), dtype=np.uint64).astype(np.uint32)
    d_pts, d_val = torch.from_numpy(pts).cuda(), torch.from_numpy(val.view(np.int32)).cuda()
    oc = T.Octree(d_pts)
    cloud, prop = T.PointCloud(d_pts), T.PointCloudProperty(d_val)
    for _ in range(4):
        w, h = (int(rng.integers(3, 400)), int(rng.integers(2, 300))) if rng.random() < 0.8 else (int(rng.choice([1920, 4096])), int(rng.choice([1080, 2160])))
        a, b = rng.uniform(0, 6.28), rng.uniform(-1.6, 1.6)
        yaw = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float32)
        dist = float(rng.choice([0.0, 0.3, 5.0, 60.0, 900.0]))
        tf = yaw @ T.rotation_x(float(b)) @ T.translation(float(rng.normal(0, 2)), float(rng.normal(0, 2)), dist)
        view, proj = T.view_from_transform(tf.astype(np.float32)), T.Camera3D(w / h).projection()
        vmin, vmax = (0, 0xFFFFFFFF) if rng.random() < 0.6 else (int(rng.integers(0, 2**31)), int(rng.integers(2**31, 2**32)))
        env = T.PointCloudEnvironment(vmin, vmax, float(rng.choice([1e-3, 0.05, 0.4, 3.0, 40.0])))
        d0, i0 = cloud.render(prop, view, proj, env, w, h)
        d1, i1 = cloud.render(prop, view, proj, env, w, h, octree=oc)
        if not (torch.equal(d0.view(torch.int32), d1.view(torch.int32)) and torch.equal(i0, i1)):
            print("FAIL", kind, len(pts), w, h, dist, env.scale, vmin, vmax, "differing pixels", int((d0.view(torch.int32) != d1.view(torch.int32)).sum()))
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "soak_splat_fail.npz"), pts=pts, val=val, view=view, proj=proj, wh=[w, h], env=[vmin, vmax, env.scale])
            raise SystemExit(1)
        cases += 1
    oc.close()
print("cases", cases)
