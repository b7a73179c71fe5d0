"""Mirror of the kNN consumer in treee/src/calculations.rs: the `curve` branch of
SegmentInformation::update (:328-373), fast_eigenvalues (:467-498) and map_to_u32 (:455-457)."""
import numpy as np

from . import _lib
from .context import F32_MAX, default_context, host_f32, is_device, ptr
from .knn import KDTree

NEIGHBORS_COUNT = 31  # calculations.rs:278
NEIGHBORS_MAX_DISTANCE = F32_MAX  # calculations.rs:279


def curvature(points, k=NEIGHBORS_COUNT, max_distance=NEIGHBORS_MAX_DISTANCE, ctx=None):
    """curve[i] = 3*eig3/(eig1+eig2+eig3) of the covariance of the k nearest neighbours of point i."""
    if is_device(points):
        tree = KDTree.new(points, ctx)
        out = tree.curvature_dev(k, max_distance, ctx)
        tree.ctx.sync()
        tree.close()
        return out
    p = host_f32(points, 3)
    out = np.zeros(len(p), np.float32)
    _lib.check(_lib.lib().treee_curvature(ptr(p), len(p), k, max_distance, ptr(out)))
    return out


def curvature_segments(points, segment, n_segments, k=NEIGHBORS_COUNT, max_distance=NEIGHBORS_MAX_DISTANCE, ctx=None):
    """`curve` of every point inside its own segment: SegmentInformation::update(calc_curve=true) for all
    segments in one pass (calculations.rs:328-373 per segment, interactive/mod.rs:224-231)."""
    if is_device(points):
        tree = KDTree.new_segments(points, segment, n_segments, ctx)
        out = tree.curvature_dev(k, max_distance, ctx)
        tree.ctx.sync()
        tree.close()
        return out
    p = host_f32(points, 3)
    sg = np.ascontiguousarray(segment, dtype=np.uint32)
    assert len(sg) == len(p)
    out = np.zeros(len(p), np.float32)
    _lib.check(_lib.lib().treee_curvature_segments(ptr(p), ptr(sg), len(p), int(n_segments), k, max_distance, ptr(out)))
    return out


def map_to_u32_dev(values, ctx=None):
    """map_to_u32 over a device tensor: (value * u32::MAX as f32) as u32 (saturating)."""
    import torch

    c = ctx or default_context(values.device)
    out = torch.empty(values.shape, dtype=torch.int32, device=values.device)
    _lib.check(_lib.lib().treee_map_to_u32_dev(c.handle, ptr(values), values.numel(), ptr(out)))
    return out
