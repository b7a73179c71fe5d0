"""Mirror of treee/src/calculations.rs: the `curve` branch of SegmentInformation::update (:328-373),
fast_eigenvalues (:467-498), map_to_u32 (:455-457) -- and the rest of the Calculations phase around it:
SegmentData::new (:133-175), SegmentInformation::new (:210-276) / update (:278-327) and
DisplayData::new (interactive/mod.rs:1002-1039), for every segment of a segmented cloud in one call."""
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


def curvature_ties(points, k=NEIGHBORS_COUNT, max_distance=NEIGHBORS_MAX_DISTANCE):
    """(curve, tie flags) of a host cloud (treee_curvature_ties): flag 0 rows are the reference's `curve`, bit 0 rows may
    differ by the f32 summation order, bit 1 rows by another (equally far) k-th neighbour."""
    p = host_f32(points, 3)
    out, flags = np.zeros(len(p), np.float32), np.zeros(len(p), np.uint8)
    _lib.check(_lib.lib().treee_curvature_ties(ptr(p), len(p), k, max_distance, ptr(out), ptr(flags)))
    return out, flags


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


GROUND, TRUNK, CROWN = 0, 1, 2  # Classification (calculations.rs:45-58, bincode variant indices)


class SegmentsData:
    """What `Calculations::new` leaves in `Shared.segments` (calculations.rs:97-118), struct-of-arrays over the segments:
    min / max [S,3], info [S,4] = (trunk_height, crown_height, ground_sep, crown_sep), classification [n];
    plus `CalculationProperties` (expansion, height [n]) when asked for."""

    def __init__(self, seg_min, seg_max, info, classification, expansion, height):
        self.min, self.max, self.info = seg_min, seg_max, info
        self.classification, self.expansion, self.height = classification, expansion, height


def segments_calculate(points, segment, n_segments, properties=True, ctx=None):
    """SegmentData::new for every segment (and SegmentInformation::update(calc_curve=false) when `properties`).
    numpy in -> numpy out; torch CUDA tensors in -> tensors out (no synchronisation)."""
    S = int(n_segments)
    if is_device(points):
        import torch

        from .context import dev_f32

        p = dev_f32(points, 3)
        n, dev = p.shape[0], p.device
        sg = segment.to(torch.int32).contiguous()
        assert sg.shape[0] == n and sg.device == dev
        c = ctx or default_context(dev)
        mn = torch.empty((S, 3), dtype=torch.float32, device=dev)
        mx = torch.empty((S, 3), dtype=torch.float32, device=dev)
        info = torch.empty((S, 4), dtype=torch.float32, device=dev)
        cls = torch.empty((n,), dtype=torch.int32, device=dev)
        exp = torch.empty((n,), dtype=torch.float32, device=dev) if properties else None
        hgt = torch.empty((n,), dtype=torch.float32, device=dev) if properties else None
        _lib.check(_lib.lib().treee_segments_calculate_dev(c.handle, ptr(p), ptr(sg), n, S, ptr(mn), ptr(mx), ptr(info), ptr(cls), ptr(exp), ptr(hgt)))
        return SegmentsData(mn, mx, info, cls, exp, hgt)
    p = host_f32(points, 3)
    sg = np.ascontiguousarray(segment, dtype=np.uint32)
    assert len(sg) == len(p)
    mn, mx, info = np.zeros((S, 3), np.float32), np.zeros((S, 3), np.float32), np.zeros((S, 4), np.float32)
    cls = np.zeros(len(p), np.uint32)
    exp = np.zeros(len(p), np.float32) if properties else None
    hgt = np.zeros(len(p), np.float32) if properties else None
    _lib.check(_lib.lib().treee_segments_calculate(ptr(p), ptr(sg), len(p), S, ptr(mn), ptr(mx), ptr(info), ptr(cls), ptr(exp), ptr(hgt)))
    return SegmentsData(mn, mx, info, cls, exp, hgt)


def display_values_dev(segment, n_segments, classification=None, curve=None, expansion=None, height=None, ctx=None):
    """DisplayData::new (interactive/mod.rs:1002-1039) over device tensors -> dict of int32 tensors (u32 bit patterns)
    that `PointCloudProperty` takes as the splat's per-point value."""
    import torch

    sg = segment.to(torch.int32).contiguous()
    n, dev = sg.shape[0], sg.device
    c = ctx or default_context(dev)
    ins = {"classification": classification, "curve": curve, "expansion": expansion, "height": height}
    outs = {k: (torch.empty((n,), dtype=torch.int32, device=dev) if v is not None else None) for k, v in ins.items()}
    cls = None if classification is None else classification.to(torch.int32).contiguous()
    _lib.check(_lib.lib().treee_display_values_dev(c.handle, ptr(sg), n, int(n_segments), ptr(cls), ptr(curve), ptr(expansion), ptr(height),
                                                   ptr(outs["classification"]), ptr(outs["curve"]), ptr(outs["expansion"]), ptr(outs["height"])))
    return outs
