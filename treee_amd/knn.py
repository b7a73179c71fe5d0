"""Host-side mirror of the `k-nearest` crate interface (k-nearest/src/lib.rs:8-13) on top of the
C ABI: KDTree::{new, k_nearest, nearest_to_position, nearest, empty} (kd_tree.rs:35,82,86,92,98)
and treee's wrapper NeighborsTree::{new, get} (treee/src/calculations.rs:392-419).

Same names, argument meaning and error behaviour: empty input and k == 0 raise (the reference
panics), distances are squared, `max_distance` is strict, results are ascending, "initial
values are never read".
"""
import ctypes as C

import numpy as np

from . import _lib
from .context import ENTRY_DTYPE, F32_MAX, default_context, dev_f32, host_f32, is_device, ptr


class KDTree:
    """k_nearest::KDTree<3, f32, Point3<f32>, Adapter, EuclideanDistanceSquared>."""

    def __init__(self):
        raise TypeError("use KDTree.new(points)")

    @classmethod
    def new(cls, points, ctx=None):
        """KDTree::new(&[Point]) -- copies the coordinates (kd_tree.rs:36-39).

        `points`: numpy (n,3) float32 on the host (drop-in path) or a torch CUDA tensor
        (HBM-resident path)."""
        self = object.__new__(cls)
        h = C.c_void_p()
        if is_device(points):
            self.ctx = ctx or default_context(points.device)
            p = dev_f32(points, 3)
            self.n = p.shape[0]
            _lib.check(_lib.lib().treee_knn_build_dev(self.ctx.handle, ptr(p), self.n, C.byref(h)))
            self.device = p.device
        else:
            p = host_f32(points, 3)
            self.n = p.shape[0]
            _lib.check(_lib.lib().treee_knn_build(ptr(p), self.n, C.byref(h)))
            self.ctx = None
            self.device = None
        self._h = h
        return self

    @classmethod
    def new_segments(cls, points, segment, n_segments, ctx=None):
        """One structure that behaves as `KDTree::new(&segment_points)` per segment (calculations.rs:97-118,329):
        `segment` is the u32 id of every point (the `segment: Vec<u32>` array the reference keeps beside the
        points), ids in [0, n_segments).  Device tensors only; serves self_neighbors_dev / curvature_dev."""
        import torch

        self = object.__new__(cls)
        self.ctx = ctx or default_context(points.device)
        p = dev_f32(points, 3)
        sg = segment.to(torch.int32).contiguous()
        assert sg.shape[0] == p.shape[0] and sg.device == p.device
        self.n = p.shape[0]
        h = C.c_void_p()
        _lib.check(_lib.lib().treee_knn_build_segments_dev(self.ctx.handle, ptr(p), ptr(sg), self.n, int(n_segments), C.byref(h)))
        self._h, self.device = h, p.device
        return self

    @classmethod
    def from_octree(cls, octree, ctx=None):
        """kNN structure over the cloud `octree` has already Morton-sorted (no second sort)."""
        self = object.__new__(cls)
        self.ctx = ctx or octree.ctx or default_context()
        h = C.c_void_p()
        _lib.check(_lib.lib().treee_knn_from_octree_dev(self.ctx.handle, octree._h, C.byref(h)))
        self._h, self.n, self.device = h, int(octree.n_points), self.ctx.device
        return self

    @classmethod
    def from_shard(cls, sharded_octree, ghosts, ctx=None):
        """kNN structure of one rank of a sharded cloud: the rank's own points (which `sharded_octree` already holds in
        Morton order of the common cube) followed by `ghosts` (float32 [g, 3] on the device; may be empty) -- no second
        sort of the tile (treee_knn_from_shard_dev).  Point i < n_own is own point i, n_own + j is ghost j."""
        self = object.__new__(cls)
        self.ctx = ctx or sharded_octree.ctx or default_context()
        gh = dev_f32(ghosts, 3)
        h = C.c_void_p()
        _lib.check(_lib.lib().treee_knn_from_shard_dev(self.ctx.handle, sharded_octree._h, ptr(gh) if gh.shape[0] else None, gh.shape[0], C.byref(h)))
        n = C.c_uint64()
        _lib.check(_lib.lib().treee_knn_len(h, C.byref(n)))
        self._h, self.n, self.device = h, int(n.value), self.ctx.device
        return self

    def __len__(self):
        return self.n

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().treee_knn_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host slices in, host slices out (what the Rust call sites pass) -------------------
    def k_nearest(self, point, k, max_distance=F32_MAX):
        """KDTree::k_nearest(&Point, &mut [Entry; k], max) -> (entries[k], count)."""
        out, counts = self.k_nearest_batch(host_f32(point).reshape(1, 3), k, max_distance)
        return out[0], int(counts[0])

    nearest_to_position = k_nearest  # kd_tree.rs:92-96: same thing for a raw [f32; 3]

    def k_nearest_batch(self, positions, k, max_distance=F32_MAX):
        q = host_f32(positions, 3)
        out = np.zeros((len(q), k), ENTRY_DTYPE)
        counts = np.zeros(len(q), np.uint32)
        _lib.check(_lib.lib().treee_knn_query_batch(self._h, ptr(q), len(q), k, max_distance, ptr(out), ptr(counts)))
        return out, counts

    def nearest(self, point, max_distance):
        """KDTree::nearest(&Point, max) -> Vec<Entry> ascending by distance."""
        q = host_f32(point).reshape(3)
        found = C.c_uint64(0)
        _lib.check(_lib.lib().treee_knn_radius(self._h, ptr(q), max_distance, None, 0, C.byref(found)))
        out = np.zeros(found.value, ENTRY_DTYPE)
        if found.value:
            _lib.check(_lib.lib().treee_knn_radius(self._h, ptr(q), max_distance, ptr(out), found.value, C.byref(found)))
        return out

    def empty(self, point, max_distance):
        """KDTree::empty(&Point, max) -> bool."""
        return bool(self.empty_batch(host_f32(point).reshape(1, 3), max_distance)[0])

    def empty_batch(self, positions, max_distance):
        q = host_f32(positions, 3)
        out = np.zeros(len(q), np.uint8)
        _lib.check(_lib.lib().treee_knn_empty(self._h, ptr(q), len(q), max_distance, ptr(out)))
        return out.astype(bool)

    # -- HBM-resident forms ------------------------------------------------------------------
    def _ctx(self, ctx):
        c = ctx or self.ctx
        if c is None:
            c = default_context()
        return c

    def k_nearest_dev(self, positions, k, max_distance=F32_MAX, ctx=None):
        """device query batch: returns (dist[nq,k] f32, index[nq,k] u32 as int64 view-free tensors, counts)."""
        import torch

        q = dev_f32(positions, 3)
        nq = q.shape[0]
        dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        index = torch.empty((nq, k), dtype=torch.int32, device=q.device)
        counts = torch.empty((nq,), dtype=torch.int32, device=q.device)
        _lib.check(
            _lib.lib().treee_knn_query_batch_dev(self._ctx(ctx).handle, self._h, ptr(q), nq, k, max_distance, ptr(dist), ptr(index), ptr(counts))
        )
        return dist, index, counts

    def self_neighbors_dev(self, k, max_distance=F32_MAX, ctx=None, want_dist=True, want_index=True, ties=False):
        """every tree point queries itself (the loop of calculations.rs:332-340).  `ties=True` also returns the per-row tie
        flags (uint8; treee_knn_self_ties_dev: bit 0 = equal distances inside the k, bit 1 = the k-th and the next equally far)."""
        import torch

        c = self._ctx(ctx)
        dev = c.device
        dist = torch.empty((self.n, k), dtype=torch.float32, device=dev) if want_dist else None
        index = torch.empty((self.n, k), dtype=torch.int32, device=dev) if want_index else None
        counts = torch.empty((self.n,), dtype=torch.int32, device=dev)
        if ties:
            flags = torch.empty((self.n,), dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib().treee_knn_self_ties_dev(c.handle, self._h, k, max_distance, ptr(dist), ptr(index), ptr(counts), ptr(flags)))
            return dist, index, counts, flags
        _lib.check(_lib.lib().treee_knn_self_dev(c.handle, self._h, k, max_distance, ptr(dist), ptr(index), ptr(counts)))
        return dist, index, counts

    def curvature_ties_dev(self, k=31, max_distance=F32_MAX, ctx=None, out=None):
        """(curve, tie flags) in one pass (treee_curvature_ties_dev)"""
        import torch

        c = self._ctx(ctx)
        if out is None:
            out = torch.empty((self.n,), dtype=torch.float32, device=c.device)
        flags = torch.empty((self.n,), dtype=torch.uint8, device=c.device)
        _lib.check(_lib.lib().treee_curvature_ties_dev(c.handle, self._h, k, max_distance, ptr(out), ptr(flags)))
        return out, flags

    def curvature_dev(self, k=31, max_distance=F32_MAX, ctx=None, out=None, n_own=None):
        """`n_own`: only the first n_own points (original order) are queried -- a shard's ghost points behind them
        serve as neighbours only."""
        import torch

        c = self._ctx(ctx)
        if out is None:
            out = torch.empty((self.n if n_own is None else n_own,), dtype=torch.float32, device=c.device)
        if n_own is None:
            _lib.check(_lib.lib().treee_curvature_dev(c.handle, self._h, k, max_distance, ptr(out)))
        else:
            assert out.numel() >= n_own
            _lib.check(_lib.lib().treee_curvature_own_dev(c.handle, self._h, int(n_own), k, max_distance, ptr(out)))
        return out


class NeighborsTree:
    """treee/src/calculations.rs:392-419."""

    def __init__(self, points):
        self.tree = KDTree.new(points)

    @classmethod
    def new(cls, points):
        return cls(points)

    def get(self, index, data, k, max_distance):
        """NeighborsTree::get(index, data, location, max) -> &location[0..l]"""
        out, count = self.tree.k_nearest(np.asarray(data)[index], k, max_distance)
        return out[:count]
