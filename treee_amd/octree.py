"""Morton sort + octree + LOD (build-defined stage; the only
reference hook is the cube (corner, size) of render/src/camera_3d.rs:20-51)."""
import ctypes as C

import numpy as np

from . import _lib
from .context import NODE_DTYPE, default_context, dev_f32, host_f32, is_device, ptr

_ARR = {
    "keys": (_lib.OCT_KEYS, np.uint32, 1),
    "perm": (_lib.OCT_PERM, np.uint32, 1),
    "nodes": (_lib.OCT_NODES, NODE_DTYPE, 1),
    "leaf": (_lib.OCT_LEAF, np.uint32, 1),
    "lod": (_lib.OCT_LOD, np.uint32, 1),
    "sorted_xyz": (_lib.OCT_SORTED_XYZ, np.float32, 4),
    "top_nodes": (_lib.OCT_TOP_NODES, NODE_DTYPE, 1),
    "top_lod": (_lib.OCT_TOP_LOD, np.uint32, 1),
}


class Octree:
    def __init__(self, points, corner=None, size=0.0, leaf_cap=256, lod_cap=256, ctx=None, defer=False):
        """`defer=True` (device clouds): only the Morton sort now (treee_sort_dev); `complete()` builds the node table
        later -- e.g. on a context of a second stream while the kNN kernel of the first runs."""
        h = C.c_void_p()
        c = None if corner is None else host_f32(corner).reshape(3)
        self._deferred = None
        if is_device(points):
            self.ctx = ctx or default_context(points.device)
            p = dev_f32(points, 3)
            if defer:
                _lib.check(_lib.lib().treee_sort_dev(self.ctx.handle, ptr(p), p.shape[0], ptr(c), size, C.byref(h)))
                self._deferred = (leaf_cap, lod_cap)
            else:
                _lib.check(_lib.lib().treee_sort_build_dev(self.ctx.handle, ptr(p), p.shape[0], ptr(c), size, leaf_cap, lod_cap, C.byref(h)))
        else:
            assert not defer, "defer=True needs a device cloud"
            self.ctx = None
            p = host_f32(points, 3)
            _lib.check(_lib.lib().treee_sort_build(ptr(p), p.shape[0], ptr(c), size, leaf_cap, lod_cap, C.byref(h)))
        self._h = h
        self._refresh()

    @classmethod
    def with_knn(cls, points, corner=None, size=0.0, leaf_cap=256, lod_cap=256, ctx=None):
        """(octree, kNN structure) of one device cloud from ONE call (treee_sort_build_knn_dev): one Morton sort, the node
        table on the context's auxiliary stream beside the record gather and the kNN structure's own work.  The same arrays as
        `Octree(points)` + `KDTree.from_octree(octree)`."""
        from .knn import KDTree

        self = object.__new__(cls)
        self._deferred = None
        self.ctx = ctx or default_context(points.device)
        c = None if corner is None else host_f32(corner).reshape(3)
        p = dev_f32(points, 3)
        h, hk = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().treee_sort_build_knn_dev(self.ctx.handle, ptr(p), p.shape[0], ptr(c), size, leaf_cap, lod_cap, C.byref(h), C.byref(hk)))
        self._h = h
        self._refresh()
        tree = object.__new__(KDTree)
        tree.ctx, tree._h, tree.n, tree.device = self.ctx, hk, int(self.n_points), self.ctx.device
        return self, tree

    def _refresh(self):
        info = _lib.OctreeInfo()
        _lib.check(_lib.lib().treee_octree_info_get(self._h, C.byref(info)))
        self.n_points, self.n_nodes, self.n_lod = info.n_points, info.n_nodes, info.n_lod
        self.corner = np.array(list(info.corner), np.float32)
        self.size = float(info.size)
        self.leaf_cap, self.lod_cap = info.leaf_cap, info.lod_cap

    def complete(self, ctx=None):
        """node table, leaf ids and LOD lists of a deferred octree (treee_octree_complete_dev), on `ctx` (default: the
        sort's context).  Synchronises `ctx`'s stream only."""
        assert self._deferred is not None, "not a deferred octree (or already completed)"
        c = ctx or self.ctx
        _lib.check(_lib.lib().treee_octree_complete_dev(c.handle, self._h, self._deferred[0], self._deferred[1]))
        self._deferred = None
        self._refresh()
        return self

    def array(self, name):
        """copy one of keys / perm / nodes / leaf / lod / sorted_xyz to the host."""
        what, dt, cols = _ARR[name]
        count = {"nodes": self.n_nodes, "lod": self.n_lod, "top_nodes": getattr(self, "n_top_nodes", 0),
                 "top_lod": getattr(self, "n_top_lod", 0)}.get(name, self.n_points)
        out = np.zeros((count, cols) if cols > 1 else (count,), dt)
        if count:
            _lib.check(_lib.lib().treee_octree_copy(self._h, what, ptr(out), out.nbytes))
        return out

    def dev_ptr(self, name):
        p = C.c_void_p()
        _lib.check(_lib.lib().treee_octree_dev_ptr(self._h, _ARR[name][0], C.byref(p)))
        return p.value

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().treee_octree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedOctree(Octree):
    """One rank's piece of the octree of a cloud that is spread over the ranks of `comm` (treee_shard_octree_build_dev;
    SURVEY.md section 8 row E).  Collective: every rank constructs it with its own points.  `assemble` puts the pieces
    of all ranks together into the single-GPU tables (tests, exports)."""

    def __init__(self, points, comm, corner=None, size=0.0, leaf_cap=256, lod_cap=256, top_level=0, ctx=None):
        h = C.c_void_p()
        c = None if corner is None else host_f32(corner).reshape(3)
        self._deferred = None
        self.ctx = ctx or default_context(points.device)
        self.comm = comm
        p = dev_f32(points, 3)
        _lib.check(_lib.lib().treee_shard_octree_build_dev(self.ctx.handle, comm.handle, ptr(p), p.shape[0], ptr(c), size, leaf_cap, lod_cap, top_level, C.byref(h)))
        self._h = h
        self._refresh()
        self.info, self.rank_info, self.levels = _lib.ShardPlanInfo(), _lib.ShardPlanRank(), _lib.ShardLevels()
        _lib.check(_lib.lib().treee_shard_octree_info_get(self._h, C.byref(self.info), C.byref(self.rank_info), C.byref(self.levels)))
        self.n_top_nodes, self.n_top_lod = self.info.n_top_nodes, self.info.n_top_lod

    def release_local(self):
        """gives back the rank's own Morton-sorted input (kept for KDTree.from_shard; 24 B per point) -- call it once the kNN
        structure is built (treee_shard_octree_release_local)"""
        _lib.check(_lib.lib().treee_shard_octree_release_local(self._h))

    def piece(self):
        """this rank's arrays and numbering as plain host data (what `assemble` consumes)"""
        lv = self.levels
        return {
            "first": int(self.rank_info.first), "count": int(self.rank_info.count), "index_base": int(self.rank_info.index_base),
            "top_level": int(self.info.top_level), "n_total": int(self.info.n_total),
            "n_nodes_total": int(lv.n_nodes_total), "n_lod_total": int(lv.n_lod_total),
            "own_node_first": list(lv.own_node_first), "own_node_count": list(lv.own_node_count),
            "own_lod_first": list(lv.own_lod_first), "own_lod_count": list(lv.own_lod_count),
            "keys": self.array("keys"), "perm": self.array("perm"), "leaf": self.array("leaf"), "sorted_xyz": self.array("sorted_xyz"),
            "nodes": self.array("nodes"), "lod": self.array("lod"), "top_nodes": self.array("top_nodes"), "top_lod": self.array("top_lod"),
        }


def assemble(pieces):
    """pieces (ShardedOctree.piece() of every rank, rank order) -> dict(keys, perm, sorted_xyz, leaf, nodes, lod) of the
    whole cloud's octree: the top from any rank (they must agree), every level below it rank after rank."""
    p0 = pieces[0]
    for p in pieces[1:]:
        assert np.array_equal(p["top_nodes"].view(np.uint32), p0["top_nodes"].view(np.uint32)) and np.array_equal(p["top_lod"], p0["top_lod"])
    n, nn, nl, L = p0["n_total"], p0["n_nodes_total"], p0["n_lod_total"], p0["top_level"]
    out = {"keys": np.zeros(n, np.uint32), "perm": np.zeros(n, np.uint32), "leaf": np.zeros(n, np.uint32), "sorted_xyz": np.zeros((n, 4), np.float32),
           "nodes": np.zeros(nn, NODE_DTYPE), "lod": np.zeros(nl, np.uint32)}
    filled_nodes, filled_lod = np.zeros(nn, bool), np.zeros(nl, bool)
    nt, ntl = len(p0["top_nodes"]), len(p0["top_lod"])
    out["nodes"][:nt], out["lod"][:ntl] = p0["top_nodes"], p0["top_lod"]
    filled_nodes[:nt], filled_lod[:ntl] = True, True
    for p in pieces:
        a, c = p["first"], p["count"]
        for k in ("keys", "perm", "leaf", "sorted_xyz"):
            out[k][a : a + c] = p[k]
        no = lo = 0
        for lvl in range(L, _lib.SHARD_LEVELS):
            cn, cl = p["own_node_count"][lvl], p["own_lod_count"][lvl]
            g, gl = p["own_node_first"][lvl], p["own_lod_first"][lvl]
            assert not filled_nodes[g : g + cn].any() and not filled_lod[gl : gl + cl].any(), "two ranks claim the same node or LOD slots"
            out["nodes"][g : g + cn] = p["nodes"][no : no + cn]
            out["lod"][gl : gl + cl] = p["lod"][lo : lo + cl]
            filled_nodes[g : g + cn], filled_lod[gl : gl + cl] = True, True
            no, lo = no + cn, lo + cl
        assert no == len(p["nodes"]) and lo == len(p["lod"])
    assert filled_nodes.all() and filled_lod.all(), "node or LOD slots nobody owns"
    return out
