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
        count = {"nodes": self.n_nodes, "lod": self.n_lod}.get(name, self.n_points)
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
