"""Mirror of the Segmenting phase (treee/src/segmenting.rs): `segmentation` (:110-235) over the y-slices of a loaded
cloud -> one segment id per point, the `segment: Vec<u32>` of `Event::PointCloud` (treee/src/program.rs:18-22) that the
Calculations phase, the per-segment kNN and the splat consume.  Host code in the library, as in the reference
(csrc/segmenting.cpp lists the four places where the reference's randomness is replaced by fixed rules)."""
import ctypes as C

import numpy as np

from . import _lib
from .context import host_f32, ptr

DEFAULT_MAX_DISTANCE = 0.75  # segmenting.rs:16
DELETED_INDEX = 0  # interactive/mod.rs:19


def segment_slices(points, bbox_min=None, bbox_max=None, max_distance=DEFAULT_MAX_DISTANCE, seed=0):
    """-> (segment u32[n], n_segments).  `bbox_min` / `bbox_max`: the loaded cloud's min / max (x, y, z) as `Loading` keeps
    them (loading.rs:16-17; default: of `points`); only x and z are used (segmenting.rs:148-154)."""
    p = host_f32(points, 3)
    if bbox_min is None or bbox_max is None:  # the library takes the box from the points (numpy's min / max over 1.5 M rows cost a third of the call)
        assert bbox_min is None and bbox_max is None
        bmin = bmax = None
    else:
        lo, hi = np.asarray(bbox_min, np.float32), np.asarray(bbox_max, np.float32)
        bmin = np.ascontiguousarray([lo[0], lo[2]], dtype=np.float32)
        bmax = np.ascontiguousarray([hi[0], hi[2]], dtype=np.float32)
    seg = np.empty(len(p), np.uint32)
    nseg = C.c_uint32(0)
    _lib.check(_lib.lib().treee_segment_slices(ptr(p), len(p), ptr(bmin), ptr(bmax), float(max_distance), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(seg),
                                               C.byref(nseg)))
    return seg, int(nseg.value)
