"""ctypes/numpy front-end of the CPU ORACLE (test infrastructure, never shipped).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package `treee_amd` must not.  Parity status: unpinned (see
oracle/treee_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libtreee_oracle.so")

ENTRY_DTYPE = np.dtype([("distance", "<f4"), ("_pad", "<u4"), ("index", "<u8")])
NODE_DTYPE = np.dtype(
    [(n, "<u4") for n in ("key_prefix", "level", "first", "count", "first_child", "child_mask", "parent", "lod_first")]
)
INVALID = np.uint64(0xFFFFFFFFFFFFFFFF)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class LasHeader(C.Structure):
    _fields_ = [
        ("number_of_point_records", C.c_uint64),
        ("offset_to_point_data", C.c_uint32),
        ("header_size", C.c_uint16),
        ("point_data_record_length", C.c_uint16),
        ("point_data_record_format", C.c_uint8),
        ("version_major", C.c_uint8),
        ("version_minor", C.c_uint8),
        ("scale", C.c_double * 3),
        ("offset", C.c_double * 3),
        ("min", C.c_double * 3),
        ("max", C.c_double * 3),
        ("center", C.c_double * 3),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, u64, u32, f32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_int
        L.orc_kdtree_new.restype = vp
        L.orc_kdtree_new.argtypes = [vp, u64]
        L.orc_kdtree_free.argtypes = [vp]
        L.orc_kdtree_len.restype = u64
        L.orc_kdtree_len.argtypes = [vp]
        L.orc_kdtree_dump.argtypes = [vp, vp, vp]
        L.orc_kdtree_k_nearest.restype = u64
        L.orc_kdtree_k_nearest.argtypes = [vp, vp, vp, u64, f32]
        L.orc_kdtree_k_nearest_batch.argtypes = [vp, vp, u64, vp, vp, u64, f32, i32]
        L.orc_kdtree_nearest.restype = u64
        L.orc_kdtree_nearest.argtypes = [vp, vp, f32, vp, u64]
        L.orc_kdtree_empty.restype = i32
        L.orc_kdtree_empty.argtypes = [vp, vp, f32]
        L.orc_brute_k_nearest.restype = u64
        L.orc_brute_k_nearest.argtypes = [vp, u64, vp, vp, u64, f32]
        L.orc_fast_eigenvalues.argtypes = [vp, vp]
        L.orc_curve_from_neighbors.restype = f32
        L.orc_curve_from_neighbors.argtypes = [vp, vp, u64]
        L.orc_curvature.argtypes = [vp, u64, u64, f32, vp, i32]
        L.orc_curve_from_lists.argtypes = [vp, vp, vp, u64, u64, vp, i32]
        L.orc_curve_from_lists_trig.argtypes = [vp, vp, vp, u64, u64, vp, i32, i32]
        L.orc_trig_f32.argtypes = [vp, u64, vp, vp, vp, vp]
        L.orc_map_to_u32.restype = u32
        L.orc_map_to_u32.argtypes = [f32]
        L.orc_decode_las_records.argtypes = [vp, u64, u32, vp, vp, vp, vp]
        L.orc_las_header_parse.restype = i32
        L.orc_las_header_parse.argtypes = [vp, u64, C.POINTER(LasHeader)]
        L.orc_slice_keys.argtypes = [vp, u64, vp]
        L.orc_slice_bin.restype = u64
        L.orc_slice_bin.argtypes = [vp, u64, vp, vp, vp, u64]
        L.orc_perspective.argtypes = [f32, f32, f32, f32, vp]
        L.orc_splat.argtypes = [vp, vp, u64, vp, vp, f32, u32, u32, u32, u32, vp, vp, vp]
        L.orc_eye_dome.argtypes = [vp, u32, u32, f32, vp]
        L.orc_morton_keys.argtypes = [vp, u64, vp, f32, vp]
        L.orc_hilbert_keys.argtypes = [vp, u64, vp, f32, vp]
        L.orc_sort_keys.argtypes = [vp, u64, vp, vp]
        L.orc_octree_build.restype = u64
        L.orc_octree_build.argtypes = [vp, u64, u32, u32, vp, u64, vp, vp, u64, vp]
        L.orc_octree_build_from.restype = u64
        L.orc_octree_build_from.argtypes = [vp, u64, u32, u32, vp, u64, vp, u64, vp, vp, u64, vp]
        L.orc_hull_insert_all.restype = u64
        L.orc_hull_insert_all.argtypes = [vp, u64, vp, u64, vp]
        L.orc_size_areas.restype = u64
        L.orc_size_areas.argtypes = [vp, u64, vp, i32, f32, f32, vp, u64]
        L.orc_segment_data.argtypes = [vp, u64, vp, vp, vp, vp]
        L.orc_segment_update.argtypes = [vp, vp, u64, f32, f32, vp, vp, vp]
        L.orc_display_values.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


class KDTree:
    """k_nearest::KDTree<3, f32, ...> (k-nearest/src/kd_tree.rs:18-42)."""

    def __init__(self, points):
        self.points = _f32(points, 3)
        if len(self.points) == 0:
            raise ValueError("KDTree::new panics on empty input (kd_tree.rs:53,60)")
        self._h = lib().orc_kdtree_new(_p(self.points), len(self.points))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kdtree_free(self._h)
            self._h = None

    def __len__(self):
        return int(lib().orc_kdtree_len(self._h))

    def dump(self):
        n = len(self)
        xyz = np.empty((n, 3), np.float32)
        idx = np.empty(n, np.uint64)
        lib().orc_kdtree_dump(self._h, _p(xyz), _p(idx))
        return xyz, idx

    def k_nearest(self, q, k, max_distance):
        q = _f32(q).reshape(3)
        out = np.zeros(k, ENTRY_DTYPE)
        c = lib().orc_kdtree_k_nearest(self._h, _p(q), _p(out), k, max_distance)
        return out, int(c)

    def k_nearest_batch(self, q, k, max_distance, threads=1):
        q = _f32(q, 3)
        out = np.zeros((len(q), k), ENTRY_DTYPE)
        counts = np.zeros(len(q), np.uint32)
        lib().orc_kdtree_k_nearest_batch(self._h, _p(q), len(q), _p(out), _p(counts), k, max_distance, threads)
        return out, counts

    def nearest(self, q, max_distance):
        q = _f32(q).reshape(3)
        cap = len(self)
        out = np.zeros(cap, ENTRY_DTYPE)
        n = lib().orc_kdtree_nearest(self._h, _p(q), max_distance, _p(out), cap)
        return out[: int(n)]

    def empty(self, q, max_distance):
        q = _f32(q).reshape(3)
        return bool(lib().orc_kdtree_empty(self._h, _p(q), max_distance))


def brute_k_nearest(points, q, k, max_distance):
    points = _f32(points, 3)
    q = _f32(q).reshape(3)
    out = np.zeros(k, ENTRY_DTYPE)
    c = lib().orc_brute_k_nearest(_p(points), len(points), _p(q), _p(out), k, max_distance)
    return out, int(c)


def fast_eigenvalues(m):
    m = _f32(m).reshape(9)
    out = np.zeros(3, np.float32)
    lib().orc_fast_eigenvalues(_p(m), _p(out))
    return out


def curve_from_neighbors(points, entries):
    points = _f32(points, 3)
    entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
    return float(lib().orc_curve_from_neighbors(_p(points), _p(entries), len(entries)))


def curve_from_lists(points, index, counts=None, threads=1, libm=False):
    """calculations.rs:342-368 over given neighbour lists (index [nq, k] u32, first counts[i] valid), summed in list order.
    libm=True: acos / cos of this host's libm (what Rust's f32::acos / cos forward to) instead of the shared header"""
    points = _f32(points, 3)
    index = np.ascontiguousarray(index, dtype=np.uint32)
    nq, k = index.shape
    cnt = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
    out = np.zeros(nq, np.float32)
    lib().orc_curve_from_lists_trig(_p(points), _p(index), None if cnt is None else _p(cnt), nq, k, _p(out), threads, 1 if libm else 0)
    return out


def trig_f32(x):
    """(shared acos, shared cos, libm acosf, libm cosf) of the f32 array x (include/treee_f32_trig.h)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    outs = [np.zeros(len(x), np.float32) for _ in range(4)]
    lib().orc_trig_f32(_p(x), len(x), *[_p(o) for o in outs])
    return outs


def curvature(points, k=31, max_distance=np.finfo(np.float32).max, threads=1):
    points = _f32(points, 3)
    out = np.zeros(len(points), np.float32)
    lib().orc_curvature(_p(points), len(points), k, max_distance, _p(out), threads)
    return out


def map_to_u32(v):
    return int(lib().orc_map_to_u32(float(v)))


def las_header_parse(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    h = LasHeader()
    rc = lib().orc_las_header_parse(_p(buf), len(buf), C.byref(h))
    if rc != 0:
        raise ValueError("CorruptFile")
    return h


def decode_las_records(rec, n, stride, scale, offset, center):
    rec = np.ascontiguousarray(np.frombuffer(bytes(rec), dtype=np.uint8))
    s, o, c = (np.ascontiguousarray(a, dtype=np.float64) for a in (scale, offset, center))
    out = np.zeros((n, 3), np.float32)
    lib().orc_decode_las_records(_p(rec), n, stride, _p(s), _p(o), _p(c), _p(out))
    return out


def slice_keys(points):
    points = _f32(points, 3)
    out = np.zeros(len(points), np.int64)
    lib().orc_slice_keys(_p(points), len(points), _p(out))
    return out


def slice_bin(points):
    points = _f32(points, 3)
    n = len(points)
    mn = C.c_int64(0)
    order = np.zeros(n, np.uint64)
    layers = lib().orc_slice_bin(_p(points), n, C.byref(mn), _p(order), None, 0)
    offsets = np.zeros(layers + 1, np.uint64)
    lib().orc_slice_bin(_p(points), n, C.byref(mn), _p(order), _p(offsets), layers + 1)
    return int(mn.value), order, offsets


def perspective(aspect, fovy, znear, zfar):
    out = np.zeros(16, np.float32)
    lib().orc_perspective(aspect, fovy, znear, zfar, _p(out))
    return out


def splat(points, value, view, proj, scale, vmin, vmax, w, h):
    points = _f32(points, 3)
    value = np.ascontiguousarray(value, dtype=np.uint32)
    view = _f32(view).reshape(16)
    proj = _f32(proj).reshape(16)
    depth = np.zeros((h, w), np.float32)
    ids = np.zeros((h, w), np.uint32)
    index = np.zeros((h, w), np.uint32)
    lib().orc_splat(_p(points), _p(value), len(points), _p(view), _p(proj), scale, vmin, vmax, w, h, _p(depth), _p(ids), _p(index))
    return depth, ids, index


def eye_dome(depth, strength):
    depth = _f32(depth)
    h, w = depth.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_eye_dome(_p(depth), w, h, strength, _p(out))
    return out


def morton_keys(points, corner, size):
    points = _f32(points, 3)
    corner = _f32(corner).reshape(3)
    out = np.zeros(len(points), np.uint32)
    lib().orc_morton_keys(_p(points), len(points), _p(corner), size, _p(out))
    return out


def hilbert_keys(points, corner, size):
    points = _f32(points, 3)
    corner = _f32(corner).reshape(3)
    out = np.zeros(len(points), np.uint32)
    lib().orc_hilbert_keys(_p(points), len(points), _p(corner), size, _p(out))
    return out


def sort_keys(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    sk = np.zeros(len(keys), np.uint32)
    perm = np.zeros(len(keys), np.uint32)
    lib().orc_sort_keys(_p(keys), len(keys), _p(sk), _p(perm))
    return sk, perm


def octree_build(sorted_keys, leaf_cap=256, lod_cap=256):
    sorted_keys = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    n = len(sorted_keys)
    lod_total = C.c_uint64(0)
    nn = lib().orc_octree_build(_p(sorted_keys), n, leaf_cap, lod_cap, None, 0, None, None, 0, C.byref(lod_total))
    nodes = np.zeros(nn, NODE_DTYPE)
    leaf = np.zeros(n, np.uint32)
    lod = np.zeros(lod_total.value, np.uint32)
    lib().orc_octree_build(_p(sorted_keys), n, leaf_cap, lod_cap, _p(nodes), nn, _p(leaf), _p(lod), len(lod), C.byref(lod_total))
    return nodes, leaf, lod


def octree_build_from(sorted_keys, roots, leaf_cap=256, lod_cap=256):
    """the build continued from `roots` (NODE_DTYPE, one level, `first` relative to sorted_keys): local node table (roots
    first), leaf[n] (local node index; positions no root covers keep 0xFFFFFFFF), LOD list (local positions)"""
    sorted_keys = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    roots = np.ascontiguousarray(roots, dtype=NODE_DTYPE)
    n = len(sorted_keys)
    lod_total = C.c_uint64(0)
    nn = lib().orc_octree_build_from(_p(sorted_keys), n, leaf_cap, lod_cap, _p(roots), len(roots), None, 0, None, None, 0, C.byref(lod_total))
    nodes = np.zeros(nn, NODE_DTYPE)
    leaf = np.full(n, 0xFFFFFFFF, np.uint32)
    lod = np.zeros(lod_total.value, np.uint32)
    lib().orc_octree_build_from(_p(sorted_keys), n, leaf_cap, lod_cap, _p(roots), len(roots), _p(nodes), nn, _p(leaf), _p(lod), len(lod),
                                C.byref(lod_total))
    return nodes, leaf, lod


# ---- Calculations phase (calc_oracle.cpp) ------------------------------------------------------------
def hull_insert_all(xz):
    """Tree::new(xz[0]) then Tree::insert for the rest, in order -> (hull vertices (m,2), area)."""
    xz = _f32(xz, 2)
    out = np.zeros((len(xz) + 2, 2), np.float32)
    area = np.zeros(1, np.float32)
    m = lib().orc_hull_insert_all(_p(xz), len(xz), _p(out), len(out), _p(area))
    return out[: int(m)], float(area[0])


def size_areas(points, min_y, height, cls=None, want_cls=2):
    points = _f32(points, 3)
    c = None if cls is None else np.ascontiguousarray(cls, dtype=np.uint8)
    m = lib().orc_size_areas(_p(points), len(points), _p(c), want_cls, min_y, height, None, 0)
    out = np.zeros(int(m), np.float32)
    lib().orc_size_areas(_p(points), len(points), _p(c), want_cls, min_y, height, _p(out), len(out))
    return out


def segment_data(points):
    """SegmentData::new for one segment -> (min[3], max[3], info[4], classification u8[n])."""
    points = _f32(points, 3)
    mn, mx, info = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(4, np.float32)
    cls = np.zeros(len(points), np.uint8)
    lib().orc_segment_data(_p(points), len(points), _p(mn), _p(mx), _p(info), _p(cls))
    return mn, mx, info, cls


def segment_update(points, cls, min_y, max_y, info):
    points = _f32(points, 3)
    cls = np.ascontiguousarray(cls, dtype=np.uint8)
    info = _f32(info).reshape(4)
    exp, hgt = np.zeros(len(points), np.float32), np.zeros(len(points), np.float32)
    lib().orc_segment_update(_p(points), _p(cls), len(points), min_y, max_y, _p(info), _p(exp), _p(hgt))
    return exp, hgt


def display_values(cls, curve, expansion, height):
    cls = np.ascontiguousarray(cls, dtype=np.uint8)
    curve, expansion, height = _f32(curve), _f32(expansion), _f32(height)
    n = len(cls)
    out = [np.zeros(n, np.uint32) for _ in range(4)]
    lib().orc_display_values(_p(cls), _p(curve), _p(expansion), _p(height), n, *[_p(o) for o in out])
    return out
