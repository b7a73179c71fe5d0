"""ctypes binding of libtreee_hip.so (the C ABI declared in include/treee_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at load
time, and every compute call fails with TREEE_ERR_HIP when no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TREEE_HIP_LIB") or os.path.join(_HERE, "lib", "libtreee_hip.so")  # env: A/B builds

TREEE_OK, TREEE_ERR_INVALID, TREEE_ERR_HIP, TREEE_ERR_EMPTY, TREEE_ERR_UNSUPPORTED, TREEE_ERR_NOMEM = range(6)
_ERR_NAMES = {1: "INVALID", 2: "HIP", 3: "EMPTY", 4: "UNSUPPORTED", 5: "NOMEM"}

OCT_KEYS, OCT_PERM, OCT_NODES, OCT_LEAF, OCT_LOD, OCT_SORTED_XYZ, OCT_TOP_NODES, OCT_TOP_LOD = range(8)
SHARD_LEVELS = 11


class TreeeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"treee_hip error {code} ({_ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class Entry(C.Structure):  # treee_entry == repr(C) mirror of k_nearest::Entry<f32>
    _fields_ = [("distance", C.c_float), ("_pad", C.c_uint32), ("index", C.c_uint64)]


class OctreeInfo(C.Structure):
    _fields_ = [
        ("n_points", C.c_uint64),
        ("n_nodes", C.c_uint64),
        ("n_lod", C.c_uint64),
        ("corner", C.c_float * 3),
        ("size", C.c_float),
        ("leaf_cap", C.c_uint32),
        ("lod_cap", C.c_uint32),
    ]


class ShardPlanInfo(C.Structure):  # treee_shard_plan_info
    _fields_ = [("n_total", C.c_uint64), ("world", C.c_uint32), ("top_level", C.c_uint32), ("n_top_nodes", C.c_uint32),
                ("n_top_lod", C.c_uint32), ("n_level_nodes", C.c_uint32), ("leaf_cap", C.c_uint32), ("lod_cap", C.c_uint32)]


class ShardPlanRank(C.Structure):  # treee_shard_plan_rank
    _fields_ = [("cell_begin", C.c_uint32), ("cell_end", C.c_uint32), ("first", C.c_uint64), ("count", C.c_uint64),
                ("index_base", C.c_uint64), ("level_node_first", C.c_uint32), ("level_node_count", C.c_uint32)]


class ShardLevels(C.Structure):  # treee_shard_levels
    _fields_ = [("n_nodes_total", C.c_uint64), ("n_lod_total", C.c_uint64)] + [
        (name, C.c_uint32 * SHARD_LEVELS) for name in ("node_begin", "own_node_first", "own_node_count", "lod_begin", "own_lod_first", "own_lod_count")]


_CB_GATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_CB_ALLTOALL = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                           C.POINTER(C.c_uint64))
_CB_REDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)


class CommHostOps(C.Structure):  # treee_comm_host_ops
    _fields_ = [("user", C.c_void_p), ("all_gather", _CB_GATHER), ("all_to_all_v", _CB_ALLTOALL), ("all_reduce_min_u64", _CB_REDUCE)]


# every symbol include/treee_hip.h declares: name -> (restype, argtypes)
_vp, _u64, _u32, _i32, _f32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_float
_pp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "treee_abi_version": (_i32, []),
    "treee_last_error": (C.c_char_p, []),
    "treee_device_count": (_i32, [C.POINTER(_i32)]),
    "treee_ctx_create": (_i32, [_i32, _vp, _pp]),
    "treee_ctx_destroy": (_i32, [_vp]),
    "treee_ctx_sync": (_i32, [_vp]),
    "treee_ctx_reserve": (_i32, [_vp, _u64]),
    "treee_knn_build": (_i32, [_vp, _u64, _pp]),
    "treee_knn_build_dev": (_i32, [_vp, _vp, _u64, _pp]),
    "treee_knn_from_octree_dev": (_i32, [_vp, _vp, _pp]),
    "treee_knn_from_shard_dev": (_i32, [_vp, _vp, _vp, _u64, _pp]),
    "treee_knn_build_segments_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, C.POINTER(_vp)]),
    "treee_knn_free": (_i32, [_vp]),
    "treee_knn_len": (_i32, [_vp, C.POINTER(_u64)]),
    "treee_knn_query_batch": (_i32, [_vp, _vp, _u64, _u32, _f32, _vp, _vp]),
    "treee_knn_query_batch_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _f32, _vp, _vp, _vp]),
    "treee_knn_self_dev": (_i32, [_vp, _vp, _u32, _f32, _vp, _vp, _vp]),
    "treee_knn_self_ties_dev": (_i32, [_vp, _vp, _u32, _f32, _vp, _vp, _vp, _vp]),
    "treee_knn_radius": (_i32, [_vp, _vp, _f32, _vp, _u64, C.POINTER(_u64)]),
    "treee_knn_empty": (_i32, [_vp, _vp, _u64, _f32, _vp]),
    "treee_curvature": (_i32, [_vp, _u64, _u32, _f32, _vp]),
    "treee_curvature_segments": (_i32, [_vp, _vp, _u64, _u32, _u32, _f32, _vp]),
    "treee_curvature_dev": (_i32, [_vp, _vp, _u32, _f32, _vp]),
    "treee_curvature_ties_dev": (_i32, [_vp, _vp, _u32, _f32, _vp, _vp]),
    "treee_curvature_ties": (_i32, [_vp, _u64, _u32, _f32, _vp, _vp]),
    "treee_curvature_own_dev": (_i32, [_vp, _vp, _u64, _u32, _f32, _vp]),
    "treee_map_to_u32_dev": (_i32, [_vp, _vp, _u64, _vp]),
    "treee_sort_build": (_i32, [_vp, _u64, _vp, _f32, _u32, _u32, _pp]),
    "treee_sort_build_dev": (_i32, [_vp, _vp, _u64, _vp, _f32, _u32, _u32, _pp]),
    "treee_sort_dev": (_i32, [_vp, _vp, _u64, _vp, _f32, _pp]),
    "treee_octree_complete_dev": (_i32, [_vp, _vp, _u32, _u32]),
    "treee_sort_build_knn_dev": (_i32, [_vp, _vp, _u64, _vp, _f32, _u32, _u32, _pp, _pp]),
    "treee_octree_free": (_i32, [_vp]),
    "treee_octree_info_get": (_i32, [_vp, C.POINTER(OctreeInfo)]),
    "treee_octree_copy": (_i32, [_vp, _i32, _vp, _u64]),
    "treee_octree_dev_ptr": (_i32, [_vp, _i32, _pp]),
    "treee_splat": (_i32, [_vp, _vp, _u64, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "treee_splat_dev": (_i32, [_vp, _vp, _vp, _u64, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _vp, _i32, _u32, _vp, _vp]),
    "treee_splat_octree_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _vp, _i32, _u32, _vp, _vp]),
    "treee_splat_knn_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _vp, _i32, _u32, _vp, _vp]),
    "treee_splat_resolve_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _vp, _vp]),
    "treee_perspective": (_i32, [_f32, _f32, _f32, _f32, _vp]),
    "treee_eye_dome_dev": (_i32, [_vp, _vp, _u32, _u32, _f32, _vp]),
    "treee_decode_las_records": (_i32, [_vp, _u64, _u32, _vp, _vp, _vp, _vp]),
    "treee_decode_las_records_dev": (_i32, [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp]),
    "treee_slice_keys": (_i32, [_vp, _u64, _vp]),
    "treee_slice_keys_dev": (_i32, [_vp, _vp, _u64, _vp]),
    "treee_laz_chunk_table": (_i32, [_vp, _u64, _u64, _u32, _u64, _vp, _vp, _u64, C.POINTER(_u64)]),
    "treee_laz_decompress_chunks": (_i32, [_vp, _u64, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _u64, _u32, _vp, _u64]),
    "treee_slice_bin": (_i32, [_vp, _u64, C.POINTER(C.c_int64), C.POINTER(_u64), _vp, _vp, _u64]),
    "treee_slice_bin_dev": (_i32, [_vp, _vp, _u64, C.POINTER(C.c_int64), C.POINTER(_u64), _vp, _vp, _u64]),
    "treee_select_xz_dev": (_i32, [_vp, _vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    "treee_segment_slices": (_i32, [_vp, _u64, _vp, _vp, _f32, _u64, _vp, C.POINTER(_u32)]),
    "treee_segments_calculate": (_i32, [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "treee_segments_calculate_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "treee_display_values_dev": (_i32, [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "treee_comm_create_host": (_i32, [C.POINTER(CommHostOps), _i32, _i32, _pp]),
    "treee_rccl_available": (_i32, []),
    "treee_rccl_unique_id": (_i32, [_vp]),
    "treee_comm_create_rccl": (_i32, [_vp, _i32, _i32, _i32, _pp]),
    "treee_comm_wrap_rccl": (_i32, [_vp, _i32, _i32, _pp]),
    "treee_comm_destroy": (_i32, [_vp]),
    "treee_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "treee_shard_plan_create": (_i32, [_vp, _i32, _u32, _u32, _u32, _pp]),
    "treee_shard_plan_free": (_i32, [_vp]),
    "treee_shard_plan_info_get": (_i32, [_vp, C.POINTER(ShardPlanInfo)]),
    "treee_shard_plan_rank_get": (_i32, [_vp, _i32, C.POINTER(ShardPlanRank)]),
    "treee_shard_plan_top_nodes": (_i32, [_vp, _vp]),
    "treee_shard_plan_level_nodes": (_i32, [_vp, _i32, _vp]),
    "treee_shard_plan_exchange": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "treee_shard_plan_levels": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(ShardLevels)]),
    "treee_shard_cube_dev": (_i32, [_vp, _vp, _vp, _u64, _vp, C.POINTER(_f32)]),
    "treee_shard_octree_build_dev": (_i32, [_vp, _vp, _vp, _u64, _vp, _f32, _u32, _u32, _u32, _pp]),
    "treee_shard_octree_release_local": (_i32, [_vp]),
    "treee_shard_octree_info_get": (_i32, [_vp, C.POINTER(ShardPlanInfo), C.POINTER(ShardPlanRank), C.POINTER(ShardLevels)]),
    "treee_shard_exchange_ghosts_dev": (_i32, [_vp, _vp, _vp, _u64, _vp, _u32, _f32, _u64, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "treee_frame_reduce_min_dev": (_i32, [_vp, _vp, _vp, _u64]),
    "treee_curvature_segments_multi": (_i32, [_vp, _u32, _vp, _vp, _u64, _u32, _u32, _f32, _vp]),
    "treee_segments_calculate_multi": (_i32, [_vp, _u32, _vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load libtreee_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  treee_amd has no CPU fallback."
            )
        try:
            # torch ships its own libamdhip64.so.7; load it FIRST so that libtreee_hip.so binds to
            # the same HIP runtime (same SONAME) instead of pulling in /opt/rocm's second copy --
            # two runtimes in one process cannot share device pointers or streams.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        if L.treee_abi_version() != 1:
            raise ImportError("libtreee_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc != TREEE_OK:
        raise TreeeError(rc, lib().treee_last_error().decode("utf-8", "replace"))
