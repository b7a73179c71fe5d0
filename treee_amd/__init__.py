"""treee_amd -- MI355X-native hot path of antonWetzel/treee behind a C ABI (include/treee_hip.h).

Only what the path needs: csrc/ (HIP kernels + C ABI), lib/ (the built libtreee_hip.so) and thin
Python mirrors of the reference interfaces (k-nearest KDTree, render PointCloud, laz Chunk).
The HIP library is mandatory: importing a mirror class without it raises ImportError.
"""
from ._lib import LIB_PATH, SYMBOLS, TreeeError, lib  # noqa: F401
from .context import ENTRY_DTYPE, F32_MAX, INVALID_INDEX, NODE_DTYPE, Context, default_context  # noqa: F401
from .knn import KDTree, NeighborsTree  # noqa: F401
from .calculations import (  # noqa: F401
    SegmentsData,
    curvature,
    curvature_segments,
    curvature_ties,
    display_values_dev,
    map_to_u32_dev,
    segments_calculate,
)
from .octree import Octree, ShardedOctree  # noqa: F401
from .render import (  # noqa: F401
    Camera3D,
    PointCloud,
    PointCloudEnvironment,
    PointCloudProperty,
    eye_dome_dev,
    rotation_x,
    translation,
    view_from_transform,
)
from .laz import Chunk, CorruptFile, Header, Laz, slice_bin, slice_keys  # noqa: F401
from .export import rust_display_f32, save_points, segment_save  # noqa: F401
from .segmenting import segment_slices  # noqa: F401

__version__ = "0.1.0"
