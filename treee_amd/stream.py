"""Out-of-core tile stream (BASELINE.json configs[4]: a cloud far larger than one wants resident, streamed as
uncompressed LAS tiles): host LAS records -> HBM over PCIe on a copy stream, overlapped with the previous
tile's decode -> Morton sort -> octree/LOD [-> kNN + curvature] -> splat on the compute stream.

What the reference does with such a file: `Laz::read` hands 50 000-point chunks to worker threads
(treee/src/laz.rs:130-180) which decode them (`Chunk::read`, :193-213) and push them to the renderer as separate
point clouds (treee/src/loading.rs:91-115, program.rs:295-298).  Here a tile (millions of points) plays the role
of a chunk: every tile gets its own octree (a chunked octree: the list of tile roots is the top level) and is
drawn into one shared depth/id frame, exactly like the reference draws chunk after chunk into one framebuffer.

Two HIP streams, two staging buffers, events both ways -- no host synchronisation per tile except the one inside
the octree build (its node count sizes the node table).
"""
import ctypes as C

import numpy as np

from . import _lib
from .context import Context, F32_MAX, ptr
from .knn import KDTree
from .octree import Octree


class Tile:
    """One uncompressed-LAS tile held in (pinned) host memory: `records` is a uint8 tensor of n * stride bytes
    whose first 12 bytes per record are i32 LE x, y, z (laz.rs:202-213)."""

    def __init__(self, records, n, stride, scale, offset, center):
        self.records, self.n, self.stride = records, int(n), int(stride)
        self.scale, self.offset, self.center = (np.ascontiguousarray(a, dtype=np.float64) for a in (scale, offset, center))


class TileStream:
    def __init__(self, device, max_points, max_stride=20, width=1920, height=1080, knn_k=0, leaf_cap=256, lod_cap=256):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.compute = torch.cuda.Stream(self.device)
        self.copy = torch.cuda.Stream(self.device)
        self.ctx = Context(self.device.index, self.compute)
        self.ctx.reserve(int(max_points * 1.1) * 64 + (64 << 20))
        self.max_points, self.max_stride = int(max_points), int(max_stride)
        self.raw = [torch.empty((self.max_points * self.max_stride,), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.xyz = torch.empty((self.max_points, 3), dtype=torch.float32, device=self.device)
        self.value = torch.empty((self.max_points,), dtype=torch.int32, device=self.device)
        self.curve = torch.empty((self.max_points,), dtype=torch.float32, device=self.device) if knn_k else None
        self.packed = torch.empty((height, width), dtype=torch.int64, device=self.device)
        self.w, self.h, self.knn_k = width, height, int(knn_k)
        self.leaf_cap, self.lod_cap = leaf_cap, lod_cap
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]

    def run(self, tiles, view, proj, env, on_tile=None):
        """Streams `tiles` (an iterable of Tile).  Returns a dict with the tile roots (corner, size, node / lod
        counts per tile) and the totals; the frame is left in self.packed (depth_bits << 32 | global index)."""
        torch, lib = self.torch, _lib.lib()
        roots, total, index_base = [], 0, 0
        it = iter(tiles)
        pending = next(it, None)
        if pending is not None:
            self._upload(pending, 0, first=True)
        i = 0
        with torch.cuda.stream(self.compute):
            # n = 0, accumulate = 0: clears the packed frame (render/src/window.rs:153-159)
            _lib.check(lib.treee_splat_dev(self.ctx.handle, None, None, 0, ptr(view), ptr(proj), env.scale, env.min, env.max, self.w, self.h,
                                           ptr(self.packed), 0, 0, None, None))
        while pending is not None:
            tile, slot = pending, i & 1
            pending = next(it, None)
            if pending is not None:  # next tile's PCIe copy runs under this tile's kernels
                self._upload(pending, slot ^ 1, first=i == 0)
            with torch.cuda.stream(self.compute):
                self.compute.wait_event(self.copied[slot])
                n = tile.n
                xyz = self.xyz[:n]
                _lib.check(lib.treee_decode_las_records_dev(self.ctx.handle, ptr(self.raw[slot]), n, tile.stride, ptr(tile.scale),
                                                            ptr(tile.offset), ptr(tile.center), ptr(xyz)))
                self.consumed[slot].record(self.compute)  # the staging buffer may be overwritten from here on
                oc = Octree(xyz, leaf_cap=self.leaf_cap, lod_cap=self.lod_cap, ctx=self.ctx)
                if self.knn_k:
                    tree = KDTree.from_octree(oc, self.ctx)
                    tree.curvature_dev(self.knn_k, F32_MAX, self.ctx, out=self.curve[:n])
                    _lib.check(lib.treee_map_to_u32_dev(self.ctx.handle, ptr(self.curve), n, ptr(self.value)))
                    tree.close()
                else:
                    self.value[:n].fill_(0)
                if index_base + n >= 0xFFFFFFFF:
                    # the packed frame carries 32-bit global point ids and 0xFFFFFFFF is its "no point" sentinel
                    raise OverflowError(f"tile {i}: global point id {index_base + n - 1} does not fit the packed frame's 32-bit id "
                                        "(split the stream into runs of < 2^32 - 1 points)")
                _lib.check(lib.treee_splat_dev(self.ctx.handle, ptr(xyz), ptr(self.value), n, ptr(view), ptr(proj), env.scale, env.min, env.max,
                                               self.w, self.h, ptr(self.packed), 1, index_base, None, None))
                roots.append({"corner": oc.corner.tolist(), "size": oc.size, "points": n, "nodes": int(oc.n_nodes), "lod": int(oc.n_lod)})
                if on_tile is not None:
                    on_tile(i, tile, oc, xyz)
                oc.close()
            total += n
            index_base += n
            i += 1
        self.compute.synchronize()
        return {"tiles": roots, "points": total}

    def _upload(self, tile, slot, first):
        torch = self.torch
        nbytes = tile.n * tile.stride
        assert tile.n <= self.max_points and tile.stride <= self.max_stride
        with torch.cuda.stream(self.copy):
            if not first:
                self.copy.wait_event(self.consumed[slot])
            self.raw[slot][:nbytes].copy_(tile.records[:nbytes], non_blocking=True)
            self.copied[slot].record(self.copy)

    def close(self):
        self.ctx.close()


def las_tile_from_cloud(xyz, scale=0.001, stride=20, pin=True):
    """Quantise an internal-frame cloud (device tensor, x/z horizontal, y up) to LAS format-0-like records and
    bring them to (pinned) host memory: LAS (x, y, z) = (x, -z, y) (inverse of laz.rs:210), centre 0."""
    import torch

    n = xyz.shape[0]
    q = torch.round(xyz.double() / scale).to(torch.int32)
    rec = torch.zeros((n, stride // 4), dtype=torch.int32, device=xyz.device)
    rec[:, 0] = q[:, 0]
    rec[:, 1] = -q[:, 2]
    rec[:, 2] = q[:, 1]
    host = torch.empty((n * stride,), dtype=torch.uint8, pin_memory=pin)
    host.copy_(rec.view(torch.uint8).reshape(-1))
    return Tile(host, n, stride, (scale,) * 3, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
