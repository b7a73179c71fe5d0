"""Out-of-core tile stream (BASELINE.json configs[4]: a cloud far larger than one wants resident, streamed as
uncompressed LAS tiles, or as LAZ tiles decompressed on host threads one tile ahead): host LAS records -> HBM over PCIe on a copy stream, overlapped with the previous
tile's decode -> Morton sort -> octree/LOD [-> kNN + curvature] -> splat on the compute stream.

What the reference does with such a file: `Laz::read` hands 50 000-point chunks to worker threads
(treee/src/laz.rs:130-180) which decode them (`Chunk::read`, :193-213) and push them to the renderer as separate
point clouds (treee/src/loading.rs:91-115, program.rs:295-298).  Here a tile (millions of points) plays the role
of a chunk: every tile gets its own octree (a chunked octree: the list of tile roots is the top level) and is
drawn into one shared depth/id frame, exactly like the reference draws chunk after chunk into one framebuffer.

Two HIP streams, two staging buffers, events both ways -- no host synchronisation per tile except the one inside
the octree build (its node count sizes the node table).

Two modes:
* `TileStream` -- PER TILE, NO GHOST POINTS: one octree per tile (each with its own cube), kNN inside a tile only, so a
  point near a tile border gets its neighbours from one side.  That is what the reference does with its chunks (every chunk
  is its own point cloud), and the cheapest way to draw a frame of a cloud that does not fit; it is NOT the octree or the
  neighbourhoods of the whole cloud.
* `WholeCloudStream` -- ONE CLOUD: the tiles are decoded as they arrive into one resident xyz buffer (1.005 G points are
  12 GB of a 288 GB MI355X; "out of core" is about the host side and the PCIe link, which stay streamed and overlapped),
  then ONE Morton sort / octree / LOD with one cube and global numbering, the kNN structure from that sort, curvature of
  every point with its true neighbours across tile borders, one splat.  Same result, bit for bit, as the single call on the
  concatenated cloud (tests/test_gpu_parity.py::test_whole_cloud_stream_is_the_single_call).
"""
import ctypes as C

import numpy as np

from . import _lib
from .context import Context, F32_MAX, ptr
from .knn import KDTree  # noqa: F401  (the handle type Octree.with_knn returns)
from .octree import Octree


class Tile:
    """One uncompressed-LAS tile held in (pinned) host memory: `records` is a uint8 tensor of n * stride bytes
    whose first 12 bytes per record are i32 LE x, y, z (laz.rs:202-213)."""

    def __init__(self, records, n, stride, scale, offset, center):
        self.records, self.n, self.stride = records, int(n), int(stride)
        self.scale, self.offset, self.center = (np.ascontiguousarray(a, dtype=np.float64) for a in (scale, offset, center))


class LazTile:
    """One LAZ-compressed tile: `image` is the file (bytes).  The stream decompresses it on host threads into a pinned
    staging buffer (treee_laz_decompress_chunks: what the reference's rayon tasks do, laz.rs:134-160) while the GPU works on
    the tile before it."""

    def __init__(self, image, center=None):
        from .laz import Laz

        self.src = Laz(image, center)
        self.n = int(sum(c for _, c in self.src.chunks))
        self.stride = int(self.src.point_length)
        self.scale, self.offset, self.center = (np.ascontiguousarray(a, dtype=np.float64) for a in (self.src.scale, self.src.offset, self.src.center))
        self.records = None  # set by the stream: the pinned buffer the records were decompressed into

    def decompress_into(self, out, threads=0):
        """out: uint8 tensor (pinned host memory) of at least n * stride bytes"""
        src, v = self.src, self.src.vlr
        items = np.asarray(v["items"], dtype=np.uint16).reshape(-1, 3)
        types, sizes, versions = (np.ascontiguousarray(items[:, i]) for i in range(3))
        starts = np.asarray([s for s, _ in src.chunks], dtype=np.uint64)
        counts = np.asarray([c for _, c in src.chunks], dtype=np.uint64)
        nbytes = np.asarray(src._chunk_bytes, dtype=np.uint64)
        buf = np.frombuffer(src.data, np.uint8)
        if out.numel() * out.element_size() < self.n * self.stride:
            raise ValueError(f"decompress_into: the buffer holds {out.numel() * out.element_size()} bytes, the tile needs {self.n * self.stride}")
        # dst_len = the buffer's real capacity: the library refuses records that would not fit (no Python assert to strip)
        _lib.check(_lib.lib().treee_laz_decompress_chunks(ptr(buf), len(buf), ptr(types), ptr(sizes), ptr(versions), len(items), ptr(starts), ptr(nbytes),
                                                          ptr(counts), len(src.chunks), threads, ptr(out), out.numel() * out.element_size()))
        self.records = out


class _Staging:
    """what both modes share: the copy stream, two device staging buffers for raw records, events both ways, and the
    LAZ side (two pinned buffers, one worker decompressing a tile ahead)"""

    def _init_staging(self, device, max_points, max_stride):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.compute = torch.cuda.Stream(self.device)
        self.copy = torch.cuda.Stream(self.device)
        self.max_points, self.max_stride = int(max_points), int(max_stride)
        self.raw = [torch.empty((self.max_points * self.max_stride,), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self.pinned = None  # two pinned staging buffers for LAZ tiles, allocated when the first one arrives
        self.pool = None  # one worker: the next LAZ tile is decompressed (on the library's host threads) while this one runs
        self.laz_threads = 0

    def _decoded(self, tiles):
        """passes LAS tiles through; a LAZ tile is handed on once its records are in pinned memory, and the decompression
        of the LAZ tile after it has been started by then (one tile ahead, alternating staging buffers)"""
        torch = self.torch
        ahead, k = None, 0  # (tile, future) being decompressed

        def start(tile):
            nonlocal k
            from concurrent.futures import ThreadPoolExecutor

            if self.pinned is None:
                self.pinned = [torch.empty((self.max_points * self.max_stride,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
                self.pinned_read = [torch.cuda.Event() for _ in range(2)]  # the H2D copy that last read each staging buffer
                self.pool = ThreadPoolExecutor(max_workers=1)
            if tile.n > self.max_points or tile.stride > self.max_stride:
                raise ValueError(f"tile of {tile.n} x {tile.stride} B exceeds the stream's {self.max_points} x {self.max_stride} B")
            slot = k & 1
            k += 1
            self.pinned_read[slot].synchronize()  # (two LAZ tiles ago)
            return tile, self.pool.submit(tile.decompress_into, self.pinned[slot], self.laz_threads)

        for tile in tiles:
            if isinstance(tile, LazTile):
                nxt = start(tile)
                if ahead is not None:
                    ahead[1].result()
                    yield ahead[0]
                ahead = nxt
            else:
                if ahead is not None:
                    ahead[1].result()
                    yield ahead[0]
                    ahead = None
                yield tile
        if ahead is not None:
            ahead[1].result()
            yield ahead[0]

    def _upload(self, tile, slot, first):
        torch = self.torch
        nbytes = tile.n * tile.stride
        if tile.n > self.max_points or tile.stride > self.max_stride:
            raise ValueError(f"tile of {tile.n} x {tile.stride} B exceeds the stream's {self.max_points} x {self.max_stride} B")
        with torch.cuda.stream(self.copy):
            if not first:
                self.copy.wait_event(self.consumed[slot])
            self.raw[slot][:nbytes].copy_(tile.records[:nbytes], non_blocking=True)
            self.copied[slot].record(self.copy)
            if self.pinned is not None:
                for k in range(2):
                    if tile.records is self.pinned[k]:
                        self.pinned_read[k].record(self.copy)


    def _ingest_into(self, xyz, tiles, ctx):
        """decodes `tiles` one after the other into the resident buffer `xyz` ([capacity, 3] f32 on the device) on the compute
        stream, the PCIe copy of tile i + 1 under the decode of tile i.  Returns (points, [points per tile]); nothing is
        synchronised -- the caller orders its consumers behind self.compute."""
        torch, lib = self.torch, _lib.lib()
        capacity = xyz.shape[0]
        it = self._decoded(iter(tiles))
        pending = next(it, None)
        if pending is not None:
            self._upload(pending, 0, first=True)
        base, i, sizes = 0, 0, []
        while pending is not None:
            tile, slot = pending, i & 1
            pending = next(it, None)
            if pending is not None:  # next tile's PCIe copy runs under this tile's decode
                self._upload(pending, slot ^ 1, first=i == 0)
            n = tile.n
            if base + n > capacity:
                raise ValueError(f"tile {i}: {base + n} points exceed the resident buffer's {capacity}")
            with torch.cuda.stream(self.compute):
                self.compute.wait_event(self.copied[slot])
                _lib.check(lib.treee_decode_las_records_dev(ctx.handle, ptr(self.raw[slot]), n, tile.stride, ptr(tile.scale),
                                                            ptr(tile.offset), ptr(tile.center), ptr(xyz[base:base + n])))
                self.consumed[slot].record(self.compute)  # the staging buffer may be overwritten from here on
            sizes.append(n)
            base += n
            i += 1
        return base, sizes


class TileIngest(_Staging):
    """The ingest half alone: LAS / LAZ tiles from host memory -> one resident xyz buffer on the device (what a rank of the
    sharded path does with ITS tiles before the sharded step: bench.py --stream-tiles).  `ingest` returns the number of
    points; the caller's stream must wait for `self.compute` (ingest does that for torch's current stream)."""

    def __init__(self, device, capacity, max_tile_points, max_stride=20):
        self._init_staging(device, max_tile_points, max_stride)
        self.ctx = Context(self.device.index, self.compute)
        self.xyz = self.torch.empty((int(capacity), 3), dtype=self.torch.float32, device=self.device)

    def ingest(self, tiles):
        n, self.sizes = self._ingest_into(self.xyz, tiles, self.ctx)
        self.torch.cuda.current_stream(self.device).wait_stream(self.compute)
        return n

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
        self.ctx.close()


def split_into_las_tiles(xyz, max_tile_points, scale=0.001, stride=20):
    """a device cloud cut into runs of at most max_tile_points points, each handed over as a pinned LAS tile (las_tile_from_cloud)"""
    return [las_tile_from_cloud(xyz[a:a + max_tile_points], scale, stride) for a in range(0, xyz.shape[0], max_tile_points)]


class TileStream(_Staging):
    """PER TILE, NO GHOST POINTS (module docstring): every tile its own octree and its own neighbourhoods."""

    def __init__(self, device, max_points, max_stride=20, width=1920, height=1080, knn_k=0, leaf_cap=256, lod_cap=256):
        self._init_staging(device, max_points, max_stride)
        torch = self.torch
        self.ctx = Context(self.device.index, self.compute)
        # (with kNN on, a tile's octree node table is completed beside the kNN structure's own build by the ONE call
        # treee_sort_build_knn_dev -- the library's auxiliary stream; rounds 3-5a: a second context on a torch side stream)
        self.ctx.reserve(int(max_points * 1.1) * (88 if knn_k else 64) + (96 << 20))
        self.xyz = torch.empty((self.max_points, 3), dtype=torch.float32, device=self.device)
        self.value = torch.empty((self.max_points,), dtype=torch.int32, device=self.device)
        self.curve = torch.empty((self.max_points,), dtype=torch.float32, device=self.device) if knn_k else None
        self.packed = torch.empty((height, width), dtype=torch.int64, device=self.device)
        self.w, self.h, self.knn_k = width, height, int(knn_k)
        self.leaf_cap, self.lod_cap = leaf_cap, lod_cap

    def run(self, tiles, view, proj, env, on_tile=None):
        """Streams `tiles` (an iterable of Tile).  Returns a dict with the tile roots (corner, size, node / lod
        counts per tile) and the totals; the frame is left in self.packed (depth_bits << 32 | global index)."""
        torch, lib = self.torch, _lib.lib()
        roots, total, index_base = [], 0, 0
        it = self._decoded(iter(tiles))
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
                if self.knn_k:
                    oc, tree = Octree.with_knn(xyz, leaf_cap=self.leaf_cap, lod_cap=self.lod_cap, ctx=self.ctx)
                    tree.curvature_dev(self.knn_k, F32_MAX, self.ctx, out=self.curve[:n])
                    _lib.check(lib.treee_map_to_u32_dev(self.ctx.handle, ptr(self.curve), n, ptr(self.value)))
                    tree.close()
                else:
                    oc = Octree(xyz, leaf_cap=self.leaf_cap, lod_cap=self.lod_cap, ctx=self.ctx)
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

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
        self.ctx.close()


class WholeCloudStream(_Staging):
    """ONE CLOUD (module docstring): the tiles are decoded into one resident xyz buffer as they arrive -- the PCIe copy of
    tile i + 1 runs under the decode of tile i --, then the whole path runs ONCE over the concatenation: Morton sort +
    octree / LOD with one cube (treee_sort_dev / treee_octree_complete_dev on a side stream) [-> kNN structure from that
    sort -> curvature -> map_to_u32] -> splat of the sorted records -> frame.  Global point index = position in the
    concatenation (points of earlier tiles first), as in the per-tile mode's frame.

    After `run` the handles stay open for the caller (`self.octree`, `self.tree`, `self.curve`, `self.value`,
    `self.packed`, `self.xyz[:n]`) until the next `run` or `close`."""

    def __init__(self, device, total_points, max_tile_points, max_stride=20, width=1920, height=1080, knn_k=31, leaf_cap=256, lod_cap=256):
        self._init_staging(device, max_tile_points, max_stride)
        torch = self.torch
        self.capacity = int(total_points)
        if self.capacity >= (1 << 31):
            raise OverflowError(f"{self.capacity} points: a cloud holds at most 2^31 - 1 points (include/treee_hip.h)")
        self.ctx = Context(self.device.index, self.compute)
        self.xyz = torch.empty((self.capacity, 3), dtype=torch.float32, device=self.device)
        self.value = torch.empty((self.capacity,), dtype=torch.int32, device=self.device)
        self.curve = torch.empty((self.capacity,), dtype=torch.float32, device=self.device) if knn_k else None
        self.packed = torch.empty((height, width), dtype=torch.int64, device=self.device)
        self.w, self.h, self.knn_k = width, height, int(knn_k)
        self.leaf_cap, self.lod_cap = leaf_cap, lod_cap
        self.octree = self.tree = None

    def _drop(self):
        for name in ("tree", "octree"):
            h = getattr(self, name, None)
            if h is not None:
                h.close()
            setattr(self, name, None)

    def run(self, tiles, view, proj, env):
        """Streams `tiles` into the resident cloud and builds once.  Returns {"points", "tiles", "nodes", "lod", "corner",
        "size", "stages_ms": {...}} (HIP events on the compute stream; "ingest" = first copy to last decode, i.e. the
        PCIe-bound part)."""
        torch, lib = self.torch, _lib.lib()
        self._drop()
        ev = {k: torch.cuda.Event(enable_timing=True) for k in ("t0", "ingested", "structure", "knn", "splat")}
        ev["t0"].record(self.compute)
        total, sizes = self._ingest_into(self.xyz, tiles, self.ctx)
        res = {"points": total, "tiles": sizes, "mode": "one cloud: one cube, one octree, neighbours across tile borders"}
        with torch.cuda.stream(self.compute):
            ev["ingested"].record(self.compute)
            # n = 0, accumulate = 0: clears the packed frame (render/src/window.rs:153-159)
            _lib.check(lib.treee_splat_dev(self.ctx.handle, None, None, 0, ptr(view), ptr(proj), env.scale, env.min, env.max, self.w, self.h,
                                           ptr(self.packed), 0, 0, None, None))
            if total == 0:
                self.compute.synchronize()
                return res
            cloud = self.xyz[:total]
            if self.knn_k:
                oc, tree = Octree.with_knn(cloud, leaf_cap=self.leaf_cap, lod_cap=self.lod_cap, ctx=self.ctx)
                ev["structure"].record(self.compute)
                tree.curvature_dev(self.knn_k, F32_MAX, self.ctx, out=self.curve[:total])
                _lib.check(lib.treee_map_to_u32_dev(self.ctx.handle, ptr(self.curve), total, ptr(self.value)))
                ev["knn"].record(self.compute)
                _lib.check(lib.treee_splat_knn_dev(self.ctx.handle, tree._h, ptr(self.value), ptr(view), ptr(proj), env.scale, env.min, env.max,
                                                   self.w, self.h, ptr(self.packed), 1, 0, None, None))
                self.tree = tree
            else:
                oc = Octree(cloud, leaf_cap=self.leaf_cap, lod_cap=self.lod_cap, ctx=self.ctx)
                ev["structure"].record(self.compute)
                self.value[:total].fill_(0)
                ev["knn"].record(self.compute)
                _lib.check(lib.treee_splat_octree_dev(self.ctx.handle, oc._h, ptr(self.value), ptr(view), ptr(proj), env.scale, env.min, env.max,
                                                      self.w, self.h, ptr(self.packed), 1, 0, None, None))
            ev["splat"].record(self.compute)
            self.octree = oc
        self.compute.synchronize()
        res.update(nodes=int(oc.n_nodes), lod=int(oc.n_lod), corner=oc.corner.tolist(), size=oc.size)
        res["stages_ms"] = {"ingest_decode": ev["t0"].elapsed_time(ev["ingested"]), "sort_octree_knn_structure": ev["ingested"].elapsed_time(ev["structure"]),
                            "knn_curve": ev["structure"].elapsed_time(ev["knn"]), "splat": ev["knn"].elapsed_time(ev["splat"])}
        return res

    def close(self):
        self._drop()
        if self.pool is not None:
            self.pool.shutdown()
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
