"""Spatial sharding of one cloud over the GPUs of a node.

One process per GPU.  The cloud is cut into x/z tiles (y = height is never cut, like the
forest plots of section 8 D); every rank runs the whole pipeline on its own tile.  The only
data-path exchange is the ghost-point all-gather: points within `halo` metres of a tile face
are gathered (padded to the largest contribution -- RCCL all-gather needs equal counts) and
every rank keeps the ghosts that fall inside its own dilated tile.  xGMI is point-to-point and
the halo volume is a few MB per rank (0.3 M points at 150 M / 8 GPUs), so one padded
all-gather is far below link bandwidth; no other collective touches point data.

Exactness with the reference's unbounded radius (max_distance = f32::MAX,
treee/src/calculations.rs:279): a query's result is exact iff its k-th neighbour distance is
<= its distance to the outer rim of the halo; `halo_is_sufficient` checks exactly that and the
caller widens the halo and repeats otherwise (1 round on forest data).

The frame buffer is combined with an element-wise MIN all-reduce on the packed
(depth_bits << 32 | index) words, which is what a single-GPU draw of the whole cloud would
have produced (indices are made globally unique with index_base).
"""
import math

import torch
import torch.distributed as dist


class Comm:
    """treee_comm of the C ABI (include/treee_hip.h, "many GPUs -- communicator"): what the library's sharded entry points
    talk to the other ranks through.

    * `Comm.rccl(group)`  -- the library's own RCCL communicator (ncclCommInitRank inside libtreee_hip.so; payloads stay in
      HBM, collectives run on the context's stream).  torch.distributed only carries the 128-byte unique id to the ranks.
    * `Comm.host(group)`  -- host callbacks over any torch.distributed group (gloo in the tests and rehearsals): the
      library stages device data through host memory and calls back here for the transport.
    """

    def __init__(self, handle, rank, world, keep=None):
        self._h, self.rank, self.world, self._keep = handle, rank, world, keep

    @property
    def handle(self):
        return self._h

    def info(self):
        """(rank, world, is_rccl) as the communicator itself reports them (treee_comm_info)"""
        import ctypes as C

        from . import _lib

        r, w, flag = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().treee_comm_info(self._h, C.byref(r), C.byref(w), C.byref(flag)))
        return int(r.value), int(w.value), bool(flag.value)

    @property
    def is_rccl(self):
        return self.info()[2]

    @classmethod
    def rccl(cls, group=None, device=None):
        """The library's RCCL communicator over `group`'s ranks.  Every step that can fail on ONE rank is followed by an
        agreement over the torch group before the next collective step, so that all ranks raise together (and a caller can
        fall back to `Comm.host` on all of them) instead of some waiting in a broadcast or inside ncclCommInitRank for a
        rank that has already given up: (1) every rank probes librccl by itself (treee_rccl_available: no collective),
        (2) rank 0 creates the unique id and broadcasts it with a status byte, (3) ncclCommInitRank, (4) the outcome."""
        import ctypes as C

        import numpy as np

        from . import _lib

        live = dist.is_initialized()
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if live else (0, 1)
        on_gpu = live and dist.get_backend(group) != "gloo"

        def agree(ok, what):
            """MIN over the ranks of a local success flag; raises the same error everywhere when any rank failed"""
            msg = "" if ok else _lib.lib().treee_last_error().decode("utf-8", "replace")
            if world > 1:
                t = torch.tensor([1 if ok else 0], dtype=torch.int32)
                if on_gpu:
                    t = t.cuda()
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                all_ok = int(t.item()) == 1
            else:
                all_ok = bool(ok)
            if not all_ok:
                raise _lib.TreeeError(_lib.TREEE_ERR_UNSUPPORTED, f"RCCL transport: {what} failed on " + (f"this rank ({msg})" if not ok else "another rank"))

        agree(_lib.lib().treee_rccl_available() == _lib.TREEE_OK, "opening librccl")
        uid = np.zeros(129, np.uint8)  # [128] = rank 0's status
        if rank == 0:
            uid[128] = 1 if _lib.lib().treee_rccl_unique_id(uid.ctypes.data_as(C.c_void_p)) == _lib.TREEE_OK else 0
        if world > 1:
            t = torch.from_numpy(uid)
            if on_gpu:
                t = t.cuda()
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = t.cpu().numpy()
        if uid[128] != 1:  # (the same byte on every rank)
            raise _lib.TreeeError(_lib.TREEE_ERR_UNSUPPORTED, "RCCL transport: ncclGetUniqueId failed on rank 0")
        dev = torch.cuda.current_device() if device is None else torch.device(device).index
        h = C.c_void_p()
        rc = _lib.lib().treee_comm_create_rccl(np.ascontiguousarray(uid).ctypes.data_as(C.c_void_p), rank, world, dev, C.byref(h))
        try:
            agree(rc == _lib.TREEE_OK, "ncclCommInitRank")
        except _lib.TreeeError:
            if rc == _lib.TREEE_OK:
                _lib.lib().treee_comm_destroy(h)
            raise
        return cls(h, rank, world)

    @classmethod
    def auto(cls, group=None, device=None):
        """RCCL when every rank of the group can open it, host callbacks over the group otherwise -- the same choice on
        every rank (Comm.rccl agrees before it initialises anything)."""
        from . import _lib

        try:
            return cls.rccl(group, device)
        except _lib.TreeeError as e:
            import sys

            print(f"# treee: {e}; host-staged transport instead", file=sys.stderr)
            return cls.host(group)

    @classmethod
    def host(cls, group=None):
        """host-staged transport over `group` (any backend that moves CPU tensors; a world of one needs no group)"""
        import ctypes as C

        import numpy as np

        from . import _lib

        live = dist.is_initialized()
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if live else (0, 1)
        if live and world > 1 and _cannot_carry_host_tensors(group):
            # an nccl(= RCCL)-only group moves device tensors only, and these callbacks hand it host buffers (the
            # fallback of Comm.auto under `init_process_group("nccl")` built a communicator whose every collective failed
            # inside the callback).  The callbacks get a gloo group over the same ranks.  new_group is collective over the
            # DEFAULT group, so this is only done for it; a caller with an nccl sub-group passes a gloo group of its own.
            if group is not None and group is not dist.group.WORLD:
                raise _lib.TreeeError(_lib.TREEE_ERR_UNSUPPORTED, "Comm.host: the group's backend cannot move host tensors; pass a gloo group over the same ranks")
            group = dist.new_group(backend="gloo")

        def view(p, nbytes):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[: int(nbytes)]

        def all_gather(_user, send, recv, nbytes):
            try:
                src, dst = view(send, nbytes), view(recv, nbytes * world)
                if world == 1:
                    dst[:] = src
                    return 0
                parts = [torch.empty(int(nbytes), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(src.copy()), group=group)
                for r, part in enumerate(parts):
                    dst[r * nbytes : (r + 1) * nbytes] = part.numpy()
                return 0
            except Exception:  # never unwind into C
                import traceback

                traceback.print_exc()
                return 1

        def all_to_all_v(_user, send, s_off, s_len, recv, r_off, r_len):
            try:
                so, sl, ro, rl = ([int(a[r]) for r in range(world)] for a in (s_off, s_len, r_off, r_len))
                src = view(send, max(o + n for o, n in zip(so, sl)))
                dst = view(recv, max(o + n for o, n in zip(ro, rl)))
                if world == 1:
                    dst[ro[0] : ro[0] + rl[0]] = src[so[0] : so[0] + sl[0]]
                    return 0
                outs = [torch.from_numpy(src[so[r] : so[r] + sl[r]].copy()) for r in range(world)]
                ins = [torch.empty(rl[r], dtype=torch.uint8) for r in range(world)]
                # gloo has no all_to_all: pairwise exchange, lower rank sends first (every pair is ordered the same way)
                for peer in range(world):
                    if peer == rank:
                        ins[peer].copy_(outs[peer])
                    elif rank < peer:
                        dist.send(outs[peer], dst=_global(group, peer), group=group)
                        dist.recv(ins[peer], src=_global(group, peer), group=group)
                    else:
                        dist.recv(ins[peer], src=_global(group, peer), group=group)
                        dist.send(outs[peer], dst=_global(group, peer), group=group)
                for r in range(world):
                    dst[ro[r] : ro[r] + rl[r]] = ins[r].numpy()
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        def all_reduce_min_u64(_user, buf, count):
            try:
                if world == 1:
                    return 0
                a = np.ctypeslib.as_array(buf, shape=(int(count),))
                t = torch.from_numpy(a.view(np.int64))  # values < 2^63 (depth < 1.0): the signed order is the unsigned one
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        ops = _lib.CommHostOps(None, _lib._CB_GATHER(all_gather), _lib._CB_ALLTOALL(all_to_all_v), _lib._CB_REDUCE(all_reduce_min_u64))
        h = C.c_void_p()
        _lib.check(_lib.lib().treee_comm_create_host(C.byref(ops), rank, world, C.byref(h)))
        return cls(h, rank, world, keep=ops)  # (the callbacks must outlive the handle)

    @classmethod
    def for_group(cls, group=None):
        """RCCL when the group runs on it, host callbacks otherwise (gloo)."""
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            return cls.auto(group)
        return cls.host(group)

    def close(self):
        if getattr(self, "_h", None):
            from . import _lib

            _lib.lib().treee_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardPlan:
    """treee_shard_plan of the C ABI: the host arithmetic of the sharded octree (owners of the Morton cells, what travels
    where, the replicated top, the global numbering).  Pure host code -- usable without a GPU."""

    def __init__(self, cell_counts, top_level, leaf_cap=256, lod_cap=256):
        import ctypes as C

        import numpy as np

        from . import _lib

        cc = np.ascontiguousarray(cell_counts, dtype=np.uint32)
        assert cc.ndim == 2 and cc.shape[1] == 8 ** top_level
        self.world = cc.shape[0]
        h = C.c_void_p()
        _lib.check(_lib.lib().treee_shard_plan_create(cc.ctypes.data_as(C.c_void_p), self.world, top_level, leaf_cap, lod_cap, C.byref(h)))
        self._h = h
        self.info = _lib.ShardPlanInfo()
        _lib.check(_lib.lib().treee_shard_plan_info_get(h, C.byref(self.info)))

    def rank(self, r):
        import ctypes as C

        from . import _lib

        out = _lib.ShardPlanRank()
        _lib.check(_lib.lib().treee_shard_plan_rank_get(self._h, r, C.byref(out)))
        return out

    def top_nodes(self):
        import ctypes as C

        import numpy as np

        from . import _lib
        from .context import NODE_DTYPE

        out = np.zeros(self.info.n_top_nodes, NODE_DTYPE)
        _lib.check(_lib.lib().treee_shard_plan_top_nodes(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def level_nodes(self, r):
        import ctypes as C

        import numpy as np

        from . import _lib
        from .context import NODE_DTYPE

        out = np.zeros(self.rank(r).level_node_count, NODE_DTYPE)
        _lib.check(_lib.lib().treee_shard_plan_level_nodes(self._h, r, out.ctypes.data_as(C.c_void_p)))
        return out

    def exchange(self, r):
        """(send_first, send_count, recv_count), each [world]"""
        import ctypes as C

        import numpy as np

        from . import _lib

        a, b, c = (np.zeros(self.world, np.uint64) for _ in range(3))
        _lib.check(_lib.lib().treee_shard_plan_exchange(self._h, r, *(x.ctypes.data_as(C.c_void_p) for x in (a, b, c))))
        return a, b, c

    def levels(self, level_counts, lod_counts, r):
        import ctypes as C

        import numpy as np

        from . import _lib

        lc = np.ascontiguousarray(level_counts, dtype=np.uint32)
        dc = np.ascontiguousarray(lod_counts, dtype=np.uint32)
        assert lc.shape == dc.shape == (self.world, _lib.SHARD_LEVELS)
        out = _lib.ShardLevels()
        _lib.check(_lib.lib().treee_shard_plan_levels(self._h, lc.ctypes.data_as(C.c_void_p), dc.ctypes.data_as(C.c_void_p), r, C.byref(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            from . import _lib

            _lib.lib().treee_shard_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_reduce_min(packed, comm, ctx=None):
    """treee_frame_reduce_min_dev: element-wise MIN of the packed frames of all ranks, in place"""
    from . import _lib
    from .context import default_context, ptr

    c = ctx or default_context(packed.device)
    _lib.check(_lib.lib().treee_frame_reduce_min_dev(c.handle, comm.handle, ptr(packed), packed.numel()))
    return packed


def _cannot_carry_host_tensors(group=None):
    """True for a group whose backend only moves device tensors (nccl = RCCL)"""
    return dist.get_backend(group) == "nccl"


def _global(group, r):
    return dist.get_global_rank(group, r) if group is not None else r


def _host_staged(group=None):
    """gloo cannot move CUDA tensors: rehearsal runs (several ranks sharing one GPU) stage the
    collectives through host memory.  With nccl (= RCCL over xGMI) tensors stay in HBM."""
    return dist.get_backend(group) == "gloo"


def _all_reduce(t, op, group=None):
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def _all_gather_equal(send, world_size, group=None):
    """all-gather of equally shaped tensors -> [world, ...] on send's device."""
    if send.is_cuda and not _host_staged(group):
        recv = torch.empty((world_size,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv.view(-1, *send.shape[1:]), send, group=group)
        return recv
    h = send.cpu()
    parts = [torch.empty_like(h) for _ in range(world_size)]
    dist.all_gather(parts, h, group=group)
    return torch.stack(parts).to(send.device)


def tile_grid(world_size):
    """tiles along x and z: 1->1x1, 2->2x1, 4->2x2, 8->4x2 (SURVEY.md 8E example)."""
    gx = 1
    while gx * gx < world_size:
        gx *= 2
    gz = max(1, world_size // gx)
    if gx * gz != world_size:  # non power-of-two world: a single row
        gx, gz = world_size, 1
    return gx, gz


def tile_origin(rank, world_size, plot):
    """centre (x, z) of rank's tile when every tile is a plot x plot square."""
    gx, gz = tile_grid(world_size)
    ix, iz = rank % gx, rank // gx
    return ((ix - (gx - 1) / 2.0) * plot, (iz - (gz - 1) / 2.0) * plot)


def tile_bounds(rank, world_size, plot):
    ox, oz = tile_origin(rank, world_size, plot)
    return (ox - plot / 2.0, ox + plot / 2.0, oz - plot / 2.0, oz + plot / 2.0)


def _interior_faces(rank, world_size):
    """which of the 4 faces (x-, x+, z-, z+) touch another tile."""
    gx, gz = tile_grid(world_size)
    ix, iz = rank % gx, rank // gx
    return (ix > 0, ix < gx - 1, iz > 0, iz < gz - 1)


def select_boundary(points, bounds, faces, halo):
    """points of this tile within `halo` of a face shared with a neighbour tile."""
    x0, x1, z0, z1 = bounds
    x, z = points[:, 0], points[:, 2]
    m = torch.zeros(points.shape[0], dtype=torch.bool, device=points.device)
    if faces[0]:
        m |= x < x0 + halo
    if faces[1]:
        m |= x >= x1 - halo
    if faces[2]:
        m |= z < z0 + halo
    if faces[3]:
        m |= z >= z1 - halo
    return m


_INF = float("inf")


def select_xz(points, keep_box, drop_box, ctx=None):
    """points (float32 [n,3], CUDA) whose (x, z) is inside keep_box and not inside drop_box, in input
    order: the library's fused flag / scan / compact (treee_select_xz_dev).  Synchronises to learn the count."""
    import numpy as np

    from . import _lib
    from .context import default_context, ptr

    pts = points.contiguous()
    n = pts.shape[0]
    c = ctx or default_context(pts.device)
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=pts.device)
    count = torch.zeros((1,), dtype=torch.int32, device=pts.device)
    keep = np.asarray(keep_box, dtype=np.float32)
    drop = np.asarray(drop_box, dtype=np.float32)
    _lib.check(_lib.lib().treee_select_xz_dev(c.handle, ptr(pts), n, ptr(keep), ptr(drop), ptr(out), n, ptr(count)))
    return out[: int(count.item())]


def exchange_ghosts(points, bounds, halo, rank, world_size, group=None, ctx=None):
    """the ghost-point all-gather.  Returns the ghosts (float32 [g,3]) that lie inside this
    rank's tile dilated by `halo` (own points excluded).  CUDA tensors go through the library's fused
    selection kernels; CPU tensors (gloo tests) through the equivalent torch expressions."""
    if world_size == 1:
        return points[:0]
    faces = _interior_faces(rank, world_size)
    x0, x1, z0, z1 = bounds
    if points.is_cuda:
        interior = (x0 + halo if faces[0] else -_INF, x1 - halo if faces[1] else _INF, z0 + halo if faces[2] else -_INF, z1 - halo if faces[3] else _INF)
        mine = select_xz(points, (-_INF, _INF, -_INF, _INF), interior, ctx)
    else:
        mine = points[select_boundary(points, bounds, faces, halo)].contiguous()
    count = torch.tensor([mine.shape[0]], dtype=torch.int64, device=points.device)
    counts = [int(c) for c in _all_gather_equal(count, world_size, group).flatten().tolist()]
    cap = max(max(counts), 1)
    send = torch.full((cap, 3), float("nan"), dtype=torch.float32, device=points.device)  # NaN padding is never selected
    send[: mine.shape[0]] = mine
    recv = _all_gather_equal(send, world_size, group)  # padded: RCCL all-gather needs equal counts
    dilated = (x0 - halo, x1 + halo, z0 - halo, z1 + halo)
    keep = []
    for part in (recv[:rank], recv[rank + 1 :]):  # every peer's block, own block skipped
        flat = part.reshape(-1, 3)
        if flat.shape[0] == 0:
            continue
        if points.is_cuda:
            keep.append(select_xz(flat, dilated, (0.0, 0.0, 0.0, 0.0), ctx))
        else:
            inside = (flat[:, 0] >= dilated[0]) & (flat[:, 0] < dilated[1]) & (flat[:, 2] >= dilated[2]) & (flat[:, 2] < dilated[3])
            keep.append(flat[inside])
    return torch.cat(keep) if keep else points[:0]


class GhostExchanger:
    """The per-step ghost exchange for clouds resident in HBM: `treee_shard_exchange_ghosts_dev` behind a buffer policy.

    * the cloud lives in a buffer with room for the ghosts behind the own points (`cloud[:n]` = own points,
      `cloud[n:n+g]` = ghosts), so nothing is concatenated;
    * ONE collective per step (an all-gather of fixed-capacity blocks whose header row carries the sender's count) and
      ONE host round trip (the ghost count the sort needs and every sender's count);
    * a sender that outgrew the capacity is seen by all ranks in the same data, so all of them grow the buffers together
      and repeat.
    """

    def __init__(self, points, bounds, halo, rank, world_size, ctx=None, group=None, margin=1.5, comm=None):
        from .context import default_context

        assert points.is_cuda and world_size > 1
        self.rank, self.world, self.group, self.bounds = rank, world_size, group, bounds
        self.ctx = ctx or default_context(points.device)
        self.comm = comm or Comm.for_group(group)
        self.n = points.shape[0]
        self.dev = points.device
        self.margin = margin
        f = _interior_faces(rank, world_size)
        self.faces = (1 if f[0] else 0) | (2 if f[1] else 0) | (4 if f[2] else 0) | (8 if f[3] else 0)
        self.cap = 0
        self.cloud = points.contiguous()
        ghosts, most = self._call(halo, 0, 0)  # counted once: nothing is written with zero capacities
        self._allocate(points, int(most * margin) + 64)

    def _allocate(self, points, cap):
        self.cap = cap
        self.ghost_cap = (self.world - 1) * cap
        self.cloud = torch.empty((self.n + self.ghost_cap, 3), dtype=torch.float32, device=self.dev)
        self.cloud[: self.n] = points

    def _call(self, halo, send_cap, ghost_cap):
        import ctypes as C

        import numpy as np

        from . import _lib
        from .context import ptr

        box = np.asarray(self.bounds, dtype=np.float32)
        g, most = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().treee_shard_exchange_ghosts_dev(self.ctx.handle, self.comm.handle, ptr(self.cloud), self.n, ptr(box), self.faces, float(halo),
                                                              send_cap, ghost_cap, C.byref(g), C.byref(most)))
        return int(g.value), int(most.value)

    def exchange(self, halo):
        """refreshes the ghosts behind the own points; returns cloud[: n + ghosts]."""
        while True:
            ghosts, most = self._call(halo, self.cap, self.ghost_cap)
            if most <= self.cap and ghosts <= self.ghost_cap:
                return self.cloud[: self.n + ghosts]
            self._allocate(self.cloud[: self.n].clone(), int(max(most, self.cap) * self.margin) + 64)  # same decision on every rank


def halo_is_sufficient(points, kth_dist_sq, bounds, halo, rank, world_size, group=None):
    """True on every rank iff every own query's k-th neighbour lies inside the region the halo
    covers: sqrt(d_k) <= distance to the nearest shared face + halo."""
    if world_size == 1:
        return True
    faces = _interior_faces(rank, world_size)
    x0, x1, z0, z1 = bounds
    x, z = points[:, 0], points[:, 2]
    inf = torch.full_like(x, math.inf)
    reach = torch.minimum(
        torch.minimum(x - x0 if faces[0] else inf, x1 - x if faces[1] else inf),
        torch.minimum(z - z0 if faces[2] else inf, z1 - z if faces[3] else inf),
    ) + halo
    bad = (torch.sqrt(kth_dist_sq.to(torch.float64)) > reach.to(torch.float64)).any().to(torch.int32).reshape(1)
    _all_reduce(bad, dist.ReduceOp.MAX, group)
    return int(bad.item()) == 0


def reduce_packed_min(packed, group=None):
    """element-wise MIN of the packed (depth_bits << 32 | index) frame buffers of all ranks.
    depth < 1.0 keeps the sign bit clear, so the signed int64 order equals the unsigned one."""
    return _all_reduce(packed, dist.ReduceOp.MIN, group)


class _Done:
    def wait(self):
        return None


def reduce_packed_min_async(packed, group=None):
    """`reduce_packed_min` as an asynchronous collective: RCCL runs it on the process group's own stream while the
    caller's stream keeps computing; `.wait()` makes the current stream wait for it.  (gloo rehearsal: host-staged,
    synchronous.)"""
    if packed.is_cuda and _host_staged(group):
        _all_reduce(packed, dist.ReduceOp.MIN, group)
        return _Done()
    try:
        return dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):  # a backend without asynchronous collectives: do it in line
        dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=group)
        return _Done()


def partition_segments(counts, world):
    """SURVEY 8 E, first form: the reference's kNN units are whole segments (one KD-tree per segment,
    calculations.rs:97-118), so they shard with no exchange at all.  Greedy longest-processing-time assignment of
    segments to ranks by point count; deterministic (ties: lower segment id first, lower rank first).
    Returns owner[n_segments] (int64 numpy)."""
    import numpy as np

    counts = np.asarray(counts, dtype=np.int64)
    owner = np.zeros(len(counts), np.int64)
    load = np.zeros(world, np.int64)
    for sg in np.lexsort((np.arange(len(counts)), -counts)):
        r = int(np.argmin(load))
        owner[sg] = r
        load[r] += counts[sg]
    return owner
