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
    """The per-step form of `exchange_ghosts` for clouds resident in HBM: ONE collective and ONE host round trip
    per step instead of two and four.

    * the cloud lives in a buffer with room for the ghosts behind the own points (`cloud[:n]` = own points,
      `cloud[n:n+g]` = ghosts), so nothing is concatenated;
    * the send block has a fixed capacity sized by one counted exchange at construction (times `margin`); its
      row 0 is a header (x = the sender's count as int32 bits, y = z = NaN: never inside a box), unused rows are
      NaN;
    * receivers blank their own block and run ONE fused selection over all blocks straight into the ghost region;
    * the host reads (ghost count, every sender's count) in one copy; a sender that outgrew the capacity is seen
      by all ranks in the same data, so all of them grow the buffers together and repeat.
    """

    def __init__(self, points, bounds, halo, rank, world_size, ctx=None, group=None, margin=1.5):
        from .context import default_context

        assert points.is_cuda and world_size > 1
        self.rank, self.world, self.group, self.bounds = rank, world_size, group, bounds
        self.ctx = ctx or default_context(points.device)
        self.n = points.shape[0]
        self.dev = points.device
        self.margin = margin
        self.cnt = torch.zeros((2,), dtype=torch.int32, device=self.dev)
        mine = self._select_boundary(points.contiguous(), halo, None, 0)  # counted once (synchronises)
        count = torch.tensor([mine], dtype=torch.int64, device=self.dev)
        most = int(_all_gather_equal(count, world_size, group).max().item())
        self._allocate(points, int(most * margin) + 64)

    def _allocate(self, points, cap):
        self.cap = cap
        self.ghost_cap = (self.world - 1) * cap
        self.send = torch.empty((cap + 1, 3), dtype=torch.float32, device=self.dev)
        self.recv = torch.empty((self.world, cap + 1, 3), dtype=torch.float32, device=self.dev)
        self.cloud = torch.empty((self.n + self.ghost_cap, 3), dtype=torch.float32, device=self.dev)
        self.cloud[: self.n] = points

    def _select_boundary(self, points, halo, out, cap):
        """own points within `halo` of a shared face -> out (or just counted when out is None)."""
        import numpy as np

        from . import _lib
        from .context import ptr

        faces = _interior_faces(self.rank, self.world)
        x0, x1, z0, z1 = self.bounds
        keep = np.asarray((-_INF, _INF, -_INF, _INF), dtype=np.float32)
        drop = np.asarray((x0 + halo if faces[0] else -_INF, x1 - halo if faces[1] else _INF, z0 + halo if faces[2] else -_INF,
                           z1 - halo if faces[3] else _INF), dtype=np.float32)
        _lib.check(_lib.lib().treee_select_xz_dev(self.ctx.handle, ptr(points), points.shape[0], ptr(keep), ptr(drop), ptr(out), cap, ptr(self.cnt)))
        return int(self.cnt[0].item()) if out is None else None

    def exchange(self, halo):
        """refreshes the ghosts behind the own points; returns cloud[: n + ghosts]."""
        import numpy as np

        from . import _lib
        from .context import ptr

        x0, x1, z0, z1 = self.bounds
        while True:
            cap = self.cap
            self.send.fill_(float("nan"))
            self._select_boundary(self.cloud[: self.n], halo, self.send[1:], cap)
            self.send.view(torch.int32)[0, 0:1].copy_(self.cnt[0:1])
            if _host_staged(self.group):
                self.recv.copy_(_all_gather_equal(self.send, self.world, self.group))
            else:
                dist.all_gather_into_tensor(self.recv.view(-1, 3), self.send, group=self.group)
            heads = self.recv.view(torch.int32)[:, 0, 0].clone()
            self.recv[self.rank].fill_(float("nan"))
            keep = np.asarray((x0 - halo, x1 + halo, z0 - halo, z1 + halo), dtype=np.float32)
            drop = np.zeros(4, np.float32)
            flat = self.recv.view(-1, 3)
            _lib.check(_lib.lib().treee_select_xz_dev(self.ctx.handle, ptr(flat), flat.shape[0], ptr(keep), ptr(drop), ptr(self.cloud[self.n :]),
                                                      self.ghost_cap, ptr(self.cnt[1:])))
            info = torch.cat([self.cnt[1:2], heads]).tolist()  # the step's one host round trip
            ghosts, most = info[0], max(info[1:])
            if most <= cap and ghosts <= self.ghost_cap:
                return self.cloud[: self.n + ghosts]
            self._allocate(self.cloud[: self.n].clone(), int(max(most, cap) * self.margin) + 64)  # same decision on every rank


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
