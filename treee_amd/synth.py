"""Seeded synthetic clouds of SURVEY.md section 8 row D (the reference ships no data; its one
example file is fetched over the network, treee/src/empty.rs:72).

Coordinates are in the reference's post-load frame (y up, x/z horizontal, metres, f32;
treee/src/laz.rs:202-213) and quantised to 1 mm like a LAS file with scale 0.001, which
provokes exact distance ties.  Counter-based SplitMix64 keeps every point a pure function
of (seed, index), so shards of one cloud can be generated independently on every rank.
"""
import numpy as np

SEED_UNIFORM = 0x7EEE0001
SEED_FOREST = 0x7EEE000F
SEED_FOREST_150M = 0x7EEE0096

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def _u01(seed, stream, idx):
    """uniform [0,1) with 24 bits: (u >> 40) * 2^-24 (f64 holding an exact f32 value)."""
    base = _splitmix64(np.uint64(seed) ^ np.uint64((stream * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over="ignore"):
        u = _splitmix64((base + idx.astype(np.uint64)) & _M)
    return (u >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def _quantise(v):
    return (np.round(v * 1000.0) / 1000.0).astype(np.float32)


def uniform_cloud(n=1_000_000, seed=SEED_UNIFORM, start=0):
    """cfg 1: x,z ~ U[-50,50), y ~ U[0,30)."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    out = np.empty((n, 3), np.float32)
    out[:, 0] = _quantise(_u01(seed, 0, idx) * 100.0 - 50.0)
    out[:, 1] = _quantise(_u01(seed, 1, idx) * 30.0)
    out[:, 2] = _quantise(_u01(seed, 2, idx) * 100.0 - 50.0)
    return out


def _ground(x, z):
    return 0.5 * np.sin(x / 17.0) + 0.5 * np.cos(z / 23.0)


def _fold(v, h, xp):
    """reflect coordinates that overshoot the plot [-h, h) back inside (crowns of edge trees), so
    that plots tile the plane without overlap and every shard owns exactly its own tile."""
    v = xp.where(v >= h, 2 * h - v - 1e-3, v)
    return xp.where(v < -h, -2 * h - v, v)


def forest_cloud(n=15_000_000, plot=200.0, seed=SEED_FOREST, origin=(0.0, 0.0), start=0, chunk=4_000_000):
    """cfg 2/3: forest plot of `plot` x `plot` metres centred on `origin` (x, z).

    20 % ground points on a gently rolling surface, trees on a 4 m jittered grid, each tree
    25 % trunk points (cylinder) and 75 % crown points (ellipsoid).  Returns
    (xyz float32 [n,3], segment uint32 [n]) with segment = tree id (0 = ground)."""
    xyz = np.empty((n, 3), np.float32)
    seg = np.empty(n, np.uint32)
    side = max(1, int(round(plot / 4.0)))
    ntrees = side * side
    t = np.arange(ntrees, dtype=np.uint64)
    tx = ((t % np.uint64(side)).astype(np.float64) + 0.5) * (plot / side) - plot / 2 + (_u01(seed, 10, t) * 3.0 - 1.5)
    tz = ((t // np.uint64(side)).astype(np.float64) + 0.5) * (plot / side) - plot / 2 + (_u01(seed, 11, t) * 3.0 - 1.5)
    theight = 12.0 + 16.0 * _u01(seed, 12, t)
    trad = 0.12 + 0.23 * _u01(seed, 13, t)
    crx = 1.5 + 1.5 * _u01(seed, 14, t)
    cry = 0.45 * theight / 2.0
    tground = _ground(tx, tz)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = np.arange(start + lo, start + hi, dtype=np.uint64)
        kind = _u01(seed, 0, idx)
        tree = np.minimum((_u01(seed, 1, idx) * ntrees).astype(np.int64), ntrees - 1)
        part = _u01(seed, 2, idx)
        u3, u4, u5, u6, u7 = (_u01(seed, s, idx) for s in (3, 4, 5, 6, 7))
        # ground
        gx = u3 * plot - plot / 2
        gz = u4 * plot - plot / 2
        noise = np.sqrt(-2.0 * np.log(1.0 - u5)) * np.cos(2.0 * np.pi * u6) * 0.03
        gy = _ground(gx, gz) + noise
        # trunk: cylinder up to 55 % of the height
        ang = 2.0 * np.pi * u3
        kx = tx[tree] + trad[tree] * np.cos(ang)
        kz = tz[tree] + trad[tree] * np.sin(ang)
        ky = tground[tree] + u4 * 0.55 * theight[tree]
        # crown: uniform in an ellipsoid whose top touches the tree height
        rr = np.cbrt(u5)
        ct = 2.0 * u6 - 1.0
        st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
        ph = 2.0 * np.pi * u7
        cx = tx[tree] + crx[tree] * rr * st * np.cos(ph)
        cz = tz[tree] + crx[tree] * rr * st * np.sin(ph)
        cy = tground[tree] + theight[tree] - cry[tree] + cry[tree] * rr * ct
        is_ground = kind < 0.2
        is_trunk = (~is_ground) & (part < 0.25)
        x = _fold(np.where(is_ground, gx, np.where(is_trunk, kx, cx)), plot / 2, np) + origin[0]
        y = np.where(is_ground, gy, np.where(is_trunk, ky, cy))
        z = _fold(np.where(is_ground, gz, np.where(is_trunk, kz, cz)), plot / 2, np) + origin[1]
        xyz[lo:hi, 0] = _quantise(x)
        xyz[lo:hi, 1] = _quantise(y)
        xyz[lo:hi, 2] = _quantise(z)
        seg[lo:hi] = np.where(is_ground, 0, tree + 1).astype(np.uint32)
    return xyz, seg


def las_bytes(xyz_las_int, scale=(0.001, 0.001, 0.001), offset=(0.0, 0.0, 0.0), fmt=0, record_length=20, version=(1, 2), extra_header=0):
    """Build an uncompressed LAS image from integer records (n,3) int32 -- fixture writer
    (layout of treee/src/laz.rs:228-268; the header is padded to the 375 bytes the reference
    always reads)."""
    import struct

    ints = np.ascontiguousarray(xyz_las_int, dtype=np.int32)
    n = len(ints)
    header_size = 375 + extra_header
    h = bytearray(header_size)
    h[0:4] = b"LASF"
    h[24], h[25] = version
    struct.pack_into("<HII", h, 94, header_size, header_size, 0)
    h[104] = fmt
    struct.pack_into("<H", h, 105, record_length)
    struct.pack_into("<I", h, 107, n if version[1] < 4 else 0)
    struct.pack_into("<3d", h, 131, *scale)
    struct.pack_into("<3d", h, 155, *offset)
    w = ints.astype(np.float64) * np.asarray(scale) + np.asarray(offset)
    mx, mn = (w.max(0), w.min(0)) if n else (np.zeros(3), np.zeros(3))
    struct.pack_into("<6d", h, 179, mx[0], mn[0], mx[1], mn[1], mx[2], mn[2])
    struct.pack_into("<Q", h, 247, n)
    rec = np.zeros((n, record_length), np.uint8)
    rec[:, :12] = ints.view(np.uint8).reshape(n, 12)
    if record_length > 12:  # deterministic filler for the attribute bytes
        rec[:, 12:] = (np.arange(n)[:, None] * 7 + np.arange(record_length - 12)[None, :]) & 0xFF
    return bytes(h) + rec.tobytes()


# ---- torch (device) generator: same construction, evaluated on the GPU ---------------------------
# Used by bench.py / the multi-GPU path where 15-150 M points must be produced in seconds.
# sin/cos/log run in the device's f64 libm, so coordinates may differ from the numpy
# generator in the last quantisation step for a handful of points; parity tests always use
# the array they are handed, never a regenerated one.

def _splitmix64_t(x):
    import torch

    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)

    z = x + torch.tensor(0x9E3779B97F4A7C15 - (1 << 64), dtype=torch.int64, device=x.device)
    z = (z ^ lsr(z, 30)) * torch.tensor(0xBF58476D1CE4E5B9 - (1 << 64), dtype=torch.int64, device=x.device)
    z = (z ^ lsr(z, 27)) * torch.tensor(0x94D049BB133111EB - (1 << 64), dtype=torch.int64, device=x.device)
    return z ^ lsr(z, 31)


def _u01_t(seed, stream, idx):
    import torch

    base = int(_splitmix64(np.uint64(seed) ^ np.uint64((stream * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)))
    if base >= 1 << 63:
        base -= 1 << 64
    u = _splitmix64_t(idx + torch.tensor(base, dtype=torch.int64, device=idx.device))
    return ((u >> 40) & ((1 << 24) - 1)).to(torch.float64) * (1.0 / 16777216.0)


def forest_cloud_torch(n, device, plot=200.0, seed=SEED_FOREST, origin=(0.0, 0.0), start=0, chunk=8_000_000):
    """forest_cloud evaluated with torch on `device`; returns (xyz float32 [n,3], segment int32 [n])."""
    import math

    import torch

    side = max(1, int(round(plot / 4.0)))
    ntrees = side * side
    t = torch.arange(ntrees, dtype=torch.int64, device=device)
    cell = plot / side
    tx = ((t % side).to(torch.float64) + 0.5) * cell - plot / 2 + (_u01_t(seed, 10, t) * 3.0 - 1.5)
    tz = ((t // side).to(torch.float64) + 0.5) * cell - plot / 2 + (_u01_t(seed, 11, t) * 3.0 - 1.5)
    theight = 12.0 + 16.0 * _u01_t(seed, 12, t)
    trad = 0.12 + 0.23 * _u01_t(seed, 13, t)
    crx = 1.5 + 1.5 * _u01_t(seed, 14, t)
    cry = 0.45 * theight / 2.0

    def ground(x, z):
        return 0.5 * torch.sin(x / 17.0) + 0.5 * torch.cos(z / 23.0)

    tground = ground(tx, tz)
    xyz = torch.empty((n, 3), dtype=torch.float32, device=device)
    seg = torch.empty((n,), dtype=torch.int32, device=device)

    def q(v):
        return (torch.round(v * 1000.0) / 1000.0).to(torch.float32)

    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = torch.arange(start + lo, start + hi, dtype=torch.int64, device=device)
        kind = _u01_t(seed, 0, idx)
        tree = torch.clamp((_u01_t(seed, 1, idx) * ntrees).to(torch.int64), max=ntrees - 1)
        part = _u01_t(seed, 2, idx)
        u3, u4, u5, u6, u7 = (_u01_t(seed, s, idx) for s in (3, 4, 5, 6, 7))
        gx = u3 * plot - plot / 2
        gz = u4 * plot - plot / 2
        gy = ground(gx, gz) + torch.sqrt(-2.0 * torch.log(1.0 - u5)) * torch.cos(2.0 * math.pi * u6) * 0.03
        ang = 2.0 * math.pi * u3
        kx = tx[tree] + trad[tree] * torch.cos(ang)
        kz = tz[tree] + trad[tree] * torch.sin(ang)
        ky = tground[tree] + u4 * 0.55 * theight[tree]
        rr = torch.pow(u5, 1.0 / 3.0)
        ct = 2.0 * u6 - 1.0
        st = torch.sqrt(torch.clamp(1.0 - ct * ct, min=0.0))
        ph = 2.0 * math.pi * u7
        cx = tx[tree] + crx[tree] * rr * st * torch.cos(ph)
        cz = tz[tree] + crx[tree] * rr * st * torch.sin(ph)
        cy = tground[tree] + theight[tree] - cry[tree] + cry[tree] * rr * ct
        is_ground = kind < 0.2
        is_trunk = (~is_ground) & (part < 0.25)
        xyz[lo:hi, 0] = q(_fold(torch.where(is_ground, gx, torch.where(is_trunk, kx, cx)), plot / 2, torch) + origin[0])
        xyz[lo:hi, 1] = q(torch.where(is_ground, gy, torch.where(is_trunk, ky, cy)))
        xyz[lo:hi, 2] = q(_fold(torch.where(is_ground, gz, torch.where(is_trunk, kz, cz)), plot / 2, torch) + origin[1])
        seg[lo:hi] = torch.where(is_ground, torch.zeros_like(tree), tree + 1).to(torch.int32)
    return xyz, seg
