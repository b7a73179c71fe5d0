"""The reference's on-disk project format `.ipc`: `bincode` 1.3.3 with its default options
(little-endian, fixed-width integers, u64 length prefixes, enum variants as u32, `Option` as a u8 tag) of

    InteractiveSave { segments: HashMap<u32, SegmentData>, deleted: SegmentData,
                      world_offset: Point3<f64>, source_location: String }          treee/src/interactive/mod.rs:44-50
    SegmentData     { points: Vec<Point3<f32>>, classifications: Vec<Classification>, info: SegmentInformation,
                      min: Point3<f32>, max: Point3<f32>, coords: Option<(f64, f64)> }  treee/src/calculations.rs:35-43
    SegmentInformation { trunk_height, crown_height, ground_sep, crown_sep: f32 }     treee/src/calculations.rs:191-197
    Classification  { Ground = 0, Trunk = 1, Crown = 2 }                              treee/src/calculations.rs:55-59

written by `bincode::serialize_into` (interactive/mod.rs:393,668) and read by `bincode::deserialize_from` (:276,302).

How a nalgebra 0.33 `Point3<T>` serialises is the one detail the snapshot cannot settle (nalgebra is a crates.io dependency,
not vendored; SURVEY.md section 8 F2 flags it).  `Point` forwards to its coordinate vector, the vector to its
`ArrayStorage`, and which serde call that makes decides the bytes under bincode:

    "seq"   `serialize_seq(Some(3))` -> a u64 length 3, then the three coordinates: 20 bytes per Point3<f32>, 32 per
            Point3<f64>.  What SURVEY section 8 F2 believes nalgebra 0.33 does; the WRITER'S DEFAULT.
    "flat"  a tuple / fixed array -> the three coordinates and nothing else: 12 / 24 bytes.

The reader accepts both and reports which one it met (`save["layout"]`; the two cannot be confused: a flat file read as
"seq" would need its first coordinate pair to be the bit pattern of the integer 3); the writer takes the layout as a
parameter (default: the one the project was read with, else "seq").  NEITHER is verified against a file written by the
reference -- none exists here; both are pinned by the bincode rules and hand-assembled vectors (tests/test_host_logic.py).
The library never writes this format on its own: this module only lets the Python host read a saved project into the
(points, segment) pair the GPU path takes (`treee_amd.curvature_segments`) and write one back unchanged.
"""
import struct

import numpy as np

GROUND, TRUNK, CROWN = 0, 1, 2


class _Reader:
    def __init__(self, data):
        self.b, self.o = memoryview(data), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("unexpected end of .ipc data")  # bincode: io::ErrorKind::UnexpectedEof
        v = self.b[self.o : self.o + n]
        self.o += n
        return v

    def u8(self):
        return self.take(1)[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def arr(self, dtype, count):
        dt = np.dtype(dtype).newbyteorder("<")
        return np.frombuffer(self.take(dt.itemsize * count), dtype=dt, count=count).copy()

    def points3(self, dtype, count, layout):
        """`count` Point3<dtype> in a row -> [count, 3]"""
        if layout == "flat":
            return self.arr(dtype, 3 * count).reshape(count, 3)
        dt = np.dtype([("len", "<u8"), ("xyz", np.dtype(dtype).newbyteorder("<"), 3)])
        rec = np.frombuffer(self.take(dt.itemsize * count), dtype=dt, count=count)
        if count and not (rec["len"] == 3).all():
            raise ValueError("Point3 without its length prefix 3")  # bincode: invalid length, expected 3 elements
        return rec["xyz"].copy().reshape(count, 3)


def _read_segment(r, layout):
    n = r.u64()
    points = r.points3(np.float32, n, layout)
    m = r.u64()
    classifications = r.arr(np.uint32, m)
    if m and classifications.max() > CROWN:
        raise ValueError("invalid Classification variant")  # bincode: invalid value, expected variant index 0 <= i < 3
    info = dict(zip(("trunk_height", "crown_height", "ground_sep", "crown_sep"), r.arr(np.float32, 4).tolist()))
    mn, mx = r.points3(np.float32, 1, layout)[0], r.points3(np.float32, 1, layout)[0]
    tag = r.u8()
    if tag > 1:
        raise ValueError("invalid Option tag")
    coords = tuple(r.arr(np.float64, 2).tolist()) if tag else None
    return {"points": points, "classifications": classifications, "info": info, "min": mn, "max": mx, "coords": coords}


def _read_as(data, layout):
    r = _Reader(data)
    count = r.u64()
    segments = {}
    for _ in range(count):
        key = r.u32()
        segments[key] = _read_segment(r, layout)
    deleted = _read_segment(r, layout)
    world_offset = r.points3(np.float64, 1, layout)[0]
    source_location = bytes(r.take(r.u64())).decode("utf-8")
    if r.o != len(r.b):
        raise ValueError("trailing bytes after InteractiveSave")
    return {"segments": segments, "deleted": deleted, "world_offset": world_offset, "source_location": source_location, "layout": layout}


def read_ipc(data, layout="auto"):
    """bytes of a `.ipc` file -> {"segments": {id: segment}, "deleted": segment, "world_offset": f64[3],
    "source_location": str, "layout": "seq" | "flat"}; a segment is the dict built by `_read_segment` (numpy arrays,
    little-endian).  `layout`: "seq", "flat" (module docstring) or "auto" = whichever of the two the bytes are."""
    if layout in ("seq", "flat"):
        return _read_as(data, layout)
    if layout != "auto":
        raise ValueError("layout must be 'seq', 'flat' or 'auto'")
    try:
        return _read_as(data, "seq")
    except (ValueError, UnicodeDecodeError) as first:
        try:
            return _read_as(data, "flat")
        except (ValueError, UnicodeDecodeError):
            raise first from None


def _points3_bytes(a, dtype, layout):
    a = np.ascontiguousarray(a, dtype=dtype).reshape(-1, 3)
    if layout == "flat":
        return a.tobytes()
    rec = np.empty(len(a), np.dtype([("len", "<u8"), ("xyz", dtype, 3)]))
    rec["len"], rec["xyz"] = 3, a
    return rec.tobytes()


def _write_segment(out, seg, layout):
    pts = np.ascontiguousarray(seg["points"], dtype="<f4").reshape(-1, 3)
    cls = np.ascontiguousarray(seg["classifications"], dtype="<u4")
    out.append(struct.pack("<Q", len(pts)))
    out.append(_points3_bytes(pts, "<f4", layout))
    out.append(struct.pack("<Q", len(cls)))
    out.append(cls.tobytes())
    info = seg["info"]
    out.append(struct.pack("<4f", info["trunk_height"], info["crown_height"], info["ground_sep"], info["crown_sep"]))
    out.append(_points3_bytes(seg["min"], "<f4", layout))
    out.append(_points3_bytes(seg["max"], "<f4", layout))
    if seg["coords"] is None:
        out.append(b"\x00")
    else:
        out.append(b"\x01" + struct.pack("<2d", *seg["coords"]))


def write_ipc(save, layout=None):
    """inverse of read_ipc (HashMap iteration order is unspecified in the reference; here: insertion order).
    `layout`: "seq" or "flat" (module docstring); None = the layout `save` was read with, "seq" for a project built here."""
    layout = layout or save.get("layout") or "seq"
    if layout not in ("seq", "flat"):
        raise ValueError("layout must be 'seq' or 'flat'")
    out = [struct.pack("<Q", len(save["segments"]))]
    for key, seg in save["segments"].items():
        out.append(struct.pack("<I", key))
        _write_segment(out, seg, layout)
    _write_segment(out, save["deleted"], layout)
    out.append(_points3_bytes(save["world_offset"], "<f8", layout))
    loc = save["source_location"].encode("utf-8")
    out.append(struct.pack("<Q", len(loc)) + loc)
    return b"".join(out)


def to_cloud(save):
    """(points f32[n,3], segment u32[n] dense ids, ids list): the `(points, segment: Vec<u32>)` pair of a saved
    project in the order of `ids` -- what `treee_amd.curvature_segments(points, segment, len(ids))` takes."""
    ids = list(save["segments"].keys())
    pts = [save["segments"][k]["points"] for k in ids]
    seg = [np.full(len(p), i, np.uint32) for i, p in enumerate(pts)]
    if not pts:
        return np.zeros((0, 3), np.float32), np.zeros(0, np.uint32), ids
    return np.concatenate(pts).astype(np.float32), np.concatenate(seg), ids
