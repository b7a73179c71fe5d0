"""The reference's on-disk project format `.ipc`: `bincode` 1.3.3 with its default options
(little-endian, fixed-width integers, u64 length prefixes, enum variants as u32, `Option` as a u8 tag) of

    InteractiveSave { segments: HashMap<u32, SegmentData>, deleted: SegmentData,
                      world_offset: Point3<f64>, source_location: String }          treee/src/interactive/mod.rs:44-50
    SegmentData     { points: Vec<Point3<f32>>, classifications: Vec<Classification>, info: SegmentInformation,
                      min: Point3<f32>, max: Point3<f32>, coords: Option<(f64, f64)> }  treee/src/calculations.rs:35-43
    SegmentInformation { trunk_height, crown_height, ground_sep, crown_sep: f32 }     treee/src/calculations.rs:191-197
    Classification  { Ground = 0, Trunk = 1, Crown = 2 }                              treee/src/calculations.rs:55-59

written by `bincode::serialize_into` (interactive/mod.rs:393,668) and read by `bincode::deserialize_from` (:276,302).
A nalgebra `Point3<T>` serialises as its three coordinates without a prefix.  The library never writes this format
on its own -- the layout on disk stays the reference's; this module only lets the Python host read a saved project
into the (points, segment) pair the GPU path takes (`treee_amd.curvature_segments`) and write one back unchanged.

No sample file ships with the reference, so the byte layout is pinned by the published bincode rules and a
hand-assembled vector (tests/test_host_logic.py), not by a reference fixture.
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


def _read_segment(r):
    n = r.u64()
    points = r.arr(np.float32, 3 * n).reshape(n, 3)
    m = r.u64()
    classifications = r.arr(np.uint32, m)
    if m and classifications.max() > CROWN:
        raise ValueError("invalid Classification variant")  # bincode: invalid value, expected variant index 0 <= i < 3
    info = dict(zip(("trunk_height", "crown_height", "ground_sep", "crown_sep"), r.arr(np.float32, 4).tolist()))
    mn, mx = r.arr(np.float32, 3), r.arr(np.float32, 3)
    tag = r.u8()
    if tag > 1:
        raise ValueError("invalid Option tag")
    coords = tuple(r.arr(np.float64, 2).tolist()) if tag else None
    return {"points": points, "classifications": classifications, "info": info, "min": mn, "max": mx, "coords": coords}


def read_ipc(data):
    """bytes of a `.ipc` file -> {"segments": {id: segment}, "deleted": segment, "world_offset": f64[3],
    "source_location": str}; a segment is the dict built by `_read_segment` (numpy arrays, little-endian)."""
    r = _Reader(data)
    count = r.u64()
    segments = {}
    for _ in range(count):
        key = r.u32()
        segments[key] = _read_segment(r)
    deleted = _read_segment(r)
    world_offset = r.arr(np.float64, 3)
    source_location = bytes(r.take(r.u64())).decode("utf-8")
    if r.o != len(r.b):
        raise ValueError("trailing bytes after InteractiveSave")
    return {"segments": segments, "deleted": deleted, "world_offset": world_offset, "source_location": source_location}


def _write_segment(out, seg):
    pts = np.ascontiguousarray(seg["points"], dtype="<f4").reshape(-1, 3)
    cls = np.ascontiguousarray(seg["classifications"], dtype="<u4")
    out.append(struct.pack("<Q", len(pts)))
    out.append(pts.tobytes())
    out.append(struct.pack("<Q", len(cls)))
    out.append(cls.tobytes())
    info = seg["info"]
    out.append(struct.pack("<4f", info["trunk_height"], info["crown_height"], info["ground_sep"], info["crown_sep"]))
    out.append(np.ascontiguousarray(seg["min"], dtype="<f4").tobytes())
    out.append(np.ascontiguousarray(seg["max"], dtype="<f4").tobytes())
    if seg["coords"] is None:
        out.append(b"\x00")
    else:
        out.append(b"\x01" + struct.pack("<2d", *seg["coords"]))


def write_ipc(save):
    """inverse of read_ipc (HashMap iteration order is unspecified in the reference; here: insertion order)."""
    out = [struct.pack("<Q", len(save["segments"]))]
    for key, seg in save["segments"].items():
        out.append(struct.pack("<I", key))
        _write_segment(out, seg)
    _write_segment(out, save["deleted"])
    out.append(np.ascontiguousarray(save["world_offset"], dtype="<f8").tobytes())
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
