"""Exports of the Interactive phase (SURVEY.md section 8 row F2 iii), host side:

  save_points      treee/src/interactive/mod.rs:1069-1103  ASCII PLY of a segment's points + CalculationProperties,
                   axes un-swizzled to LAS order (x, -z, y)
  segment_save     treee/src/interactive/mod.rs:610-626 + calculations.rs:62-73  `SegmentSave` as serde_json pretty text

Numbers are written the way Rust's `{}` (Display) writes an f32: the shortest decimal string that reads back to the same
f32, positional notation without exponent, no trailing ".0" -- numpy's `format_float_positional(unique=True)` is the
same shortest-digits algorithm.  NaN -> "NaN", infinities -> "inf" / "-inf".  The reference has no sample files; the
format is pinned by the cited lines only (parity unpinned)."""
import io
import json

import numpy as np

from .calculations import CROWN, GROUND, TRUNK  # noqa: F401


def rust_display_f32(v):
    """`format!("{}", v)` for an f32."""
    v = np.float32(v)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "inf" if v > 0 else "-inf"
    return np.format_float_positional(v, unique=True, trim="-")


def save_points(points, classifications, expansion, height, curve, valid=None, out=None):
    """save_points(saver, points, classifications, calculations_properties, valid): `valid` is a classification
    (GROUND / TRUNK / CROWN) or None for every point.  Returns the text (or writes it to `out`)."""
    p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    cls = np.asarray(classifications)
    keep = np.ones(len(p), bool) if valid is None else cls == valid
    w = out or io.StringIO()
    w.write("ply\nformat ascii 1.0\n")
    w.write(f"element vertex {int(keep.sum())}\n")
    for name in ("x", "y", "z", "expansion", "height", "curve"):
        w.write(f"property float {name}\n")
    w.write("end_header\n")
    f = rust_display_f32
    for i in np.nonzero(keep)[0]:
        w.write(f"{f(p[i, 0])} {f(-p[i, 2])} {f(p[i, 1])} {f(expansion[i])} {f(height[i])} {f(curve[i])}\n")
    return None if out else w.getvalue()


def _json_f32(v):
    # serde_json writes a finite f32 with ryu's shortest digits and always a fractional part or an exponent
    # ("1.0", "0.3", "1e20"); non-finite values become null
    v = np.float32(v)
    if not np.isfinite(v):
        return None
    return float(np.format_float_positional(v, unique=True, trim="0")) if 1e-5 <= abs(v) < 1e16 or v == 0 else float(v)


def segment_save(info, seg_min, seg_max, offset, longitude=None, latitude=None):
    """`SegmentSave` (calculations.rs:62-73: `info` flattened, then min, max, offset, latitude, longitude -- the
    last two only when known) through serde_json::to_writer_pretty (two-space indent)."""
    d = {}
    for k, v in zip(("trunk_height", "crown_height", "ground_sep", "crown_sep"), info):
        d[k] = _json_f32(v)
    d["min"] = [_json_f32(v) for v in seg_min]
    d["max"] = [_json_f32(v) for v in seg_max]
    d["offset"] = [float(v) for v in offset]
    if latitude is not None:
        d["latitude"] = float(latitude)
    if longitude is not None:
        d["longitude"] = float(longitude)
    return json.dumps(d, indent=2)
