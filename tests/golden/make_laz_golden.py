"""Generates tests/golden/laz_fmt3.npz: a small LAZ image (point format 3 + 3 extra bytes, 600 records in chunks of 250)
written by the independent Python encoder (oracle/laz_encoder.py), with the records it encodes.  The reference holds no LAZ
file and the crate it uses is not in the snapshot (parity with LASzip's own files is unpinned: tests/test_laz.py); the
fixture pins the library's decoder and both encoders against silent drift.  Run:  python tests/golden/make_laz_golden.py"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import laz_encoder as E  # noqa: E402
from treee_amd import synth  # noqa: E402


def main():
    from test_laz import flight_lines

    rec = flight_lines(600, 2024, 3, 37)
    img = bytearray(synth.las_bytes(np.zeros((600, 3), np.int32), fmt=3, record_length=37))
    (off,) = struct.unpack_from("<I", img, 96)
    img[off:] = rec.tobytes()
    laz = E.laz_bytes(bytes(img), chunk_size=250)
    np.savez_compressed(os.path.join(HERE, "laz_fmt3.npz"), laz=np.frombuffer(laz, np.uint8), las=np.frombuffer(bytes(img), np.uint8), records=rec)
    print("laz golden written:", len(laz), "bytes for", rec.size)


if __name__ == "__main__":
    main()
