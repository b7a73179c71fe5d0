"""LAZ sources (laz.rs:77-94,146-160).  The reference leaves the format to the crate `laz` 0.9.2 (absent from the snapshot) and
no LAZ file or tool exists here, so the library's decoder (treee_amd/csrc/laz_codec.cpp, host code) is checked against an
encoder written independently from the same published format (oracle/laz_encoder.py): parity with real LASzip files is
UNPINNED.  CPU tests: the decoder runs on host threads."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def T():
    import __graft_entry__ as G

    G.build()
    import treee_amd

    return treee_amd


def flight_lines(n, seed, fmt, record_length):
    """records shaped like an airborne scan: points along scan lines with small steps, returns 1..3 of n, slowly varying
    intensity / angle, GPS times in three interleaved sequences with an occasional jump, RGB near-grey with excursions"""
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, record_length), np.uint8)
    x = np.cumsum(rng.integers(-40, 60, n)).astype(np.int64) + 120_000
    y = np.cumsum(rng.integers(-5, 6, n)).astype(np.int64) * 3 - 50_000
    z = (2_000 + 300 * np.sin(np.arange(n) / 37.0) + rng.integers(-25, 25, n)).astype(np.int64)
    jump = rng.random(n) < 0.01  # a new flight line: large steps
    x += np.cumsum(np.where(jump, rng.integers(-2_000_000, 2_000_000, n), 0))
    nret = rng.integers(1, 4, n)
    ret = np.minimum(rng.integers(1, 4, n), nret)
    bitbyte = (ret | (nret << 3) | ((np.arange(n) // 97 % 2) << 6) | ((rng.random(n) < 0.02).astype(np.int64) << 7)).astype(np.uint8)
    intensity = np.clip(np.cumsum(rng.integers(-30, 31, n)) + 900, 0, 65535).astype(np.uint16)
    cls = rng.choice([1, 2, 5], n, p=[0.2, 0.5, 0.3]).astype(np.uint8)
    angle = np.clip(np.cumsum(rng.integers(-1, 2, n)), -90, 90).astype(np.int8)
    user = (rng.random(n) < 0.05).astype(np.uint8) * rng.integers(0, 256, n).astype(np.uint8)
    psid = (17 + (np.cumsum(jump) % 5)).astype(np.uint16)
    rec[:, 0:4] = x.astype(np.int32).view(np.uint8).reshape(n, 4)
    rec[:, 4:8] = y.astype(np.int32).view(np.uint8).reshape(n, 4)
    rec[:, 8:12] = z.astype(np.int32).view(np.uint8).reshape(n, 4)
    rec[:, 12:14] = intensity.view(np.uint8).reshape(n, 2)
    rec[:, 14], rec[:, 15], rec[:, 16], rec[:, 17] = bitbyte, cls, angle.view(np.uint8), user
    rec[:, 18:20] = psid.view(np.uint8).reshape(n, 2)
    o = 20
    if fmt in (1, 3):
        seq = rng.integers(0, 3, n)
        base = np.array([3.1e8, 3.1e8 + 5_000.0, 9.9e8])[seq]
        step = np.array([1.0e-5, 2.5e-5, 4.0e-6])[seq]
        t = base + np.arange(n) * step * np.where(rng.random(n) < 0.05, rng.integers(-12, 700, n), 1)
        t[rng.random(n) < 0.03] = 0.0  # an unset time
        t = np.where(rng.random(n) < 0.1, np.roll(t, 1), t)  # repeated times
        rec[:, o : o + 8] = t.astype(np.float64).view(np.uint8).reshape(n, 8)
        o += 8
    if fmt in (2, 3):
        grey = np.clip(np.cumsum(rng.integers(-400, 401, n)) + 30_000, 0, 65535)
        rgb = np.stack([grey, grey, grey], 1)
        col = rng.random(n) < 0.4
        rgb[col] = np.clip(rgb[col] + rng.integers(-3000, 3000, (int(col.sum()), 3)), 0, 65535)
        rgb[rng.random(n) < 0.1] &= 0xFF00
        rec[:, o : o + 6] = rgb.astype(np.uint16).view(np.uint8).reshape(n, 6)
        o += 6
    if record_length > o:
        rec[:, o:] = (np.cumsum(rng.integers(0, 3, (n, record_length - o)), 0) & 255).astype(np.uint8)
    return rec


def las_image(T, rec, fmt):
    from treee_amd import synth

    n, rl = rec.shape
    img = bytearray(synth.las_bytes(np.zeros((n, 3), np.int32), fmt=fmt, record_length=rl))
    (off,) = struct.unpack_from("<I", img, 96)
    img[off:] = rec.tobytes()
    return bytes(img)


@pytest.mark.parametrize("fmt,rl", [(0, 20), (1, 28), (2, 26), (3, 34), (0, 23), (3, 40)])
def test_laz_round_trip_through_the_independent_encoder(T, fmt, rl):
    from oracle import laz_encoder as E

    rec = flight_lines(2_300, 100 + fmt + rl, fmt, rl)
    las = las_image(T, rec, fmt)
    laz = E.laz_bytes(las, chunk_size=700)
    assert len(laz) < 0.75 * len(las)  # (it does compress)
    src = T.Laz(laz)
    assert src.vlr["chunk_size"] == 700 and [c for _, c in src.chunks] == [700, 700, 700, 200]
    got = []
    src.read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, rl)), threads=3)
    assert np.array_equal(np.concatenate(got), rec)
    assert np.array_equal(src.center, T.Laz(las).center) and src.point_length == rl  # (header fields are the LAS file's)


def test_laz_chunk_layouts_and_limits(T):
    from oracle import laz_encoder as E
    from treee_amd.laz import CorruptFile

    rec = flight_lines(1_000, 7, 1, 28)
    las = las_image(T, rec, 1)
    for kw in ({"variable": [1, 0, 499, 500]}, {"chunk_size": 1_000}, {"chunk_size": 5_000}, {"chunk_size": 1}, {"chunk_size": 333, "table_at_end_marker": True}):
        src = T.Laz(E.laz_bytes(las, **kw))
        got = []
        src.read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, 28)), threads=2)
        assert np.array_equal(np.concatenate([g for g in got if len(g)]), rec), kw
    # the header's point count clips the chunks (laz.rs:88-89)
    laz = bytearray(E.laz_bytes(las, chunk_size=400))
    struct.pack_into("<I", laz, 107, 650)
    src = T.Laz(bytes(laz))
    assert [c for _, c in src.chunks] == [400, 250, 0]
    got = []
    src.read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, 28)))
    assert np.array_equal(np.concatenate(got), rec[:400].tolist() + rec[400:650].tolist())
    # damage: a chunk cut short, an item list the library does not build, a record that is not a laszip record
    laz = E.laz_bytes(las, chunk_size=400)
    cut = bytearray(laz)
    (off,) = struct.unpack_from("<I", cut, 96)
    (table,) = struct.unpack_from("<q", cut, off)
    with pytest.raises(RuntimeError):  # the table's offset points behind the file
        T.Laz(bytes(cut[: table - 10])).read(lambda ch: None)
    hs = struct.unpack_from("<H", laz, 94)[0]
    layered = bytearray(laz)
    struct.pack_into("<H", layered, hs + 54 + 34 + 4, 3)  # first item: version 3
    with pytest.raises(NotImplementedError):
        T.Laz(bytes(layered))
    pointwise = bytearray(laz)
    struct.pack_into("<H", pointwise, hs + 54, 1)  # compressor 1 (no chunks)
    with pytest.raises(NotImplementedError):
        T.Laz(bytes(pointwise))
    short = bytearray(laz)
    struct.pack_into("<H", short, hs + 20, 8)  # record length after payload 8
    with pytest.raises(CorruptFile):
        T.Laz(bytes(short))


def test_range_coder_known_answer(T):
    """the smallest stream by hand: two POINT10 records that differ in nothing -- one 'changed' symbol 0 of 64 (interval
    [0, 512 * 2^17)), then three correctors of k = 0 / bit 0: the encoder's bytes are reproducible and decode back"""
    from oracle import laz_encoder as E

    rec = np.zeros((2, 20), np.uint8)
    rec[:, 14] = 1 | (1 << 3)
    chunk = E.compress_chunk(rec, E.items_of(0, 20))
    assert chunk[:20] == rec[0].tobytes() and len(chunk) <= 20 + 8
    # after `changed` = 0: length = (2^32-1 >> 15) * distribution[1] with distribution[1] = 2^15 / 64 = 512
    enc = E.Encoder()
    enc.symbol(E.SymbolModel(64), 0)
    assert enc.base == 0 and enc.length == (0xFFFFFFFF >> 15) * 512
    las = las_image(T, rec, 0)
    got = []
    T.Laz(E.laz_bytes(las, chunk_size=2)).read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, 20)))
    assert np.array_equal(got[0], rec)


@pytest.mark.parametrize("fmt,rl", [(0, 20), (1, 28), (2, 26), (3, 37)])
def test_laz_round_trip_of_noise_and_extremes(T, fmt, rl):
    """records of uniformly random bytes (every corrector width up to 32 bits, GPS times that fit no sequence, all 64
    'changed' masks), then runs of extreme values: INT32_MIN / MAX coordinates, times at the i64 limits, saturated colours"""
    from oracle import laz_encoder as E

    rng = np.random.default_rng(fmt * 31 + rl)
    rec = rng.integers(0, 256, (1_500, rl), dtype=np.uint8)
    ext = np.zeros((200, rl), np.uint8)
    vals = np.array([-(1 << 31), (1 << 31) - 1, 0, -1, 1], dtype=np.int64)
    for c in range(3):
        ext[:, 4 * c : 4 * c + 4] = vals[rng.integers(0, 5, 200)].astype(np.int32).view(np.uint8).reshape(200, 4)
    ext[:, 12:20] = rng.choice(np.array([0, 255], np.uint8), (200, 8))
    if rl > 20:
        big = np.array([-(1 << 63), (1 << 63) - 1, 0, 1 << 32, -(1 << 32) - 1], dtype=np.int64)
        ext[:, 20:28] = big[rng.integers(0, 5, 200)].view(np.uint8).reshape(200, 8)[:, : min(8, rl - 20)] if rl >= 28 else ext[:, 20:28]
        ext[:, 20:] ^= rng.choice(np.array([0, 255], np.uint8), (200, rl - 20))
    rec = np.concatenate([rec, ext, rec[:100]])
    las = las_image(T, rec, fmt)
    got = []
    T.Laz(E.laz_bytes(las, chunk_size=600)).read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, rl)), threads=4)
    assert np.array_equal(np.concatenate(got), rec)


@pytest.mark.parametrize("fmt,rl", [(0, 20), (1, 28), (2, 26), (3, 34), (3, 41)])
def test_library_encoder_writes_the_bytes_of_the_independent_one(T, fmt, rl):
    """the C++ fixture compressor (tests/cpp/laz_encode_fixture.cpp: fixtures / benchmark input) against oracle/laz_encoder.py (Python): the same file, byte for
    byte -- three statements of the format (two encoders, one decoder) agree.  Then noise, which needs the retry path."""
    from oracle import laz_encoder as E
    from laz_fixture import laz_encode

    rec = np.concatenate([flight_lines(1_700, 40 + fmt, fmt, rl), np.random.default_rng(fmt).integers(0, 256, (300, rl), dtype=np.uint8)])
    las = las_image(T, rec, fmt)
    assert laz_encode(las, chunk_size=450, threads=3) == E.laz_bytes(las, chunk_size=450)
    noise = las_image(T, np.random.default_rng(9).integers(0, 256, (4_000, rl), dtype=np.uint8), fmt)
    laz = laz_encode(noise, chunk_size=1_000)
    got = []
    T.Laz(laz).read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, rl)))
    (off,) = struct.unpack_from("<I", noise, 96)
    assert np.concatenate(got).tobytes() == noise[off:]
    with pytest.raises(RuntimeError):
        laz_encode(b"LASX" + las[4:])


def test_laz_decoder_survives_damaged_streams(T):
    """bytes flipped inside the range-coded data, a chunk table that claims billions of chunks, a table cut short: an error
    or wrong records, never a crash or a hang (the decoder's model invariants do not depend on the data)"""
    from oracle import laz_encoder as E

    rec = flight_lines(3_000, 3, 3, 34)
    laz = bytearray(E.laz_bytes(las_image(T, rec, 3), chunk_size=1_000))
    (off,) = struct.unpack_from("<I", laz, 96)
    (table,) = struct.unpack_from("<q", laz, off)
    rng = np.random.default_rng(1)
    for _ in range(40):
        bad = bytearray(laz)
        for pos in rng.integers(off + 8, table, 6):
            bad[pos] ^= int(rng.integers(1, 256))
        got = []
        try:
            T.Laz(bytes(bad)).read(lambda ch: got.append(np.asarray(ch.slice)), threads=2)
            assert sum(len(g) for g in got) == rec.size  # (decoded to something of the right size)
        except RuntimeError:
            pass
    bad = bytearray(laz)
    struct.pack_into("<I", bad, table + 4, 0xFFFFFFF0)
    with pytest.raises(RuntimeError):
        T.Laz(bytes(bad))
    with pytest.raises(RuntimeError):
        T.Laz(bytes(laz[: table + 9]))


def test_laz_reader_fuzz(T):
    """random bytes written anywhere into LAZ images (header, laszip record, chunks, chunk table), files cut short: an
    exception or a read of the promised size, never a crash (the same loop ran 3 600 files under ASan + UBSan:
    tests/sanitize_host.sh)"""
    from oracle import laz_encoder as E
    from treee_amd._lib import TreeeError
    from treee_amd.laz import CorruptFile

    rng = np.random.default_rng(5)
    outcomes = set()
    for fmt, rl in ((0, 20), (3, 37)):
        laz = E.laz_bytes(las_image(T, flight_lines(900, fmt, fmt, rl), fmt), chunk_size=400)
        for it in range(120):
            bad = bytearray(laz)
            lo, hi = [(0, len(bad)), (94, 480), (len(bad) - 40, len(bad)), (375, len(bad))][it % 4]
            for pos in rng.integers(lo, hi, int(rng.integers(1, 8))):
                bad[pos] = int(rng.integers(0, 256))
            if it % 8 == 3:
                bad = bad[: int(rng.integers(380, len(bad)))]
            try:
                with np.errstate(all="ignore"):
                    src = T.Laz(bytes(bad))
                got = []
                src.read(lambda ch: got.append(len(ch.slice)), threads=2)
                assert sum(got) == sum(c for _, c in src.chunks) * src.point_length
                outcomes.add("ok")
            except (TreeeError, CorruptFile, NotImplementedError, MemoryError, OverflowError, ValueError) as e:
                outcomes.add(type(e).__name__)
    assert {"ok", "TreeeError"} <= outcomes


def test_laz_golden_fixture(T):
    """tests/golden/laz_fmt3.npz (make_laz_golden.py): the committed LAZ image decodes to the committed records, and both
    encoders still write exactly that image"""
    from oracle import laz_encoder as E
    from laz_fixture import laz_encode

    g = np.load(os.path.join(ROOT, "tests", "golden", "laz_fmt3.npz"))
    laz, las, rec = g["laz"].tobytes(), g["las"].tobytes(), g["records"]
    got = []
    T.Laz(laz).read(lambda ch: got.append(np.asarray(ch.slice).reshape(-1, rec.shape[1])))
    assert [len(x) for x in got] == [250, 250, 100] and np.array_equal(np.concatenate(got), rec)
    assert E.laz_bytes(las, chunk_size=250) == laz and laz_encode(las, chunk_size=250) == laz
