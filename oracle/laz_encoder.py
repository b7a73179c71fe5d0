"""CPU ORACLE (test infrastructure): a LASzip ENCODER written independently of treee_amd/csrc/laz_codec.cpp, from the same
published format (M. Isenburg, "LASzip: lossless compression of LiDAR data", PE&RS 79(2), 2013; the range coder of A. Said's
FastAC that LASzip adopts): pointwise chunked compressor, items POINT10 / GPSTIME11 / RGB12 / BYTE, version 2.  It exists so
that the library's decoder can be checked on something: the reference hands LAZ to the crate `laz` 0.9.2, which is not in
the snapshot, and no LAZ file or tool exists in the build container.
PARITY STATUS: unpinned -- encoder and decoder restate the same document; a real LAZ file has never been through either.
Pure Python, slow: small clouds only."""
import struct

import numpy as np

AC_MIN_LENGTH, AC_MAX_LENGTH = 1 << 24, 0xFFFFFFFF
M32 = 0xFFFFFFFF


class BitModel:
    def __init__(self):
        self.update_cycle = self.bits_until_update = 4
        self.bit_0_prob, self.bit_0_count, self.bit_count = 1 << 12, 1, 2

    def update(self):
        self.bit_count += self.update_cycle
        if self.bit_count > (1 << 13):
            self.bit_count = (self.bit_count + 1) >> 1
            self.bit_0_count = (self.bit_0_count + 1) >> 1
            if self.bit_0_count == self.bit_count:
                self.bit_count += 1
        scale = 0x80000000 // self.bit_count
        self.bit_0_prob = ((self.bit_0_count * scale) & M32) >> (31 - 13)
        self.update_cycle = min((5 * self.update_cycle) >> 2, 64)
        self.bits_until_update = self.update_cycle


class SymbolModel:
    def __init__(self, symbols):
        self.symbols, self.last_symbol = symbols, symbols - 1
        self.count = [1] * symbols
        self.distribution = [0] * symbols
        self.total_count, self.update_cycle = 0, symbols
        self.update()
        self.symbols_until_update = self.update_cycle = (symbols + 6) >> 1

    def update(self):
        self.total_count += self.update_cycle
        if self.total_count > (1 << 15):
            self.total_count = 0
            for k in range(self.symbols):
                self.count[k] = (self.count[k] + 1) >> 1
                self.total_count += self.count[k]
        scale, acc = 0x80000000 // self.total_count, 0
        for k in range(self.symbols):
            self.distribution[k] = ((scale * acc) & M32) >> (31 - 15)
            acc += self.count[k]
        self.update_cycle = min((5 * self.update_cycle) >> 2, (self.symbols + 6) << 3)
        self.symbols_until_update = self.update_cycle


class Encoder:
    """32-bit range encoder; carries ripple back through the bytes already written"""

    def __init__(self):
        self.out, self.base, self.length = bytearray(), 0, AC_MAX_LENGTH

    def _carry(self):
        i = len(self.out) - 1
        while self.out[i] == 0xFF:
            self.out[i] = 0
            i -= 1
        self.out[i] += 1

    def _advance(self, add):
        b = self.base + add
        self.base = b & M32
        if b > M32:
            self._carry()

    def _renorm(self):
        while True:
            self.out.append(self.base >> 24)
            self.base = (self.base << 8) & M32
            self.length = (self.length << 8) & M32
            if self.length >= AC_MIN_LENGTH:
                break

    def bit(self, m, sym):
        x = m.bit_0_prob * (self.length >> 13)
        if sym == 0:
            self.length = x
            m.bit_0_count += 1
        else:
            self._advance(x)
            self.length -= x
        if self.length < AC_MIN_LENGTH:
            self._renorm()
        m.bits_until_update -= 1
        if m.bits_until_update == 0:
            m.update()

    def symbol(self, m, sym):
        assert 0 <= sym < m.symbols
        if sym == m.last_symbol:
            x = m.distribution[sym] * (self.length >> 15)
            self._advance(x)
            self.length -= x
        else:
            self.length >>= 15
            x = m.distribution[sym] * self.length
            self._advance(x)
            self.length = m.distribution[sym + 1] * self.length - x
        if self.length < AC_MIN_LENGTH:
            self._renorm()
        m.count[sym] += 1
        m.symbols_until_update -= 1
        if m.symbols_until_update == 0:
            m.update()

    def short(self, sym):
        self.length >>= 16
        self._advance(sym * self.length)
        if self.length < AC_MIN_LENGTH:
            self._renorm()

    def bits(self, bits, sym):
        if bits > 19:
            self.short(sym & 0xFFFF)
            sym >>= 16
            bits -= 16
        self.length >>= bits
        self._advance(sym * self.length)
        if self.length < AC_MIN_LENGTH:
            self._renorm()

    def int32(self, sym):
        self.short(sym & 0xFFFF)
        self.short(sym >> 16)

    def done(self):
        another = True
        if self.length > 2 * AC_MIN_LENGTH:
            self._advance(AC_MIN_LENGTH)
            self.length = AC_MIN_LENGTH >> 1
        else:
            self._advance(AC_MIN_LENGTH >> 1)
            self.length = AC_MIN_LENGTH >> 9
            another = False
        self._renorm()
        self.out += b"\0\0" + (b"\0" if another else b"")  # what the decoder reads ahead
        return bytes(self.out)


def i32(v):
    v &= M32
    return v - (1 << 32) if v >= (1 << 31) else v


class IntegerCompressor:
    def __init__(self, enc, bits=16, contexts=1, bits_high=8):
        self.enc, self.bits_high, self.k = enc, bits_high, 0
        if bits and bits < 32:
            self.corr_bits, self.corr_range = bits, 1 << bits
            self.corr_min = -(self.corr_range // 2)
            self.corr_max = self.corr_min + self.corr_range - 1
        else:
            self.corr_bits, self.corr_range, self.corr_min, self.corr_max = 32, 0, -(1 << 31), (1 << 31) - 1
        self.m_bits = [SymbolModel(self.corr_bits + 1) for _ in range(contexts)]
        self.m_corr0 = BitModel()
        self.m_corr = [None] + [SymbolModel(1 << min(i, bits_high)) for i in range(1, self.corr_bits + 1)]

    def compress(self, pred, real, context=0):
        corr = i32(real - pred)
        if corr < self.corr_min:
            corr += self.corr_range
        elif corr > self.corr_max:
            corr -= self.corr_range
        c1 = -corr if corr <= 0 else corr - 1
        k = c1.bit_length()  # significant bits of the corrector
        self.k = k
        self.enc.symbol(self.m_bits[context], k)
        if k == 0:
            self.enc.bit(self.m_corr0, corr)
        elif k < 32:
            c = corr + ((1 << k) - 1) if corr < 0 else corr - 1
            if k <= self.bits_high:
                self.enc.symbol(self.m_corr[k], c)
            else:
                k1 = k - self.bits_high
                self.enc.symbol(self.m_corr[k], c >> k1)
                self.enc.bits(k1, c & ((1 << k1) - 1))


class Median5:
    def __init__(self):
        self.v, self.high = [0] * 5, True

    def get(self):
        return self.v[2]

    def add(self, x):
        v = self.v
        if self.high:
            if x < v[2]:
                v[4], v[3] = v[3], v[2]
                if x < v[0]:
                    v[2], v[1], v[0] = v[1], v[0], x
                elif x < v[1]:
                    v[2], v[1] = v[1], x
                else:
                    v[2] = x
            else:
                if x < v[3]:
                    v[4], v[3] = v[3], x
                else:
                    v[4] = x
                self.high = False
        else:
            if v[2] < x:
                v[0], v[1] = v[1], v[2]
                if v[4] < x:
                    v[2], v[3], v[4] = v[3], v[4], x
                elif v[3] < x:
                    v[2], v[3] = v[3], x
                else:
                    v[2] = x
            else:
                if v[1] < x:
                    v[0], v[1] = v[1], x
                else:
                    v[0] = x
                self.high = True


RETURN_MAP = [[15, 14, 13, 12, 11, 10, 9, 8], [14, 0, 1, 3, 6, 10, 10, 9], [13, 1, 2, 4, 7, 11, 11, 10], [12, 3, 4, 5, 8, 12, 12, 11],
              [11, 6, 7, 8, 9, 13, 13, 12], [10, 10, 11, 12, 13, 14, 14, 13], [9, 10, 11, 12, 13, 14, 15, 14], [8, 9, 10, 11, 12, 13, 14, 15]]
RETURN_LEVEL = [[abs(r - n) for r in range(8)] for n in range(8)]


class Point10Writer:
    size, type = 20, 6

    def __init__(self, enc, first):
        self.enc = enc
        self.last = list(struct.unpack("<iiiHBBbBH", first))
        self.last[3] = 0  # the intensity context starts at zero
        self.last_intensity = [0] * 16
        self.dx, self.dy = [Median5() for _ in range(16)], [Median5() for _ in range(16)]
        self.last_height = [0] * 8
        self.m_changed = SymbolModel(64)
        self.m_scan = [SymbolModel(256), SymbolModel(256)]
        self.m_bit, self.m_cls, self.m_user = {}, {}, {}
        self.ic_intensity, self.ic_psid = IntegerCompressor(enc, 16, 4), IntegerCompressor(enc, 16, 1)
        self.ic_dx, self.ic_dy, self.ic_z = IntegerCompressor(enc, 32, 2), IntegerCompressor(enc, 32, 22), IntegerCompressor(enc, 32, 20)

    def write(self, rec):
        x, y, z, intensity, bitbyte, cls, angle, user, psid = struct.unpack("<iiiHBBbBH", rec)
        last = self.last
        r, n = bitbyte & 7, (bitbyte >> 3) & 7
        m, lvl = RETURN_MAP[n][r], RETURN_LEVEL[n][r]
        changed = ((last[4] != bitbyte) << 5) | ((self.last_intensity[m] != intensity) << 4) | ((last[5] != cls) << 3) | ((last[6] != angle) << 2) | \
                  ((last[7] != user) << 1) | (last[8] != psid)
        enc = self.enc
        enc.symbol(self.m_changed, changed)
        if changed & 32:
            enc.symbol(self.m_bit.setdefault(last[4], SymbolModel(256)), bitbyte)
        if changed & 16:
            self.ic_intensity.compress(self.last_intensity[m], intensity, min(m, 3))
            self.last_intensity[m] = intensity
        if changed & 8:
            enc.symbol(self.m_cls.setdefault(last[5], SymbolModel(256)), cls)
        if changed & 4:
            enc.symbol(self.m_scan[(bitbyte >> 6) & 1], (angle - last[6]) & 255)
        if changed & 2:
            enc.symbol(self.m_user.setdefault(last[7], SymbolModel(256)), user)
        if changed & 1:
            self.ic_psid.compress(last[8], psid)
        d = i32(x - last[0])
        self.ic_dx.compress(self.dx[m].get(), d, int(n == 1))
        self.dx[m].add(d)
        kb = self.ic_dx.k
        d = i32(y - last[1])
        self.ic_dy.compress(self.dy[m].get(), d, int(n == 1) + ((kb & ~1) if kb < 20 else 20))
        self.dy[m].add(d)
        kb = (self.ic_dx.k + self.ic_dy.k) // 2
        self.ic_z.compress(self.last_height[lvl], z, int(n == 1) + ((kb & ~1) if kb < 18 else 18))
        self.last_height[lvl] = z
        self.last = [x, y, z, intensity, bitbyte, cls, angle, user, psid]


GPS_MULTI, GPS_MINUS = 500, -10
GPS_UNCHANGED, GPS_FULL, GPS_TOTAL = GPS_MULTI - GPS_MINUS + 1, GPS_MULTI - GPS_MINUS + 2, GPS_MULTI - GPS_MINUS + 6


def fits32(v):
    return -(1 << 31) <= v < (1 << 31)


class GpsTime11Writer:
    size, type = 8, 7

    def __init__(self, enc, first):
        self.enc, self.last, self.next = enc, 0, 0
        self.t = [struct.unpack("<q", first)[0], 0, 0, 0]
        self.diff, self.extreme = [0] * 4, [0] * 4
        self.m_multi, self.m_0diff, self.ic = SymbolModel(GPS_TOTAL), SymbolModel(6), IntegerCompressor(enc, 32, 9)

    def _full(self, this, model, code):
        self.enc.symbol(model, code)
        self.ic.compress(i32((self.t[self.last] & ((1 << 64) - 1)) >> 32), i32((this & ((1 << 64) - 1)) >> 32), 8)
        self.enc.int32(this & M32)
        self.next = (self.next + 1) & 3
        self.last = self.next
        self.diff[self.last] = self.extreme[self.last] = 0

    def _bump(self, d):
        self.extreme[self.last] += 1
        if self.extreme[self.last] > 3:
            self.diff[self.last], self.extreme[self.last] = d, 0

    def write(self, rec):
        this = struct.unpack("<q", rec)[0]
        while True:
            last = self.last
            if self.diff[last] == 0:
                if this == self.t[last]:
                    self.enc.symbol(self.m_0diff, 0)
                    return
                d64 = this - self.t[last]
                if fits32(d64):
                    self.enc.symbol(self.m_0diff, 1)
                    self.ic.compress(0, d64, 0)
                    self.diff[last], self.extreme[last] = d64, 0
                else:
                    other = next((i for i in (1, 2, 3) if fits32(this - self.t[(last + i) & 3])), None)
                    if other is not None:
                        self.enc.symbol(self.m_0diff, other + 2)
                        self.last = (last + other) & 3
                        continue
                    self._full(this, self.m_0diff, 2)
                self.t[self.last] = this
                return
            if this == self.t[last]:
                self.enc.symbol(self.m_multi, GPS_UNCHANGED)
                return
            d64 = this - self.t[last]
            if fits32(d64):
                f = np.float32(d64) / np.float32(self.diff[last])
                multi = int(f + np.float32(0.5)) if f >= 0 else int(f - np.float32(0.5))
                ld = self.diff[last]
                if multi == 1:
                    self.enc.symbol(self.m_multi, 1)
                    self.ic.compress(ld, d64, 1)
                    self.extreme[last] = 0
                elif multi > 0:
                    if multi < GPS_MULTI:
                        self.enc.symbol(self.m_multi, multi)
                        self.ic.compress(i32(multi * ld), d64, 2 if multi < 10 else 3)
                    else:
                        self.enc.symbol(self.m_multi, GPS_MULTI)
                        self.ic.compress(i32(GPS_MULTI * ld), d64, 4)
                        self._bump(d64)
                elif multi < 0:
                    if multi > GPS_MINUS:
                        self.enc.symbol(self.m_multi, GPS_MULTI - multi)
                        self.ic.compress(i32(multi * ld), d64, 5)
                    else:
                        self.enc.symbol(self.m_multi, GPS_MULTI - GPS_MINUS)
                        self.ic.compress(i32(GPS_MINUS * ld), d64, 6)
                        self._bump(d64)
                else:
                    self.enc.symbol(self.m_multi, 0)
                    self.ic.compress(0, d64, 7)
                    self._bump(d64)
            else:
                other = next((i for i in (1, 2, 3) if fits32(this - self.t[(last + i) & 3])), None)
                if other is not None:
                    self.enc.symbol(self.m_multi, GPS_FULL + other)
                    self.last = (last + other) & 3
                    continue
                self._full(this, self.m_multi, GPS_FULL)
            self.t[self.last] = this
            return


def clamp8(v):
    return 0 if v <= 0 else (255 if v >= 255 else v)


def half(v):  # C integer division by two (towards zero)
    return int(v / 2)


class Rgb12Writer:
    size, type = 6, 8

    def __init__(self, enc, first):
        self.enc, self.last = enc, list(struct.unpack("<HHH", first))
        self.m_used, self.m_diff = SymbolModel(128), [SymbolModel(256) for _ in range(6)]

    def write(self, rec):
        it, last, enc = struct.unpack("<HHH", rec), self.last, self.enc
        lo = [v & 255 for v in it]
        hi = [v >> 8 for v in it]
        llo = [v & 255 for v in last]
        lhi = [v >> 8 for v in last]
        sym = (llo[0] != lo[0]) | ((lhi[0] != hi[0]) << 1) | ((llo[1] != lo[1]) << 2) | ((lhi[1] != hi[1]) << 3) | ((llo[2] != lo[2]) << 4) | \
              ((lhi[2] != hi[2]) << 5) | ((lo[0] != lo[1] or lo[0] != lo[2] or hi[0] != hi[1] or hi[0] != hi[2]) << 6)
        enc.symbol(self.m_used, sym)
        diff_l = diff_h = 0
        if sym & 1:
            diff_l = lo[0] - llo[0]
            enc.symbol(self.m_diff[0], diff_l & 255)
        if sym & 2:
            diff_h = hi[0] - lhi[0]
            enc.symbol(self.m_diff[1], diff_h & 255)
        if sym & 64:
            if sym & 4:
                enc.symbol(self.m_diff[2], (lo[1] - clamp8(diff_l + llo[1])) & 255)
            if sym & 16:
                diff_l = half(diff_l + lo[1] - llo[1])
                enc.symbol(self.m_diff[4], (lo[2] - clamp8(diff_l + llo[2])) & 255)
            if sym & 8:
                enc.symbol(self.m_diff[3], (hi[1] - clamp8(diff_h + lhi[1])) & 255)
            if sym & 32:
                diff_h = half(diff_h + hi[1] - lhi[1])
                enc.symbol(self.m_diff[5], (hi[2] - clamp8(diff_h + lhi[2])) & 255)
        self.last = list(it)


class BytesWriter:
    type = 0

    def __init__(self, enc, first):
        self.enc, self.last, self.size = enc, list(first), len(first)
        self.m = [SymbolModel(256) for _ in first]

    def write(self, rec):
        for i, b in enumerate(rec):
            self.enc.symbol(self.m[i], (b - self.last[i]) & 255)
            self.last[i] = b


def items_of(point_format, record_length):
    """(type, size, version) of the items of a LAS 1.0-1.3 point format, extra bytes as one BYTE item"""
    base = {0: [(6, 20)], 1: [(6, 20), (7, 8)], 2: [(6, 20), (8, 6)], 3: [(6, 20), (7, 8), (8, 6)]}[point_format]
    extra = record_length - sum(s for _, s in base)
    assert extra >= 0
    return [(t, s, 2) for t, s in base] + ([(0, extra, 2)] if extra else [])


def compress_chunk(records, items):
    """records: (n, record_length) uint8 -> the chunk's bytes (first record raw, the rest range-coded)"""
    rec0 = records[0].tobytes()
    if len(records) == 1:
        return rec0
    enc, writers, o = Encoder(), [], 0
    for t, s, _ in items:
        cls = {6: Point10Writer, 7: GpsTime11Writer, 8: Rgb12Writer, 0: BytesWriter}[t]
        writers.append((cls(enc, rec0[o : o + s]), o, s))
        o += s
    for r in records[1:]:
        b = r.tobytes()
        for w, o, s in writers:
            w.write(b[o : o + s])
    return rec0 + enc.done()


def chunk_table(byte_counts, point_counts=None):
    enc = Encoder()
    ic = IntegerCompressor(enc, 32, 2)
    pb = pp = 0
    for i, b in enumerate(byte_counts):
        if point_counts is not None:
            ic.compress(pp, point_counts[i], 0)
            pp = point_counts[i]
        ic.compress(pb, b, 1)
        pb = b
    return struct.pack("<II", 0, len(byte_counts)) + enc.done()


def laz_bytes(las_image, chunk_size=50_000, variable=None, table_at_end_marker=False):
    """an uncompressed LAS image (treee_amd.synth.las_bytes layout: no VLRs) -> the LAZ image of the same records.
    `variable`: list of chunk point counts (chunk_size becomes 0xFFFFFFFF in the LAZ record).  `table_at_end_marker`: the
    chunk table offset slot holds -1 and the file's last 8 bytes the real offset (a writer that could not seek back)."""
    header_size, offset, n_vlr = struct.unpack_from("<HII", las_image, 94)
    fmt = las_image[104]
    (rl,) = struct.unpack_from("<H", las_image, 105)
    (n,) = struct.unpack_from("<Q", las_image, 247)
    (legacy,) = struct.unpack_from("<I", las_image, 107)
    n = legacy or n
    assert n_vlr == 0
    rec = np.frombuffer(las_image, np.uint8, n * rl, offset).reshape(n, rl)
    items = items_of(fmt & 0x3F, rl)
    counts = list(variable) if variable is not None else [min(chunk_size, n - s) for s in range(0, n, chunk_size)]
    assert sum(counts) == n
    chunks, s = [], 0
    for c in counts:
        chunks.append(compress_chunk(rec[s : s + c], items) if c else b"")
        s += c
    payload = struct.pack("<HHBBHIIqqH", 2, 0, 3, 4, 3, 0, 0xFFFFFFFF if variable is not None else chunk_size, -1, -1, len(items))
    payload += b"".join(struct.pack("<HHH", t, sz, v) for t, sz, v in items)
    vlr = bytearray(54)
    vlr[2:16] = b"laszip encoded"
    struct.pack_into("<HH", vlr, 18, 22204, len(payload))
    head = bytearray(las_image[:header_size])
    new_offset = header_size + len(vlr) + len(payload)
    struct.pack_into("<I", head, 96, new_offset)
    struct.pack_into("<I", head, 100, 1)
    head[104] = fmt | 0x80  # LASzip marks the point format
    body = b"".join(chunks)
    table_pos = new_offset + 8 + len(body)
    table = chunk_table([len(c) for c in chunks], counts if variable is not None else None)
    if table_at_end_marker:
        return bytes(head) + bytes(vlr) + payload + struct.pack("<q", -1) + body + table + struct.pack("<q", table_pos)
    return bytes(head) + bytes(vlr) + payload + struct.pack("<q", table_pos) + body + table
