"""Mirror of the ingest seam: treee/src/laz.rs (`Laz::new`, `Laz::read`, `Chunk::read`), and the y-slice binning of
treee/src/loading.rs:91-115.

LAZ sources: the reference hands decompression to the third-party crate `laz` 0.9.2 (not in the snapshot).  The library
restates the published LASzip format for LAS 1.0-1.3 files (pointwise chunked, items POINT10 / GPSTIME11 / RGB12 / BYTE,
version 2: treee_amd/csrc/laz_codec.cpp, host threads like the reference's rayon tasks); layered LAS 1.4 items raise
NotImplementedError.  Parity unpinned: no LAZ file or tool exists in the build container."""
import struct

import numpy as np

from . import _lib
from .context import default_context, dev_f32, host_f32, is_device, ptr

DEFAULT_CHUNK_SIZE = 50_000  # laz.rs:98
HEADER_BYTES = 375  # the packed struct read regardless of version (laz.rs:228-277)


class CorruptFile(Exception):  # Error::CorruptFile
    pass


class Header:
    """laz.rs:228-287 (fixed 375-byte LAS-1.4-shaped layout)."""

    def __init__(self, data):
        if len(data) < HEADER_BYTES:
            raise CorruptFile("short header")
        b = bytes(data[:HEADER_BYTES])
        if b[0:4] != b"LASF":
            raise CorruptFile("signature")
        self.version_major, self.version_minor = b[24], b[25]
        self.header_size, self.offset_to_point_data, self.number_of_variable_length_records = struct.unpack_from("<HII", b, 94)
        self.point_data_record_format = b[104]
        (self.point_data_record_length,) = struct.unpack_from("<H", b, 105)
        (legacy,) = struct.unpack_from("<I", b, 107)
        self.scale = np.array(struct.unpack_from("<3d", b, 131))
        self.offset = np.array(struct.unpack_from("<3d", b, 155))
        max_x, min_x, max_y, min_y, max_z, min_z = struct.unpack_from("<6d", b, 179)
        (self.number_of_point_records,) = struct.unpack_from("<Q", b, 247)
        if legacy != 0:  # laz.rs:279-281
            self.number_of_point_records = legacy
        self.min = np.array([min_x, min_z, -max_y])  # laz.rs:72
        self.max = np.array([max_x, max_z, -min_y])  # laz.rs:73


class Chunk:
    """laz.rs:184-213: raw record bytes + the decode parameters."""

    def __init__(self, data, point_length, offset, scale, center):
        self.slice, self.point_length = data, point_length
        self.offset, self.scale, self.center = (np.ascontiguousarray(a, dtype=np.float64) for a in (offset, scale, center))

    def read(self, ctx=None):
        """Chunk::read -> Vec<Point3<f32>> (numpy in, numpy out; torch uint8 CUDA in, torch out)."""
        if is_device(self.slice):
            import torch

            n = self.slice.numel() // self.point_length
            c = ctx or default_context(self.slice.device)
            out = torch.empty((n, 3), dtype=torch.float32, device=self.slice.device)
            _lib.check(
                _lib.lib().treee_decode_las_records_dev(
                    c.handle, ptr(self.slice), n, self.point_length, ptr(self.scale), ptr(self.offset), ptr(self.center), ptr(out)
                )
            )
            return out
        rec = np.frombuffer(self.slice, dtype=np.uint8) if not isinstance(self.slice, np.ndarray) else self.slice
        rec = np.ascontiguousarray(rec)
        n = len(rec) // self.point_length
        out = np.zeros((n, 3), np.float32)
        _lib.check(
            _lib.lib().treee_decode_las_records(ptr(rec), n, self.point_length, ptr(self.scale), ptr(self.offset), ptr(self.center), ptr(out))
        )
        return out


class Laz:
    """Laz::new(source, center) over an in-memory .las image (bytes)."""

    def __init__(self, data, center=None):
        self.data = bytes(data)
        h = self.header = Header(self.data)
        self.point_length = h.point_data_record_length
        self.scale, self.offset = h.scale, h.offset
        self.center = (h.min + h.max) * 0.5 if center is None else np.asarray(center, dtype=np.float64)  # na::center
        total = h.number_of_point_records
        self.vlr = self._laszip_vlr()
        self._records = None  # decompressed records of a LAZ source (filled by the first read)
        if self.vlr is not None:  # laz.rs:80-94
            self._laz_chunks(total)
            self.min = (h.min - self.center).astype(np.float32)
            self.max = (h.max - self.center).astype(np.float32)
            self.world_offset = self.center
            return
        need = h.offset_to_point_data + total * self.point_length
        if len(self.data) < need:  # the reference's read_exact fails on a short file (laz.rs:146-170)
            raise CorruptFile(f"file holds {len(self.data)} bytes, header promises {need}")
        self.chunks = []  # laz.rs:96-107
        start = h.offset_to_point_data
        for _ in range(total // DEFAULT_CHUNK_SIZE):
            self.chunks.append((start, DEFAULT_CHUNK_SIZE))
            start += DEFAULT_CHUNK_SIZE * self.point_length
        if total % DEFAULT_CHUNK_SIZE:
            self.chunks.append((start, total % DEFAULT_CHUNK_SIZE))
        self.min = (h.min - self.center).astype(np.float32)
        self.max = (h.max - self.center).astype(np.float32)
        self.world_offset = self.center

    def _laszip_vlr(self):
        """laz.rs:77-79: the file is LAZ iff one of its variable length records is the laszip record
        (`laz` 0.9.2 `read_vlrs_and_get_laszip_vlr`: user id "laszip encoded", record id 22204); the scan starts
        at header_size and walks number_of_variable_length_records 54-byte VLR headers.  Returns the record's fields
        (compressor, coder, chunk_size, items [(type, size, version)]) or None."""
        h, pos = self.header, self.header.header_size
        for _ in range(h.number_of_variable_length_records):
            if pos + 54 > len(self.data):
                return None
            user_id = self.data[pos + 2 : pos + 18].split(b"\0", 1)[0]
            record_id, length = struct.unpack_from("<HH", self.data, pos + 18)
            if user_id == b"laszip encoded" and record_id == 22204:
                body = self.data[pos + 54 : pos + 54 + length]
                if len(body) < 34:
                    raise CorruptFile("laszip record shorter than its fixed part")
                compressor, coder, _, _, _, _, chunk_size, _, _, n_items = struct.unpack_from("<HHBBHIIqqH", body, 0)
                if len(body) < 34 + 6 * n_items:
                    raise CorruptFile("laszip record shorter than its item list")
                items = [struct.unpack_from("<HHH", body, 34 + 6 * i) for i in range(n_items)]
                return {"compressor": compressor, "coder": coder, "chunk_size": chunk_size, "items": items}
            pos += 54 + length
        return None

    def _laz_chunks(self, total):
        """laz.rs:80-94: chunk table -> (start, point count) per chunk, the counts clipped to the header's total"""
        import ctypes as C

        v, h = self.vlr, self.header
        if v["compressor"] != 2 or v["coder"] != 0:
            raise NotImplementedError(f"LASzip compressor {v['compressor']} / coder {v['coder']}: only the pointwise chunked arithmetic form is built")
        if any(ver != 2 for _, _, ver in v["items"]) or any(t not in (0, 6, 7, 8) for t, _, _ in v["items"]):
            raise NotImplementedError(f"LASzip items {v['items']}: only POINT10, GPSTIME11, RGB12 and BYTE of version 2 are built")
        if sum(s for _, s, _ in v["items"]) != self.point_length:
            raise CorruptFile("the laszip items do not add up to the point record length")
        buf = np.frombuffer(self.data, np.uint8)
        nc = C.c_uint64(0)
        _lib.check(_lib.lib().treee_laz_chunk_table(ptr(buf), len(buf), h.offset_to_point_data, v["chunk_size"], total, None, None, 0, C.byref(nc)))
        nb, npnt = np.zeros(nc.value, np.uint64), np.zeros(nc.value, np.uint64)
        if nc.value:
            _lib.check(_lib.lib().treee_laz_chunk_table(ptr(buf), len(buf), h.offset_to_point_data, v["chunk_size"], total, ptr(nb), ptr(npnt), nc.value, C.byref(nc)))
        self.chunks, self._chunk_bytes = [], []
        # Where the reference stands: laz.rs:76-80 seeks to header_size, scans the VLRs and calls ChunkTable::read_from AT THE
        # POSITION THE SCAN ENDED, i.e. header_size + the VLRs' bytes; the chunks follow 8 bytes later.  A well-formed file has
        # offset_to_point_data there.  DEVIATION (documented): this reader takes the header's offset_to_point_data --
        # a file with padding between its last VLR and the point data is read here and fails in the reference.
        start = h.offset_to_point_data + 8  # the reader's position after ChunkTable::read_from
        for b, c in zip(nb.tolist(), npnt.tolist()):
            self.chunks.append((start, c))
            self._chunk_bytes.append(b)
            start += b

    def _decompress(self, threads=0):
        v = self.vlr
        items = np.asarray(v["items"], dtype=np.uint16).reshape(-1, 3)
        types, sizes, versions = (np.ascontiguousarray(items[:, i]) for i in range(3))
        starts = np.asarray([s for s, _ in self.chunks], dtype=np.uint64)
        counts = np.asarray([c for _, c in self.chunks], dtype=np.uint64)
        nbytes = np.asarray(self._chunk_bytes, dtype=np.uint64)
        out = np.zeros(int(counts.sum()) * self.point_length, np.uint8)
        buf = np.frombuffer(self.data, np.uint8)
        _lib.check(_lib.lib().treee_laz_decompress_chunks(ptr(buf), len(buf), ptr(types), ptr(sizes), ptr(versions), len(items), ptr(starts), ptr(nbytes),
                                                          ptr(counts), len(self.chunks), threads, ptr(out), len(out)))
        self._records = out

    def total(self):
        return len(self.chunks)

    def read(self, cb, threads=0):
        """Laz::read(cb): hands one Chunk per (start, length) to the callback.  A LAZ source is decompressed first, all
        chunks at once on `threads` host threads (0 = all) -- the reference's rayon tasks (laz.rs:134-160)."""
        if self.vlr is not None:
            if self._records is None:
                self._decompress(threads)
            o = 0
            for _, l in self.chunks:
                cb(Chunk(self._records[o : o + l * self.point_length], self.point_length, self.offset, self.scale, self.center))
                o += l * self.point_length
            return
        for s, l in self.chunks:
            cb(Chunk(self.data[s : s + l * self.point_length], self.point_length, self.offset, self.scale, self.center))


def slice_keys(points, ctx=None):
    """loading.rs:99: idx = p.y.floor() as isize."""
    if is_device(points):
        import torch

        p = dev_f32(points, 3)
        c = ctx or default_context(p.device)
        out = torch.empty((p.shape[0],), dtype=torch.int64, device=p.device)
        _lib.check(_lib.lib().treee_slice_keys_dev(c.handle, ptr(p), p.shape[0], ptr(out)))
        return out
    p = host_f32(points, 3)
    out = np.zeros(len(p), np.int64)
    _lib.check(_lib.lib().treee_slice_keys(ptr(p), len(p), ptr(out)))
    return out


def slice_bin(points, ctx=None):
    """Shared.slices (loading.rs:28,97-101) densified like segmenting.rs:121-133.
    Returns (min_key, order[n] u32, offsets[layers+1] u64); device tensors for a device cloud."""
    import ctypes as C

    if is_device(points):
        import torch

        p = dev_f32(points, 3)
        c = ctx or default_context(p.device)
        n = p.shape[0]
        mn, layers = C.c_int64(0), C.c_uint64(0)
        order = torch.empty((n,), dtype=torch.int32, device=p.device)
        offsets = torch.empty((4097,), dtype=torch.int64, device=p.device)  # one call unless the cloud has > 4096 layers
        if n:
            rc = _lib.lib().treee_slice_bin_dev(c.handle, ptr(p), n, C.byref(mn), C.byref(layers), ptr(order), ptr(offsets), offsets.shape[0])
            if rc != 0 and layers.value + 1 > offsets.shape[0]:
                offsets = torch.empty((layers.value + 1,), dtype=torch.int64, device=p.device)
                rc = _lib.lib().treee_slice_bin_dev(c.handle, ptr(p), n, C.byref(mn), C.byref(layers), ptr(order), ptr(offsets), offsets.shape[0])
            _lib.check(rc)
        offsets = offsets[: layers.value + 1 if n else 0]
        return int(mn.value), order, offsets
    p = host_f32(points, 3)
    n = len(p)
    mn, layers = C.c_int64(0), C.c_uint64(0)
    _lib.check(_lib.lib().treee_slice_bin(ptr(p), n, C.byref(mn), C.byref(layers), None, None, 0))
    order = np.zeros(n, np.uint32)
    offsets = np.zeros(layers.value + 1 if n else 0, np.uint64)
    if n:
        _lib.check(_lib.lib().treee_slice_bin(ptr(p), n, C.byref(mn), C.byref(layers), ptr(order), ptr(offsets), len(offsets)))
    return int(mn.value), order, offsets
