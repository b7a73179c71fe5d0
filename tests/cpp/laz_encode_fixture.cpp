// laz_encode_fixture.cpp -- TEST INFRASTRUCTURE, not part of the product: the LASzip COMPRESSOR (pointwise chunked, item version 2:
// POINT10 / GPSTIME11 / RGB12 / BYTE).  The reference never writes LAZ (treee/src/laz.rs only reads), so nothing of this has a
// place in libtreee_hip.so (rounds 2-4 exported it as treee_laz_encode): it writes the fixtures of
// tests/test_laz.py and the LAZ tiles bench.py streams.  The mirror image of the library's decoder (treee_amd/csrc/laz_codec.cpp),
// model for model, over the shared model definitions of treee_amd/csrc/laz_models.h; pinned by the independent Python encoder
// oracle/laz_encoder.py, whose files it writes byte for byte (tests/test_laz.py).  Built by __graft_entry__.build_laz_fixture().
#include "../../treee_amd/csrc/laz_models.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <memory>
#include <thread>

using namespace treee_laz;

namespace {

enum { FX_OK = 0, FX_INVALID = 1, FX_UNSUPPORTED = 4, FX_NOMEM = 5 }; // (the library's codes, include/treee_hip.h)
thread_local char g_err[512] = "";
int32_t fx_err(int32_t code, const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}
unsigned fx_threads() {
	unsigned n = std::thread::hardware_concurrency();
	return n == 0 ? 1u : (n > 32u ? 32u : n);
}

// ---- the compressor (fixtures and benchmark input: the reference only ever reads LAZ) -------------------------------------
// The mirror image of the decoder above, model for model; kept beside it so that every model rule exists once per direction.

struct Encoder {
	std::vector<uint8_t> out;
	uint32_t base = 0, length = AC_MAX_LENGTH;
	void carry() {
		size_t i = out.size() - 1;
		while (out[i] == 0xFFu) out[i--] = 0;
		++out[i];
	}
	void advance(uint32_t add) {
		const uint32_t init = base;
		base += add;
		if (init > base) carry();
	}
	void renorm() {
		do {
			out.push_back((uint8_t)(base >> 24));
			base <<= 8;
		} while ((length <<= 8) < AC_MIN_LENGTH);
	}
	void bit(BitModel &m, uint32_t sym) {
		const uint32_t x = m.bit_0_prob * (length >> BM_LENGTH_SHIFT);
		if (sym == 0) {
			length = x;
			++m.bit_0_count;
		} else {
			advance(x);
			length -= x;
		}
		if (length < AC_MIN_LENGTH) renorm();
		if (--m.bits_until_update == 0) m.update();
	}
	void symbol(SymbolModel &m, uint32_t sym) {
		if (sym == m.last_symbol) {
			const uint32_t x = m.distribution[sym] * (length >> DM_LENGTH_SHIFT);
			advance(x);
			length -= x;
		} else {
			const uint32_t x = m.distribution[sym] * (length >>= DM_LENGTH_SHIFT);
			advance(x);
			length = m.distribution[sym + 1] * length - x;
		}
		if (length < AC_MIN_LENGTH) renorm();
		++m.symbol_count[sym];
		if (--m.symbols_until_update == 0) m.update();
	}
	void write_short(uint32_t sym) {
		advance(sym * (length >>= 16));
		if (length < AC_MIN_LENGTH) renorm();
	}
	void write_bits(uint32_t bits, uint32_t sym) {
		if (bits > 19) {
			write_short(sym & 0xFFFFu);
			sym >>= 16;
			bits -= 16;
		}
		advance(sym * (length >>= bits));
		if (length < AC_MIN_LENGTH) renorm();
	}
	void write_int(uint32_t sym) {
		write_short(sym & 0xFFFFu);
		write_short(sym >> 16);
	}
	void done() {
		bool another = true;
		if (length > 2 * AC_MIN_LENGTH) {
			advance(AC_MIN_LENGTH);
			length = AC_MIN_LENGTH >> 1;
		} else {
			advance(AC_MIN_LENGTH >> 1);
			length = AC_MIN_LENGTH >> 9;
			another = false;
		}
		renorm();
		out.push_back(0); // what the decoder reads ahead
		out.push_back(0);
		if (another) out.push_back(0);
	}
};

struct IntegerCompressor {
	IntegerModels m; // the same models, driven from the other side
	int32_t corr_max;
	uint32_t k = 0;
	IntegerCompressor(uint32_t bits, uint32_t contexts) : m(bits, contexts) { corr_max = m.corr_range ? (int32_t)(m.corr_min + (int32_t)m.corr_range - 1) : INT32_MAX; }
	void compress(Encoder &e, int32_t pred, int32_t real, uint32_t context = 0) {
		int32_t corr = (int32_t)((uint32_t)real - (uint32_t)pred);
		if (corr < m.corr_min)
			corr = (int32_t)((uint32_t)corr + m.corr_range);
		else if (corr > corr_max)
			corr = (int32_t)((uint32_t)corr - m.corr_range);
		uint32_t c1 = corr <= 0 ? (uint32_t)0 - (uint32_t)corr : (uint32_t)corr - 1u;
		k = 0;
		while (c1) {
			c1 >>= 1;
			++k;
		}
		e.symbol(m.m_bits[context], k);
		if (k == 0) {
			e.bit(m.m_corrector0, (uint32_t)corr);
		} else if (k < 32) {
			uint32_t c = corr < 0 ? (uint32_t)corr + ((1u << k) - 1u) : (uint32_t)corr - 1u;
			if (k <= m.bits_high) {
				e.symbol(m.m_corrector[k], c);
			} else {
				const uint32_t k1 = k - m.bits_high;
				e.symbol(m.m_corrector[k], c >> k1);
				e.write_bits(k1, c & ((1u << k1) - 1u));
			}
		}
	}
};

struct ItemWriter {
	virtual ~ItemWriter() = default;
	virtual void write(Encoder &e, const uint8_t *rec) = 0;
};

struct Point10Writer : ItemWriter {
	uint8_t last[20];
	uint16_t last_intensity[16];
	StreamingMedian5 last_x_diff[16], last_y_diff[16];
	int32_t last_height[8];
	SymbolModel m_changed{64}, m_scan_angle[2] = {SymbolModel(256), SymbolModel(256)};
	std::vector<SymbolModel> m_bit_byte = std::vector<SymbolModel>(256), m_classification = std::vector<SymbolModel>(256),
	                         m_user_data = std::vector<SymbolModel>(256);
	IntegerCompressor ic_intensity{16, 4}, ic_point_source{16, 1}, ic_dx{32, 2}, ic_dy{32, 22}, ic_z{32, 20};
	explicit Point10Writer(const uint8_t *first) {
		std::memset(last_intensity, 0, sizeof(last_intensity));
		std::memset(last_height, 0, sizeof(last_height));
		std::memcpy(last, first, 20);
		last[12] = last[13] = 0;
	}
	static SymbolModel &lazy(std::vector<SymbolModel> &v, uint32_t i) {
		if (!v[i].live()) v[i].init(256);
		return v[i];
	}
	void write(Encoder &e, const uint8_t *it) override {
		const uint32_t r = it[14] & 7u, n = (it[14] >> 3) & 7u, m = RETURN_MAP[n][r], l = RETURN_LEVEL[n][r];
		const uint16_t intensity = rd_u16(it + 12);
		const uint32_t changed = ((uint32_t)(last[14] != it[14]) << 5) | ((uint32_t)(last_intensity[m] != intensity) << 4) | ((uint32_t)(last[15] != it[15]) << 3) |
		                         ((uint32_t)(last[16] != it[16]) << 2) | ((uint32_t)(last[17] != it[17]) << 1) | (uint32_t)(rd_u16(last + 18) != rd_u16(it + 18));
		e.symbol(m_changed, changed);
		if (changed & 32) e.symbol(lazy(m_bit_byte, last[14]), it[14]);
		if (changed & 16) {
			ic_intensity.compress(e, last_intensity[m], intensity, m < 3 ? m : 3);
			last_intensity[m] = intensity;
		}
		if (changed & 8) e.symbol(lazy(m_classification, last[15]), it[15]);
		if (changed & 4) e.symbol(m_scan_angle[(it[14] >> 6) & 1u], u8_fold((int)it[16] - (int)last[16]));
		if (changed & 2) e.symbol(lazy(m_user_data, last[17]), it[17]);
		if (changed & 1) ic_point_source.compress(e, rd_u16(last + 18), rd_u16(it + 18));
		int32_t diff = (int32_t)((uint32_t)rd_i32(it + 0) - (uint32_t)rd_i32(last + 0));
		ic_dx.compress(e, last_x_diff[m].get(), diff, n == 1);
		last_x_diff[m].add(diff);
		uint32_t k_bits = ic_dx.k;
		diff = (int32_t)((uint32_t)rd_i32(it + 4) - (uint32_t)rd_i32(last + 4));
		ic_dy.compress(e, last_y_diff[m].get(), diff, (n == 1) + (k_bits < 20 ? (k_bits & ~1u) : 20u));
		last_y_diff[m].add(diff);
		k_bits = (ic_dx.k + ic_dy.k) / 2;
		ic_z.compress(e, last_height[l], rd_i32(it + 8), (n == 1) + (k_bits < 18 ? (k_bits & ~1u) : 18u));
		last_height[l] = rd_i32(it + 8);
		std::memcpy(last, it, 20);
	}
};

struct GpsTime11Writer : ItemWriter {
	uint32_t last = 0, next = 0;
	int64_t t[4] = {0, 0, 0, 0};
	int32_t diff[4] = {0, 0, 0, 0}, extreme[4] = {0, 0, 0, 0};
	SymbolModel m_multi{(uint32_t)GPS_MULTI_TOTAL}, m_0diff{6};
	IntegerCompressor ic{32, 9};
	explicit GpsTime11Writer(const uint8_t *first) { std::memcpy(&t[0], first, 8); }
	static bool fits32(int64_t v) { return v == (int64_t)(int32_t)v; }
	void full(Encoder &e, int64_t now, SymbolModel &model, uint32_t code) {
		e.symbol(model, code);
		ic.compress(e, (int32_t)((uint64_t)t[last] >> 32), (int32_t)((uint64_t)now >> 32), 8);
		e.write_int((uint32_t)(uint64_t)now);
		next = (next + 1) & 3;
		last = next;
		diff[last] = 0;
		extreme[last] = 0;
	}
	void bump(int32_t d) {
		if (++extreme[last] > 3) {
			diff[last] = d;
			extreme[last] = 0;
		}
	}
	int other_sequence(int64_t now) const {
		for (int i = 1; i < 4; ++i)
			if (fits32((int64_t)((uint64_t)now - (uint64_t)t[(last + i) & 3]))) return i;
		return 0;
	}
	void write(Encoder &e, const uint8_t *rec) override {
		int64_t now;
		std::memcpy(&now, rec, 8);
		for (;;) {
			const int64_t d64 = (int64_t)((uint64_t)now - (uint64_t)t[last]);
			if (diff[last] == 0) {
				if (now == t[last]) {
					e.symbol(m_0diff, 0);
					return;
				}
				if (fits32(d64)) {
					e.symbol(m_0diff, 1);
					ic.compress(e, 0, (int32_t)d64, 0);
					diff[last] = (int32_t)d64;
					extreme[last] = 0;
				} else if (const int o = other_sequence(now)) {
					e.symbol(m_0diff, (uint32_t)o + 2);
					last = (last + (uint32_t)o) & 3;
					continue;
				} else {
					full(e, now, m_0diff, 2);
				}
				t[last] = now;
				return;
			}
			if (now == t[last]) {
				e.symbol(m_multi, (uint32_t)GPS_MULTI_UNCHANGED);
				return;
			}
			if (fits32(d64)) {
				const int32_t d = (int32_t)d64, ld = diff[last];
				const float f = (float)d / (float)ld;
				const int32_t multi = f >= 0 ? (int32_t)(f + 0.5f) : (int32_t)(f - 0.5f);
				if (multi == 1) {
					e.symbol(m_multi, 1);
					ic.compress(e, ld, d, 1);
					extreme[last] = 0;
				} else if (multi > 0) {
					if (multi < GPS_MULTI) {
						e.symbol(m_multi, (uint32_t)multi);
						ic.compress(e, (int32_t)((uint32_t)multi * (uint32_t)ld), d, multi < 10 ? 2 : 3);
					} else {
						e.symbol(m_multi, (uint32_t)GPS_MULTI);
						ic.compress(e, (int32_t)((uint32_t)GPS_MULTI * (uint32_t)ld), d, 4);
						bump(d);
					}
				} else if (multi < 0) {
					if (multi > GPS_MULTI_MINUS) {
						e.symbol(m_multi, (uint32_t)(GPS_MULTI - multi));
						ic.compress(e, (int32_t)((uint32_t)multi * (uint32_t)ld), d, 5);
					} else {
						e.symbol(m_multi, (uint32_t)(GPS_MULTI - GPS_MULTI_MINUS));
						ic.compress(e, (int32_t)((uint32_t)GPS_MULTI_MINUS * (uint32_t)ld), d, 6);
						bump(d);
					}
				} else {
					e.symbol(m_multi, 0);
					ic.compress(e, 0, d, 7);
					bump(d);
				}
			} else if (const int o = other_sequence(now)) {
				e.symbol(m_multi, (uint32_t)(GPS_MULTI_CODE_FULL + o));
				last = (last + (uint32_t)o) & 3;
				continue;
			} else {
				full(e, now, m_multi, (uint32_t)GPS_MULTI_CODE_FULL);
			}
			t[last] = now;
			return;
		}
	}
};

struct Rgb12Writer : ItemWriter {
	uint16_t last[3];
	SymbolModel m_used{128}, m_diff[6] = {SymbolModel(256), SymbolModel(256), SymbolModel(256), SymbolModel(256), SymbolModel(256), SymbolModel(256)};
	explicit Rgb12Writer(const uint8_t *first) {
		for (int i = 0; i < 3; ++i) last[i] = rd_u16(first + 2 * i);
	}
	void write(Encoder &e, const uint8_t *rec) override {
		int lo[3], hi[3], llo[3], lhi[3];
		for (int i = 0; i < 3; ++i) {
			const uint16_t v = rd_u16(rec + 2 * i);
			lo[i] = v & 255;
			hi[i] = v >> 8;
			llo[i] = last[i] & 255;
			lhi[i] = last[i] >> 8;
		}
		const uint32_t sym = (uint32_t)(llo[0] != lo[0]) | ((uint32_t)(lhi[0] != hi[0]) << 1) | ((uint32_t)(llo[1] != lo[1]) << 2) | ((uint32_t)(lhi[1] != hi[1]) << 3) |
		                     ((uint32_t)(llo[2] != lo[2]) << 4) | ((uint32_t)(lhi[2] != hi[2]) << 5) |
		                     ((uint32_t)(lo[0] != lo[1] || lo[0] != lo[2] || hi[0] != hi[1] || hi[0] != hi[2]) << 6);
		e.symbol(m_used, sym);
		int diff_l = 0, diff_h = 0;
		if (sym & 1) {
			diff_l = lo[0] - llo[0];
			e.symbol(m_diff[0], u8_fold(diff_l));
		}
		if (sym & 2) {
			diff_h = hi[0] - lhi[0];
			e.symbol(m_diff[1], u8_fold(diff_h));
		}
		if (sym & 64) {
			if (sym & 4) e.symbol(m_diff[2], u8_fold(lo[1] - u8_clamp(diff_l + llo[1])));
			if (sym & 16) {
				diff_l = (diff_l + lo[1] - llo[1]) / 2;
				e.symbol(m_diff[4], u8_fold(lo[2] - u8_clamp(diff_l + llo[2])));
			}
			if (sym & 8) e.symbol(m_diff[3], u8_fold(hi[1] - u8_clamp(diff_h + lhi[1])));
			if (sym & 32) {
				diff_h = (diff_h + hi[1] - lhi[1]) / 2;
				e.symbol(m_diff[5], u8_fold(hi[2] - u8_clamp(diff_h + lhi[2])));
			}
		}
		for (int i = 0; i < 3; ++i) last[i] = rd_u16(rec + 2 * i);
	}
};

struct BytesWriter : ItemWriter {
	std::vector<uint8_t> last;
	std::vector<SymbolModel> m;
	BytesWriter(const uint8_t *first, uint32_t n) : last(first, first + n), m(n) {
		for (auto &x : m) x.init(256);
	}
	void write(Encoder &e, const uint8_t *rec) override {
		for (size_t i = 0; i < last.size(); ++i) {
			e.symbol(m[i], u8_fold((int)rec[i] - (int)last[i]));
			last[i] = rec[i];
		}
	}
};

// one chunk of records -> its bytes (first record raw, the rest range-coded)
void compress_chunk(const uint8_t *rec, uint64_t n, const uint16_t *types, const uint16_t *sizes, uint32_t n_items, uint32_t rl, std::vector<uint8_t> &out) {
	out.assign(rec, rec + (n ? rl : 0));
	if (n < 2) return;
	std::vector<std::unique_ptr<ItemWriter>> w;
	uint32_t o = 0;
	for (uint32_t i = 0; i < n_items; ++i) {
		switch (types[i]) {
		case ITEM_POINT10: w.emplace_back(new Point10Writer(rec + o)); break;
		case ITEM_GPSTIME11: w.emplace_back(new GpsTime11Writer(rec + o)); break;
		case ITEM_RGB12: w.emplace_back(new Rgb12Writer(rec + o)); break;
		default: w.emplace_back(new BytesWriter(rec + o, sizes[i])); break;
		}
		o += sizes[i];
	}
	Encoder e;
	e.out.reserve((size_t)(n * rl / 3));
	for (uint64_t p = 1; p < n; ++p) {
		o = 0;
		for (uint32_t i = 0; i < n_items; ++i) {
			w[i]->write(e, rec + p * rl + o);
			o += sizes[i];
		}
	}
	e.done();
	out.insert(out.end(), e.out.begin(), e.out.end());
}


// An uncompressed LAS image (LAS 1.0-1.3, point formats 0-3, any extra bytes) -> the LAZ image of the same records: the
// laszip record appended to the variable length records, the compression bit set in the point format, chunks of
// `chunk_size` points compressed on `threads` host threads (0 = all), the chunk table behind them.  The reference never
// writes LAZ; this exists for fixtures and for the streamed-LAZ benchmark input.  `out` needs *out_len_out bytes: a call
// whose `cap` is too small fails with FX_INVALID after setting it (random records can grow by a few per cent).
static int32_t laz_encode_impl(const uint8_t *las, uint64_t las_len, uint32_t chunk_size, uint32_t threads, uint8_t *out, uint64_t cap, uint64_t *out_len_out) {
	if (!las || !out_len_out) return fx_err(FX_INVALID, "laz_encode_fixture: null argument");
	*out_len_out = 0;
	if (las_len < 375 || std::memcmp(las, "LASF", 4) != 0) return fx_err(FX_INVALID, "laz_encode_fixture: not a LAS image (laz.rs:274-284)");
	if (chunk_size == 0 || chunk_size == 0xFFFFFFFFu) return fx_err(FX_INVALID, "laz_encode_fixture: chunk size %u", chunk_size);
	uint32_t offset, n_vlr, legacy;
	uint64_t n;
	std::memcpy(&offset, las + 96, 4);
	std::memcpy(&n_vlr, las + 100, 4);
	std::memcpy(&legacy, las + 107, 4);
	std::memcpy(&n, las + 247, 8);
	if (legacy) n = legacy; // laz.rs:279-281
	const uint32_t fmt = las[104] & 0x3Fu, rl = rd_u16(las + 105);
	if (fmt > 3) return fx_err(FX_UNSUPPORTED, "laz_encode_fixture: point format %u (0-3 are built)", fmt);
	uint16_t types[4], sizes[4], versions[4] = {2, 2, 2, 2};
	uint32_t n_items = 0, base = 20;
	types[n_items] = ITEM_POINT10, sizes[n_items++] = 20;
	if (fmt == 1 || fmt == 3) types[n_items] = ITEM_GPSTIME11, sizes[n_items++] = 8, base += 8;
	if (fmt == 2 || fmt == 3) types[n_items] = ITEM_RGB12, sizes[n_items++] = 6, base += 6;
	if (rl < base) return fx_err(FX_INVALID, "laz_encode_fixture: record length %u below the %u bytes of format %u", rl, base, fmt);
	if (rl > base) types[n_items] = ITEM_BYTE, sizes[n_items++] = (uint16_t)(rl - base);
	if (offset > las_len || n > (las_len - offset) / rl) return fx_err(FX_INVALID, "laz_encode_fixture: the header promises more records than the image holds");
	const uint8_t *rec = las + offset;
	const uint64_t n_chunks = (n + chunk_size - 1) / chunk_size;
	std::vector<std::vector<uint8_t>> chunks(n_chunks);
	{
		uint32_t nt = threads ? threads : fx_threads();
		if (nt == 0) nt = 1;
		if ((uint64_t)nt > n_chunks) nt = (uint32_t)(n_chunks ? n_chunks : 1);
		std::atomic<uint64_t> next{0};
		std::atomic<bool> threw{false};
		auto work = [&]() {
			try {
				for (;;) {
					const uint64_t c = next.fetch_add(1);
					if (c >= n_chunks || threw) return;
					const uint64_t first = c * chunk_size, cnt = std::min<uint64_t>(chunk_size, n - first);
					compress_chunk(rec + first * rl, cnt, types, sizes, n_items, rl, chunks[c]);
				}
			} catch (...) {
				threw = true;
			}
		};
		if (nt == 1) {
			work();
		} else {
			std::vector<std::thread> pool;
			try {
				for (uint32_t t = 1; t < nt; ++t) pool.emplace_back(work);
			} catch (...) {
			}
			work();
			for (auto &t : pool) t.join();
		}
		if (threw) return fx_err(FX_NOMEM, "laz_encode_fixture: out of memory");
	}
	Encoder te; // chunk table: version 0, count, byte sizes (context 1, each predicted by the one before)
	if (n_chunks) {
		IntegerCompressor ic(32, 2);
		uint32_t prev = 0;
		for (auto &c : chunks) {
			ic.compress(te, (int32_t)prev, (int32_t)(uint32_t)c.size(), 1);
			prev = (uint32_t)c.size();
		}
		te.done();
	}
	uint64_t body = 0;
	for (auto &c : chunks) body += c.size();
	const uint32_t vlr_len = 54 + 34 + 6 * n_items;
	const uint64_t new_offset = (uint64_t)offset + vlr_len, table_pos = new_offset + 8 + body;
	const uint64_t total = table_pos + 8 + te.out.size();
	*out_len_out = total;
	if (new_offset > 0xFFFFFFFFull) return fx_err(FX_UNSUPPORTED, "laz_encode_fixture: header area beyond 4 GiB");
	if (!out || cap < total) return fx_err(FX_INVALID, "laz_encode_fixture: the image needs %llu bytes, room for %llu", (unsigned long long)total, (unsigned long long)cap);
	std::memcpy(out, las, offset);
	const uint32_t off32 = (uint32_t)new_offset, nv = n_vlr + 1;
	std::memcpy(out + 96, &off32, 4);
	std::memcpy(out + 100, &nv, 4);
	out[104] = (uint8_t)(las[104] | 0x80u); // LASzip marks the point format
	uint8_t *v = out + offset;
	std::memset(v, 0, vlr_len);
	std::memcpy(v + 2, "laszip encoded", 14);
	wr_u16(v + 18, 22204);
	wr_u16(v + 20, (uint16_t)(34 + 6 * n_items));
	uint8_t *b = v + 54; // compressor 2 (pointwise chunked), coder 0, version 3.4.3, options 0, chunk size, no special EVLRs, items
	wr_u16(b + 0, 2);
	wr_u16(b + 2, 0);
	b[4] = 3, b[5] = 4;
	wr_u16(b + 6, 3);
	std::memset(b + 8, 0, 4);
	std::memcpy(b + 12, &chunk_size, 4);
	const int64_t minus1 = -1;
	std::memcpy(b + 16, &minus1, 8);
	std::memcpy(b + 24, &minus1, 8);
	wr_u16(b + 32, (uint16_t)n_items);
	for (uint32_t i = 0; i < n_items; ++i) {
		wr_u16(b + 34 + 6 * i, types[i]);
		wr_u16(b + 36 + 6 * i, sizes[i]);
		wr_u16(b + 38 + 6 * i, versions[i]);
	}
	uint8_t *p = out + new_offset;
	const int64_t tp = (int64_t)table_pos;
	std::memcpy(p, &tp, 8);
	p += 8;
	for (auto &c : chunks) {
		std::memcpy(p, c.data(), c.size());
		p += c.size();
	}
	const uint32_t zero = 0, nc32 = (uint32_t)n_chunks;
	std::memcpy(p, &zero, 4);
	std::memcpy(p + 4, &nc32, 4);
	if (!te.out.empty()) std::memcpy(p + 8, te.out.data(), te.out.size());
	return FX_OK;
}


} // namespace

extern "C" {

int32_t treee_test_laz_encode(const uint8_t *las, uint64_t las_len, uint32_t chunk_size, uint32_t threads, uint8_t *out, uint64_t cap, uint64_t *out_len_out) {
	try {
		return laz_encode_impl(las, las_len, chunk_size, threads, out, cap, out_len_out);
	} catch (const std::exception &e) {
		return fx_err(FX_NOMEM, "laz_encode_fixture: %s", e.what());
	}
}

const char *treee_test_laz_last_error(void) { return g_err; }

} // extern "C"
