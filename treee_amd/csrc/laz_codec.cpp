// laz_codec.cpp -- LASzip point decompression (host code; like segmenting.cpp it is in the library because the reference
// does this work on host threads, treee/src/laz.rs:130-180: one chunk per rayon task, decompress_many, then next_point).
//
// The reference delegates it to the third-party crate `laz` 0.9.2 (Cargo.lock:1393-1394; call sites laz.rs:1-8,77-94,146-160),
// whose source is NOT in the snapshot.  What follows restates the published LASzip format (M. Isenburg, "LASzip: lossless
// compression of LiDAR data", PE&RS 79(2), 2013, and the LGPL reference implementation it describes) for the subset a
// LAS 1.0-1.3 file uses: compressor "pointwise chunked" (2), arithmetic coder (0), items POINT10 / GPSTIME11 / RGB12 / BYTE,
// item version 2.  Layered LAS 1.4 items (version 3/4, point formats 6-10) are not built: TREEE_ERR_UNSUPPORTED.
// PARITY STATUS: UNPINNED.  No LAZ file, tool or library exists in the build container; the decoder is checked against an
// independently written encoder (oracle/laz_encoder.py, test infrastructure) and nothing else.
//
// Structure (the published algorithm):
//   * range decoder: 32-bit `value` / `length`, renormalised bytewise below 2^24; adaptive binary models (13-bit
//     probability, update cycle 4 -> 64) and adaptive multi-symbol models (15-bit cumulative distribution, update cycle
//     (n+6)/2 -> 8(n+6), counts halved past 2^15)
//   * integer decompressor: a symbol k = number of significant bits of the corrector (context-selected model), then the
//     corrector itself from model k (its high 8 bits; lower bits raw), signed folding, wrap to the value range
//   * POINT10: a 6-bit "changed" mask; return-number map / level tables select the contexts; x and y differences are
//     predicted by streaming medians of 5, z by the last height of the return level
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "laz_models.h"

using namespace treee_laz;

namespace {

struct Truncated {}; // thrown when a chunk's bytes run out (caught at the chunk boundary: TREEE_ERR_INVALID)

struct Decoder {
	const uint8_t *p = nullptr, *end = nullptr;
	uint32_t value = 0, length = AC_MAX_LENGTH;
	uint8_t byte() {
		if (p >= end) throw Truncated();
		return *p++;
	}
	void init(const uint8_t *src, const uint8_t *src_end) {
		p = src;
		end = src_end;
		length = AC_MAX_LENGTH;
		value = (uint32_t)byte() << 24;
		value |= (uint32_t)byte() << 16;
		value |= (uint32_t)byte() << 8;
		value |= (uint32_t)byte();
	}
	void renorm() {
		do {
			value = (value << 8) | byte();
		} while ((length <<= 8) < AC_MIN_LENGTH);
	}
	uint32_t bit(BitModel &m) {
		const uint32_t x = m.bit_0_prob * (length >> BM_LENGTH_SHIFT);
		const uint32_t sym = value >= x;
		if (sym == 0) {
			length = x;
			++m.bit_0_count;
		} else {
			value -= x;
			length -= x;
		}
		if (length < AC_MIN_LENGTH) renorm();
		if (--m.bits_until_update == 0) m.update();
		return sym;
	}
	// the largest s with distribution[s] * length <= value: by bisection, started from a table slice for the larger models
	uint32_t symbol(SymbolModel &m) {
		uint32_t y = length, x = 0, sym = 0, n = m.symbols;
		length >>= DM_LENGTH_SHIFT;
		if (m.table_size) {
			const uint32_t dv = value / length, t = dv >> m.table_shift;
			sym = m.table[t];
			n = m.table[t + 1] + 1;
			while (n > sym + 1) {
				const uint32_t k = (sym + n) >> 1;
				if (m.distribution[k] > dv)
					n = k;
				else
					sym = k;
			}
			x = m.distribution[sym] * length;
			if (sym != m.last_symbol) y = m.distribution[sym + 1] * length;
		} else {
			uint32_t k = n >> 1;
			do {
				const uint32_t z = length * m.distribution[k];
				if (z > value) {
					n = k;
					y = z;
				} else {
					sym = k;
					x = z;
				}
			} while ((k = (sym + n) >> 1) != sym);
		}
		value -= x;
		length = y - x;
		if (length < AC_MIN_LENGTH) renorm();
		++m.symbol_count[sym];
		if (--m.symbols_until_update == 0) m.update();
		return sym;
	}
	uint32_t read_short() {
		const uint32_t sym = value / (length >>= 16);
		value -= length * sym;
		if (length < AC_MIN_LENGTH) renorm();
		return sym;
	}
	uint32_t read_bits(uint32_t bits) {
		if (bits > 19) {
			const uint32_t lo = read_short();
			return (read_bits(bits - 16) << 16) | lo;
		}
		const uint32_t sym = value / (length >>= bits);
		value -= length * sym;
		if (length < AC_MIN_LENGTH) renorm();
		return sym;
	}
	uint32_t read_int() {
		const uint32_t lo = read_short();
		return (read_short() << 16) | lo;
	}
};

struct IntegerDecompressor : IntegerModels {
	using IntegerModels::IntegerModels;
	int32_t corrector(Decoder &d, SymbolModel &mb) {
		k = d.symbol(mb);
		if (k == 0) return (int32_t)d.bit(m_corrector0);
		if (k >= 32) return corr_min;
		int32_t c;
		if (k <= bits_high) {
			c = (int32_t)d.symbol(m_corrector[k]);
		} else {
			const uint32_t k1 = k - bits_high;
			c = (int32_t)d.symbol(m_corrector[k]);
			const int32_t c1 = (int32_t)d.read_bits(k1);
			c = (int32_t)(((uint32_t)c << k1) | (uint32_t)c1);
		}
		if (c >= (int32_t)(1u << (k - 1)))
			c += 1;
		else
			c -= (int32_t)((1u << k) - 1u);
		return c;
	}
	int32_t decompress(Decoder &d, int32_t pred, uint32_t context = 0) {
		int32_t real = (int32_t)((uint32_t)pred + (uint32_t)corrector(d, m_bits[context])); // wraps like the I32 sum it restates
		if (real < 0)
			real = (int32_t)((uint32_t)real + corr_range);
		else if ((uint32_t)real >= corr_range)
			real = (int32_t)((uint32_t)real - corr_range);
		return real;
	}
};

struct Item {
	virtual ~Item() = default;
	virtual void init(const uint8_t *first) = 0; // the chunk's first record, stored raw
	virtual void read(Decoder &d, uint8_t *out) = 0;
};

// x y z i32, intensity u16, (return number : 3, number of returns : 3, scan direction : 1, edge : 1), classification u8,
// scan angle rank i8, user data u8, point source id u16
struct Point10 : Item {
	uint8_t last[20];
	uint16_t last_intensity[16];
	StreamingMedian5 last_x_diff[16], last_y_diff[16];
	int32_t last_height[8];
	SymbolModel m_changed{64}, m_scan_angle[2] = {SymbolModel(256), SymbolModel(256)};
	std::vector<SymbolModel> m_bit_byte = std::vector<SymbolModel>(256), m_classification = std::vector<SymbolModel>(256),
	                         m_user_data = std::vector<SymbolModel>(256);
	IntegerDecompressor ic_intensity{16, 4}, ic_point_source{16, 1}, ic_dx{32, 2}, ic_dy{32, 22}, ic_z{32, 20};
	void init(const uint8_t *first) override {
		std::memset(last_intensity, 0, sizeof(last_intensity));
		std::memset(last_height, 0, sizeof(last_height));
		std::memcpy(last, first, 20);
		last[12] = last[13] = 0; // the intensity context starts at zero whatever the first record holds
	}
	void read(Decoder &d, uint8_t *out) override {
		const uint32_t changed = d.symbol(m_changed);
		uint32_t r, n, m, l;
		if (changed) {
			if (changed & 32) {
				SymbolModel &mm = m_bit_byte[last[14]];
				if (!mm.live()) mm.init(256);
				last[14] = (uint8_t)d.symbol(mm);
			}
			r = last[14] & 7u;
			n = (last[14] >> 3) & 7u;
			m = RETURN_MAP[n][r];
			l = RETURN_LEVEL[n][r];
			if (changed & 16) {
				const uint16_t v = (uint16_t)ic_intensity.decompress(d, last_intensity[m], m < 3 ? m : 3);
				wr_u16(last + 12, v);
				last_intensity[m] = v;
			} else {
				wr_u16(last + 12, last_intensity[m]);
			}
			if (changed & 8) {
				SymbolModel &mm = m_classification[last[15]];
				if (!mm.live()) mm.init(256);
				last[15] = (uint8_t)d.symbol(mm);
			}
			if (changed & 4) {
				const int val = (int)d.symbol(m_scan_angle[(last[14] >> 6) & 1u]);
				last[16] = u8_fold(val + last[16]);
			}
			if (changed & 2) {
				SymbolModel &mm = m_user_data[last[17]];
				if (!mm.live()) mm.init(256);
				last[17] = (uint8_t)d.symbol(mm);
			}
			if (changed & 1) wr_u16(last + 18, (uint16_t)ic_point_source.decompress(d, rd_u16(last + 18)));
		} else {
			r = last[14] & 7u;
			n = (last[14] >> 3) & 7u;
			m = RETURN_MAP[n][r];
			l = RETURN_LEVEL[n][r];
		}
		int32_t median = last_x_diff[m].get();
		int32_t diff = ic_dx.decompress(d, median, n == 1);
		wr_i32(last + 0, (int32_t)((uint32_t)rd_i32(last + 0) + (uint32_t)diff));
		last_x_diff[m].add(diff);
		median = last_y_diff[m].get();
		uint32_t k_bits = ic_dx.k;
		diff = ic_dy.decompress(d, median, (n == 1) + (k_bits < 20 ? (k_bits & ~1u) : 20u));
		wr_i32(last + 4, (int32_t)((uint32_t)rd_i32(last + 4) + (uint32_t)diff));
		last_y_diff[m].add(diff);
		k_bits = (ic_dx.k + ic_dy.k) / 2;
		const int32_t z = ic_z.decompress(d, last_height[l], (n == 1) + (k_bits < 18 ? (k_bits & ~1u) : 18u));
		wr_i32(last + 8, z);
		last_height[l] = z;
		std::memcpy(out, last, 20);
	}
};


struct GpsTime11 : Item {
	uint32_t last = 0, next = 0;
	uint64_t last_gpstime[4];
	int32_t last_diff[4], multi_extreme_counter[4];
	SymbolModel m_multi{(uint32_t)GPS_MULTI_TOTAL}, m_0diff{6};
	IntegerDecompressor ic{32, 9};
	void init(const uint8_t *first) override {
		last = next = 0;
		std::memset(last_gpstime, 0, sizeof(last_gpstime));
		std::memset(last_diff, 0, sizeof(last_diff));
		std::memset(multi_extreme_counter, 0, sizeof(multi_extreme_counter));
		std::memcpy(&last_gpstime[0], first, 8);
	}
	void add(int32_t diff) { last_gpstime[last] += (uint64_t)(int64_t)diff; } // (wraps like the i64 sum it restates; damaged streams get here with anything)
	void full(Decoder &d) { // a time that fits no sequence: upper half predicted from the current sequence, lower half raw
		next = (next + 1) & 3;
		last_gpstime[next] = (uint64_t)(uint32_t)ic.decompress(d, (int32_t)(last_gpstime[last] >> 32), 8);
		last_gpstime[next] = (last_gpstime[next] << 32) | d.read_int();
		last = next;
		last_diff[last] = 0;
		multi_extreme_counter[last] = 0;
	}
	void read(Decoder &d, uint8_t *out) override {
		for (;;) { // (a "switch sequence" symbol restarts the read with another of the four sequences)
			if (last_diff[last] == 0) { // the last difference was zero
				const int multi = (int)d.symbol(m_0diff);
				if (multi == 1) { // the difference fits 32 bits
					last_diff[last] = ic.decompress(d, 0, 0);
					add(last_diff[last]);
					multi_extreme_counter[last] = 0;
				} else if (multi == 2) {
					full(d);
				} else if (multi > 2) {
					last = (last + (uint32_t)multi - 2) & 3;
					continue;
				}
			} else {
				int multi = (int)d.symbol(m_multi);
				if (multi == 1) {
					add(ic.decompress(d, last_diff[last], 1));
					multi_extreme_counter[last] = 0;
				} else if (multi < GPS_MULTI_UNCHANGED) {
					int32_t gdiff;
					if (multi == 0) {
						gdiff = ic.decompress(d, 0, 7);
						if (++multi_extreme_counter[last] > 3) {
							last_diff[last] = gdiff;
							multi_extreme_counter[last] = 0;
						}
					} else if (multi < GPS_MULTI) {
						gdiff = ic.decompress(d, (int32_t)((uint32_t)multi * (uint32_t)last_diff[last]), multi < 10 ? 2 : 3);
					} else if (multi == GPS_MULTI) {
						gdiff = ic.decompress(d, (int32_t)((uint32_t)GPS_MULTI * (uint32_t)last_diff[last]), 4);
						if (++multi_extreme_counter[last] > 3) {
							last_diff[last] = gdiff;
							multi_extreme_counter[last] = 0;
						}
					} else {
						multi = GPS_MULTI - multi;
						if (multi > GPS_MULTI_MINUS) {
							gdiff = ic.decompress(d, (int32_t)((uint32_t)multi * (uint32_t)last_diff[last]), 5);
						} else {
							gdiff = ic.decompress(d, (int32_t)((uint32_t)GPS_MULTI_MINUS * (uint32_t)last_diff[last]), 6);
							if (++multi_extreme_counter[last] > 3) {
								last_diff[last] = gdiff;
								multi_extreme_counter[last] = 0;
							}
						}
					}
					add(gdiff);
				} else if (multi == GPS_MULTI_CODE_FULL) {
					full(d);
				} else if (multi > GPS_MULTI_CODE_FULL) {
					last = (last + (uint32_t)multi - (uint32_t)GPS_MULTI_CODE_FULL) & 3;
					continue;
				}
			}
			break;
		}
		std::memcpy(out, &last_gpstime[last], 8);
	}
};

struct Rgb12 : Item {
	uint16_t last[3];
	SymbolModel m_used{128}, m_diff[6] = {SymbolModel(256), SymbolModel(256), SymbolModel(256), SymbolModel(256), SymbolModel(256), SymbolModel(256)};
	void init(const uint8_t *first) override {
		for (int i = 0; i < 3; ++i) last[i] = rd_u16(first + 2 * i);
	}
	void read(Decoder &d, uint8_t *out) override {
		uint16_t it[3];
		const uint32_t sym = d.symbol(m_used);
		if (sym & 1)
			it[0] = u8_fold((int)d.symbol(m_diff[0]) + (last[0] & 255));
		else
			it[0] = last[0] & 0xFF;
		if (sym & 2)
			it[0] |= (uint16_t)(u8_fold((int)d.symbol(m_diff[1]) + (last[0] >> 8)) << 8);
		else
			it[0] |= last[0] & 0xFF00;
		if (sym & 64) {
			int diff = (it[0] & 0xFF) - (last[0] & 0xFF);
			if (sym & 4)
				it[1] = u8_fold((int)d.symbol(m_diff[2]) + u8_clamp(diff + (last[1] & 255)));
			else
				it[1] = last[1] & 0xFF;
			if (sym & 16) {
				diff = (diff + ((it[1] & 0xFF) - (last[1] & 0xFF))) / 2;
				it[2] = u8_fold((int)d.symbol(m_diff[4]) + u8_clamp(diff + (last[2] & 255)));
			} else {
				it[2] = last[2] & 0xFF;
			}
			diff = (it[0] >> 8) - (last[0] >> 8);
			if (sym & 8)
				it[1] |= (uint16_t)(u8_fold((int)d.symbol(m_diff[3]) + u8_clamp(diff + (last[1] >> 8))) << 8);
			else
				it[1] |= last[1] & 0xFF00;
			if (sym & 32) {
				diff = (diff + ((it[1] >> 8) - (last[1] >> 8))) / 2;
				it[2] |= (uint16_t)(u8_fold((int)d.symbol(m_diff[5]) + u8_clamp(diff + (last[2] >> 8))) << 8);
			} else {
				it[2] |= last[2] & 0xFF00;
			}
		} else {
			it[1] = it[0];
			it[2] = it[0];
		}
		for (int i = 0; i < 3; ++i) {
			last[i] = it[i];
			wr_u16(out + 2 * i, it[i]);
		}
	}
};

struct Bytes : Item {
	std::vector<uint8_t> last;
	std::vector<SymbolModel> m;
	explicit Bytes(uint32_t n) : last(n), m(n) {
		for (auto &x : m) x.init(256);
	}
	void init(const uint8_t *first) override { std::memcpy(last.data(), first, last.size()); }
	void read(Decoder &d, uint8_t *out) override {
		for (size_t i = 0; i < last.size(); ++i) out[i] = last[i] = u8_fold((int)last[i] + (int)d.symbol(m[i]));
	}
};


int32_t make_items(const uint16_t *types, const uint16_t *sizes, const uint16_t *versions, uint32_t n_items, std::vector<std::unique_ptr<Item>> &items) {
	for (uint32_t i = 0; i < n_items; ++i) {
		if (versions[i] != 2) return treee::set_err(TREEE_ERR_UNSUPPORTED, "LASzip item %u: version %u (only the pointwise version 2 is built)", types[i], versions[i]);
		switch (types[i]) {
		case ITEM_POINT10:
			if (sizes[i] != 20) return treee::set_err(TREEE_ERR_INVALID, "LASzip POINT10 item of %u bytes", sizes[i]);
			items.emplace_back(new Point10);
			break;
		case ITEM_GPSTIME11:
			if (sizes[i] != 8) return treee::set_err(TREEE_ERR_INVALID, "LASzip GPSTIME11 item of %u bytes", sizes[i]);
			items.emplace_back(new GpsTime11);
			break;
		case ITEM_RGB12:
			if (sizes[i] != 6) return treee::set_err(TREEE_ERR_INVALID, "LASzip RGB12 item of %u bytes", sizes[i]);
			items.emplace_back(new Rgb12);
			break;
		case ITEM_BYTE:
			if (sizes[i] == 0) return treee::set_err(TREEE_ERR_INVALID, "LASzip BYTE item of 0 bytes");
			items.emplace_back(new Bytes(sizes[i]));
			break;
		default:
			return treee::set_err(TREEE_ERR_UNSUPPORTED, "LASzip item type %u is not built (POINT10, GPSTIME11, RGB12, BYTE are)", types[i]);
		}
	}
	return TREEE_OK;
}

// one chunk: the first record raw, then the range-coded rest (a fresh decoder and fresh models per chunk)
int32_t decompress_chunk(const uint8_t *src, uint64_t src_len, const uint16_t *types, const uint16_t *sizes, const uint16_t *versions, uint32_t n_items,
                         uint64_t n_points, uint8_t *dst) {
	if (n_points == 0) return TREEE_OK;
	std::vector<std::unique_ptr<Item>> items;
	TREEE_TRY(make_items(types, sizes, versions, n_items, items));
	uint32_t rec = 0;
	for (uint32_t i = 0; i < n_items; ++i) rec += sizes[i];
	if (src_len < rec) return treee::set_err(TREEE_ERR_INVALID, "LASzip chunk of %llu bytes cannot hold its first record", (unsigned long long)src_len);
	std::memcpy(dst, src, rec);
	uint32_t o = 0;
	for (uint32_t i = 0; i < n_items; ++i) {
		items[i]->init(src + o);
		o += sizes[i];
	}
	if (n_points == 1) return TREEE_OK;
	try {
		Decoder d;
		d.init(src + rec, src + src_len);
		for (uint64_t p = 1; p < n_points; ++p) {
			uint8_t *out = dst + p * rec;
			o = 0;
			for (uint32_t i = 0; i < n_items; ++i) {
				items[i]->read(d, out + o);
				o += sizes[i];
			}
		}
	} catch (const Truncated &) {
		return treee::set_err(TREEE_ERR_INVALID, "LASzip chunk ends inside its range-coded data");
	}
	return TREEE_OK;
}


} // namespace

namespace { // the bodies of the entry points below (which only add the exception barrier)

// The chunk table behind the point data (`laz` 0.9.2 ChunkTable::read_from, laz.rs:80): the i64 at offset_to_point_data is
// its position (-1: the file's last 8 bytes hold it); there a version u32 (0), the number of chunks u32, and the chunks'
// byte sizes range-coded with a 32-bit integer decompressor (context 1, each predicted by the one before; with a variable
// chunk size the point counts come first in context 0).
static int32_t treee_laz_chunk_table_impl(const uint8_t *file, uint64_t file_len, uint64_t offset_to_point_data, uint32_t chunk_size, uint64_t total_points,
                              uint64_t *byte_counts_out, uint64_t *point_counts_out, uint64_t cap, uint64_t *n_chunks_out) {
	if (!file || !n_chunks_out) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_chunk_table: null argument");
	*n_chunks_out = 0;
	if (offset_to_point_data + 8 > file_len) return treee::set_err(TREEE_ERR_INVALID, "LAZ: no room for the chunk table offset");
	int64_t pos;
	std::memcpy(&pos, file + offset_to_point_data, 8);
	if (pos == -1) {
		if (file_len < 8) return treee::set_err(TREEE_ERR_INVALID, "LAZ: short file");
		std::memcpy(&pos, file + file_len - 8, 8);
	}
	if (pos < 0 || (uint64_t)pos + 8 > file_len) return treee::set_err(TREEE_ERR_INVALID, "LAZ: chunk table offset %lld outside the file", (long long)pos);
	uint32_t version, number;
	std::memcpy(&version, file + pos, 4);
	std::memcpy(&number, file + pos + 4, 4);
	if (version != 0) return treee::set_err(TREEE_ERR_UNSUPPORTED, "LAZ: chunk table version %u", version);
	if ((uint64_t)number > file_len) return treee::set_err(TREEE_ERR_INVALID, "LAZ: the chunk table claims %u chunks in a file of %llu bytes", number, (unsigned long long)file_len);
	*n_chunks_out = number;
	if (number == 0 || (!byte_counts_out && !point_counts_out)) return TREEE_OK; // (a call that only asks for the count)
	if (!byte_counts_out || !point_counts_out || cap < number) return treee::set_err(TREEE_ERR_INVALID, "LAZ: %u chunks, room for %llu", number, (unsigned long long)cap);
	try {
		Decoder d;
		d.init(file + pos + 8, file + file_len);
		IntegerDecompressor ic(32, 2);
		const bool variable = chunk_size == 0xFFFFFFFFu;
		uint32_t prev_bytes = 0, prev_points = 0;
		uint64_t remaining = total_points;
		for (uint32_t i = 0; i < number; ++i) {
			if (variable) prev_points = (uint32_t)ic.decompress(d, (int32_t)prev_points, 0);
			prev_bytes = (uint32_t)ic.decompress(d, (int32_t)prev_bytes, 1);
			byte_counts_out[i] = prev_bytes;
			const uint64_t pts = variable ? prev_points : chunk_size;
			point_counts_out[i] = pts < remaining ? pts : remaining; // laz.rs:88-89
			remaining -= point_counts_out[i];
		}
	} catch (const Truncated &) {
		return treee::set_err(TREEE_ERR_INVALID, "LAZ: the chunk table ends inside its range-coded data");
	}
	return TREEE_OK;
}

// Laz::read for a LAZ source (laz.rs:130-180): every chunk decompressed into its place in `dst` (records of
// sum(item_sizes) bytes, chunk after chunk), one chunk per task on `threads` host threads (0 = all).
static int32_t treee_laz_decompress_chunks_impl(const uint8_t *file, uint64_t file_len, const uint16_t *item_types, const uint16_t *item_sizes, const uint16_t *item_versions,
                                    uint32_t n_items, const uint64_t *chunk_starts, const uint64_t *chunk_bytes, const uint64_t *chunk_points, uint64_t n_chunks,
                                    uint32_t threads, uint8_t *dst, uint64_t dst_len) {
	if (n_chunks == 0) return TREEE_OK;
	if (!file || !item_types || !item_sizes || !item_versions || !chunk_starts || !chunk_bytes || !chunk_points || !dst)
		return treee::set_err(TREEE_ERR_INVALID, "treee_laz_decompress_chunks: null argument");
	uint64_t rec = 0, total = 0;
	for (uint32_t i = 0; i < n_items; ++i) rec += item_sizes[i];
	std::vector<uint64_t> first(n_chunks);
	for (uint64_t c = 0; c < n_chunks; ++c) {
		first[c] = total;
		total += chunk_points[c];
		if (chunk_starts[c] > file_len || chunk_bytes[c] > file_len - chunk_starts[c])
			return treee::set_err(TREEE_ERR_INVALID, "LAZ: chunk %llu lies outside the file", (unsigned long long)c);
	}
	if (total * rec > dst_len) return treee::set_err(TREEE_ERR_INVALID, "LAZ: %llu records of %llu bytes, room for %llu bytes", (unsigned long long)total, (unsigned long long)rec, (unsigned long long)dst_len);
	{ // item list checked once on the calling thread (its error text is thread-local)
		std::vector<std::unique_ptr<Item>> probe;
		TREEE_TRY(make_items(item_types, item_sizes, item_versions, n_items, probe));
	}
	uint32_t nt = threads ? threads : treee::host_threads();
	if (nt == 0) nt = 1;
	if ((uint64_t)nt > n_chunks) nt = (uint32_t)n_chunks;
	std::atomic<uint64_t> next{0};
	std::atomic<int64_t> failed{-1};
	std::atomic<bool> threw{false};
	auto work = [&]() {
		for (;;) {
			const uint64_t c = next.fetch_add(1);
			if (c >= n_chunks || failed.load() >= 0) return;
			int32_t rc;
			try { // (nothing may leave a std::thread's function: an escaping exception is std::terminate)
				rc = decompress_chunk(file + chunk_starts[c], chunk_bytes[c], item_types, item_sizes, item_versions, n_items, chunk_points[c], dst + first[c] * rec);
			} catch (...) {
				threw = true;
				rc = TREEE_ERR_NOMEM;
			}
			if (rc != TREEE_OK) {
				int64_t none = -1;
				failed.compare_exchange_strong(none, (int64_t)c);
			}
		}
	};
	try {
		if (nt == 1) {
			work();
		} else {
			std::vector<std::thread> pool;
			try {
				for (uint32_t t = 1; t < nt; ++t) pool.emplace_back(work);
			} catch (...) { // (no more threads to be had: the ones that exist and this one do the work)
			}
			work();
			for (auto &t : pool) t.join();
		}
		if (threw) return treee::set_err(TREEE_ERR_NOMEM, "LAZ: out of memory while decompressing chunk %lld", (long long)failed.load());
		if (failed.load() >= 0) { // decode that chunk again here for its message
			const uint64_t c = (uint64_t)failed.load();
			return decompress_chunk(file + chunk_starts[c], chunk_bytes[c], item_types, item_sizes, item_versions, n_items, chunk_points[c], dst + first[c] * rec);
		}
	} catch (const std::exception &e) {
		return treee::set_err(TREEE_ERR_NOMEM, "LAZ: %s", e.what());
	}
	return TREEE_OK;
}

} // namespace

extern "C" {

int32_t treee_laz_chunk_table(const uint8_t *file, uint64_t file_len, uint64_t offset_to_point_data, uint32_t chunk_size, uint64_t total_points,
                              uint64_t *byte_counts_out, uint64_t *point_counts_out, uint64_t cap, uint64_t *n_chunks_out) {
	try { // nothing unwinds across the C ABI
		return treee_laz_chunk_table_impl(file, file_len, offset_to_point_data, chunk_size, total_points, byte_counts_out, point_counts_out, cap, n_chunks_out);
	} catch (const std::exception &e) {
		return treee::set_err(TREEE_ERR_NOMEM, "treee_laz_chunk_table: %s", e.what());
	}
}

int32_t treee_laz_decompress_chunks(const uint8_t *file, uint64_t file_len, const uint16_t *item_types, const uint16_t *item_sizes, const uint16_t *item_versions,
                                    uint32_t n_items, const uint64_t *chunk_starts, const uint64_t *chunk_bytes, const uint64_t *chunk_points, uint64_t n_chunks,
                                    uint32_t threads, uint8_t *dst, uint64_t dst_len) {
	try { // nothing unwinds across the C ABI
		return treee_laz_decompress_chunks_impl(file, file_len, item_types, item_sizes, item_versions, n_items, chunk_starts, chunk_bytes, chunk_points, n_chunks, threads, dst, dst_len);
	} catch (const std::exception &e) {
		return treee::set_err(TREEE_ERR_NOMEM, "treee_laz_decompress_chunks: %s", e.what());
	}
}

} // extern "C"
