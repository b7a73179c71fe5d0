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

namespace {

constexpr uint32_t AC_MIN_LENGTH = 0x01000000u, AC_MAX_LENGTH = 0xFFFFFFFFu;
constexpr uint32_t BM_LENGTH_SHIFT = 13, BM_MAX_COUNT = 1u << BM_LENGTH_SHIFT;
constexpr uint32_t DM_LENGTH_SHIFT = 15, DM_MAX_COUNT = 1u << DM_LENGTH_SHIFT;

struct Truncated {}; // thrown when a chunk's bytes run out (caught at the chunk boundary: TREEE_ERR_INVALID)

struct BitModel {
	uint32_t update_cycle = 4, bits_until_update = 4, bit_0_prob = 1u << (BM_LENGTH_SHIFT - 1), bit_0_count = 1, bit_count = 2;
	void update() {
		if ((bit_count += update_cycle) > BM_MAX_COUNT) {
			bit_count = (bit_count + 1) >> 1;
			bit_0_count = (bit_0_count + 1) >> 1;
			if (bit_0_count == bit_count) ++bit_count;
		}
		const uint32_t scale = 0x80000000u / bit_count;
		bit_0_prob = (bit_0_count * scale) >> (31 - BM_LENGTH_SHIFT);
		update_cycle = (5 * update_cycle) >> 2;
		if (update_cycle > 64) update_cycle = 64;
		bits_until_update = update_cycle;
	}
};

struct SymbolModel {
	uint32_t symbols = 0, last_symbol = 0, total_count = 0, update_cycle = 0, symbols_until_update = 0;
	uint32_t table_size = 0, table_shift = 0; // models of more than 16 symbols: first symbol of every slice of the 15-bit range
	std::vector<uint32_t> distribution, symbol_count, table;
	SymbolModel() = default;
	explicit SymbolModel(uint32_t n) { init(n); }
	bool live() const { return symbols != 0; }
	void init(uint32_t n) {
		symbols = n;
		last_symbol = n - 1;
		distribution.assign(n, 0);
		symbol_count.assign(n, 1);
		if (n > 16) {
			uint32_t table_bits = 3;
			while (n > (1u << (table_bits + 2))) ++table_bits;
			table_size = 1u << table_bits;
			table_shift = DM_LENGTH_SHIFT - table_bits;
			table.assign(table_size + 2, 0);
		}
		total_count = 0;
		update_cycle = n;
		update();
		symbols_until_update = update_cycle = (n + 6) >> 1;
	}
	void update() {
		if ((total_count += update_cycle) > DM_MAX_COUNT) {
			total_count = 0;
			for (uint32_t k = 0; k < symbols; ++k) total_count += (symbol_count[k] = (symbol_count[k] + 1) >> 1);
		}
		uint32_t sum = 0, s = 0;
		const uint32_t scale = 0x80000000u / total_count;
		if (table_size == 0) {
			for (uint32_t k = 0; k < symbols; ++k) {
				distribution[k] = (scale * sum) >> (31 - DM_LENGTH_SHIFT);
				sum += symbol_count[k];
			}
		} else {
			for (uint32_t k = 0; k < symbols; ++k) {
				distribution[k] = (scale * sum) >> (31 - DM_LENGTH_SHIFT);
				sum += symbol_count[k];
				const uint32_t w = distribution[k] >> table_shift;
				while (s < w) table[++s] = k - 1;
			}
			table[0] = 0;
			while (s <= table_size) table[++s] = symbols - 1;
		}
		update_cycle = (5 * update_cycle) >> 2;
		const uint32_t max_cycle = (symbols + 6) << 3;
		if (update_cycle > max_cycle) update_cycle = max_cycle;
		symbols_until_update = update_cycle;
	}
};

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

struct IntegerDecompressor {
	uint32_t k = 0, bits_high = 8, corr_bits = 0, corr_range = 0;
	int32_t corr_min = 0;
	std::vector<SymbolModel> m_bits, m_corrector; // m_corrector[0] is unused (the bit model below takes its place)
	BitModel m_corrector0;
	IntegerDecompressor(uint32_t bits, uint32_t contexts) {
		if (bits && bits < 32) {
			corr_bits = bits;
			corr_range = 1u << bits;
			corr_min = -(int32_t)(corr_range / 2);
		} else {
			corr_bits = 32;
			corr_range = 0;
			corr_min = INT32_MIN;
		}
		m_bits.resize(contexts);
		for (auto &m : m_bits) m.init(corr_bits + 1);
		m_corrector.resize(corr_bits + 1);
		for (uint32_t i = 1; i <= corr_bits; ++i) m_corrector[i].init(i <= bits_high ? (1u << i) : (1u << bits_high));
	}
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

struct StreamingMedian5 {
	int32_t v[5] = {0, 0, 0, 0, 0};
	bool high = true;
	int32_t get() const { return v[2]; }
	void add(int32_t x) {
		if (high) {
			if (x < v[2]) {
				v[4] = v[3];
				v[3] = v[2];
				if (x < v[0]) {
					v[2] = v[1];
					v[1] = v[0];
					v[0] = x;
				} else if (x < v[1]) {
					v[2] = v[1];
					v[1] = x;
				} else {
					v[2] = x;
				}
			} else {
				if (x < v[3]) {
					v[4] = v[3];
					v[3] = x;
				} else {
					v[4] = x;
				}
				high = false;
			}
		} else {
			if (v[2] < x) {
				v[0] = v[1];
				v[1] = v[2];
				if (v[4] < x) {
					v[2] = v[3];
					v[3] = v[4];
					v[4] = x;
				} else if (v[3] < x) {
					v[2] = v[3];
					v[3] = x;
				} else {
					v[2] = x;
				}
			} else {
				if (v[1] < x) {
					v[0] = v[1];
					v[1] = x;
				} else {
					v[0] = x;
				}
				high = true;
			}
		}
	}
};

const uint8_t RETURN_MAP[8][8] = {{15, 14, 13, 12, 11, 10, 9, 8}, {14, 0, 1, 3, 6, 10, 10, 9}, {13, 1, 2, 4, 7, 11, 11, 10}, {12, 3, 4, 5, 8, 12, 12, 11},
                                  {11, 6, 7, 8, 9, 13, 13, 12}, {10, 10, 11, 12, 13, 14, 14, 13}, {9, 10, 11, 12, 13, 14, 15, 14}, {8, 9, 10, 11, 12, 13, 14, 15}};
const uint8_t RETURN_LEVEL[8][8] = {{0, 1, 2, 3, 4, 5, 6, 7}, {1, 0, 1, 2, 3, 4, 5, 6}, {2, 1, 0, 1, 2, 3, 4, 5}, {3, 2, 1, 0, 1, 2, 3, 4},
                                    {4, 3, 2, 1, 0, 1, 2, 3}, {5, 4, 3, 2, 1, 0, 1, 2}, {6, 5, 4, 3, 2, 1, 0, 1}, {7, 6, 5, 4, 3, 2, 1, 0}};

inline int32_t rd_i32(const uint8_t *p) {
	int32_t v;
	std::memcpy(&v, p, 4);
	return v;
}
inline void wr_i32(uint8_t *p, int32_t v) { std::memcpy(p, &v, 4); }
inline uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline void wr_u16(uint8_t *p, uint16_t v) {
	p[0] = (uint8_t)v;
	p[1] = (uint8_t)(v >> 8);
}
inline uint8_t u8_fold(int v) { return (uint8_t)(v & 255); }
inline int u8_clamp(int v) { return v <= 0 ? 0 : (v >= 255 ? 255 : v); }

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

constexpr int GPS_MULTI = 500, GPS_MULTI_MINUS = -10, GPS_MULTI_UNCHANGED = GPS_MULTI - GPS_MULTI_MINUS + 1,
              GPS_MULTI_CODE_FULL = GPS_MULTI - GPS_MULTI_MINUS + 2, GPS_MULTI_TOTAL = GPS_MULTI - GPS_MULTI_MINUS + 6;

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

enum { ITEM_BYTE = 0, ITEM_POINT10 = 6, ITEM_GPSTIME11 = 7, ITEM_RGB12 = 8 };

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
	IntegerDecompressor m; // the same models, driven from the other side
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

// An uncompressed LAS image (LAS 1.0-1.3, point formats 0-3, any extra bytes) -> the LAZ image of the same records: the
// laszip record appended to the variable length records, the compression bit set in the point format, chunks of
// `chunk_size` points compressed on `threads` host threads (0 = all), the chunk table behind them.  The reference never
// writes LAZ; this exists for fixtures and for the streamed-LAZ benchmark input.  `out` needs *out_len_out bytes: a call
// whose `cap` is too small fails with TREEE_ERR_INVALID after setting it (random records can grow by a few per cent).
static int32_t treee_laz_encode_impl(const uint8_t *las, uint64_t las_len, uint32_t chunk_size, uint32_t threads, uint8_t *out, uint64_t cap, uint64_t *out_len_out) {
	if (!las || !out_len_out) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_encode: null argument");
	*out_len_out = 0;
	if (las_len < 375 || std::memcmp(las, "LASF", 4) != 0) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_encode: not a LAS image (laz.rs:274-284)");
	if (chunk_size == 0 || chunk_size == 0xFFFFFFFFu) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_encode: chunk size %u", chunk_size);
	uint32_t offset, n_vlr, legacy;
	uint64_t n;
	std::memcpy(&offset, las + 96, 4);
	std::memcpy(&n_vlr, las + 100, 4);
	std::memcpy(&legacy, las + 107, 4);
	std::memcpy(&n, las + 247, 8);
	if (legacy) n = legacy; // laz.rs:279-281
	const uint32_t fmt = las[104] & 0x3Fu, rl = rd_u16(las + 105);
	if (fmt > 3) return treee::set_err(TREEE_ERR_UNSUPPORTED, "treee_laz_encode: point format %u (0-3 are built)", fmt);
	uint16_t types[4], sizes[4], versions[4] = {2, 2, 2, 2};
	uint32_t n_items = 0, base = 20;
	types[n_items] = ITEM_POINT10, sizes[n_items++] = 20;
	if (fmt == 1 || fmt == 3) types[n_items] = ITEM_GPSTIME11, sizes[n_items++] = 8, base += 8;
	if (fmt == 2 || fmt == 3) types[n_items] = ITEM_RGB12, sizes[n_items++] = 6, base += 6;
	if (rl < base) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_encode: record length %u below the %u bytes of format %u", rl, base, fmt);
	if (rl > base) types[n_items] = ITEM_BYTE, sizes[n_items++] = (uint16_t)(rl - base);
	if (offset > las_len || n > (las_len - offset) / rl) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_encode: the header promises more records than the image holds");
	const uint8_t *rec = las + offset;
	const uint64_t n_chunks = (n + chunk_size - 1) / chunk_size;
	std::vector<std::vector<uint8_t>> chunks(n_chunks);
	{
		uint32_t nt = threads ? threads : treee::host_threads();
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
		if (threw) return treee::set_err(TREEE_ERR_NOMEM, "treee_laz_encode: out of memory");
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
	if (new_offset > 0xFFFFFFFFull) return treee::set_err(TREEE_ERR_UNSUPPORTED, "treee_laz_encode: header area beyond 4 GiB");
	if (!out || cap < total) return treee::set_err(TREEE_ERR_INVALID, "treee_laz_encode: the image needs %llu bytes, room for %llu", (unsigned long long)total, (unsigned long long)cap);
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

int32_t treee_laz_encode(const uint8_t *las, uint64_t las_len, uint32_t chunk_size, uint32_t threads, uint8_t *out, uint64_t cap, uint64_t *out_len_out) {
	try { // nothing unwinds across the C ABI
		return treee_laz_encode_impl(las, las_len, chunk_size, threads, out, cap, out_len_out);
	} catch (const std::exception &e) {
		return treee::set_err(TREEE_ERR_NOMEM, "treee_laz_encode: %s", e.what());
	}
}

} // extern "C"
