// laz_models.h -- the adaptive models and tables of the LASzip format (M. Isenburg, "LASzip: lossless compression of LiDAR
// data", PE&RS 79(2), 2013) that BOTH directions share: the decoder of the library (laz_codec.cpp, the reference's side:
// treee/src/laz.rs:146-160 decompresses, nothing in it compresses) and the compressor that writes fixtures and benchmark input
// (tests/cpp/laz_encode_fixture.cpp, test infrastructure).  Every model rule exists once.
#pragma once
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>

namespace treee_laz {


constexpr uint32_t AC_MIN_LENGTH = 0x01000000u, AC_MAX_LENGTH = 0xFFFFFFFFu;
constexpr uint32_t BM_LENGTH_SHIFT = 13, BM_MAX_COUNT = 1u << BM_LENGTH_SHIFT;
constexpr uint32_t DM_LENGTH_SHIFT = 15, DM_MAX_COUNT = 1u << DM_LENGTH_SHIFT;


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

// the model set of one integer (de)compressor: a symbol model per context for k = the number of significant corrector bits,
// one model per k for the corrector's high bits
struct IntegerModels {
	uint32_t k = 0, bits_high = 8, corr_bits = 0, corr_range = 0;
	int32_t corr_min = 0;
	std::vector<SymbolModel> m_bits, m_corrector; // m_corrector[0] is unused (the bit model below takes its place)
	BitModel m_corrector0;
	IntegerModels(uint32_t bits, uint32_t contexts) {
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

enum { ITEM_BYTE = 0, ITEM_POINT10 = 6, ITEM_GPSTIME11 = 7, ITEM_RGB12 = 8 };

// GPSTIME11: the multiplier symbols of the time differences
constexpr int GPS_MULTI = 500, GPS_MULTI_MINUS = -10, GPS_MULTI_UNCHANGED = GPS_MULTI - GPS_MULTI_MINUS + 1,
              GPS_MULTI_CODE_FULL = GPS_MULTI - GPS_MULTI_MINUS + 2, GPS_MULTI_TOTAL = GPS_MULTI - GPS_MULTI_MINUS + 6;

} // namespace treee_laz
