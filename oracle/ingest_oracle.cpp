// ingest_oracle.cpp -- CPU ORACLE (test infrastructure): LAS record decode and y-slice
// binning.  Follows treee/src/laz.rs:33-124,193-213,228-287 and treee/src/loading.rs:91-115,
// consumer densification treee/src/segmenting.rs:121-133.  Parity unpinned (no reference
// fixtures exist); the arithmetic is +,-,*,casts only, so it is bit-reproducible.
// LAZ arithmetic decompression (crate `laz` 0.9.2, Cargo.lock:1393-1394) is third-party and
// NOT restated: only uncompressed LAS records are handled.
#include "treee_oracle.h"

#include <cmath>
#include <cstring>
#include <map>
#include <vector>

namespace {

template <class T> T rd(const uint8_t *p) {
	T v;
	std::memcpy(&v, p, sizeof(T)); // little endian host (laz.rs:270 asserts it too)
	return v;
}

inline int64_t floor_as_isize(float y) {
	// loading.rs:99  p.y.floor() as isize  (Rust `as`: saturating, NaN -> 0)
	float f = std::floor(y);
	if (f != f) return 0;
	if (f >= 9223372036854775808.0f) return INT64_MAX;
	if (f <= -9223372036854775808.0f) return INT64_MIN;
	return (int64_t)f;
}

} // namespace

extern "C" {

void orc_decode_las_records(const uint8_t *rec, uint64_t n, uint32_t stride, const double scale[3],
                            const double offset[3], const double center[3], float *xyz_out) {
	for (uint64_t i = 0; i < n; ++i) { // Chunk::next_point laz.rs:202-213
		const uint8_t *s = rec + i * (uint64_t)stride;
		int32_t x = rd<int32_t>(s + 0);
		int32_t y = rd<int32_t>(s + 4);
		int32_t z = rd<int32_t>(s + 8);
		double vx = offset[0] + (double)x * scale[0];
		double vy = offset[1] + (double)y * scale[1];
		double vz = offset[2] + (double)z * scale[2];
		// (v.x, v.z, -v.y) - center, then `as f32`
		xyz_out[3 * i + 0] = (float)(vx - center[0]);
		xyz_out[3 * i + 1] = (float)(vz - center[1]);
		xyz_out[3 * i + 2] = (float)(-vy - center[2]);
	}
}

int orc_las_header_parse(const uint8_t *b, uint64_t len, orc_las_header *h) {
	if (len < 375) return -1; // read_exact of the packed 375-byte struct (laz.rs:274-277)
	if (std::memcmp(b, "LASF", 4) != 0) return -1; // laz.rs:282-284
	h->version_major = b[24];
	h->version_minor = b[25];
	h->header_size = rd<uint16_t>(b + 94);
	h->offset_to_point_data = rd<uint32_t>(b + 96);
	h->point_data_record_format = b[104];
	h->point_data_record_length = rd<uint16_t>(b + 105);
	uint32_t legacy = rd<uint32_t>(b + 107);
	h->number_of_point_records = rd<uint64_t>(b + 247);
	if (legacy != 0) h->number_of_point_records = legacy; // laz.rs:279-281
	for (int i = 0; i < 3; ++i) {
		h->scale[i] = rd<double>(b + 131 + 8 * i);
		h->offset[i] = rd<double>(b + 155 + 8 * i);
	}
	double max_x = rd<double>(b + 179), min_x = rd<double>(b + 187);
	double max_y = rd<double>(b + 195), min_y = rd<double>(b + 203);
	double max_z = rd<double>(b + 211), min_z = rd<double>(b + 219);
	h->min[0] = min_x; // laz.rs:72-73
	h->min[1] = min_z;
	h->min[2] = -max_y;
	h->max[0] = max_x;
	h->max[1] = max_z;
	h->max[2] = -min_y;
	for (int i = 0; i < 3; ++i) h->center[i] = (h->min[i] + h->max[i]) * 0.5; // na::center
	return 0;
}

void orc_slice_keys(const float *xyz, uint64_t n, int64_t *key_out) {
	for (uint64_t i = 0; i < n; ++i) key_out[i] = floor_as_isize(xyz[3 * i + 1]);
}

uint64_t orc_slice_bin(const float *xyz, uint64_t n, int64_t *min_key_out, uint64_t *order_out,
                       uint64_t *offsets_out, uint64_t offsets_cap) {
	if (n == 0) return 0;
	std::map<int64_t, std::vector<uint64_t>> slices; // HashMap<isize, Vec<P3>> loading.rs:28
	for (uint64_t i = 0; i < n; ++i) slices[floor_as_isize(xyz[3 * i + 1])].push_back(i);
	int64_t mn = slices.begin()->first, mx = slices.rbegin()->first; // segmenting.rs:121-122
	uint64_t layers = (uint64_t)(mx - mn + 1);
	if (min_key_out) *min_key_out = mn;
	if (layers + 1 > offsets_cap) return layers;
	uint64_t pos = 0;
	for (uint64_t l = 0; l < layers; ++l) {
		offsets_out[l] = pos;
		auto it = slices.find(mn + (int64_t)l);
		if (it != slices.end())
			for (uint64_t i : it->second) order_out[pos++] = i;
	}
	offsets_out[layers] = pos;
	return layers;
}

} // extern "C"
