// calc_oracle.cpp -- CPU ORACLE (test infrastructure): the Calculations phase around the kNN consumer.
//   Tree (the convex slice hull)     treee/src/segmenting.rs:242-396 (new :258-269, insert :346-396, statistics :320-323)
//   centroid                          treee/src/segmenting.rs:526-545
//   get_size_areas                    treee/src/calculations.rs:429-452 (SLICE_WIDTH :16)
//   SegmentInformation::new           treee/src/calculations.rs:210-276
//   SegmentData::new                  treee/src/calculations.rs:133-175 (min/max, classification)
//   SegmentInformation::update        treee/src/calculations.rs:278-327 (expansion, height; the curve branch is
//                                     curvature_oracle.cpp), approximate_diameter :460-462
//   DisplayData::new                  treee/src/interactive/mod.rs:1002-1039
// Third-party arithmetic restated from published behaviour (parity unpinned, nalgebra 0.33.0): Vector2::normalize =
// v / sqrt(x*x + y*y) component-wise; dot = a.x*b.x + a.y*b.y; f32::min/max drop a NaN operand; `as usize` saturates
// (NaN -> 0, negative -> 0).  No FMA anywhere (compile with -ffp-contract=off).
#include "treee_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr float SLICE_WIDTH = 0.1f; // calculations.rs:16

struct P2 {
	float x, y;
};

inline size_t as_usize(float v) { // Rust `as usize`: saturating, NaN -> 0
	if (!(v > 0.0f)) return 0;
	if (v >= 18446744073709551616.0f) return SIZE_MAX;
	return (size_t)v;
}
inline float rs_min(float a, float b) { return std::fmin(a, b); } // f32::min: the non-NaN operand
inline float rs_max(float a, float b) { return std::fmax(a, b); }

// segmenting.rs:347-353: is `point` strictly outside the edge a -> b (outward normal (dir.y, -dir.x) normalised)
inline bool outside(P2 a, P2 b, P2 point) {
	const float dx = b.x - a.x, dy = b.y - a.y;
	const float ox = dy, oy = -dx;
	const float norm = std::sqrt(ox * ox + oy * oy);
	const float nx = ox / norm, ny = oy / norm;
	const float fx = point.x - a.x, fy = point.y - a.y;
	const float dist = nx * fx + ny * fy;
	return dist > 0.0f;
}

struct Tree { // segmenting.rs:242-247 (min / max only feed contains / distance, which this phase never calls)
	std::vector<P2> points;
	explicit Tree(P2 p) : points{p, {p.x + 0.1f, p.y}, {p.x, p.y + 0.1f}} {} // :258-269
	void insert(P2 point) {                                                      // :346-396
		std::vector<P2> &tree = points;
		const size_t len = tree.size();
		bool last = outside(tree[len - 1], tree[0], point);
		bool has_start = false, has_end = false;
		size_t start = 0, end = 0;
		for (size_t i = 0; i < len; ++i) {
			const bool out = outside(tree[i], tree[(i + 1) % len], point);
			if (!last && out) {
				start = i;
				has_start = true;
			} else if (last && !out) {
				end = i;
				has_end = true;
			}
			last = out;
		}
		if (!has_start || !has_end) return;
		if (end < start) { // splice((start + 1)..len, [point]); splice(0..end, [])
			tree.erase(tree.begin() + (start + 1), tree.end());
			tree.push_back(point);
			tree.erase(tree.begin(), tree.begin() + end);
		} else { // splice((start + 1)..end, [point])
			tree.erase(tree.begin() + (start + 1), tree.begin() + end);
			tree.insert(tree.begin() + (start + 1), point);
		}
	}
	float area() const { // statistics().area = centroid(points).1 (segmenting.rs:526-545)
		float area = 0.0f;
		const P2 a = points[0];
		for (size_t i = 1; i + 1 < points.size(); ++i) {
			const float bx = points[i].x - a.x, by = points[i].y - a.y;
			const float cx = points[i + 1].x - a.x, cy = points[i + 1].y - a.y;
			const float t_area = bx * cy - by * cx;
			area += t_area;
		}
		return area / 2.0f;
	}
};

// calculations.rs:429-452; valid == nullptr -> every point
std::vector<float> size_areas(float min, float height, const float *xyz, uint64_t n, const uint8_t *cls, int want_cls) {
	const size_t slices = as_usize(height / SLICE_WIDTH) + 1;
	std::vector<Tree *> sets(slices, nullptr);
	for (uint64_t i = 0; i < n; ++i) {
		if (cls && cls[i] != want_cls) continue;
		const float y = xyz[3 * i + 1];
		const size_t idx = as_usize((y - min) / SLICE_WIDTH);
		const P2 p{xyz[3 * i + 0], xyz[3 * i + 2]};
		if (idx >= slices) continue; // (the reference would panic; unreachable for min <= y <= max)
		if (sets[idx])
			sets[idx]->insert(p);
		else
			sets[idx] = new Tree(p);
	}
	std::vector<float> areas(slices, 0.0f);
	for (size_t s = 0; s < slices; ++s)
		if (sets[s]) {
			areas[s] = sets[s]->area();
			delete sets[s];
		}
	return areas;
}

inline int total_cmp(float a, float b) { // f32::total_cmp
	int32_t x, y;
	std::memcpy(&x, &a, 4);
	std::memcpy(&y, &b, 4);
	x ^= (int32_t)((uint32_t)(x >> 31) >> 1);
	y ^= (int32_t)((uint32_t)(y >> 31) >> 1);
	return x < y ? -1 : (x > y ? 1 : 0);
}

// calculations.rs:210-276; info = {trunk_height, crown_height, ground_sep, crown_sep}
void segment_information(const float *xyz, uint64_t n, float min, float max, float info[4]) {
	const float height = max - min;
	const float ground_max_search_height = 1.0f, ground_min_area_scale = 1.5f, min_crown_diameter = 2.0f;
	const size_t slices = as_usize(height / SLICE_WIDTH) + 1;
	const std::vector<float> areas = size_areas(min, height, xyz, n, nullptr, 0);
	// .skip((1.0 / SLICE_WIDTH) as usize).take((10.0 / SLICE_WIDTH) as usize).min_by(total_cmp): the FIRST minimum
	const size_t skip = as_usize(1.0f / SLICE_WIDTH), take = as_usize(10.0f / SLICE_WIDTH);
	bool have = false;
	float min_area = 0.5f;
	for (size_t i = skip; i < areas.size() && i < skip + take; ++i)
		if (!have || total_cmp(areas[i], min_area) < 0) {
			min_area = areas[i];
			have = true;
		}
	if (!have) min_area = 0.5f;
	min_area = rs_max(min_area, 0.5f);
	const float thresh = min_area * ground_min_area_scale;
	bool has_ground = false;
	size_t ground = 0;
	const size_t gtake = as_usize(ground_max_search_height / SLICE_WIDTH);
	for (size_t i = 0; i < areas.size() && i < gtake; ++i)
		if (areas[i] > thresh) {
			ground = i;
			has_ground = true;
			break;
		}
	size_t ground_sep = 0;
	if (has_ground) {
		for (size_t i = ground; i < areas.size() && i < slices / 2; ++i) // .take(slices / 2).skip(ground)
			if (areas[i] < thresh) {
				ground_sep = i;
				break;
			}
	}
	const float pi = 3.14159265358979323846f;
	const float half = min_crown_diameter / 2.0f;
	const float min_crown_area = pi * (half * half); // powi(2)
	size_t crown_sep = 0;
	for (size_t i = ground_sep; i < areas.size(); ++i)
		if (areas[i] > min_crown_area) {
			crown_sep = i;
			break;
		}
	const float ground_sep_f = min + (float)ground_sep * SLICE_WIDTH;
	const float crown_sep_f = min + (float)crown_sep * SLICE_WIDTH;
	info[0] = crown_sep_f - ground_sep_f;
	info[1] = max - crown_sep_f;
	info[2] = ground_sep_f;
	info[3] = crown_sep_f;
}

inline float approximate_diameter(float area) { // calculations.rs:460-462
	return 2.0f * std::sqrt(area / 3.14159265358979323846f);
}

} // namespace

extern "C" {

uint64_t orc_hull_insert_all(const float *xz, uint64_t n, float *hull_out, uint64_t cap, float *area_out) {
	if (n == 0) return 0;
	Tree t(P2{xz[0], xz[1]});
	for (uint64_t i = 1; i < n; ++i) t.insert(P2{xz[2 * i], xz[2 * i + 1]});
	if (area_out) *area_out = t.area();
	for (uint64_t i = 0; i < t.points.size() && i < cap; ++i) {
		hull_out[2 * i] = t.points[i].x;
		hull_out[2 * i + 1] = t.points[i].y;
	}
	return t.points.size();
}

uint64_t orc_size_areas(const float *xyz, uint64_t n, const uint8_t *cls, int32_t want_cls, float min, float height,
                        float *areas_out, uint64_t cap) {
	const std::vector<float> a = size_areas(min, height, xyz, n, cls, want_cls);
	for (uint64_t i = 0; i < a.size() && i < cap; ++i) areas_out[i] = a[i];
	return a.size();
}

void orc_segment_data(const float *xyz, uint64_t n, float min_out[3], float max_out[3], float info_out[4], uint8_t *class_out) {
	float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
	if (n) {
		for (int d = 0; d < 3; ++d) mn[d] = mx[d] = xyz[d];
		for (uint64_t i = 0; i < n; ++i)
			for (int d = 0; d < 3; ++d) {
				mn[d] = rs_min(mn[d], xyz[3 * i + d]);
				mx[d] = rs_max(mx[d], xyz[3 * i + d]);
			}
	}
	float info[4];
	segment_information(xyz, n, mn[1], mx[1], info);
	const float crown_sep = mx[1] - info[1];     // calculations.rs:149
	const float ground_sep = crown_sep - info[0]; // :150
	for (uint64_t i = 0; i < n; ++i) {
		const float y = xyz[3 * i + 1];
		class_out[i] = y < ground_sep ? 0 : (y < crown_sep ? 1 : 2); // Ground, Trunk, Crown
	}
	for (int d = 0; d < 3; ++d) {
		min_out[d] = mn[d];
		max_out[d] = mx[d];
	}
	for (int d = 0; d < 4; ++d) info_out[d] = info[d];
}

void orc_segment_update(const float *xyz, const uint8_t *cls, uint64_t n, float min, float max, const float info[4],
                        float *expansion_out, float *height_out) {
	const float height = max - min;
	std::vector<float> areas = size_areas(min, height, xyz, n, cls, 2); // crown points only (:299-301)
	const size_t skip = as_usize((info[3] - min) / SLICE_WIDTH);
	bool have = false;
	float crown_area = 0.0f;
	for (size_t i = skip; i < areas.size(); ++i) // max_by(total_cmp): the LAST maximum
		if (!have || total_cmp(areas[i], crown_area) >= 0) {
			crown_area = areas[i];
			have = true;
		}
	if (!have) crown_area = 0.0f;
	const float crown_diameter = approximate_diameter(crown_area);
	for (float &a : areas) a = approximate_diameter(a) / crown_diameter;
	for (uint64_t i = 0; i < n; ++i) {
		const float y = xyz[3 * i + 1];
		const size_t idx = as_usize((y - min) / SLICE_WIDTH);
		expansion_out[i] = idx < areas.size() ? areas[idx] : 0.0f;
		height_out[i] = (y - min) / height;
	}
}

/* DisplayData::new (interactive/mod.rs:1002-1039) for one segment */
void orc_display_values(const uint8_t *cls, const float *curve, const float *expansion, const float *height, uint64_t n,
                        uint32_t *cls_val, uint32_t *curve_val, uint32_t *exp_val, uint32_t *height_val) {
	bool have = false;
	float max_expansion = 0.0f; // unwrap_or_default
	for (uint64_t i = 0; i < n; ++i)
		if (!have || total_cmp(expansion[i], max_expansion) >= 0) {
			max_expansion = expansion[i];
			have = true;
		}
	for (uint64_t i = 0; i < n; ++i) {
		exp_val[i] = orc_map_to_u32(expansion[i] / max_expansion);
		curve_val[i] = orc_map_to_u32(curve[i]);
		height_val[i] = orc_map_to_u32(height[i]);
		cls_val[i] = cls[i] == 0 ? 0xFFFFFFFFu / 8u : (cls[i] == 1 ? 0xFFFFFFFFu / 8u * 3u : 0xFFFFFFFFu / 8u * 6u);
	}
}

} // extern "C"
