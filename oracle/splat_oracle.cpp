// splat_oracle.cpp -- CPU ORACLE (test infrastructure): software rasteriser restating the
// reference point-splat pipeline:
//   vertex stage    render/src/point_cloud.wgsl:37-59, base triangle render/src/point_cloud.rs:15-22
//   fragment stage  render/src/point_cloud.wgsl:72-80
//   fixed function  depth Depth32Float, compare Less, write on (render/src/point_cloud.rs:76-82),
//                   cleared to 1.0 (render/src/window.rs:153-159), wgpu clip volume 0 <= z <= w
//   camera          render/src/camera_3d.rs:16-18,59-63 (column-major mat4 pair)
//   eye dome        render/src/eye_dome.wgsl:37-67
// All three vertices of a splat share view-space z and w, so depth is constant over the splat
// and the covered set is the pixel centres inside the disc (inscribed circle of the base
// triangle, radius `scale` in view space).  With a projection whose x/y rows do not mix axes
// (nalgebra Perspective3) the disc maps to an axis-aligned screen ellipse; the closed form
// below is THE coverage rule shared with the HIP kernel (same float op order, no FMA).
// The reference has no ID buffer; id_out is the winning point's `value` (what the colour
// target encodes through the LUT, render/src/lookup.rs:20-21).  Depth ties: first drawn wins
// (compare Less), i.e. the smallest point index.
// Third-party, parity unpinned: nalgebra 0.33.0 Perspective3::new (Cargo.lock:1562-1563).
#include "treee_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline int clampi(float f, int lo, int hi) {
	if (!(f >= (float)lo)) return lo; // also NaN
	if (f >= (float)hi) return hi;
	return (int)f;
}

} // namespace

extern "C" {

void orc_perspective(float aspect, float fovy, float znear, float zfar, float out[16]) {
	// nalgebra Perspective3::new: identity; m33=0; m32=-1; set_fovy; set_aspect; set_znear_and_zfar
	std::memset(out, 0, 16 * sizeof(float));
	float m11 = 1.0f / std::tan(fovy / 2.0f);
	float m00 = m11 / aspect;
	out[0 * 4 + 0] = m00;
	out[1 * 4 + 1] = m11;
	out[2 * 4 + 2] = (zfar + znear) / (znear - zfar);
	out[3 * 4 + 2] = zfar * znear * 2.0f / (znear - zfar);
	out[2 * 4 + 3] = -1.0f;
	out[3 * 4 + 3] = 0.0f;
}

void orc_splat(const float *xyz, const uint32_t *value, uint64_t n, const float view[16],
               const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t W,
               uint32_t H, float *depth_out, uint32_t *id_out, uint32_t *index_out) {
	const size_t px_count = (size_t)W * H;
	std::vector<uint32_t> winner(px_count, 0xFFFFFFFFu);
	for (size_t i = 0; i < px_count; ++i) depth_out[i] = 1.0f; // window.rs:153-159
	const float fw = (float)W, fh = (float)H;
	for (uint64_t i = 0; i < n; ++i) {
		uint32_t val = value[i];
		if (val < vmin || vmax < val) continue; // point_cloud.wgsl:44-49
		float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
		float pos[4], clip[4];
		for (int r = 0; r < 4; ++r) // camera.view * vec4(p, 1.0)   (:51)
			pos[r] = ((view[0 + r] * x + view[4 + r] * y) + view[8 + r] * z) + view[12 + r];
		for (int r = 0; r < 4; ++r) // camera.proj * pos at the splat centre (:55)
			clip[r] = ((proj[0 + r] * pos[0] + proj[4 + r] * pos[1]) + proj[8 + r] * pos[2]) +
			          proj[12 + r] * pos[3];
		float w = clip[3];
		if (!(w > 0.0f)) continue;
		float depth = clip[2] / w;
		if (!(depth >= 0.0f && depth < 1.0f)) continue; // clip volume + Less against clear 1.0
		float ndc_x = clip[0] / w, ndc_y = clip[1] / w;
		float sx = (ndc_x * 0.5f + 0.5f) * fw;
		float sy = (0.5f - ndc_y * 0.5f) * fh;
		float rx = std::fabs(proj[0] * scale / w) * (0.5f * fw);
		float ry = std::fabs(proj[5] * scale / w) * (0.5f * fh);
		if (!(rx > 0.0f && ry > 0.0f)) continue;
		int x0 = clampi(std::floor(sx - rx - 0.5f) - 1.0f, 0, (int)W);
		int x1 = clampi(std::ceil(sx + rx - 0.5f) + 2.0f, 0, (int)W);
		int y0 = clampi(std::floor(sy - ry - 0.5f) - 1.0f, 0, (int)H);
		int y1 = clampi(std::ceil(sy + ry - 0.5f) + 2.0f, 0, (int)H);
		for (int py = y0; py < y1; ++py) {
			float v = (((float)py + 0.5f) - sy) / ry;
			float vv = v * v;
			for (int px = x0; px < x1; ++px) {
				float u = (((float)px + 0.5f) - sx) / rx;
				if (!(u * u + vv < 1.0f)) continue; // discard outside unit disc (:74-76)
				size_t o = (size_t)py * W + px;
				if (depth < depth_out[o]) { // Less; equal depth keeps the earlier draw
					depth_out[o] = depth;
					winner[o] = (uint32_t)i;
				}
			}
		}
	}
	for (size_t o = 0; o < px_count; ++o) {
		if (id_out) id_out[o] = winner[o] == 0xFFFFFFFFu ? 0u : value[winner[o]];
		if (index_out) index_out[o] = winner[o];
	}
}

void orc_eye_dome(const float *depth, uint32_t W, uint32_t H, float strength, float *alpha_out) {
	auto get_depth = [&](int x, int y) -> float { // eye_dome.wgsl:58-67, clamp-to-edge sampling
		x = x < 0 ? 0 : (x >= (int)W ? (int)W - 1 : x);
		y = y < 0 ? 0 : (y >= (int)H ? (int)H - 1 : y);
		const float near = 0.1f, far = 10000.0f;
		float d = depth[(size_t)y * W + x];
		if (d >= 1.0f) return 1.0f;
		return (2.0f * near) / (far + near - d * (far - near));
	};
	for (int y = 0; y < (int)H; ++y)
		for (int x = 0; x < (int)W; ++x) {
			float d = get_depth(x, y);
			float a;
			if (d == 1.0f) {
				a = 0.0f;
			} else {
				float m = d;
				m = std::fmax(m, get_depth(x - 1, y));
				m = std::fmax(m, get_depth(x + 1, y));
				m = std::fmax(m, get_depth(x, y - 1));
				m = std::fmax(m, get_depth(x, y + 1));
				m = std::fmin(m, d + strength);
				a = (m - d) / strength;
			}
			alpha_out[(size_t)y * W + x] = a;
		}
}

} // extern "C"
