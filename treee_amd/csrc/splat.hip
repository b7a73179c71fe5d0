// splat.hip -- compute point-splat into a depth + id buffer, replacing the instanced-triangle
// pipeline of the `render` crate:
//   vertex    render/src/point_cloud.wgsl:37-59   (value-range cull, view*p, billboard, proj)
//   fragment  render/src/point_cloud.wgsl:72-80   (unit-disc discard)
//   state     render/src/point_cloud.rs:76-82     (Depth32Float, Less, write), window.rs:153-159
//             (clear 1.0), wgpu clip volume 0 <= z <= w
//   eye dome  render/src/eye_dome.wgsl:37-67
// Coverage rule, float op order and tie rule are exactly those of oracle/splat_oracle.cpp.
//
// MI355X design: one lane per point; the depth test is a 64-bit atomicMin on
// (depth_bits << 32 | point index) -- depth in [0,1) so IEEE bits order like unsigned ints,
// and the low word makes "first drawn wins" on equal depth.  A plain (possibly stale) load
// of the pixel filters out occluded fragments first: values only decrease, so a stale value
// is conservative.  Splats whose pixel box exceeds 256 pixels are deferred to a
// workgroup-per-splat kernel so one lane never serialises a huge disc.
#include "common.h"

#include <cstdlib>

using namespace treee;

namespace {

constexpr uint64_t PACKED_CLEAR = (0x3F800000ull << 32) | 0xFFFFFFFFull; // depth 1.0, no point
constexpr int LARGE_AREA = 256;

struct SplatParams {
	float view[16], proj[16];
	float scale;
	uint32_t vmin, vmax, w, h, index_base;
};

struct Foot { // screen footprint of one splat
	float depth, sx, sy, rx, ry;
	int x0, x1, y0, y1;
};

__device__ __forceinline__ int clampi(float f, int lo, int hi) {
	if (!(f >= (float)lo)) return lo;
	if (f >= (float)hi) return hi;
	return (int)f;
}

__device__ __forceinline__ bool footprint(const SplatParams &P, float x, float y, float z, uint32_t val, Foot &f) {
	if (val < P.vmin || P.vmax < val) return false; // point_cloud.wgsl:44-49
	float pos[4], clip[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) pos[r] = ((P.view[0 + r] * x + P.view[4 + r] * y) + P.view[8 + r] * z) + P.view[12 + r];
#pragma unroll
	for (int r = 0; r < 4; ++r)
		clip[r] = ((P.proj[0 + r] * pos[0] + P.proj[4 + r] * pos[1]) + P.proj[8 + r] * pos[2]) + P.proj[12 + r] * pos[3];
	const float w = clip[3];
	if (!(w > 0.0f)) return false;
	f.depth = clip[2] / w;
	if (!(f.depth >= 0.0f && f.depth < 1.0f)) return false;
	const float fw = (float)P.w, fh = (float)P.h;
	const float ndc_x = clip[0] / w, ndc_y = clip[1] / w;
	f.sx = (ndc_x * 0.5f + 0.5f) * fw;
	f.sy = (0.5f - ndc_y * 0.5f) * fh;
	f.rx = fabsf(P.proj[0] * P.scale / w) * (0.5f * fw);
	f.ry = fabsf(P.proj[5] * P.scale / w) * (0.5f * fh);
	if (!(f.rx > 0.0f && f.ry > 0.0f)) return false;
	// Pixel px is covered only if lower < px < upper with lower = sx - rx - 0.5, upper = sx + rx - 0.5 (|u| >= 1 in exact
	// arithmetic on the floats sx, rx survives the rounding of u and u * u: 1.0 is representable).  The box is the integers of
	// (lower - SLACK, upper + SLACK): the two float sums are within 0.008 px of the exact ones up to 64 K-pixel frames, SLACK is
	// twice that.  The coverage test below -- not this box -- decides: the frame does not depend on the box as long as it is a
	// superset.  (Until round 4 the box kept a whole pixel on either side: a splat of the benchmark frame -- radius 0.47 px --
	// tested 3 x 3 pixels to cover at most one.)
	constexpr float SLACK = 0.015625f;
	f.x0 = clampi(floorf((f.sx - f.rx - 0.5f) - SLACK) + 1.0f, 0, (int)P.w);
	f.x1 = clampi(ceilf((f.sx + f.rx - 0.5f) + SLACK), 0, (int)P.w);
	f.y0 = clampi(floorf((f.sy - f.ry - 0.5f) - SLACK) + 1.0f, 0, (int)P.h);
	f.y1 = clampi(ceilf((f.sy + f.ry - 0.5f) + SLACK), 0, (int)P.h);
	return f.x0 < f.x1 && f.y0 < f.y1;
}

__device__ __forceinline__ float row_v(const Foot &f, int py) { return (((float)py + 0.5f) - f.sy) / f.ry; }
// v = row_v(f, py): the same for a whole row, so the caller computes it (and its IEEE divide) once per row
__device__ __forceinline__ void plot(unsigned long long *packed, uint32_t w, const Foot &f, int px, int py, float v, unsigned long long mine) {
	const float u = (((float)px + 0.5f) - f.sx) / f.rx;
	if (!(u * u + v * v < 1.0f)) return; // point_cloud.wgsl:74-76
	unsigned long long *p = packed + (size_t)py * w + px;
	if (mine < *p) atomicMin(p, mine); // stale read is conservative: values only decrease
}

__global__ __launch_bounds__(256) void k_splat_clear(unsigned long long *packed, uint32_t count) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < count) packed[i] = PACKED_CLEAR;
}

// REC = false: the caller's cloud, float xyz[n][3] (the layout of &[Point3<f32>]), index = position.
// REC = true: the records an octree already holds in Morton order, float4 (x, y, z, bitcast(original index)): one 16-byte
// load per lane instead of three strided 4-byte ones, and neighbouring lanes draw neighbouring pixels (their atomics meet
// in the same L2 lines).  The packed word carries the ORIGINAL index either way, so "first drawn wins" still means the
// smallest original index (render/src/point_cloud.rs:76-82) and the frame does not depend on the order points are drawn in.
template <bool REC> __device__ __forceinline__ bool load_point(const void *__restrict__ pts, const uint32_t *__restrict__ value, bool use_value, uint32_t i,
                                                              const SplatParams &P, Foot &f, uint32_t &orig) {
	float x, y, z;
	if (REC) {
		const float4 p = ((const float4 *)pts)[i];
		x = p.x, y = p.y, z = p.z;
		orig = __float_as_uint(p.w);
		if (orig == 0xFFFFFFFFu) return false; // a padding record of a segmented kNN structure (treee_splat_knn_dev): no point, no value
	} else {
		const float *xyz = (const float *)pts;
		x = xyz[3ull * i + 0], y = xyz[3ull * i + 1], z = xyz[3ull * i + 2];
		orig = i;
	}
	return footprint(P, x, y, z, use_value ? value[orig] : P.vmin, f); // (nothing culled: the values are not read at all)
}

template <bool REC>
__global__ __launch_bounds__(256) void k_splat_points(const void *__restrict__ pts, const uint32_t *__restrict__ value, bool use_value, uint32_t n,
                                                       SplatParams P, unsigned long long *packed,
                                                       uint32_t *__restrict__ large_count, uint32_t *__restrict__ large_list) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	Foot f;
	uint32_t orig;
	if (!load_point<REC>(pts, value, use_value, i, P, f, orig)) return;
	if ((f.x1 - f.x0) * (f.y1 - f.y0) > LARGE_AREA) {
		large_list[atomicAdd(large_count, 1u)] = i;
		return;
	}
	const unsigned long long mine = ((unsigned long long)__float_as_uint(f.depth) << 32) | (unsigned long long)(P.index_base + orig);
	for (int py = f.y0; py < f.y1; ++py) {
		const float v = row_v(f, py);
		if (!(v * v < 1.0f)) continue; // (u * u >= 0: no pixel of this row can pass; NaN rows never pass either)
		for (int px = f.x0; px < f.x1; ++px) plot(packed, P.w, f, px, py, v, mine);
	}
}

// The sorted records once more, with the depth test done in LDS first (cdna_hip_programming.md, Guideline 12: reduce in
// the workgroup, then one global atomic per destination).  256 Morton-consecutive points are a compact blob in space and,
// unless the camera sits inside it, a few dozen pixels on the screen: the workgroup takes the bounding box of its
// footprints, resolves all fragments that fall in it with ds_min_u64 on a tile of at most TILE_PX pixels, and sends one
// plain-load test + atomicMin per TOUCHED PIXEL to the frame -- instead of one per fragment, which is what makes drawing
// in sorted order slower than drawing in input order (same-pixel atomics of neighbouring lanes serialise at the memory
// side).  A workgroup whose box is larger draws the ordinary way.  min over (depth, original index) is order-free: the
// frame is the same, bit for bit.
#ifndef TREEE_SPLAT_PPT
#define TREEE_SPLAT_PPT 2 // (4 until round 4: with footprint boxes that are no wider than they must be, 1 / 2 / 3 / 4 / 6 / 8 records per thread: 0.169 / 0.156 / 0.158 / 0.165 / 0.207 / 0.26 ms)
#endif
#ifndef TREEE_SPLAT_TILE_PX
#define TREEE_SPLAT_TILE_PX 2048
#endif
constexpr int TILE_PX = TREEE_SPLAT_TILE_PX, TILE_PPT = TREEE_SPLAT_PPT; // pixels of the LDS tile; points per thread (256 * PPT consecutive records per workgroup)
__global__ __launch_bounds__(256) void k_splat_tile(const float4 *__restrict__ rec, const uint32_t *__restrict__ value, bool use_value, uint32_t n,
                                                     SplatParams P, unsigned long long *packed, uint32_t *__restrict__ large_count,
                                                     uint32_t *__restrict__ large_list) {
	__shared__ unsigned long long s_tile[TILE_PX];
	__shared__ int s_box[4][4];
	Foot f[TILE_PPT];
	unsigned long long mine[TILE_PPT];
	bool draw[TILE_PPT];
	int bx0 = 0x7FFFFFFF, by0 = 0x7FFFFFFF, bx1 = -1, by1 = -1;
#pragma unroll
	for (int k = 0; k < TILE_PPT; ++k) { // (the PPT records of a thread are 256 apart: every load of the wave is one contiguous KiB)
		const uint32_t i = (blockIdx.x * TILE_PPT + k) * 256u + threadIdx.x;
		uint32_t orig = 0;
		draw[k] = i < n && load_point<true>(rec, value, use_value, i, P, f[k], orig);
		if (draw[k] && (f[k].x1 - f[k].x0) * (f[k].y1 - f[k].y0) > LARGE_AREA) {
			large_list[atomicAdd(large_count, 1u)] = i;
			draw[k] = false;
		}
		mine[k] = draw[k] ? ((unsigned long long)__float_as_uint(f[k].depth) << 32) | (unsigned long long)(P.index_base + orig) : PACKED_CLEAR;
		if (draw[k]) {
			bx0 = min(bx0, f[k].x0), by0 = min(by0, f[k].y0);
			bx1 = max(bx1, f[k].x1), by1 = max(by1, f[k].y1);
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		bx0 = min(bx0, __shfl_xor(bx0, o, 64));
		by0 = min(by0, __shfl_xor(by0, o, 64));
		bx1 = max(bx1, __shfl_xor(bx1, o, 64));
		by1 = max(by1, __shfl_xor(by1, o, 64));
	}
	if ((threadIdx.x & 63u) == 0) {
		int *b = s_box[threadIdx.x >> 6];
		b[0] = bx0, b[1] = by0, b[2] = bx1, b[3] = by1;
	}
	__syncthreads();
	bx0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0]));
	by0 = min(min(s_box[0][1], s_box[1][1]), min(s_box[2][1], s_box[3][1]));
	bx1 = max(max(s_box[0][2], s_box[1][2]), max(s_box[2][2], s_box[3][2]));
	by1 = max(max(s_box[0][3], s_box[1][3]), max(s_box[2][3], s_box[3][3]));
	if (bx1 <= bx0) return; // nothing to draw (uniform)
	const int bw = bx1 - bx0, area = bw * (by1 - by0);
	const bool tiled = area <= TILE_PX; // (uniform) a larger box is drawn fragment by fragment, as k_splat_points does
	// (rows padded to a power of two -- slot -> row / column by shift and mask instead of a division -- measured slower:
	// 0.222 vs 0.201 ms for the stage; the padding costs more tile slots than the divisions cost instructions)
	if (tiled) {
		for (int t = threadIdx.x; t < area; t += 256) s_tile[t] = PACKED_CLEAR;
		__syncthreads();
	}
#pragma unroll
	for (int k = 0; k < TILE_PPT; ++k) {
		if (!draw[k]) continue;
		for (int py = f[k].y0; py < f[k].y1; ++py) {
			const float v = row_v(f[k], py);
			if (!(v * v < 1.0f)) continue;
			for (int px = f[k].x0; px < f[k].x1; ++px) {
				if (!tiled) {
					plot(packed, P.w, f[k], px, py, v, mine[k]);
					continue;
				}
				const float u = (((float)px + 0.5f) - f[k].sx) / f[k].rx;
				if (!(u * u + v * v < 1.0f)) continue; // point_cloud.wgsl:74-76
				unsigned long long *q = &s_tile[(py - by0) * bw + (px - bx0)];
#ifdef TREEE_SPLAT_LDS_PRETEST
				if (mine[k] < *q) atomicMin(q, mine[k]);
#else
				atomicMin(q, mine[k]);
#endif
			}
		}
	}
	if (!tiled) return;
	__syncthreads();
	// slot t = row * bw + col, t = threadIdx.x + 256 i: one division per thread, then (row, col) step by (256 / bw, 256 % bw)
	const int step_row = 256 / bw, step_col = 256 % bw; // (uniform)
	int row = (int)threadIdx.x / bw, col = (int)threadIdx.x % bw;
	for (int t = threadIdx.x; t < area; t += 256, row += step_row, col += step_col) {
		if (col >= bw) col -= bw, ++row;
		const unsigned long long best = s_tile[t];
		if (best == PACKED_CLEAR) continue;
		unsigned long long *q = packed + (size_t)(by0 + row) * P.w + (bx0 + col);
#ifdef TREEE_SPLAT_PRETEST
		if (best < *q) atomicMin(q, best);
#else
		atomicMin(q, best); // (no plain-load test first: the dependent load costs more than the atomics it saves -- 0.168 vs 0.187 ms)
#endif
	}
}

template <bool REC>
__global__ __launch_bounds__(256) void k_splat_large(const void *__restrict__ pts, const uint32_t *__restrict__ value, bool use_value,
                                                      SplatParams P, unsigned long long *packed,
                                                      const uint32_t *__restrict__ large_count, const uint32_t *__restrict__ large_list) {
	const uint32_t total = *large_count;
	for (uint32_t s = blockIdx.x; s < total; s += gridDim.x) {
		const uint32_t i = large_list[s];
		Foot f;
		uint32_t orig;
		if (!load_point<REC>(pts, value, use_value, i, P, f, orig)) continue;
		const unsigned long long mine = ((unsigned long long)__float_as_uint(f.depth) << 32) | (unsigned long long)(P.index_base + orig);
		const int bw = f.x1 - f.x0;
		const int area = bw * (f.y1 - f.y0);
		for (int t = threadIdx.x; t < area; t += 256) plot(packed, P.w, f, f.x0 + t % bw, f.y0 + t / bw, row_v(f, f.y0 + t / bw), mine);
	}
}

__global__ __launch_bounds__(256) void k_splat_resolve(const unsigned long long *__restrict__ packed, const uint32_t *__restrict__ value,
                                                        uint32_t n_values, uint32_t count, float *__restrict__ depth, uint32_t *__restrict__ id) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= count) return;
	const unsigned long long v = packed[i];
	const uint32_t idx = (uint32_t)v;
	if (depth) depth[i] = __uint_as_float((uint32_t)(v >> 32));
	if (id) id[i] = (idx == 0xFFFFFFFFu || idx >= n_values) ? 0u : value[idx];
}

// render/src/eye_dome.wgsl:37-67
__device__ __forceinline__ float ed_depth(const float *__restrict__ depth, int w, int h, int x, int y) {
	x = x < 0 ? 0 : (x >= w ? w - 1 : x);
	y = y < 0 ? 0 : (y >= h ? h - 1 : y);
	const float near = 0.1f, far = 10000.0f;
	const float d = depth[(size_t)y * w + x];
	if (d >= 1.0f) return 1.0f;
	return (2.0f * near) / (far + near - d * (far - near));
}

__device__ __forceinline__ float ed_lin(float d) { // linearised depth of one texel (eye_dome.wgsl:37-45)
	const float near = 0.1f, far = 10000.0f;
	if (d >= 1.0f) return 1.0f;
	return (2.0f * near) / (far + near - d * (far - near));
}

// one thread = 4 horizontally adjacent pixels: three 16-byte row loads (above, centre, below) + the two scalars
// left and right of the quad; 64 x 4 threads = a 256 x 4 pixel tile per workgroup.  Same arithmetic as the scalar
// form (ed_depth), which handles widths that are not a multiple of 4.
__global__ __launch_bounds__(256) void k_eye_dome4(const float *__restrict__ depth, uint32_t w, uint32_t h, float strength, float *__restrict__ alpha) {
	const uint32_t x0 = (blockIdx.x * 64u + (threadIdx.x & 63u)) * 4u, y = blockIdx.y * 4u + (threadIdx.x >> 6);
	if (x0 >= w || y >= h) return;
	const uint32_t yu = y == 0 ? 0 : y - 1, yd = y + 1 >= h ? h - 1 : y + 1;
	const float4 c = *(const float4 *)(depth + (size_t)y * w + x0);
	const float4 u = *(const float4 *)(depth + (size_t)yu * w + x0);
	const float4 dn = *(const float4 *)(depth + (size_t)yd * w + x0);
	const float lf = depth[(size_t)y * w + (x0 == 0 ? 0 : x0 - 1)], rt = depth[(size_t)y * w + (x0 + 4 >= w ? w - 1 : x0 + 4)];
	const float cc[6] = {ed_lin(lf), ed_lin(c.x), ed_lin(c.y), ed_lin(c.z), ed_lin(c.w), ed_lin(rt)};
	const float uu[4] = {ed_lin(u.x), ed_lin(u.y), ed_lin(u.z), ed_lin(u.w)};
	const float dd[4] = {ed_lin(dn.x), ed_lin(dn.y), ed_lin(dn.z), ed_lin(dn.w)};
	float out[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const float d = cc[j + 1];
		float a = 0.0f;
		if (d != 1.0f) {
			float m = d;
			m = fmaxf(m, cc[j]);
			m = fmaxf(m, cc[j + 2]);
			m = fmaxf(m, uu[j]);
			m = fmaxf(m, dd[j]);
			m = fminf(m, d + strength);
			a = (m - d) / strength;
		}
		out[j] = a;
	}
	*(float4 *)(alpha + (size_t)y * w + x0) = make_float4(out[0], out[1], out[2], out[3]);
}

__global__ __launch_bounds__(256) void k_eye_dome(const float *__restrict__ depth, uint32_t w, uint32_t h, float strength, float *__restrict__ alpha) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= w * h) return;
	const int x = i % w, y = i / w;
	const float d = ed_depth(depth, w, h, x, y);
	float a = 0.0f;
	if (d != 1.0f) {
		float m = d;
		m = fmaxf(m, ed_depth(depth, w, h, x - 1, y));
		m = fmaxf(m, ed_depth(depth, w, h, x + 1, y));
		m = fmaxf(m, ed_depth(depth, w, h, x, y - 1));
		m = fmaxf(m, ed_depth(depth, w, h, x, y + 1));
		m = fminf(m, d + strength);
		a = (m - d) / strength;
	}
	alpha[i] = a;
}

int32_t check_proj(const float proj[16]) {
	// x row must not read y, y row must not read x, z/w rows must not read x or y: then a
	// view-plane disc is an axis-aligned screen ellipse with constant depth
	if (proj[4] != 0.0f || proj[1] != 0.0f || proj[2] != 0.0f || proj[6] != 0.0f || proj[3] != 0.0f || proj[7] != 0.0f)
		return set_err(TREEE_ERR_UNSUPPORTED, "projection matrix mixes x/y into other rows; only perspective/orthographic forms are supported");
	return TREEE_OK;
}

} // namespace

extern "C" {

// pts: float xyz[n][3] (rec = false) or float4 records in any order whose .w is the point's index into d_value (rec = true)
static int32_t splat_any(treee_ctx *ctx, const void *d_pts, bool rec, const uint32_t *d_value, uint64_t n, uint64_t n_values, const float view[16],
                         const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h, uint64_t *d_packed,
                         int32_t accumulate, uint32_t index_base, float *d_depth, uint32_t *d_id) {
	if (n && !d_value && (vmin != 0u || vmax != 0xFFFFFFFFu || d_id))
		return set_err(TREEE_ERR_INVALID, "treee_splat_dev: d_value may only be null when nothing is culled (vmin = 0, vmax = u32::MAX) and no id buffer is asked for");
	if (w == 0 || h == 0 || (uint64_t)w * h >= (1ull << 31)) return set_err(TREEE_ERR_INVALID, "bad framebuffer size %ux%u", w, h);
	if (n >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "n=%llu exceeds 2^31-1 points per call", (unsigned long long)n);
	if (n_values >= 0xFFFFFFFFull - index_base) return set_err(TREEE_ERR_UNSUPPORTED, "point index does not fit in 32 bits");
	if (index_base != 0 && d_id)
		return set_err(TREEE_ERR_INVALID, "d_id_out needs index_base == 0; resolve multi-chunk draws with treee_splat_resolve_dev");
	TREEE_TRY(check_proj(proj));
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	const uint32_t px = w * h;
	unsigned long long *packed = (unsigned long long *)d_packed;
	if (!packed) {
		TREEE_TRY(ctx->alloc_t(px, &packed));
		accumulate = 0;
	}
	SplatParams P;
	memcpy(P.view, view, 64);
	memcpy(P.proj, proj, 64);
	P.scale = scale;
	P.vmin = vmin;
	P.vmax = vmax;
	P.w = w;
	P.h = h;
	P.index_base = index_base;
	hipStream_t st = ctx->stream;
	if (!accumulate) hipLaunchKernelGGL(k_splat_clear, dim3(div_up(px, 256)), dim3(256), 0, st, packed, px);
	if (n) {
		uint32_t *d_lc = nullptr, *d_ll = nullptr;
		TREEE_TRY(ctx->alloc_t(1, &d_lc));
		TREEE_TRY(ctx->alloc_t(n, &d_ll));
		TREEE_HIP(hipMemsetAsync(d_lc, 0, 4, st));
		const bool use_value = d_value && !(vmin == 0u && vmax == 0xFFFFFFFFu); // the cull of point_cloud.wgsl:44-49 can never fire otherwise
		if (rec) {
			static const bool no_tile = getenv("TREEE_SPLAT_NO_TILE") != nullptr; // A/B hook: fragment-by-fragment global atomics
			if (no_tile)
				hipLaunchKernelGGL(k_splat_points<true>, dim3(div_up(n, 256)), dim3(256), 0, st, d_pts, d_value, use_value, (uint32_t)n, P, packed, d_lc, d_ll);
			else
				hipLaunchKernelGGL(k_splat_tile, dim3(div_up(n, 256 * TILE_PPT)), dim3(256), 0, st, (const float4 *)d_pts, d_value, use_value, (uint32_t)n, P, packed, d_lc, d_ll);
			hipLaunchKernelGGL(k_splat_large<true>, dim3(ctx->num_cus * 4), dim3(256), 0, st, d_pts, d_value, use_value, P, packed, d_lc, d_ll);
		} else {
			hipLaunchKernelGGL(k_splat_points<false>, dim3(div_up(n, 256)), dim3(256), 0, st, d_pts, d_value, use_value, (uint32_t)n, P, packed, d_lc, d_ll);
			hipLaunchKernelGGL(k_splat_large<false>, dim3(ctx->num_cus * 4), dim3(256), 0, st, d_pts, d_value, use_value, P, packed, d_lc, d_ll);
		}
	}
	if (d_depth || d_id)
		hipLaunchKernelGGL(k_splat_resolve, dim3(div_up(px, 256)), dim3(256), 0, st, packed, d_value, (uint32_t)n_values, px, d_depth, d_id);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_splat_dev(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_value, uint64_t n, const float view[16],
                        const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h,
                        uint64_t *d_packed, int32_t accumulate, uint32_t index_base, float *d_depth, uint32_t *d_id) {
	if (!ctx || !view || !proj || (n && !d_xyz)) return set_err(TREEE_ERR_INVALID, "treee_splat_dev: null argument");
	return splat_any(ctx, d_xyz, false, d_value, n, n, view, proj, scale, vmin, vmax, w, h, d_packed, accumulate, index_base, d_depth, d_id);
}

int32_t treee_splat_octree_dev(treee_ctx *ctx, const treee_octree *o, const uint32_t *d_value, const float view[16], const float proj[16],
                               float scale, uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h, uint64_t *d_packed, int32_t accumulate,
                               uint32_t index_base, float *d_depth, uint32_t *d_id) {
	if (!ctx || !o || !view || !proj) return set_err(TREEE_ERR_INVALID, "treee_splat_octree_dev: null argument");
	std::shared_ptr<SortedCloud> sc = octree_sorted_cloud(o);
	if (sc->device != ctx->device) return set_err(TREEE_ERR_INVALID, "treee_splat_octree_dev: context of another device");
	TREEE_TRY(use_begin(ctx, sc->owner, &sc->sync));
	const int32_t rc = splat_any(ctx, sc->d_sorted, true, d_value, sc->n, octree_is_shard(o) ? 0xFFFFFFFEull - index_base : sc->n, view, proj, scale, vmin, vmax,
	                             w, h, d_packed, accumulate, index_base, d_depth, d_id);
	const int32_t e = use_end(ctx, sc->owner, &sc->sync);
	return rc != TREEE_OK ? rc : e;
}

int32_t treee_splat_knn_dev(treee_ctx *ctx, const treee_knn *t, const uint32_t *d_value, const float view[16], const float proj[16], float scale,
                            uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h, uint64_t *d_packed, int32_t accumulate, uint32_t index_base,
                            float *d_depth, uint32_t *d_id) {
	if (!ctx || !t || !view || !proj) return set_err(TREEE_ERR_INVALID, "treee_splat_knn_dev: null argument");
	uint64_t n_points = 0;
	bool segmented = false;
	std::shared_ptr<SortedCloud> sc = knn_sorted_cloud(t, &n_points, &segmented);
	if (sc->device != ctx->device) return set_err(TREEE_ERR_INVALID, "treee_splat_knn_dev: context of another device");
	TREEE_TRY(use_begin(ctx, sc->owner, &sc->sync));
	const int32_t rc = splat_any(ctx, sc->d_sorted, true, d_value, sc->n, n_points, view, proj, scale, vmin, vmax, w, h, d_packed, accumulate, index_base, d_depth,
	                             d_id);
	const int32_t e = use_end(ctx, sc->owner, &sc->sync);
	return rc != TREEE_OK ? rc : e;
}

int32_t treee_splat_resolve_dev(treee_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_value, uint64_t n_values, uint32_t w,
                                uint32_t h, float *d_depth, uint32_t *d_id) {
	if (!ctx || !d_packed) return set_err(TREEE_ERR_INVALID, "treee_splat_resolve_dev: null argument");
	if (d_id && n_values && !d_value) return set_err(TREEE_ERR_INVALID, "treee_splat_resolve_dev: value is null");
	TREEE_HIP(hipSetDevice(ctx->device));
	const uint32_t px = w * h;
	hipLaunchKernelGGL(k_splat_resolve, dim3(div_up(px, 256)), dim3(256), 0, ctx->stream, (const unsigned long long *)d_packed, d_value,
	                   (uint32_t)n_values, px, d_depth, d_id);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_splat(const float *xyz, const uint32_t *value, uint64_t n, const float view[16], const float proj[16], float scale,
                    uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h, float *depth_out, uint32_t *id_out) {
	if (!view || !proj || (n && (!xyz || !value))) return set_err(TREEE_ERR_INVALID, "treee_splat: null argument");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	const uint64_t px = (uint64_t)w * h;
	DevBuf<float> d_xyz, d_depth;
	DevBuf<uint32_t> d_val, d_id;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_TRY(d_val.alloc(n));
	TREEE_TRY(d_depth.alloc(px));
	TREEE_TRY(d_id.alloc(px));
	if (n) {
		TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
		TREEE_HIP(hipMemcpyAsync(d_val.p, value, n * 4, hipMemcpyHostToDevice, ctx->stream));
	}
	TREEE_TRY(treee_splat_dev(ctx, d_xyz.p, d_val.p, n, view, proj, scale, vmin, vmax, w, h, nullptr, 0, 0, d_depth.p, d_id.p));
	if (depth_out) TREEE_HIP(hipMemcpyAsync(depth_out, d_depth.p, px * 4, hipMemcpyDeviceToHost, ctx->stream));
	if (id_out) TREEE_HIP(hipMemcpyAsync(id_out, d_id.p, px * 4, hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_perspective(float aspect, float fovy, float znear, float zfar, float out[16]) {
	if (!out) return set_err(TREEE_ERR_INVALID, "treee_perspective: out is null");
	// nalgebra 0.33 Perspective3::new: identity, m33=0, m32=-1, set_fovy, set_aspect, set_znear_and_zfar
	memset(out, 0, 64);
	const float m11 = 1.0f / tanf(fovy / 2.0f);
	out[0] = m11 / aspect;
	out[5] = m11;
	out[10] = (zfar + znear) / (znear - zfar);
	out[14] = zfar * znear * 2.0f / (znear - zfar);
	out[11] = -1.0f;
	return TREEE_OK;
}

int32_t treee_eye_dome_dev(treee_ctx *ctx, const float *d_depth, uint32_t w, uint32_t h, float strength, float *d_alpha) {
	if (!ctx || !d_depth || !d_alpha) return set_err(TREEE_ERR_INVALID, "treee_eye_dome_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	if (w % 4 == 0 && ((uintptr_t)d_depth % 16) == 0 && ((uintptr_t)d_alpha % 16) == 0)
		hipLaunchKernelGGL(k_eye_dome4, dim3(div_up(w, 256), div_up(h, 4)), dim3(256), 0, ctx->stream, d_depth, w, h, strength, d_alpha);
	else
		hipLaunchKernelGGL(k_eye_dome, dim3(div_up((uint64_t)w * h, 256)), dim3(256), 0, ctx->stream, d_depth, w, h, strength, d_alpha);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

} // extern "C"
