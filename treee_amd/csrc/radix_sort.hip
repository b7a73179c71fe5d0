// radix_sort.hip -- Morton key generation, LSD radix sort (8-bit digits, LDS digit bins,
// wave64 ballot ranking), exclusive scan, cube reduction, sorted gather.
//
// Build-defined stage: the reference has no Morton / sort code;
// the only reference hook is the cube (corner,size) of render/src/camera_3d.rs:20-51.  The
// GPU analogue of the y-slice binning (treee/src/loading.rs:97-101) reuses the same sort.
// Bit-exact against oracle/octree_oracle.cpp (stable sort by key, ties by original index).
//
// Sort structure per 8-bit pass (all HBM-bound, integer only):
//   rs_hist      read keys                     -> table[digit][tile]          4 B/key
//   rs_scan_rows exclusive scan of each digit row, digit totals               (tiny)
//   rs_scatter   read key+val, rank in-tile with 8 wave ballots per key (match-any on the
//                digit), stage the tile sorted by digit in LDS, write runs    16 B/key
#include "common.h"

#include <cstdlib>

namespace treee {

constexpr int RS_BLOCK = 256;
#ifndef TREEE_RS_ITEMS
#define TREEE_RS_ITEMS 20 // keys per thread; a multiple of 4 (16-byte loads).  12 / 16 / 20: structure stage 1.706 / 1.633 / 1.590 ms (round 3; round 2 had 24 at 1.05 vs 1.02 for 20)
#endif
constexpr int RS_ITEMS = TREEE_RS_ITEMS;
constexpr int RS_TILE = RS_BLOCK * RS_ITEMS; // 5120 keys per workgroup
static_assert(RS_ITEMS % 4 == 0, "rs_hist loads four keys at a time");

// ---- Morton keys (quant10 / spread3: common.h) ----------------------------------------------
__global__ __launch_bounds__(256) void k_morton_keys(const float *__restrict__ xyz, uint32_t n,
                                                      float cx, float cy, float cz, float size,
                                                      uint32_t *__restrict__ keys) {
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	float x = xyz[3ull * i + 0], y = xyz[3ull * i + 1], z = xyz[3ull * i + 2];
	keys[i] = spread3(quant10(x, cx, size)) | (spread3(quant10(y, cy, size)) << 1) |
	          (spread3(quant10(z, cz, size)) << 2);
}

__global__ __launch_bounds__(256) void k_morton_keys_cube(const float *__restrict__ xyz, uint32_t n, const float *__restrict__ cube,
                                                           uint32_t *__restrict__ keys) {
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float cx = cube[0], cy = cube[1], cz = cube[2], size = cube[3];
	float x = xyz[3ull * i + 0], y = xyz[3ull * i + 1], z = xyz[3ull * i + 2];
	keys[i] = spread3(quant10(x, cx, size)) | (spread3(quant10(y, cy, size)) << 1) | (spread3(quant10(z, cz, size)) << 2);
}

// `out` may be `morton` itself when idx is null (one element per thread): neither is declared restrict
__constant__ uint16_t c_hilbert_lut[24 * 8] = {TREEE_HILBERT_LUT_INIT};
__global__ __launch_bounds__(256) void k_hilbert_keys(const uint32_t *morton, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *out) {
	__shared__ uint16_t s_lut[24 * 8];
	if (threadIdx.x < 24 * 8) s_lut[threadIdx.x] = c_hilbert_lut[threadIdx.x];
	__syncthreads();
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) out[i] = hilbert_of_morton_lut(morton[idx ? idx[i] : i], s_lut);
}

int32_t hilbert_keys_dev(treee_ctx *ctx, const uint32_t *d_morton, const uint32_t *d_idx, uint64_t n, uint32_t *d_out) {
	if (n == 0) return TREEE_OK;
	hipLaunchKernelGGL(k_hilbert_keys, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_morton, d_idx, (uint32_t)n, d_out);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t morton_keys_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float corner[3],
                        float size, uint32_t *d_keys) {
	if (n == 0) return TREEE_OK;
	hipLaunchKernelGGL(k_morton_keys, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_xyz,
	                   (uint32_t)n, corner[0], corner[1], corner[2], size, d_keys);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

// ---- cube (bounding box) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bbox_partial(const float *__restrict__ xyz, uint32_t n,
                                                       float *__restrict__ partial /*[grid][6]*/) {
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			float v = xyz[3ull * i + a];
			mn[a] = fminf(mn[a], v);
			mx[a] = fmaxf(mx[a], v);
		}
	}
	__shared__ float s[4][6];
#pragma unroll
	for (int a = 0; a < 3; ++a) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			mn[a] = fminf(mn[a], __shfl_down(mn[a], o, 64));
			mx[a] = fmaxf(mx[a], __shfl_down(mx[a], o, 64));
		}
	}
	if (lane_id() == 0) {
		for (int a = 0; a < 3; ++a) {
			s[threadIdx.x >> 6][a] = mn[a];
			s[threadIdx.x >> 6][3 + a] = mx[a];
		}
	}
	__syncthreads();
	if (threadIdx.x < 6) {
		float v = s[0][threadIdx.x];
		for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
		partial[blockIdx.x * 6 + threadIdx.x] = v;
	}
}

// partials -> (corner.xyz, size) with the rules of bbox_dev below (largest extent; degenerate or non-finite -> 1 / 0)
__global__ __launch_bounds__(256) void k_bbox_final(const float *__restrict__ partial, uint32_t grid, float *__restrict__ cube) {
	__shared__ float s[4][6];
	float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
	for (uint32_t b = threadIdx.x; b < grid; b += 256u)
#pragma unroll
		for (int a = 0; a < 6; ++a) v[a] = a < 3 ? fminf(v[a], partial[b * 6 + a]) : fmaxf(v[a], partial[b * 6 + a]);
#pragma unroll
	for (int a = 0; a < 6; ++a)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float t = __shfl_down(v[a], o, 64);
			v[a] = a < 3 ? fminf(v[a], t) : fmaxf(v[a], t);
		}
	if (lane_id() == 0)
		for (int a = 0; a < 6; ++a) s[threadIdx.x >> 6][a] = v[a];
	__syncthreads();
	if (threadIdx.x == 0) {
		float mn[3], mx[3], size = 0.0f;
		for (int a = 0; a < 3; ++a) {
			mn[a] = fminf(fminf(s[0][a], s[1][a]), fminf(s[2][a], s[3][a]));
			mx[a] = fmaxf(fmaxf(s[0][3 + a], s[1][3 + a]), fmaxf(s[2][3 + a], s[3][3 + a]));
			const float e = mx[a] - mn[a];
			if (e > size) size = e;
		}
		if (!(size > 0.0f) || !(size < INFINITY)) size = 1.0f;
		for (int a = 0; a < 3; ++a) cube[a] = (mn[a] > -INFINITY && mn[a] < INFINITY) ? mn[a] : 0.0f;
		cube[3] = size;
	}
}

int32_t bbox_partials_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, float **d_partial_out, uint32_t *grid_out) {
	const uint32_t grid = n < 256 * 1024 ? div_up(n ? n : 1, 256) : 1024;
	float *d_partial = nullptr;
	TREEE_TRY(ctx->alloc_t(grid * 6, &d_partial));
	hipLaunchKernelGGL(k_bbox_partial, dim3(grid), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, d_partial);
	TREEE_LAUNCH_CHECK();
	*d_partial_out = d_partial;
	*grid_out = grid;
	return TREEE_OK;
}

static int32_t bbox_to_device(treee_ctx *ctx, const float *d_xyz, uint64_t n, float *d_cube) {
	uint32_t grid = n < 256 * 1024 ? div_up(n ? n : 1, 256) : 1024;
	float *d_partial = nullptr;
	TREEE_TRY(ctx->alloc_t(grid * 6, &d_partial));
	hipLaunchKernelGGL(k_bbox_partial, dim3(grid), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, d_partial);
	hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, ctx->stream, d_partial, grid, d_cube);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t cube_fetch_async(treee_ctx *ctx, SortedCloud *sc, float h_cube[4]) {
	if (sc->cube_pending) TREEE_HIP(hipMemcpyAsync(h_cube, sc->d_cube, 16, hipMemcpyDeviceToHost, ctx->stream));
	return TREEE_OK;
}

int32_t cube_to_host(treee_ctx *ctx, SortedCloud *sc) {
	if (!sc->cube_pending) return TREEE_OK;
	float *h = (float *)(ctx->h_scalars + 4); // pinned (see treee_ctx::h_scalars)
	TREEE_TRY(cube_fetch_async(ctx, sc, h));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	cube_commit(sc, h);
	return TREEE_OK;
}

void cube_commit(SortedCloud *sc, const float h_cube[4]) {
	if (!sc->cube_pending) return;
	for (int a = 0; a < 3; ++a) sc->corner[a] = h_cube[a];
	sc->size = h_cube[3];
	sc->cube_pending = false;
}

int32_t bbox_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, float corner_out[3], float *size_out) {
	uint32_t grid = n < 256 * 1024 ? div_up(n ? n : 1, 256) : 1024;
	float *d_partial = nullptr;
	TREEE_TRY(ctx->alloc_t(grid * 6, &d_partial));
	hipLaunchKernelGGL(k_bbox_partial, dim3(grid), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, d_partial);
	TREEE_LAUNCH_CHECK();
	HostBuf<float> h;
	TREEE_TRY(h.alloc((size_t)grid * 6));
	TREEE_HIP(hipMemcpyAsync(h.data(), d_partial, grid * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (uint32_t b = 0; b < grid; ++b)
		for (int a = 0; a < 3; ++a) {
			mn[a] = fminf(mn[a], h[b * 6 + a]);
			mx[a] = fmaxf(mx[a], h[b * 6 + 3 + a]);
		}
	float size = 0.0f;
	for (int a = 0; a < 3; ++a) {
		corner_out[a] = mn[a];
		float e = mx[a] - mn[a];
		if (e > size) size = e;
	}
	if (!(size > 0.0f) || !(size < INFINITY)) size = 1.0f;
	for (int a = 0; a < 3; ++a)
		if (!(corner_out[a] > -INFINITY && corner_out[a] < INFINITY)) corner_out[a] = 0.0f;
	*size_out = size;
	return TREEE_OK;
}

// ---- radix sort ----------------------------------------------------------------------------
#ifndef TREEE_RS_HSETS
#define TREEE_RS_HSETS 8 // 4 / 8 / 16: structure stage 1.600 / 1.586 / 1.589 ms
#endif
__global__ __launch_bounds__(RS_BLOCK) void rs_hist(const uint32_t *__restrict__ keys, uint32_t n,
                                                    int shift, uint32_t *__restrict__ table,
                                                    uint32_t ntiles) {
	constexpr int SETS = TREEE_RS_HSETS; // digit-bin sets: 4 = one per wave, 8 = one per half wave, 16 = one per quarter
	__shared__ uint32_t h[SETS][256];
	const int tid = threadIdx.x, w = tid / (256 / SETS);
#pragma unroll
	for (int i = 0; i < SETS; ++i) h[i][tid] = 0;
	__syncthreads();
	const uint32_t base = blockIdx.x * RS_TILE;
	if (base + RS_TILE <= n) { // full tile: 16-byte loads (the order inside the tile does not matter for a histogram)
		const uint4 *k4 = (const uint4 *)(keys + base);
#pragma unroll
		for (int i = 0; i < RS_ITEMS / 4; ++i) {
			const uint4 k = k4[i * RS_BLOCK + tid];
			atomicAdd(&h[w][(k.x >> shift) & 255u], 1u);
			atomicAdd(&h[w][(k.y >> shift) & 255u], 1u);
			atomicAdd(&h[w][(k.z >> shift) & 255u], 1u);
			atomicAdd(&h[w][(k.w >> shift) & 255u], 1u);
		}
	} else {
#pragma unroll
		for (int i = 0; i < RS_ITEMS; ++i) {
			uint32_t idx = base + i * RS_BLOCK + tid;
			if (idx < n) atomicAdd(&h[w][(keys[idx] >> shift) & 255u], 1u);
		}
	}
	__syncthreads();
	uint32_t sum = 0;
#pragma unroll
	for (int i = 0; i < SETS; ++i) sum += h[i][tid];
	table[(uint32_t)tid * ntiles + blockIdx.x] = sum;
}

// Morton keys of one radix tile AND the tile's histogram of the first digit (round 5: the first pass's rs_hist re-read the 4 B
// per point this kernel had just written -- 60 MB and a launch per 15 M points).  cube: (corner.xyz, size) on the device.
__global__ __launch_bounds__(RS_BLOCK) void k_morton_keys_hist(const float *__restrict__ xyz, uint32_t n, const float *__restrict__ cube, uint32_t *__restrict__ keys,
                                                               uint32_t *__restrict__ table, uint32_t ntiles) {
	constexpr int SETS = TREEE_RS_HSETS;
	__shared__ uint32_t h[SETS][256];
	const int tid = threadIdx.x, w = tid / (256 / SETS);
#pragma unroll
	for (int i = 0; i < SETS; ++i) h[i][tid] = 0;
	__syncthreads();
	const float cx = cube[0], cy = cube[1], cz = cube[2], size = cube[3];
	const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll 4
	for (int i = 0; i < RS_ITEMS; ++i) {
		const uint32_t idx = base + i * RS_BLOCK + tid;
		if (idx < n) {
			const uint32_t k = morton30(xyz[3ull * idx + 0], xyz[3ull * idx + 1], xyz[3ull * idx + 2], cx, cy, cz, size);
			keys[idx] = k;
			atomicAdd(&h[w][k & 255u], 1u);
		}
	}
	__syncthreads();
	uint32_t sum = 0;
#pragma unroll
	for (int i = 0; i < SETS; ++i) sum += h[i][tid];
	table[(uint32_t)tid * ntiles + blockIdx.x] = sum;
}

__global__ __launch_bounds__(256) void rs_scan_rows(uint32_t *__restrict__ table, uint32_t ntiles,
                                                     uint32_t *__restrict__ totals) {
	__shared__ uint32_t part[4];
	uint32_t *row = table + (size_t)blockIdx.x * ntiles;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < ntiles; base += 1024) {
		// 4 consecutive entries per thread
		uint32_t i0 = base + threadIdx.x * 4;
		uint32_t v[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] = (i0 + j < ntiles) ? row[i0 + j] : 0u;
		uint32_t s = v[0] + v[1] + v[2] + v[3], tot;
		uint32_t ex = block_excl_scan_256(s, part, &tot) + carry;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (i0 + j < ntiles) row[i0 + j] = ex;
			ex += v[j];
		}
		carry += tot;
	}
	if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(RS_BLOCK) void rs_scatter(const uint32_t *__restrict__ keys_in,
                                                       const uint32_t *__restrict__ vals_in,
                                                       uint32_t *__restrict__ keys_out,
                                                       uint32_t *__restrict__ vals_out, uint32_t n,
                                                       int shift, const uint32_t *__restrict__ table,
                                                       const uint32_t *__restrict__ totals,
                                                       uint32_t ntiles) {
	__shared__ uint32_t s_keys[RS_TILE];
	__shared__ uint32_t s_vals[RS_TILE];
	__shared__ uint32_t s_whist[4][256];
	__shared__ uint32_t s_goff[256];
	__shared__ uint32_t s_part[4];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
	for (int i = 0; i < 4; ++i) s_whist[i][tid] = 0;
	__syncthreads();

	// wave-striped tile order: (wave, item, lane) -- each load is 64 consecutive keys
	const uint32_t tile_base = blockIdx.x * RS_TILE;
	const uint32_t base = tile_base + w * (64 * RS_ITEMS) + lane;
	uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i) {
		uint32_t idx = base + i * 64;
		bool ok = idx < n;
		key[i] = ok ? keys_in[idx] : 0xFFFFFFFFu; // padding sorts last inside the tile
		val[i] = ok ? (vals_in ? vals_in[idx] : idx) : 0u;
	}
	const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i) {
		const uint32_t d = (key[i] >> shift) & 255u;
		uint64_t peers = ~0ull; // lanes holding the same digit (match-any by 8 ballots)
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const bool bit = (d >> b) & 1u;
			const uint64_t m = __ballot(bit);
			peers &= bit ? m : ~m;
		}
		const uint32_t cnt = __popcll(peers);
		const int leader = 63 - __clzll((long long)peers);
		uint32_t old = 0;
		if (lane == leader) old = atomicAdd(&s_whist[w][d], cnt);
		old = __shfl(old, leader, 64);
		rank[i] = old + __popcll(peers & lt_mask);
	}
	__syncthreads();

	{ // digit starts inside the tile, per-wave bases, global offsets
		const uint32_t c0 = s_whist[0][tid], c1 = s_whist[1][tid], c2 = s_whist[2][tid], c3 = s_whist[3][tid];
		const uint32_t dstart = block_excl_scan_256(c0 + c1 + c2 + c3, s_part, nullptr);
		const uint32_t gstart = block_excl_scan_256(totals[tid], s_part, nullptr);
		s_whist[0][tid] = dstart;
		s_whist[1][tid] = dstart + c0;
		s_whist[2][tid] = dstart + c0 + c1;
		s_whist[3][tid] = dstart + c0 + c1 + c2;
		s_goff[tid] = gstart + table[(uint32_t)tid * ntiles + blockIdx.x] - dstart;
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i) {
		const uint32_t d = (key[i] >> shift) & 255u;
		const uint32_t pos = s_whist[w][d] + rank[i];
		s_keys[pos] = key[i];
		s_vals[pos] = val[i];
	}
	__syncthreads();
	const uint32_t tile_count = (n - tile_base) < (uint32_t)RS_TILE ? (n - tile_base) : (uint32_t)RS_TILE;
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i) {
		const uint32_t p = i * RS_BLOCK + tid;
		if (p < tile_count) {
			const uint32_t k = s_keys[p];
			const uint32_t o = s_goff[(k >> shift) & 255u] + p;
			keys_out[o] = k;
			vals_out[o] = s_vals[p];
		}
	}
}

__global__ __launch_bounds__(256) void k_iota_copy(const uint32_t *__restrict__ keys_in,
                                                    const uint32_t *__restrict__ vals_in,
                                                    uint32_t *__restrict__ keys_out,
                                                    uint32_t *__restrict__ vals_out, uint32_t n) {
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	keys_out[i] = keys_in[i];
	vals_out[i] = vals_in ? vals_in[i] : i;
}

static int32_t sort_pairs_impl(treee_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n64, int bits, uint32_t *d_keys_out,
                               uint32_t *d_vals_out, const float *d_xyz_for_keys, const float *d_cube_for_keys);
int32_t sort_pairs_dev(treee_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in,
                       uint64_t n64, int bits, uint32_t *d_keys_out, uint32_t *d_vals_out) {
	return sort_pairs_impl(ctx, d_keys_in, d_vals_in, n64, bits, d_keys_out, d_vals_out, nullptr, nullptr);
}
// 30-bit Morton keys of a cloud generated INTO d_keys_scratch and sorted with their indices: the key kernel leaves the first
// pass's histogram behind (k_morton_keys_hist)
int32_t morton_sort_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float *d_cube, uint32_t *d_keys_scratch, uint32_t *d_keys_out, uint32_t *d_perm_out) {
	return sort_pairs_impl(ctx, d_keys_scratch, nullptr, n, 30, d_keys_out, d_perm_out, d_xyz, d_cube);
}
static int32_t sort_pairs_impl(treee_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n64, int bits, uint32_t *d_keys_out,
                               uint32_t *d_vals_out, const float *d_xyz_for_keys, const float *d_cube_for_keys) {
	if (n64 == 0) return TREEE_OK;
	if (n64 >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "sort: n=%llu exceeds 2^31-1 per shard", (unsigned long long)n64);
	const uint32_t n = (uint32_t)n64;
	const int passes = bits <= 0 ? 0 : (bits + 7) / 8;
	if (passes == 0) {
		hipLaunchKernelGGL(k_iota_copy, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_keys_in, d_vals_in, d_keys_out, d_vals_out, n);
		TREEE_LAUNCH_CHECK();
		return TREEE_OK;
	}
	const uint32_t ntiles = div_up(n, RS_TILE);
	uint32_t *d_ktmp = nullptr, *d_vtmp = nullptr;
	TREEE_TRY(ctx->alloc_t(n, &d_ktmp));
	TREEE_TRY(ctx->alloc_t(n, &d_vtmp));
	// (A onesweep variant -- one global histogram, decoupled look-back per digit pass -- was measured in round 1:
	// 405 vs 418 us per 15 M-pair sort, the look-back latency cancelling the saved key re-read.  It was dropped:
	// the three-kernel pass has no inter-workgroup protocol and so no timeout path to get wrong.)
	uint32_t *d_table = nullptr, *d_totals = nullptr;
	TREEE_TRY(ctx->alloc_t((size_t)256 * ntiles, &d_table));
	TREEE_TRY(ctx->alloc_t(256, &d_totals));
	// ping-pong so that the last pass lands in the caller's output buffers
	const uint32_t *kin = d_keys_in, *vin = d_vals_in;
	for (int p = 0; p < passes; ++p) {
		const bool to_out = ((passes - 1 - p) % 2) == 0;
		uint32_t *kout = to_out ? d_keys_out : d_ktmp;
		uint32_t *vout = to_out ? d_vals_out : d_vtmp;
		const int shift = 8 * p;
		if (p == 0 && d_xyz_for_keys) // keys generated here, with the first digit's histogram
			hipLaunchKernelGGL(k_morton_keys_hist, dim3(ntiles), dim3(RS_BLOCK), 0, ctx->stream, d_xyz_for_keys, n, d_cube_for_keys, (uint32_t *)kin, d_table, ntiles);
		else
			hipLaunchKernelGGL(rs_hist, dim3(ntiles), dim3(RS_BLOCK), 0, ctx->stream, kin, n, shift, d_table, ntiles);
		hipLaunchKernelGGL(rs_scan_rows, dim3(256), dim3(256), 0, ctx->stream, d_table, ntiles, d_totals);
		hipLaunchKernelGGL(rs_scatter, dim3(ntiles), dim3(RS_BLOCK), 0, ctx->stream, kin, vin, kout, vout, n, shift, d_table, d_totals, ntiles);
		TREEE_LAUNCH_CHECK();
		kin = kout;
		vin = vout;
	}
	return TREEE_OK;
}

// ---- exclusive scan (u32) --------------------------------------------------------------------
constexpr int SC_ITEMS = 4;
constexpr int SC_TILE = 256 * SC_ITEMS;

// `live` (optional, device): only the first *live elements are non-zero / wanted; blocks past them do nothing
__global__ __launch_bounds__(256) void sc_reduce(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ sums, const uint32_t *__restrict__ live) {
	__shared__ uint32_t part[4];
	if (live) {
		n = min(n, *live);
		if (blockIdx.x * SC_TILE >= n) {
			if (threadIdx.x == 0) sums[blockIdx.x] = 0u;
			return;
		}
	}
	uint32_t i0 = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS, s = 0;
#pragma unroll
	for (int j = 0; j < SC_ITEMS; ++j) s += (i0 + j < n) ? in[i0 + j] : 0u;
	uint32_t tot;
	block_excl_scan_256(s, part, &tot);
	if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sc_scan_sums(uint32_t *__restrict__ sums, uint32_t nb, uint32_t *__restrict__ total_out) {
	__shared__ uint32_t part[4];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nb; base += 256) {
		uint32_t i = base + threadIdx.x, v = i < nb ? sums[i] : 0u, tot;
		uint32_t ex = block_excl_scan_256(v, part, &tot);
		if (i < nb) sums[i] = carry + ex;
		carry += tot;
	}
	if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(256) void sc_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n, const uint32_t *__restrict__ sums,
                                                 const uint32_t *__restrict__ live) {
	__shared__ uint32_t part[4];
	if (live) {
		n = min(n, *live);
		if (blockIdx.x * SC_TILE >= n) return;
	}
	uint32_t i0 = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS, v[SC_ITEMS], s = 0;
#pragma unroll
	for (int j = 0; j < SC_ITEMS; ++j) {
		v[j] = (i0 + j < n) ? in[i0 + j] : 0u;
		s += v[j];
	}
	uint32_t ex = block_excl_scan_256(s, part, nullptr) + sums[blockIdx.x];
#pragma unroll
	for (int j = 0; j < SC_ITEMS; ++j) {
		if (i0 + j < n) out[i0 + j] = ex;
		ex += v[j];
	}
}

// Small arrays (the top levels of the octree: 1 ... 4 096 entries): ONE workgroup walks the tiles with a running carry -- one
// launch instead of three.  The scans sit in chains of small dependent launches (the octree's levels: five launches per level),
// whose length is launches x dispatch latency, not bytes (round 5, TREEE_TRACE_BUILD).
constexpr uint32_t SC_SMALL = 4u * SC_TILE;
constexpr uint32_t SC_SUMS_ONE_GROUP = 16384; // tile sums one workgroup scans by itself (sc_scan_sums)
__global__ __launch_bounds__(256) void sc_small(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n, uint32_t *__restrict__ total_out,
                                                 const uint32_t *__restrict__ live) {
	__shared__ uint32_t part[4];
	if (live) n = min(n, *live);
	uint32_t carry = 0;
	for (uint32_t base = 0; base < n; base += SC_TILE) { // (uniform)
		const uint32_t i0 = base + threadIdx.x * SC_ITEMS;
		uint32_t v[SC_ITEMS], s = 0, tot;
#pragma unroll
		for (int j = 0; j < SC_ITEMS; ++j) {
			v[j] = (i0 + j < n) ? in[i0 + j] : 0u;
			s += v[j];
		}
		uint32_t ex = carry + block_excl_scan_256(s, part, &tot);
#pragma unroll
		for (int j = 0; j < SC_ITEMS; ++j) {
			if (i0 + j < n) out[i0 + j] = ex;
			ex += v[j];
		}
		carry += tot;
	}
	if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// (Tried for the large arrays and dropped: tile sums and their scan from one launch, the workgroup that finishes last scanning
// the sums -- its release fences are write-backs of an XCD's whole L2 on this chip, under a gather that fills it with dirty lines:
// structure stage 1.24 -> 1.43 ms.)
int32_t exclusive_scan_u32_dev(treee_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_total, const uint32_t *d_live) {
	if (n == 0) {
		if (d_total) TREEE_HIP(hipMemsetAsync(d_total, 0, 4, ctx->stream));
		return TREEE_OK;
	}
	static const bool three = getenv("TREEE_SCAN_THREE_LAUNCHES") != nullptr; // A/B and test hook: rounds 1-4
	if (n <= SC_SMALL && !three) {
		hipLaunchKernelGGL(sc_small, dim3(1), dim3(256), 0, ctx->stream, d_in, d_out, (uint32_t)n, d_total, d_live);
		TREEE_LAUNCH_CHECK();
		return TREEE_OK;
	}
	uint32_t nb = div_up(n, SC_TILE);
	uint32_t *d_sums = nullptr;
	TREEE_TRY(ctx->alloc_t(nb, &d_sums));
	hipLaunchKernelGGL(sc_reduce, dim3(nb), dim3(256), 0, ctx->stream, d_in, (uint32_t)n, d_sums, d_live);
	if (nb > SC_SUMS_ONE_GROUP) {
		// (above ~17 M entries the tile sums are an array worth scanning in parallel themselves -- in place: every thread reads its own
		// entries before it writes them.  One workgroup walking 146 000 sums of a 150 M-entry scan took 90 us, nine times per build.)
		TREEE_TRY(exclusive_scan_u32_dev(ctx, d_sums, d_sums, nb, d_total, nullptr));
	} else {
		hipLaunchKernelGGL(sc_scan_sums, dim3(1), dim3(256), 0, ctx->stream, d_sums, nb, d_total);
	}
	hipLaunchKernelGGL(sc_apply, dim3(nb), dim3(256), 0, ctx->stream, d_in, d_out, (uint32_t)n, d_sums, d_live);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

// ---- gather into sorted order ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_sorted(const float *__restrict__ xyz, const uint32_t *__restrict__ perm,
                                                        uint32_t n, float4 *__restrict__ out) {
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	uint32_t o = perm[i];
	out[i] = make_float4(xyz[3ull * o + 0], xyz[3ull * o + 1], xyz[3ull * o + 2], __uint_as_float(o));
}

int32_t gather_sorted_dev(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_perm, uint64_t n, float4 *d_sorted) {
	if (n == 0) return TREEE_OK;
	hipLaunchKernelGGL(k_gather_sorted, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_xyz, d_perm, (uint32_t)n, d_sorted);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

void SortedCloud::release() {
	if (d_keys || d_perm || d_sorted || d_cube) (void)hipSetDevice(device);
	retire(owner, &sync); // the owner's stream first waits for uses from other streams (HandleSync)
	pool_release(owner, d_keys);
	pool_release(owner, d_perm);
	pool_release(owner, d_sorted);
	pool_release(owner, d_cube);
	d_keys = d_perm = nullptr;
	d_sorted = nullptr;
	d_cube = nullptr;
}

// every LONG_RUN_STEP-th position is probed: a run longer than LONG_RUN + LONG_RUN_STEP is always found, one
// between LONG_RUN and that may or may not be (the threshold is a heuristic, the probe a 64th of the reads)
constexpr uint32_t LONG_RUN_STEP = 64;
// bit 0: a run of more than LONG_RUN equal keys (a crowded 2^-10 cell); bit 1: a run of BLOCK7_PROBE + 1 or more keys that
// share their top 21 bits (a 2^-7 block that may hold BLOCK7_MAX points or more: hilbert_order_by_blocks, knn.hip, sorts
// tiles of whole blocks in LDS and needs every block below that) -- every run of BLOCK7_MAX is found, shorter ones down to
// BLOCK7_PROBE may be
__global__ __launch_bounds__(256) void k_long_run(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ flag) {
	const uint64_t i = (uint64_t)(blockIdx.x * 256u + threadIdx.x) * LONG_RUN_STEP;
	uint32_t bits = 0u;
	if (i + LONG_RUN < n && keys[i] == keys[i + LONG_RUN]) bits |= 1u;
	if (i + BLOCK7_PROBE < n && (keys[i] >> 9) == (keys[i + BLOCK7_PROBE] >> 9)) bits |= 2u;
	// (the read keeps a crowded cloud from queueing its atomics on one word -- read at agent scope: a plain load is served from the CU's
	// L1, which never sees the other workgroups' atomics, and 1 G crowded points then queued 2.4 ms of them; one atomic per wave)
	const uint64_t any1 = __ballot((bits & 1u) != 0u), any2 = __ballot((bits & 2u) != 0u);
	const uint32_t wave_bits = (any1 ? 1u : 0u) | (any2 ? 2u : 0u);
	if (wave_bits && (threadIdx.x & 63u) == 0u && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & wave_bits) != wave_bits) atomicOr(flag, wave_bits);
}

int32_t long_run_flag_dev(treee_ctx *ctx, const uint32_t *d_keys, uint64_t n, uint32_t *d_flag) {
	TREEE_HIP(hipMemsetAsync(d_flag, 0, 4, ctx->stream));
	if (n > LONG_RUN)
		hipLaunchKernelGGL(k_long_run, dim3(div_up(div_up(n, LONG_RUN_STEP), 256)), dim3(256), 0, ctx->stream, d_keys, (uint32_t)n, d_flag);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

// Share of distinct values among sorted keys, from a sample: one window of 256 consecutive keys out of every 64 (a 64th of
// the array is read; each workgroup leaves one partial, no same-address atomics).  d_out[0] = distinct in the windows,
// d_out[1] = keys in the windows.  Only a decision is taken from it (hilbert_order_by_cells counts exactly when it runs).
constexpr uint32_t SAMPLE_EVERY = 64;
__global__ __launch_bounds__(256) void k_distinct_sample(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ partial) {
	__shared__ uint32_t s_w[4];
	const uint32_t i = blockIdx.x * (256u * SAMPLE_EVERY) + threadIdx.x;
	uint32_t c = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
	if (lane_id() == 0) s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_distinct_sum(const uint32_t *__restrict__ partial, uint32_t blocks, uint32_t n, uint32_t *__restrict__ out) {
	__shared__ uint32_t s_w[4];
	uint32_t c = 0;
	for (uint32_t b = threadIdx.x; b < blocks; b += 256u) c += partial[b];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
	if (lane_id() == 0) s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		out[0] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
		const uint32_t full = n / (256u * SAMPLE_EVERY), rest = n - full * 256u * SAMPLE_EVERY;
		out[1] = full * 256u + (rest < 256u ? rest : 256u);
	}
}
int32_t distinct_keys_dev(treee_ctx *ctx, const uint32_t *d_keys, uint64_t n, uint32_t *d_out) {
	const uint32_t blocks = div_up(n ? n : 1, 256u * SAMPLE_EVERY);
	uint32_t *d_partial = nullptr;
	TREEE_TRY(ctx->alloc_t(blocks, &d_partial));
	hipLaunchKernelGGL(k_distinct_sample, dim3(blocks), dim3(256), 0, ctx->stream, d_keys, (uint32_t)n, d_partial);
	hipLaunchKernelGGL(k_distinct_sum, dim3(1), dim3(256), 0, ctx->stream, d_partial, blocks, (uint32_t)n, d_out);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t build_sorted_cloud(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float *corner, float size, SortedCloud *sc, bool fork_after_sort,
                           bool hilbert_order) {
	sc->n = n;
	sc->device = ctx->device;
	sc->owner = ctx;
	if (n >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "n=%llu exceeds 2^31-1 points per shard", (unsigned long long)n);
	if (corner && size > 0.0f) {
		for (int a = 0; a < 3; ++a) sc->corner[a] = corner[a];
		sc->size = size;
	} else { // cube from the data: it stays on the device, the host copy follows with the build's next round trip
		TREEE_TRY(pool_alloc_t(ctx, 4, &sc->d_cube));
		TREEE_TRY(bbox_to_device(ctx, d_xyz, n, sc->d_cube));
		sc->cube_pending = true;
	}
	size_t cnt = n ? n : 1;
	TREEE_TRY(pool_alloc_t(ctx, cnt, &sc->d_keys));
	TREEE_TRY(pool_alloc_t(ctx, cnt, &sc->d_perm));
	TREEE_TRY(pool_alloc_t(ctx, cnt, &sc->d_sorted));
	uint32_t *d_keys_raw = nullptr;
	TREEE_TRY(ctx->alloc_t(cnt, &d_keys_raw));
	if (sc->cube_pending && !hilbert_order && n) {
		// (the common path -- cube from the data, Morton order: keys and the first radix histogram from one kernel)
		TREEE_TRY(morton_sort_dev(ctx, d_xyz, n, sc->d_cube, d_keys_raw, sc->d_keys, sc->d_perm));
	} else {
		if (sc->cube_pending) {
			if (n) hipLaunchKernelGGL(k_morton_keys_cube, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, sc->d_cube, d_keys_raw);
			TREEE_LAUNCH_CHECK();
		} else {
			TREEE_TRY(morton_keys_dev(ctx, d_xyz, n, sc->corner, sc->size, d_keys_raw));
		}
		if (hilbert_order) {
			TREEE_TRY(hilbert_keys_dev(ctx, d_keys_raw, nullptr, n, d_keys_raw));
			sc->hilbert = true;
		}
		TREEE_TRY(sort_pairs_dev(ctx, d_keys_raw, nullptr, n, 30, sc->d_keys, sc->d_perm));
	}
	// keys and permutation are final here: work that needs nothing else (the octree levels) may start on the context's
	// auxiliary stream while this one gathers the records
	if (fork_after_sort) TREEE_TRY(ctx->fork());
	TREEE_TRY(gather_sorted_dev(ctx, d_xyz, sc->d_perm, n, sc->d_sorted));
	return TREEE_OK;
}

} // namespace treee
