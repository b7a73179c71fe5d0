// shard.hip -- spatial-shard helper for the multi-GPU path: stable
// compaction of the points whose (x, z) lies inside a `keep` box and outside a `drop` box.
//   boundary points of a tile  = keep: everything,      drop: the tile shrunk by the halo on shared faces
//   ghosts received from peers = keep: the dilated tile, drop: nothing   (NaN padding never passes)
// A count pass per 4096-point tile, a scan of the tile counts and a write pass that re-evaluates the predicate: the cloud
// is read twice and nothing per point is stored in between; the output keeps the input order (deterministic ghost
// indices).  No reference counterpart: the
// reference is a single-process program.
#include "comm.h"
#include "common.h"

using namespace treee;

namespace {

struct Box2 {
	float x0, x1, z0, z1;
};

__device__ __forceinline__ bool in_box(const Box2 &b, float x, float z) { return x >= b.x0 && x < b.x1 && z >= b.z0 && z < b.z1; }

constexpr int SEL_ITEMS = 16, SEL_TILE = 256 * SEL_ITEMS; // points per workgroup

// pass 1: matches per tile (the predicate is cheap, so it is evaluated again in pass 2 instead of storing 60 MB of flags)
__global__ __launch_bounds__(256) void k_select_count(const float *__restrict__ xyz, uint32_t n, Box2 keep, Box2 drop, uint32_t *__restrict__ tile_count) {
	__shared__ uint32_t s_w[4];
	const uint32_t base = blockIdx.x * SEL_TILE;
	uint32_t c = 0;
#pragma unroll
	for (int it = 0; it < SEL_ITEMS; ++it) {
		const uint32_t i = base + it * 256u + threadIdx.x;
		bool f = false;
		if (i < n) {
			const float x = xyz[3ull * i + 0], z = xyz[3ull * i + 2];
			f = in_box(keep, x, z) && !in_box(drop, x, z);
		}
		c += (uint32_t)__popcll(__ballot(f)); // wave-uniform
	}
	if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) tile_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// pass 2: stable compaction.  Output order = input order: item-major inside a tile, then wave, then lane.
__global__ __launch_bounds__(256) void k_select_write(const float *__restrict__ xyz, uint32_t n, Box2 keep, Box2 drop, const uint32_t *__restrict__ tile_off,
                                                       uint32_t cap, float *__restrict__ out) {
	__shared__ uint32_t s_c[SEL_ITEMS * 4]; // matches of (item, wave), then their exclusive prefix in that order
	const uint32_t base = blockIdx.x * SEL_TILE, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	uint32_t flags = 0;
#pragma unroll
	for (int it = 0; it < SEL_ITEMS; ++it) {
		const uint32_t i = base + it * 256u + threadIdx.x;
		bool f = false;
		if (i < n) {
			const float x = xyz[3ull * i + 0], z = xyz[3ull * i + 2];
			f = in_box(keep, x, z) && !in_box(drop, x, z);
		}
		const uint64_t b = __ballot(f);
		if (lane == 0) s_c[it * 4 + w] = (uint32_t)__popcll(b);
		flags |= (f ? 1u : 0u) << it;
	}
	__syncthreads();
	if (threadIdx.x < 64) { // 64 counters: one wave scans them
		const uint32_t v = s_c[threadIdx.x];
		const uint32_t inc = wave_incl_scan(v);
		s_c[threadIdx.x] = inc - v;
	}
	__syncthreads();
	if (!flags) return;
	const uint32_t tile_base = tile_off[blockIdx.x];
#pragma unroll
	for (int it = 0; it < SEL_ITEMS; ++it) {
		const bool f = (flags >> it) & 1u;
		const uint64_t b = __ballot(f); // lanes that left hold no flag: the ballot of the remaining ones is still exact
		if (!f) continue;
		const uint32_t o = tile_base + s_c[it * 4 + w] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
		if (o >= cap) continue;
		const uint32_t i = base + it * 256u + threadIdx.x;
		out[3ull * o + 0] = xyz[3ull * i + 0];
		out[3ull * o + 1] = xyz[3ull * i + 1];
		out[3ull * o + 2] = xyz[3ull * i + 2];
	}
}



// header row of every sender's block: the count it wanted to send (int32 bits in .x)
__global__ __launch_bounds__(256) void k_block_heads(const float *__restrict__ recv, uint32_t world, uint64_t block_floats, uint32_t *__restrict__ heads) {
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r < world) heads[r] = __float_as_uint(recv[(uint64_t)r * block_floats]);
}
__global__ void k_set_head(float *__restrict__ send, const uint32_t *__restrict__ count) { send[0] = __uint_as_float(*count); }

int32_t select_xz(treee_ctx *ctx, const float *d_xyz, uint64_t n, const Box2 &keep, const Box2 &drop, float *d_out_xyz, uint64_t cap, uint32_t *d_count_out) {
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large for one call");
	if (n == 0) {
		TREEE_HIP(hipMemsetAsync(d_count_out, 0, 4, ctx->stream));
		return TREEE_OK;
	}
	const uint32_t ntiles = div_up(n, SEL_TILE);
	uint32_t *d_tiles = nullptr, *d_off = nullptr;
	TREEE_TRY(ctx->alloc_t(ntiles, &d_tiles));
	TREEE_TRY(ctx->alloc_t(ntiles, &d_off));
	hipLaunchKernelGGL(k_select_count, dim3(ntiles), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, keep, drop, d_tiles);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_tiles, d_off, ntiles, d_count_out));
	hipLaunchKernelGGL(k_select_write, dim3(ntiles), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, keep, drop, d_off,
	                   (uint32_t)(cap < 0xFFFFFFFFull ? cap : 0xFFFFFFFFull), d_out_xyz);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

} // namespace

extern "C" {

int32_t treee_select_xz_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float keep_box[4], const float drop_box[4],
                            float *d_out_xyz, uint64_t cap, uint32_t *d_count_out) {
	if (!ctx || !keep_box || !drop_box || !d_count_out || (n && !d_xyz) || (cap && !d_out_xyz))
		return set_err(TREEE_ERR_INVALID, "treee_select_xz_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	const Box2 keep = {keep_box[0], keep_box[1], keep_box[2], keep_box[3]}, drop = {drop_box[0], drop_box[1], drop_box[2], drop_box[3]};
	return select_xz(ctx, d_xyz, n, keep, drop, d_out_xyz, cap, d_count_out);
}

// The ghost-point exchange of SURVEY.md section 8 row E in one call: boundary selection, ONE all-gather of fixed-capacity
// blocks, ghost selection straight behind the own points, one host round trip for the two counts.
int32_t treee_shard_exchange_ghosts_dev(treee_ctx *ctx, treee_comm *comm, float *d_cloud, uint64_t n_own, const float tile_box[4],
                                        uint32_t shared_faces, float halo, uint64_t send_cap, uint64_t ghost_cap, uint64_t *n_ghost_out,
                                        uint64_t *most_sent_out) {
	if (!ctx || !comm || !tile_box || !n_ghost_out || !most_sent_out || (n_own && !d_cloud) || (ghost_cap && !d_cloud))
		return set_err(TREEE_ERR_INVALID, "treee_shard_exchange_ghosts_dev: null argument");
	if (!(halo >= 0.0f)) return set_err(TREEE_ERR_INVALID, "treee_shard_exchange_ghosts_dev: halo must be >= 0");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	*n_ghost_out = *most_sent_out = 0;
	const int world = comm->world, rank = comm->rank;
	if (world == 1) return TREEE_OK;
	// the ghost selection runs over the world's blocks in ONE pass whose positions are 32-bit: refuse what cannot fit before
	// anything is allocated for it
	if ((send_cap + 1) >= (1ull << 32) / (uint64_t)world)
		return set_err(TREEE_ERR_UNSUPPORTED, "treee_shard_exchange_ghosts_dev: (send_cap + 1) x world = %llu x %d rows exceed 2^32", (unsigned long long)(send_cap + 1), world);
	const float x0 = tile_box[0], x1 = tile_box[1], z0 = tile_box[2], z1 = tile_box[3], inf = INFINITY;
	const uint64_t block = (send_cap + 1) * 3; // floats per rank: header row + send_cap points
	float *d_send = nullptr, *d_recv = nullptr;
	uint32_t *d_cnt = nullptr, *d_heads = nullptr;
	TREEE_TRY(ctx->alloc_t(block, &d_send));
	TREEE_TRY(ctx->alloc_t(block * (uint64_t)world, &d_recv));
	TREEE_TRY(ctx->alloc_t(2, &d_cnt));
	TREEE_TRY(ctx->alloc_t((size_t)world, &d_heads));
	hipStream_t st = ctx->stream;
	TREEE_HIP(hipMemsetAsync(d_send, 0xFF, block * 4, st)); // NaN: never inside a box
	// own points within halo of a face shared with another tile = everything minus the tile shrunk on those faces
	const Box2 all = {-inf, inf, -inf, inf};
	const Box2 interior = {(shared_faces & 1u) ? x0 + halo : -inf, (shared_faces & 2u) ? x1 - halo : inf, (shared_faces & 4u) ? z0 + halo : -inf,
	                       (shared_faces & 8u) ? z1 - halo : inf};
	TREEE_TRY(select_xz(ctx, d_cloud, n_own, all, interior, d_send + 3, send_cap, d_cnt));
	hipLaunchKernelGGL(k_set_head, dim3(1), dim3(1), 0, st, d_send, d_cnt);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(comm->all_gather_dev(ctx, d_send, d_recv, block * 4));
	hipLaunchKernelGGL(k_block_heads, dim3(div_up((uint64_t)world, 256)), dim3(256), 0, st, d_recv, (uint32_t)world, block, d_heads);
	TREEE_LAUNCH_CHECK();
	TREEE_HIP(hipMemsetAsync(d_recv + (uint64_t)rank * block, 0xFF, block * 4, st)); // own block: not a ghost of oneself
	const Box2 dilated = {x0 - halo, x1 + halo, z0 - halo, z1 + halo}, none = {0.0f, 0.0f, 0.0f, 0.0f};
	TREEE_TRY(select_xz(ctx, d_recv, block / 3 * (uint64_t)world, dilated, none, d_cloud + 3ull * n_own, ghost_cap, d_cnt + 1));
	HostBuf<uint32_t> h;
	TREEE_TRY(h.alloc((size_t)world + 1));
	TREEE_HIP(hipMemcpyAsync(h.data(), d_cnt + 1, 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipMemcpyAsync(h.data() + 1, d_heads, (size_t)world * 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	*n_ghost_out = h[0];
	for (int r = 0; r < world; ++r)
		if (h[1 + r] > *most_sent_out) *most_sent_out = h[1 + r];
	return TREEE_OK;
}

int32_t treee_frame_reduce_min_dev(treee_ctx *ctx, treee_comm *comm, uint64_t *d_packed, uint64_t count) {
	if (!ctx || !comm || (count && !d_packed)) return set_err(TREEE_ERR_INVALID, "treee_frame_reduce_min_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	if (comm->world == 1) return TREEE_OK;
	return comm->all_reduce_min_u64_dev(ctx, d_packed, count);
}

} // extern "C"
