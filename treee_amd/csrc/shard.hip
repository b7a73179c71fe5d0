// shard.hip -- spatial-shard helper for the multi-GPU path: stable
// compaction of the points whose (x, z) lies inside a `keep` box and outside a `drop` box.
//   boundary points of a tile  = keep: everything,      drop: the tile shrunk by the halo on shared faces
//   ghosts received from peers = keep: the dilated tile, drop: nothing   (NaN padding never passes)
// One flag pass, one exclusive scan, one scatter: 3 launches instead of ~15 elementwise torch kernels,
// and the output keeps the input order (deterministic ghost indices).  No reference counterpart: the
// reference is a single-process program.
#include "common.h"

using namespace treee;

namespace {

struct Box2 {
	float x0, x1, z0, z1;
};

__device__ __forceinline__ bool in_box(const Box2 &b, float x, float z) { return x >= b.x0 && x < b.x1 && z >= b.z0 && z < b.z1; }

__global__ __launch_bounds__(256) void k_select_flag(const float *__restrict__ xyz, uint32_t n, Box2 keep, Box2 drop, uint32_t *__restrict__ flag) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float x = xyz[3ull * i + 0], z = xyz[3ull * i + 2];
	flag[i] = (in_box(keep, x, z) && !in_box(drop, x, z)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_select_scatter(const float *__restrict__ xyz, uint32_t n, const uint32_t *__restrict__ flag,
                                                         const uint32_t *__restrict__ off, uint32_t cap, float *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n || !flag[i]) return;
	const uint32_t o = off[i];
	if (o >= cap) return;
	out[3ull * o + 0] = xyz[3ull * i + 0];
	out[3ull * o + 1] = xyz[3ull * i + 1];
	out[3ull * o + 2] = xyz[3ull * i + 2];
}

} // namespace

extern "C" int32_t treee_select_xz_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float keep_box[4], const float drop_box[4],
                                       float *d_out_xyz, uint64_t cap, uint32_t *d_count_out) {
	if (!ctx || !keep_box || !drop_box || !d_count_out || (n && !d_xyz) || (cap && !d_out_xyz))
		return set_err(TREEE_ERR_INVALID, "treee_select_xz_dev: null argument");
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large for one call");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	if (n == 0) {
		TREEE_HIP(hipMemsetAsync(d_count_out, 0, 4, ctx->stream));
		return TREEE_OK;
	}
	uint32_t *d_flag = nullptr, *d_off = nullptr;
	TREEE_TRY(ctx->alloc_t(n, &d_flag));
	TREEE_TRY(ctx->alloc_t(n, &d_off));
	const Box2 keep = {keep_box[0], keep_box[1], keep_box[2], keep_box[3]}, drop = {drop_box[0], drop_box[1], drop_box[2], drop_box[3]};
	hipLaunchKernelGGL(k_select_flag, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, keep, drop, d_flag);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_flag, d_off, n, d_count_out));
	hipLaunchKernelGGL(k_select_scatter, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, d_flag, d_off,
	                   (uint32_t)(cap < 0xFFFFFFFFull ? cap : 0xFFFFFFFFull), d_out_xyz);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}
