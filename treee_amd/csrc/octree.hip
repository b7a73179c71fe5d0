// octree.hip -- octree node table + LOD sample lists from Morton-sorted keys.
// BUILD-DEFINED: no reference counterpart exists at this
// snapshot; definition and bit-exact oracle: oracle/octree_oracle.cpp.
//
// Top-down, one level per step.  A node's children are the non-empty runs of the next
// 3 key bits inside its sorted range, found by binary search (8 lanes per parent, one
// boundary each), so the per-point arrays are touched only by the final leaf-id and LOD
// kernels (4 B/point written, coalesced).
#include "comm.h"
#include "common.h"
#include <chrono>
#include "shard_plan.h"

using namespace treee;

struct treee_octree {
	std::shared_ptr<SortedCloud> sc = std::make_shared<SortedCloud>();
	treee_octree_node *d_nodes = nullptr;
	uint64_t n_nodes = 0, nodes_cap = 0;
	uint32_t *d_leaf = nullptr;
	uint32_t *d_lod = nullptr;
	uint64_t n_lod = 0;
	uint32_t leaf_cap = 256, lod_cap = 256;
	int device = 0;
	treee_ctx *owner = nullptr;
	// a rank's piece of a sharded octree (treee_shard_octree_build_dev): the arrays above hold the owned range and the own
	// nodes of levels >= top_level with GLOBAL indices; the top of the tree is replicated
	bool shard = false;
	treee_shard_plan_info sh_info{};
	treee_shard_plan_rank sh_rank{};
	treee_shard_levels sh_levels{};
	treee_octree_node *d_top_nodes = nullptr;
	uint32_t *d_top_lod = nullptr;
	// the rank's OWN input in Morton order of the common cube (step 1 of the sharded build; records carry global indices =
	// index_base + input position): what a kNN structure over own + ghost points starts from instead of sorting the tile
	// again (treee_knn_from_shard_dev).  Null when no point changed hands: `sc` then is that cloud.
	std::shared_ptr<SortedCloud> sc_local;
	bool local_released = false; // treee_shard_octree_release_local: the copy above was given back; no kNN structure from this handle any more
};
// (the octree's own arrays are only ever touched by its owner's stream and by the caller through
// treee_octree_dev_ptr, whose ordering is the caller's; the shared sorted cloud carries a HandleSync)

namespace {

constexpr uint32_t DEPTH = 10;

__device__ __forceinline__ uint32_t lower_bound_prefix(const uint32_t *__restrict__ keys, uint32_t lo, uint32_t hi,
                                                       uint32_t shift, uint32_t prefix) {
	while (lo < hi) { // first position in [lo,hi) with (key >> shift) >= prefix
		uint32_t mid = lo + ((hi - lo) >> 1);
		if ((keys[mid] >> shift) < prefix)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}

// 8 lanes per parent node of the current level; lane c finds the start of octant c.
// lvl[0] = index of the level's first node, lvl[1] = number of nodes of the level (device memory:
// the whole top-down build is enqueued without a host round trip per level)
__global__ __launch_bounds__(256) void k_oct_split(const uint32_t *__restrict__ keys, treee_octree_node *__restrict__ nodes,
                                                    const uint32_t *__restrict__ lvl, uint32_t lvl_cap, uint32_t leaf_cap,
                                                    uint32_t *__restrict__ child_cnt, uint32_t *__restrict__ bounds) {
	const uint32_t lvl_begin = lvl[0], lvl_count = lvl[1];
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	const uint32_t p = t >> 3, c = t & 7u;
	if ((blockIdx.x * 256u >> 3) >= lvl_count) { // whole block past the level: only zero the scan input
		if (p < lvl_cap && c == 0) child_cnt[p] = 0u;
		return;
	}
	const bool live = p < lvl_count;
	treee_octree_node nd;
	nd.count = 0;
	nd.level = DEPTH;
	nd.first = 0;
	nd.key_prefix = 0;
	if (live) nd = nodes[lvl_begin + p];
	const bool split = live && nd.count > leaf_cap && nd.level < DEPTH;
	uint32_t lo = nd.first;
	if (split && c > 0) {
		const uint32_t shift = 3u * (DEPTH - (nd.level + 1u));
		lo = lower_bound_prefix(keys, nd.first, nd.first + nd.count, shift, nd.key_prefix * 8u + c);
	}
	uint32_t hi = __shfl_down(lo, 1, 8);
	if (c == 7) hi = nd.first + nd.count;
	const bool nonempty = split && hi > lo;
	const uint64_t ball = __ballot(nonempty);
	const uint32_t mask = (uint32_t)(ball >> ((threadIdx.x & 63u) & ~7u)) & 0xFFu;
	if (live) {
		bounds[(size_t)p * 8 + c] = lo;
		if (c == 0) {
			child_cnt[p] = __popc(mask);
			nodes[lvl_begin + p].child_mask = mask;
		}
	} else if (p < lvl_cap && c == 0) {
		child_cnt[p] = 0u;
	}
}

// `host_pair` (or null; the last level only): pinned host words that receive the next level's descriptor too -- the build's round
// trip then needs no copy behind the kernels
__global__ __launch_bounds__(256) void k_oct_emit(treee_octree_node *__restrict__ nodes, uint32_t *__restrict__ lvl,
                                                   const uint32_t *__restrict__ child_off, const uint32_t *__restrict__ bounds,
                                                   const uint32_t *__restrict__ total, uint32_t *__restrict__ host_pair) {
	const uint32_t lvl_begin = lvl[0], lvl_count = lvl[1];
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t == 0) { // descriptor of the next level
		lvl[2] = lvl_begin + lvl_count;
		lvl[3] = *total;
		if (host_pair) {
			host_pair[0] = lvl_begin + lvl_count;
			host_pair[1] = *total;
		}
	}
	const uint32_t p = t >> 3, c = t & 7u;
	if (p >= lvl_count) return;
	const treee_octree_node nd = nodes[lvl_begin + p];
	if (nd.child_mask == 0) return;
	const uint32_t first_child = lvl_begin + lvl_count + child_off[p];
	if (c == 0) nodes[lvl_begin + p].first_child = first_child;
	if (!((nd.child_mask >> c) & 1u)) return;
	const uint32_t lo = bounds[(size_t)p * 8 + c];
	const uint32_t hi = c == 7 ? nd.first + nd.count : bounds[(size_t)p * 8 + c + 1];
	treee_octree_node ch;
	ch.key_prefix = nd.key_prefix * 8u + c;
	ch.level = nd.level + 1u;
	ch.first = lo;
	ch.count = hi - lo;
	ch.first_child = NONE32;
	ch.child_mask = 0;
	ch.parent = lvl_begin + p;
	ch.lod_first = NONE32;
	nodes[first_child + __popc(nd.child_mask & ((1u << c) - 1u))] = ch;
}

__global__ void k_oct_root(treee_octree_node *nodes, uint32_t n, uint32_t *lvl) {
	lvl[0] = 0;
	lvl[1] = 1;
	treee_octree_node r;
	r.key_prefix = 0;
	r.level = 0;
	r.first = 0;
	r.count = n;
	r.first_child = NONE32;
	r.child_mask = 0;
	r.parent = NONE32;
	r.lod_first = NONE32;
	nodes[0] = r;
}

__global__ __launch_bounds__(256) void k_oct_lod_count(const treee_octree_node *__restrict__ nodes, const uint32_t *__restrict__ lvl_last,
                                                        uint32_t nodes_cap, uint32_t lod_cap, uint32_t *__restrict__ cnt) {
	const uint32_t n_nodes = lvl_last[0] + lvl_last[1];
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= nodes_cap) return;
	if (i >= n_nodes) {
		cnt[i] = 0u;
		return;
	}
	const treee_octree_node nd = nodes[i];
	cnt[i] = nd.first_child == NONE32 ? 0u : (nd.count < lod_cap ? nd.count : lod_cap);
}

// one wave per node: leaves stamp their id over their sorted range, inner nodes emit LOD samples
__global__ __launch_bounds__(256) void k_oct_finish(treee_octree_node *__restrict__ nodes, uint32_t n_nodes,
                                                     uint32_t lod_cap, const uint32_t *__restrict__ lod_off,
                                                     uint32_t *__restrict__ leaf, uint32_t *__restrict__ lod) {
	const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (i >= n_nodes) return;
	const int lane = threadIdx.x & 63;
	const treee_octree_node nd = nodes[i];
	if (nd.first_child == NONE32) {
		for (uint32_t j = lane; j < nd.count; j += 64) leaf[nd.first + j] = i;
	} else {
		const uint32_t m = nd.count < lod_cap ? nd.count : lod_cap;
		const uint32_t off = lod_off[i];
		if (lane == 0) nodes[i].lod_first = off;
		for (uint32_t j = lane; j < m; j += 64) lod[off + j] = nd.first + (uint32_t)(((uint64_t)j * nd.count) / m);
	}
}

int32_t build_octree_on(treee_ctx *ctx, treee_octree *o);
// (the sorted cloud may belong to another context -- treee_octree_complete_dev on a second stream: HandleSync)
int32_t build_octree(treee_ctx *ctx, treee_octree *o) {
	TREEE_TRY(use_begin(ctx, o->sc->owner, &o->sc->sync));
	const int32_t rc = build_octree_on(ctx, o);
	const int32_t e = use_end(ctx, o->sc->owner, &o->sc->sync);
	return rc != TREEE_OK ? rc : e;
}
// What a build asks of the sorted keys for a kNN structure that will share the cloud -- is a Morton cell crowded, how many
// cells are occupied -- and the cube when it was computed from the data: enqueued here, landing in the context's pinned
// scalars ([0] crowded, [4..7] cube, [8..9] cells); cloud_flags_commit after whichever synchronisation follows.
int32_t cloud_flags_enqueue(treee_ctx *ctx, SortedCloud *sc) {
	hipStream_t st = ctx->stream;
	uint32_t *h = ctx->h_scalars;
	uint32_t *d_crowded = nullptr;
	TREEE_TRY(ctx->alloc_t(1, &d_crowded));
	TREEE_TRY(long_run_flag_dev(ctx, sc->d_keys, sc->n, d_crowded));
	TREEE_HIP(hipMemcpyAsync(h + 0, d_crowded, 4, hipMemcpyDeviceToHost, st));
	uint32_t *d_cells = nullptr; // occupied cells (distinct keys): lets that structure order cells instead of points
	TREEE_TRY(ctx->alloc_t(2, &d_cells));
	TREEE_TRY(distinct_keys_dev(ctx, sc->d_keys, sc->n, d_cells));
	TREEE_HIP(hipMemcpyAsync(h + 8, d_cells, 8, hipMemcpyDeviceToHost, st));
	TREEE_TRY(cube_fetch_async(ctx, sc, (float *)(h + 4)));
	return TREEE_OK;
}
void cloud_flags_commit(treee_ctx *ctx, SortedCloud *sc) {
	const uint32_t *h = ctx->h_scalars;
	sc->crowded = (int)(h[0] & 1u);
	sc->block7_big = (int)((h[0] >> 1) & 1u);
	sc->cell_share = h[9] ? (float)h[8] / (float)h[9] : 1.0f;
	cube_commit(sc, (const float *)(h + 4));
}

// The top-down build in two halves around its only host round trip (node count + LOD total, pinned scalars [1..3]):
// everything up to the copies is enqueued by octree_levels_enqueue, the leaf ids and LOD lists by octree_finish once the
// stream has been synchronised.
struct OctreePending {
	uint32_t *d_loff = nullptr;
};
int32_t octree_levels_enqueue(treee_ctx *ctx, treee_octree *o, OctreePending *pend) {
	const uint64_t n = o->sc->n;
	hipStream_t st = ctx->stream;
	// Capacity per level: a node is split only if it holds more than leaf_cap points, so a level has at
	// most n/(leaf_cap+1) split nodes and the next one at most 8x that many nodes (and never more than
	// 8^level).  With these bounds every level can be enqueued without knowing the previous count.
	const uint64_t per_level = 8ull * (n / ((uint64_t)o->leaf_cap + 1ull)) + 8ull;
	uint32_t lvl_cap[DEPTH + 1];
	uint64_t nodes_cap = 0, pow8 = 1;
	for (uint32_t l = 0; l <= DEPTH; ++l) {
		lvl_cap[l] = (uint32_t)(pow8 < per_level ? pow8 : per_level);
		nodes_cap += lvl_cap[l];
		if (pow8 < (1ull << 40)) pow8 *= 8ull;
	}
	TREEE_TRY(pool_alloc_t(ctx, nodes_cap, &o->d_nodes));
	o->nodes_cap = nodes_cap;
	uint32_t *d_lvl = nullptr, *d_total = nullptr;
	TREEE_TRY(ctx->alloc_t(2 * (DEPTH + 2), &d_lvl));
	TREEE_TRY(ctx->alloc_t(2, &d_total));
	hipLaunchKernelGGL(k_oct_root, dim3(1), dim3(1), 0, st, o->d_nodes, (uint32_t)n, d_lvl);
	for (uint32_t level = 0; level < DEPTH; ++level) {
		const uint32_t cap = lvl_cap[level];
		uint32_t *d_cnt = nullptr, *d_off = nullptr, *d_bounds = nullptr;
		TREEE_TRY(ctx->alloc_t(cap, &d_cnt));
		TREEE_TRY(ctx->alloc_t(cap, &d_off));
		TREEE_TRY(ctx->alloc_t((size_t)cap * 8, &d_bounds));
		const uint32_t grid = div_up((uint64_t)cap * 8, 256);
		hipLaunchKernelGGL(k_oct_split, dim3(grid), dim3(256), 0, st, o->sc->d_keys, o->d_nodes, d_lvl + 2 * level, cap, o->leaf_cap, d_cnt, d_bounds);
		TREEE_LAUNCH_CHECK();
		TREEE_TRY(exclusive_scan_u32_dev(ctx, d_cnt, d_off, cap, d_total, d_lvl + 2 * level + 1)); // only the level's nodes
		hipLaunchKernelGGL(k_oct_emit, dim3(grid), dim3(256), 0, st, o->d_nodes, d_lvl + 2 * level, d_off, d_bounds, d_total,
		                   level + 1 == DEPTH ? ctx->h_scalars + 1 : (uint32_t *)nullptr); // pinned [1..2]: the last level's (begin, count)
		TREEE_LAUNCH_CHECK();
	}
	// LOD offsets, then the only host round trip of the build: node count + LOD total
	uint32_t *d_lcnt = nullptr;
	TREEE_TRY(ctx->alloc_t(nodes_cap, &d_lcnt));
	TREEE_TRY(ctx->alloc_t(nodes_cap, &pend->d_loff));
	hipLaunchKernelGGL(k_oct_lod_count, dim3(div_up(nodes_cap, 256)), dim3(256), 0, st, o->d_nodes, d_lvl + 2 * DEPTH, (uint32_t)nodes_cap, o->lod_cap, d_lcnt);
	TREEE_LAUNCH_CHECK();
	// (the scan's total goes straight to the pinned word [3]: the kernels write the host's copy themselves, no copy is queued)
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_lcnt, pend->d_loff, nodes_cap, ctx->h_scalars + 3));
	return TREEE_OK;
}
int32_t octree_finish(treee_ctx *ctx, treee_octree *o, const OctreePending &pend) {
	const uint32_t *h = ctx->h_scalars;
	const uint32_t lod_total = h[3];
	o->n_nodes = (uint64_t)h[1] + h[2];
	o->n_lod = lod_total;
	const uint32_t nn = (uint32_t)o->n_nodes;
	TREEE_TRY(pool_alloc_t(ctx, o->sc->n, &o->d_leaf));
	TREEE_TRY(pool_alloc_t(ctx, (size_t)(lod_total ? lod_total : 1), &o->d_lod));
	hipLaunchKernelGGL(k_oct_finish, dim3(div_up(nn, 4)), dim3(256), 0, ctx->stream, o->d_nodes, nn, o->lod_cap, pend.d_loff, o->d_leaf, o->d_lod);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}
int32_t build_octree_on(treee_ctx *ctx, treee_octree *o) {
	if (o->sc->n == 0) return cube_to_host(ctx, o->sc.get());
	OctreePending pend;
	TREEE_TRY(octree_levels_enqueue(ctx, o, &pend));
	// asked here so that a kNN structure sharing this cloud needs no round trip of its own
	TREEE_TRY(cloud_flags_enqueue(ctx, o->sc.get()));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	cloud_flags_commit(ctx, o->sc.get());
	return octree_finish(ctx, o, pend);
}


// ---- the sharded octree (treee_hip.h "the sharded octree") ---------------------------------
// points of every level-L cell in a rank's locally sorted keys: two binary searches per cell
__global__ __launch_bounds__(256) void k_cell_counts(const uint32_t *__restrict__ keys, uint32_t n, uint32_t shift, uint32_t ncell,
                                                      uint32_t *__restrict__ counts) {
	const uint32_t c = blockIdx.x * 256u + threadIdx.x;
	if (c >= ncell) return;
	const uint32_t a = lower_bound_prefix(keys, 0u, n, shift, c);
	counts[c] = lower_bound_prefix(keys, a, n, shift, c + 1u) - a;
}
// local original index -> global index (records' .w and / or a permutation array)
__global__ __launch_bounds__(256) void k_add_index_base(float4 *__restrict__ rec, uint32_t *__restrict__ perm, uint32_t n, uint32_t base) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	if (rec) rec[i].w = __uint_as_float(__float_as_uint(rec[i].w) + base);
	if (perm) perm[i] += base;
}
__global__ __launch_bounds__(256) void k_keys_from_records(const float4 *__restrict__ rec, uint32_t n, float cx, float cy, float cz, float size,
                                                            uint32_t *__restrict__ keys) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float4 p = rec[i];
	keys[i] = morton30(p.x, p.y, p.z, cx, cy, cz, size);
}
// received records (runs of the senders, rank order) -> the owned range in (key, global index) order
__global__ __launch_bounds__(256) void k_gather_records(const uint32_t *__restrict__ from, const float4 *__restrict__ in, uint32_t n,
                                                         float4 *__restrict__ out, uint32_t *__restrict__ perm) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float4 p = in[from[i]];
	out[i] = p;
	perm[i] = __float_as_uint(p.w);
}
__global__ void k_forest_roots(uint32_t *lvl, uint32_t n_roots) {
	lvl[0] = 0;
	lvl[1] = n_roots;
}
// own nodes and LOD entries per level: out[l] = nodes, out[LEVELS + l] = LOD entries (level table: pairs (begin, count))
__global__ void k_level_sums(const uint32_t *__restrict__ lvl, uint32_t first_level, const uint32_t *__restrict__ lcnt,
                             const uint32_t *__restrict__ loff, uint32_t *__restrict__ out) {
	const uint32_t l = threadIdx.x;
	if (l >= TREEE_SHARD_LEVELS) return;
	uint32_t begin = 0, count = 0;
	if (l >= first_level) {
		begin = lvl[2 * l];
		count = lvl[2 * l + 1];
	}
	out[l] = count;
	out[TREEE_SHARD_LEVELS + l] = count ? loff[begin + count - 1u] + lcnt[begin + count - 1u] - loff[begin] : 0u;
	out[2 * TREEE_SHARD_LEVELS + l] = begin;
}
struct ShardFix {
	uint32_t L, g0;
	uint32_t lbegin[TREEE_SHARD_LEVELS];    // first local node of a level
	uint32_t own_first[TREEE_SHARD_LEVELS]; // its global index
	uint32_t lod_first[TREEE_SHARD_LEVELS]; // global LOD offset of the rank's first entry of a level
};
// local numbering -> global numbering, in place
__global__ __launch_bounds__(256) void k_shard_fix(treee_octree_node *__restrict__ nodes, uint32_t n_nodes, const uint32_t *__restrict__ loff, ShardFix f) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_nodes) return;
	treee_octree_node nd = nodes[i];
	const uint32_t l = nd.level;
	nd.first += f.g0;
	if (l > f.L) nd.parent = f.own_first[l - 1u] + (nd.parent - f.lbegin[l - 1u]);
	if (nd.first_child != NONE32) {
		nd.first_child = f.own_first[l + 1u] + (nd.first_child - f.lbegin[l + 1u]);
		nd.lod_first = f.lod_first[l] + (loff[i] - loff[f.lbegin[l]]);
	}
	nodes[i] = nd;
}
// one wave per own node (global fields already in place): leaves stamp their GLOBAL index over their part of the owned
// range, inner nodes emit their LOD samples as global sorted positions (stored in local node order)
__global__ __launch_bounds__(256) void k_shard_finish(const treee_octree_node *__restrict__ nodes, uint32_t n_nodes, uint32_t lod_cap,
                                                       const uint32_t *__restrict__ loff, ShardFix f, uint32_t *__restrict__ leaf,
                                                       uint32_t *__restrict__ lod) {
	const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (i >= n_nodes) return;
	const int lane = threadIdx.x & 63;
	const treee_octree_node nd = nodes[i];
	if (nd.first_child == NONE32) {
		const uint32_t gid = f.own_first[nd.level] + (i - f.lbegin[nd.level]);
		for (uint32_t j = lane; j < nd.count; j += 64) leaf[nd.first - f.g0 + j] = gid;
	} else {
		const uint32_t m = nd.count < lod_cap ? nd.count : lod_cap;
		const uint32_t off = loff[i];
		for (uint32_t j = lane; j < m; j += 64) lod[off + j] = nd.first + (uint32_t)(((uint64_t)j * nd.count) / m);
	}
}
// the replicated top: LOD samples of its inner nodes; a leaf above the shard level stamps the part of the owned range it covers
__global__ __launch_bounds__(256) void k_top_finish(const treee_octree_node *__restrict__ top, uint32_t n_top, uint32_t lod_cap, uint32_t g0,
                                                     uint32_t n_own, uint32_t *__restrict__ leaf, uint32_t *__restrict__ top_lod) {
	const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (i >= n_top) return;
	const int lane = threadIdx.x & 63;
	const treee_octree_node nd = top[i];
	if (nd.first_child == NONE32) {
		const uint32_t a = nd.first > g0 ? nd.first : g0;
		const uint64_t e0 = (uint64_t)nd.first + nd.count, e1 = (uint64_t)g0 + n_own;
		const uint32_t b = (uint32_t)(e0 < e1 ? e0 : e1);
		for (uint32_t j = a + lane; j < b; j += 64) leaf[j - g0] = i;
	} else {
		const uint32_t m = nd.count < lod_cap ? nd.count : lod_cap;
		for (uint32_t j = lane; j < m; j += 64) top_lod[nd.lod_first + j] = nd.first + (uint32_t)(((uint64_t)j * nd.count) / m);
	}
}

int32_t shard_cube(treee_ctx *ctx, treee_comm *comm, const float *d_xyz, uint64_t n, float corner[3], float *size_out) {
	float mine[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
	if (n) {
		float *d_partial = nullptr;
		uint32_t grid = 0;
		TREEE_TRY(bbox_partials_dev(ctx, d_xyz, n, &d_partial, &grid));
		HostBuf<float> h;
		TREEE_TRY(h.alloc((size_t)grid * 6));
		TREEE_HIP(hipMemcpyAsync(h.data(), d_partial, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		for (uint32_t b = 0; b < grid; ++b)
			for (int a = 0; a < 3; ++a) {
				mine[a] = fminf(mine[a], h[b * 6 + a]);
				mine[3 + a] = fmaxf(mine[3 + a], h[b * 6 + 3 + a]);
			}
	}
	HostBuf<float> all;
	TREEE_TRY(all.alloc((size_t)comm->world * 6));
	TREEE_TRY(comm->all_gather_host(ctx, mine, all.data(), sizeof(mine)));
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, size = 0.0f;
	for (int r = 0; r < comm->world; ++r)
		for (int a = 0; a < 3; ++a) {
			mn[a] = fminf(mn[a], all[(size_t)r * 6 + a]);
			mx[a] = fmaxf(mx[a], all[(size_t)r * 6 + 3 + a]);
		}
	for (int a = 0; a < 3; ++a) { // the rules of bbox_dev (radix_sort.hip): largest extent; degenerate or non-finite -> 1 / 0
		const float e = mx[a] - mn[a];
		if (e > size) size = e;
		corner[a] = (mn[a] > -INFINITY && mn[a] < INFINITY) ? mn[a] : 0.0f;
	}
	if (!(size > 0.0f) || !(size < INFINITY)) size = 1.0f;
	*size_out = size;
	return TREEE_OK;
}

int32_t shard_build(treee_ctx *ctx, treee_comm *comm, const float *d_xyz, uint64_t n, const float *corner_in, float size_in, uint32_t top_level,
                    treee_octree *o) {
	hipStream_t st = ctx->stream;
	const int world = comm->world, rank = comm->rank;
	const uint32_t L = top_level, ncell = 1u << (3u * L);
	float corner[3] = {0, 0, 0}, size = size_in;
	if (corner_in && size_in > 0.0f)
		for (int a = 0; a < 3; ++a) corner[a] = corner_in[a];
	else
		TREEE_TRY(shard_cube(ctx, comm, d_xyz, n, corner, &size));
	// 1. the rank's input in (key, input order): the ordinary Morton sort with the common cube
	auto local_keep = std::make_shared<SortedCloud>();
	SortedCloud &local = *local_keep;
	TREEE_TRY(build_sorted_cloud(ctx, d_xyz, n, corner, size, &local));
	// 2. points per level-L cell, all ranks
	uint32_t *d_counts = nullptr;
	TREEE_TRY(ctx->alloc_t(ncell, &d_counts));
	hipLaunchKernelGGL(k_cell_counts, dim3(div_up(ncell, 256)), dim3(256), 0, st, local.d_keys, (uint32_t)n, 30u - 3u * L, ncell, d_counts);
	TREEE_LAUNCH_CHECK();
	std::vector<uint32_t> h_counts(ncell), all_counts((size_t)world * ncell);
	TREEE_HIP(hipMemcpyAsync(h_counts.data(), d_counts, (size_t)ncell * 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	TREEE_TRY(comm->all_gather_host(ctx, h_counts.data(), all_counts.data(), (size_t)ncell * 4));
	// 3. the plan: owners, the top of the tree, what travels
	treee_shard_plan *plan_raw = nullptr;
	TREEE_TRY(treee_shard_plan_create(all_counts.data(), world, L, o->leaf_cap, o->lod_cap, &plan_raw));
	std::unique_ptr<treee_shard_plan> plan(plan_raw);
	TREEE_TRY(treee_shard_plan_info_get(plan.get(), &o->sh_info));
	TREEE_TRY(treee_shard_plan_rank_get(plan.get(), rank, &o->sh_rank));
	const uint64_t n_own = o->sh_rank.count;
	const uint32_t g0 = (uint32_t)o->sh_rank.first, base = (uint32_t)o->sh_rank.index_base;
	std::vector<uint64_t> send_first(world), send_count(world), recv_count(world);
	TREEE_TRY(treee_shard_plan_exchange(plan.get(), rank, send_first.data(), send_count.data(), recv_count.data()));
	bool anything_moves = false; // (a property of the all-gathered counts: every rank decides alike)
	for (int s = 0; s < world && !anything_moves; ++s) {
		const uint32_t *c = all_counts.data() + (size_t)s * ncell;
		for (uint32_t i = 0; i < ncell && !anything_moves; ++i)
			anything_moves = c[i] != 0 && !(i >= plan->cell_begin[s] && i < plan->cell_begin[s + 1]);
	}
	SortedCloud *sc = o->sc.get();
	sc->n = n_own;
	sc->device = ctx->device;
	sc->owner = ctx;
	for (int a = 0; a < 3; ++a) sc->corner[a] = corner[a];
	sc->size = size;
	sc->crowded = 0;
	if (!anything_moves) { // the input already is tiled by the owners' key ranges: the local sort is the owned range
		if (n) hipLaunchKernelGGL(k_add_index_base, dim3(div_up(n, 256)), dim3(256), 0, st, local.d_sorted, local.d_perm, (uint32_t)n, base);
		TREEE_LAUNCH_CHECK();
		std::swap(sc->d_keys, local.d_keys);
		std::swap(sc->d_perm, local.d_perm);
		std::swap(sc->d_sorted, local.d_sorted);
	} else {
		// 4. one all-to-all of 16-byte records (x, y, z, global index); a rank's pieces are contiguous in its sorted order
		if (n) hipLaunchKernelGGL(k_add_index_base, dim3(div_up(n, 256)), dim3(256), 0, st, local.d_sorted, (uint32_t *)nullptr, (uint32_t)n, base);
		TREEE_LAUNCH_CHECK();
		std::vector<uint64_t> s_off(world), s_len(world), r_off(world), r_len(world);
		uint64_t acc = 0;
		for (int r = 0; r < world; ++r) {
			s_off[r] = send_first[r] * 16;
			s_len[r] = send_count[r] * 16;
			r_off[r] = acc * 16;
			r_len[r] = recv_count[r] * 16;
			acc += recv_count[r];
		}
		if (acc != n_own) return set_err(TREEE_ERR_INVALID, "sharded octree: the plan's receive counts (%llu) and owned range (%llu) disagree", (unsigned long long)acc, (unsigned long long)n_own);
		const size_t cnt = n_own ? n_own : 1;
		float4 *d_recv = nullptr;
		uint32_t *d_keys_raw = nullptr, *d_from = nullptr;
		TREEE_TRY(ctx->alloc_t(cnt, &d_recv));
		TREEE_TRY(ctx->alloc_t(cnt, &d_keys_raw));
		TREEE_TRY(ctx->alloc_t(cnt, &d_from));
		TREEE_TRY(comm->all_to_all_v_dev(ctx, local.d_sorted, s_off.data(), s_len.data(), d_recv, r_off.data(), r_len.data()));
		// 5. the received runs are sorted by (key, global index) each and arrive in rank order, so one stable sort by key of
		// their concatenation is the owned range of the whole cloud's order
		TREEE_TRY(pool_alloc_t(ctx, cnt, &sc->d_keys));
		TREEE_TRY(pool_alloc_t(ctx, cnt, &sc->d_perm));
		TREEE_TRY(pool_alloc_t(ctx, cnt, &sc->d_sorted));
		if (n_own) {
			hipLaunchKernelGGL(k_keys_from_records, dim3(div_up(n_own, 256)), dim3(256), 0, st, d_recv, (uint32_t)n_own, corner[0], corner[1], corner[2], size, d_keys_raw);
			TREEE_LAUNCH_CHECK();
			TREEE_TRY(sort_pairs_dev(ctx, d_keys_raw, nullptr, n_own, 30, sc->d_keys, d_from));
			hipLaunchKernelGGL(k_gather_records, dim3(div_up(n_own, 256)), dim3(256), 0, st, d_from, d_recv, (uint32_t)n_own, sc->d_sorted, sc->d_perm);
			TREEE_LAUNCH_CHECK();
		}
	}
	// 6. the local forest: the rank's nodes of level L are its roots; everything below them is the ordinary build
	const uint32_t n_roots = o->sh_rank.level_node_count;
	const uint64_t per_level = 8ull * (n_own / ((uint64_t)o->leaf_cap + 1ull)) + 8ull;
	uint32_t lvl_cap[DEPTH + 1] = {};
	uint64_t nodes_cap = 0, pow8 = 1;
	for (uint32_t l = 0; l <= DEPTH; ++l) {
		if (l >= L) {
			lvl_cap[l] = (uint32_t)(pow8 < per_level ? pow8 : per_level);
			if (l == L && lvl_cap[l] < n_roots) lvl_cap[l] = n_roots; // (a root's parent may hold most of its points elsewhere)
			nodes_cap += lvl_cap[l];
		}
		pow8 *= 8ull;
	}
	TREEE_TRY(pool_alloc_t(ctx, nodes_cap ? nodes_cap : 1, &o->d_nodes));
	o->nodes_cap = nodes_cap;
	std::vector<treee_octree_node> roots(n_roots ? n_roots : 1);
	TREEE_TRY(treee_shard_plan_level_nodes(plan.get(), rank, roots.data()));
	if (n_roots) TREEE_HIP(hipMemcpyAsync(o->d_nodes, roots.data(), (size_t)n_roots * sizeof(treee_octree_node), hipMemcpyHostToDevice, st));
	uint32_t *d_lvl = nullptr, *d_total = nullptr;
	TREEE_TRY(ctx->alloc_t(2 * (DEPTH + 2), &d_lvl));
	TREEE_TRY(ctx->alloc_t(2, &d_total));
	hipLaunchKernelGGL(k_forest_roots, dim3(1), dim3(1), 0, st, d_lvl + 2 * L, n_roots);
	for (uint32_t level = L; level < DEPTH; ++level) {
		const uint32_t cap = lvl_cap[level];
		uint32_t *d_cnt = nullptr, *d_off = nullptr, *d_bounds = nullptr;
		TREEE_TRY(ctx->alloc_t(cap, &d_cnt));
		TREEE_TRY(ctx->alloc_t(cap, &d_off));
		TREEE_TRY(ctx->alloc_t((size_t)cap * 8, &d_bounds));
		const uint32_t grid = div_up((uint64_t)cap * 8, 256);
		hipLaunchKernelGGL(k_oct_split, dim3(grid), dim3(256), 0, st, sc->d_keys, o->d_nodes, d_lvl + 2 * level, cap, o->leaf_cap, d_cnt, d_bounds);
		TREEE_LAUNCH_CHECK();
		TREEE_TRY(exclusive_scan_u32_dev(ctx, d_cnt, d_off, cap, d_total, d_lvl + 2 * level + 1));
		hipLaunchKernelGGL(k_oct_emit, dim3(grid), dim3(256), 0, st, o->d_nodes, d_lvl + 2 * level, d_off, d_bounds, d_total, (uint32_t *)nullptr);
		TREEE_LAUNCH_CHECK();
	}
	uint32_t *d_lcnt = nullptr, *d_loff = nullptr, *d_sums = nullptr;
	TREEE_TRY(ctx->alloc_t(nodes_cap ? nodes_cap : 1, &d_lcnt));
	TREEE_TRY(ctx->alloc_t(nodes_cap ? nodes_cap : 1, &d_loff));
	TREEE_TRY(ctx->alloc_t(3 * TREEE_SHARD_LEVELS, &d_sums));
	if (nodes_cap) {
		hipLaunchKernelGGL(k_oct_lod_count, dim3(div_up(nodes_cap, 256)), dim3(256), 0, st, o->d_nodes, d_lvl + 2 * DEPTH, (uint32_t)nodes_cap, o->lod_cap, d_lcnt);
		TREEE_LAUNCH_CHECK();
		TREEE_TRY(exclusive_scan_u32_dev(ctx, d_lcnt, d_loff, nodes_cap, d_total));
	}
	hipLaunchKernelGGL(k_level_sums, dim3(1), dim3(64), 0, st, d_lvl, L, d_lcnt, d_loff, d_sums);
	TREEE_LAUNCH_CHECK();
	uint32_t h_sums[3 * TREEE_SHARD_LEVELS];
	TREEE_HIP(hipMemcpyAsync(h_sums, d_sums, sizeof(h_sums), hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	// 7. every rank's nodes and LOD entries per level -> the global numbering
	std::vector<uint32_t> all_sums((size_t)world * 2 * TREEE_SHARD_LEVELS), lc((size_t)world * TREEE_SHARD_LEVELS), dc((size_t)world * TREEE_SHARD_LEVELS);
	TREEE_TRY(comm->all_gather_host(ctx, h_sums, all_sums.data(), 2 * TREEE_SHARD_LEVELS * 4));
	for (int r = 0; r < world; ++r)
		for (uint32_t l = 0; l < TREEE_SHARD_LEVELS; ++l) {
			lc[(size_t)r * TREEE_SHARD_LEVELS + l] = all_sums[(size_t)r * 2 * TREEE_SHARD_LEVELS + l];
			dc[(size_t)r * TREEE_SHARD_LEVELS + l] = all_sums[(size_t)r * 2 * TREEE_SHARD_LEVELS + TREEE_SHARD_LEVELS + l];
		}
	TREEE_TRY(treee_shard_plan_levels(plan.get(), lc.data(), dc.data(), rank, &o->sh_levels));
	ShardFix fix{};
	fix.L = L;
	fix.g0 = g0;
	uint64_t n_nodes = 0, n_lod = 0;
	for (uint32_t l = 0; l < TREEE_SHARD_LEVELS; ++l) {
		fix.lbegin[l] = h_sums[2 * TREEE_SHARD_LEVELS + l];
		fix.own_first[l] = o->sh_levels.own_node_first[l];
		fix.lod_first[l] = o->sh_levels.own_lod_first[l];
		if (l >= L) {
			n_nodes += h_sums[l];
			n_lod += h_sums[TREEE_SHARD_LEVELS + l];
		}
	}
	o->n_nodes = n_nodes;
	o->n_lod = n_lod;
	// 8. leaf ids, LOD lists, the replicated top
	const uint32_t n_top = o->sh_info.n_top_nodes, n_top_lod = o->sh_info.n_top_lod;
	TREEE_TRY(pool_alloc_t(ctx, n_own ? n_own : 1, &o->d_leaf));
	TREEE_TRY(pool_alloc_t(ctx, n_lod ? n_lod : 1, &o->d_lod));
	TREEE_TRY(pool_alloc_t(ctx, n_top ? n_top : 1, &o->d_top_nodes));
	TREEE_TRY(pool_alloc_t(ctx, n_top_lod ? n_top_lod : 1, &o->d_top_lod));
	if (n_top) TREEE_HIP(hipMemcpyAsync(o->d_top_nodes, plan->top.data(), (size_t)n_top * sizeof(treee_octree_node), hipMemcpyHostToDevice, st));
	if (n_nodes) {
		hipLaunchKernelGGL(k_shard_fix, dim3(div_up(n_nodes, 256)), dim3(256), 0, st, o->d_nodes, (uint32_t)n_nodes, d_loff, fix);
		hipLaunchKernelGGL(k_shard_finish, dim3(div_up(n_nodes, 4)), dim3(256), 0, st, o->d_nodes, (uint32_t)n_nodes, o->lod_cap, d_loff, fix, o->d_leaf, o->d_lod);
		TREEE_LAUNCH_CHECK();
	}
	if (n_top) {
		hipLaunchKernelGGL(k_top_finish, dim3(div_up(n_top, 4)), dim3(256), 0, st, o->d_top_nodes, n_top, o->lod_cap, g0, (uint32_t)n_own, o->d_leaf, o->d_top_lod);
		TREEE_LAUNCH_CHECK();
	}
	TREEE_HIP(hipStreamSynchronize(st)); // (the uploads above read host vectors that end with this frame)
	o->shard = true;
	if (anything_moves) o->sc_local = local_keep; // (otherwise its arrays were handed to `sc` above)
	return TREEE_OK;
}

} // namespace

namespace treee {
std::shared_ptr<SortedCloud> octree_sorted_cloud(const treee_octree *o) { return o->sc; }
bool octree_is_shard(const treee_octree *o) { return o && o->shard; }
std::shared_ptr<SortedCloud> octree_shard_local(const treee_octree *o, uint32_t *index_base) {
	if (!o || !o->shard) return nullptr;
	*index_base = (uint32_t)o->sh_rank.index_base;
	if (o->local_released) return nullptr;
	return o->sc_local ? o->sc_local : o->sc;
}
} // namespace treee

extern "C" {

int32_t treee_sort_build_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float corner[3], float size,
                             uint32_t leaf_cap, uint32_t lod_cap, treee_octree **out) {
	if (!ctx || !out || (n && !d_xyz)) return set_err(TREEE_ERR_INVALID, "treee_sort_build_dev: null argument");
	*out = nullptr;
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_octree *o = new treee_octree;
	o->device = ctx->device;
	o->owner = ctx;
	o->leaf_cap = leaf_cap ? leaf_cap : 256;
	o->lod_cap = lod_cap ? lod_cap : 256;
	// the octree levels need the sorted keys only: they run on the context's auxiliary stream beside the gather of the
	// sorted records (k_gather_sorted is a random 12-B gather, 0.33 ms per 15 M points; the levels are ~50 small dependent
	// launches that leave the GPU idle).  Joined before returning.
	int32_t rc = build_sorted_cloud(ctx, d_xyz, n, corner, size, o->sc.get(), /*fork_after_sort=*/true);
	if (rc == TREEE_OK) {
		ctx->on_aux(true);
		rc = build_octree(ctx, o);
	}
	const int32_t jr = ctx->join();
	if (rc == TREEE_OK) rc = jr;
	if (rc != TREEE_OK) {
		treee_octree_free(o);
		return rc;
	}
	*out = o;
	return TREEE_OK;
}

// One call, both consumers of the Morton sort, and the host's launches in the order that keeps the GPU fed (round 5; the three
// calls treee_sort_dev / treee_knn_from_octree_dev / treee_octree_complete_dev on two contexts do the same work, but the host
// waits for the sort's flags before it has enqueued anything else and the octree's ~65 small launches come last -- the
// stage then ends when the host has finished launching them, 1.35 ms per 15 M points):
//   stream:  sort | record gather ..................... | kNN order by blocks, boxes, dispatch table
//   aux:          | crowded-cell flags, cube, block tables of the kNN order | octree levels (small dependent launches) ... | leaf ids, LOD lists
//   host:    enqueues all of the first two rows, waits for the FLAGS only (they are done long before the gather), enqueues
//            the kNN structure behind the gather, then waits for the node count and enqueues the last two kernels.
int32_t treee_sort_build_knn_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float corner[3], float size, uint32_t leaf_cap, uint32_t lod_cap,
                                 treee_octree **octree_out, treee_knn **knn_out) {
	if (!ctx || !octree_out || !knn_out || (n && !d_xyz)) return set_err(TREEE_ERR_INVALID, "treee_sort_build_knn_dev: null argument");
	*octree_out = nullptr;
	*knn_out = nullptr;
	if (n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_octree *o = new (std::nothrow) treee_octree;
	if (!o) return set_err(TREEE_ERR_HIP, "treee_sort_build_knn_dev: out of host memory");
	o->device = ctx->device;
	o->owner = ctx;
	o->leaf_cap = leaf_cap ? leaf_cap : 256;
	o->lod_cap = lod_cap ? lod_cap : 256;
	treee_knn *t = nullptr;
	OctreePending pend;
	BlockTables tables;
	// TREEE_TRACE_BUILD=1: where each stream and the host are at the call's milestones (timing events; stderr) -- a measuring hook
	static const bool trace = getenv("TREEE_TRACE_BUILD") != nullptr;
	struct Mark {
		const char *what;
		hipEvent_t ev;
		double host_us;
	};
	Mark marks[12];
	int n_marks = 0;
	const auto t_host0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (!trace || n_marks == 12) return;
		hipEvent_t ev = nullptr;
		if (hipEventCreate(&ev) != hipSuccess || hipEventRecord(ev, ctx->stream) != hipSuccess) return;
		marks[n_marks++] = {what, ev, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count()};
	};
	mark("start (stream)");
	int32_t rc = build_sorted_cloud(ctx, d_xyz, n, corner, size, o->sc.get(), /*fork_after_sort=*/true);
	mark("sort + gather (stream)");
	if (rc == TREEE_OK) {
		ctx->on_aux(true);
		mark("after the sort (aux)");
		rc = cloud_flags_enqueue(ctx, o->sc.get());
		mark("flags (aux)");
		if (rc == TREEE_OK) rc = knn_block_tables_enqueue(ctx, o->sc.get(), &tables);
		mark("block tables (aux)");
		if (rc == TREEE_OK && hipEventRecord(ctx->ev_mid, ctx->stream) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "treee_sort_build_knn_dev: event");
		if (rc == TREEE_OK) rc = octree_levels_enqueue(ctx, o, &pend);
		mark("octree levels (aux)");
		ctx->on_aux(false);
		// (the kNN structure reads the block tables: behind the point of aux where they are made, not behind the levels)
		if (hipStreamWaitEvent(ctx->stream, ctx->ev_mid, 0) != hipSuccess && rc == TREEE_OK) rc = set_err(TREEE_ERR_HIP, "treee_sort_build_knn_dev: wait");
		if (rc == TREEE_OK && hipEventSynchronize(ctx->ev_mid) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "treee_sort_build_knn_dev: synchronise (flags)");
		if (rc == TREEE_OK) {
			cloud_flags_commit(ctx, o->sc.get());
			mark("host has the flags (stream)");
			if (rc == TREEE_OK) rc = knn_from_sorted_cloud(ctx, o->sc, &t, &tables); // on the context's stream, behind the gather
			mark("kNN structure (stream)");
		}
		ctx->on_aux(true);
		if (rc == TREEE_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "treee_sort_build_knn_dev: synchronise (node count)");
		if (rc == TREEE_OK) rc = octree_finish(ctx, o, pend);
		mark("leaf ids + LOD lists (aux)");
	}
	const int32_t jr = ctx->join();
	if (rc == TREEE_OK) rc = jr;
	if (trace && n_marks) {
		(void)hipStreamSynchronize(ctx->stream);
		for (int i = 0; i < n_marks; ++i) {
			const Mark &m = marks[i];
			float ms = 0.0f;
			(void)hipEventElapsedTime(&ms, marks[0].ev, m.ev);
			fprintf(stderr, "[treee build] %-32s gpu %8.1f us   host enqueued it at %8.1f us\n", m.what, ms * 1000.0f, m.host_us);
		}
		for (int i = 0; i < n_marks; ++i) (void)hipEventDestroy(marks[i].ev);
	}
	if (rc != TREEE_OK) {
		if (t) treee_knn_free(t);
		treee_octree_free(o);
		return rc;
	}
	*octree_out = o;
	*knn_out = t;
	return TREEE_OK;
}

int32_t treee_sort_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float corner[3], float size, treee_octree **out) {
	if (!ctx || !out || (n && !d_xyz)) return set_err(TREEE_ERR_INVALID, "treee_sort_dev: null argument");
	*out = nullptr;
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_octree *o = new treee_octree;
	o->device = ctx->device;
	o->owner = ctx;
	o->leaf_cap = o->lod_cap = 0; // no node table yet (treee_octree_complete_dev)
	SortedCloud *sc = o->sc.get();
	int32_t rc = build_sorted_cloud(ctx, d_xyz, n, corner, size, sc);
	// one round trip brings what a kNN structure over this cloud and the handle's info need on the host: the cube when it
	// was computed from the data, and whether a Morton cell is crowded (refine_dense_cells) -- so neither asks again
	if (rc == TREEE_OK && (sc->cube_pending || n > LONG_RUN)) {
		uint32_t *h = ctx->h_scalars, *d_flag = nullptr;
		float *h_cube = (float *)(h + 4);
		rc = ctx->alloc_t(1, &d_flag);
		if (rc == TREEE_OK) rc = long_run_flag_dev(ctx, sc->d_keys, n, d_flag);
		if (rc == TREEE_OK && hipMemcpyAsync(h, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "flag copy");
		uint32_t *d_cells = nullptr;
		if (rc == TREEE_OK) rc = ctx->alloc_t(2, &d_cells);
		if (rc == TREEE_OK) rc = distinct_keys_dev(ctx, sc->d_keys, n, d_cells);
		if (rc == TREEE_OK && hipMemcpyAsync(h + 8, d_cells, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "cell count copy");
		if (rc == TREEE_OK) rc = cube_fetch_async(ctx, sc, h_cube);
		if (rc == TREEE_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "treee_sort_dev: synchronise");
		if (rc == TREEE_OK) {
			sc->crowded = (int)(h[0] & 1u);
			sc->block7_big = (int)((h[0] >> 1) & 1u);
			sc->cell_share = h[9] ? (float)h[8] / (float)h[9] : 1.0f;
			cube_commit(sc, h_cube);
		}
	}
	if (rc == TREEE_OK) rc = mark_ready(ctx, &sc->sync); // what a completion on another stream waits for
	if (rc != TREEE_OK) {
		treee_octree_free(o);
		return rc;
	}
	*out = o;
	return TREEE_OK;
}

int32_t treee_octree_complete_dev(treee_ctx *ctx, treee_octree *o, uint32_t leaf_cap, uint32_t lod_cap) {
	if (!ctx || !o) return set_err(TREEE_ERR_INVALID, "treee_octree_complete_dev: null argument");
	if (o->d_nodes || o->leaf_cap) return set_err(TREEE_ERR_INVALID, "treee_octree_complete_dev: the handle already has its node table");
	if (ctx->device != o->device) return set_err(TREEE_ERR_INVALID, "treee_octree_complete_dev: context of another device");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	o->owner = ctx; // the node table, leaf ids and LOD lists come from (and return to) this context's pool
	o->leaf_cap = leaf_cap ? leaf_cap : 256;
	o->lod_cap = lod_cap ? lod_cap : 256;
	const int32_t rc = build_octree(ctx, o);
	if (rc != TREEE_OK) { // back to the sort-only state
		pool_release(o->owner, o->d_nodes);
		pool_release(o->owner, o->d_leaf);
		pool_release(o->owner, o->d_lod);
		o->d_nodes = nullptr;
		o->d_leaf = o->d_lod = nullptr;
		o->n_nodes = o->n_lod = 0;
		o->leaf_cap = o->lod_cap = 0;
	}
	return rc;
}

int32_t treee_sort_build(const float *xyz, uint64_t n, const float corner[3], float size, uint32_t leaf_cap,
                         uint32_t lod_cap, treee_octree **out) {
	if (!out || (n && !xyz)) return set_err(TREEE_ERR_INVALID, "treee_sort_build: null argument");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_xyz;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
	TREEE_TRY(treee_sort_build_dev(ctx, d_xyz.p, n, corner, size, leaf_cap, lod_cap, out));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_shard_octree_release_local(treee_octree *o) {
	if (!o) return set_err(TREEE_ERR_INVALID, "treee_shard_octree_release_local: null argument");
	if (!o->shard) return set_err(TREEE_ERR_INVALID, "treee_shard_octree_release_local: not a sharded octree");
	o->sc_local.reset(); // (the arrays return to their owner's pool behind the uses its stream knows of: SortedCloud::release)
	o->local_released = true;
	return TREEE_OK;
}

int32_t treee_shard_cube_dev(treee_ctx *ctx, treee_comm *comm, const float *d_xyz, uint64_t n, float corner_out[3], float *size_out) {
	if (!ctx || !comm || !corner_out || !size_out || (n && !d_xyz)) return set_err(TREEE_ERR_INVALID, "treee_shard_cube_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return shard_cube(ctx, comm, d_xyz, n, corner_out, size_out);
}

int32_t treee_shard_octree_build_dev(treee_ctx *ctx, treee_comm *comm, const float *d_xyz, uint64_t n, const float corner[3], float size,
                                     uint32_t leaf_cap, uint32_t lod_cap, uint32_t top_level, treee_octree **out) {
	if (!ctx || !comm || !out || (n && !d_xyz)) return set_err(TREEE_ERR_INVALID, "treee_shard_octree_build_dev: null argument");
	*out = nullptr;
	if (top_level == 0) top_level = 5;
	if (top_level > 6) return set_err(TREEE_ERR_INVALID, "treee_shard_octree_build_dev: top_level %u outside [1, 6]", top_level);
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_octree *o = new treee_octree;
	o->device = ctx->device;
	o->owner = ctx;
	o->leaf_cap = leaf_cap ? leaf_cap : 256;
	o->lod_cap = lod_cap ? lod_cap : 256;
	int32_t rc;
	try {
		rc = shard_build(ctx, comm, d_xyz, n, corner, size, top_level, o);
	} catch (const std::exception &e) {
		rc = set_err(TREEE_ERR_NOMEM, "treee_shard_octree_build_dev: %s", e.what());
	}
	if (rc != TREEE_OK) {
		(void)hipStreamSynchronize(ctx->stream); // nothing in flight may still read this frame's host buffers
		treee_octree_free(o);
		return rc;
	}
	*out = o;
	return TREEE_OK;
}

int32_t treee_shard_octree_info_get(const treee_octree *o, treee_shard_plan_info *info, treee_shard_plan_rank *rank, treee_shard_levels *levels) {
	if (!o) return set_err(TREEE_ERR_INVALID, "treee_shard_octree_info_get: null argument");
	if (!o->shard) return set_err(TREEE_ERR_INVALID, "treee_shard_octree_info_get: not a sharded octree");
	if (info) *info = o->sh_info;
	if (rank) *rank = o->sh_rank;
	if (levels) *levels = o->sh_levels;
	return TREEE_OK;
}

int32_t treee_octree_free(treee_octree *o) {
	if (!o) return TREEE_OK;
	(void)hipSetDevice(o->device);
	o->sc.reset();
	o->sc_local.reset();
	pool_release(o->owner, o->d_nodes);
	pool_release(o->owner, o->d_leaf);
	pool_release(o->owner, o->d_lod);
	pool_release(o->owner, o->d_top_nodes);
	pool_release(o->owner, o->d_top_lod);
	delete o;
	return TREEE_OK;
}

int32_t treee_octree_info_get(const treee_octree *o, treee_octree_info *info) {
	if (!o || !info) return set_err(TREEE_ERR_INVALID, "treee_octree_info_get: null argument");
	info->n_points = o->sc->n;
	info->n_nodes = o->n_nodes;
	info->n_lod = o->n_lod;
	for (int a = 0; a < 3; ++a) info->corner[a] = o->sc->corner[a];
	info->size = o->sc->size;
	info->leaf_cap = o->leaf_cap;
	info->lod_cap = o->lod_cap;
	return TREEE_OK;
}

static int32_t oct_array(const treee_octree *o, int32_t what, const void **p, uint64_t *bytes) {
	switch (what) {
	case TREEE_OCT_KEYS: *p = o->sc->d_keys; *bytes = o->sc->n * 4; break;
	case TREEE_OCT_PERM: *p = o->sc->d_perm; *bytes = o->sc->n * 4; break;
	case TREEE_OCT_NODES: *p = o->d_nodes; *bytes = o->n_nodes * sizeof(treee_octree_node); break;
	case TREEE_OCT_LEAF: *p = o->d_leaf; *bytes = o->sc->n * 4; break;
	case TREEE_OCT_LOD: *p = o->d_lod; *bytes = o->n_lod * 4; break;
	case TREEE_OCT_SORTED_XYZ: *p = o->sc->d_sorted; *bytes = o->sc->n * 16; break;
	case TREEE_OCT_TOP_NODES: *p = o->d_top_nodes; *bytes = o->shard ? o->sh_info.n_top_nodes * sizeof(treee_octree_node) : 0; break;
	case TREEE_OCT_TOP_LOD: *p = o->d_top_lod; *bytes = o->shard ? (uint64_t)o->sh_info.n_top_lod * 4 : 0; break;
	default: return set_err(TREEE_ERR_INVALID, "unknown octree array id %d", what);
	}
	return TREEE_OK;
}

int32_t treee_octree_copy(const treee_octree *o, int32_t what, void *dst, uint64_t dst_bytes) {
	if (!o || !dst) return set_err(TREEE_ERR_INVALID, "treee_octree_copy: null argument");
	const void *p = nullptr;
	uint64_t bytes = 0;
	TREEE_TRY(oct_array(o, what, &p, &bytes));
	if (dst_bytes < bytes) return set_err(TREEE_ERR_INVALID, "destination too small: %llu < %llu", (unsigned long long)dst_bytes, (unsigned long long)bytes);
	TREEE_HIP(hipSetDevice(o->device));
	TREEE_HIP(hipDeviceSynchronize());
	if (bytes) TREEE_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
	return TREEE_OK;
}

int32_t treee_octree_dev_ptr(const treee_octree *o, int32_t what, void **d_ptr_out) {
	if (!o || !d_ptr_out) return set_err(TREEE_ERR_INVALID, "treee_octree_dev_ptr: null argument");
	const void *p = nullptr;
	uint64_t bytes = 0;
	TREEE_TRY(oct_array(o, what, &p, &bytes));
	*d_ptr_out = (void *)p;
	return TREEE_OK;
}

} // extern "C"
