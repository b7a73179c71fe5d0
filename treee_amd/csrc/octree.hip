// octree.hip -- octree node table + LOD sample lists from Morton-sorted keys.
// BUILD-DEFINED: no reference counterpart exists at this
// snapshot; definition and bit-exact oracle: oracle/octree_oracle.cpp.
//
// Top-down, one level per step.  A node's children are the non-empty runs of the next
// 3 key bits inside its sorted range, found by binary search (8 lanes per parent, one
// boundary each), so the per-point arrays are touched only by the final leaf-id and LOD
// kernels (4 B/point written, coalesced).
#include "common.h"

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

__global__ __launch_bounds__(256) void k_oct_emit(treee_octree_node *__restrict__ nodes, uint32_t *__restrict__ lvl,
                                                   const uint32_t *__restrict__ child_off, const uint32_t *__restrict__ bounds,
                                                   const uint32_t *__restrict__ total) {
	const uint32_t lvl_begin = lvl[0], lvl_count = lvl[1];
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t == 0) { // descriptor of the next level
		lvl[2] = lvl_begin + lvl_count;
		lvl[3] = *total;
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
int32_t build_octree_on(treee_ctx *ctx, treee_octree *o) {
	const uint64_t n = o->sc->n;
	if (n == 0) return cube_to_host(ctx, o->sc.get());
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
		hipLaunchKernelGGL(k_oct_emit, dim3(grid), dim3(256), 0, st, o->d_nodes, d_lvl + 2 * level, d_off, d_bounds, d_total);
		TREEE_LAUNCH_CHECK();
	}
	// LOD offsets, then the only host round trip of the build: node count + LOD total
	uint32_t *d_lcnt = nullptr, *d_loff = nullptr;
	TREEE_TRY(ctx->alloc_t(nodes_cap, &d_lcnt));
	TREEE_TRY(ctx->alloc_t(nodes_cap, &d_loff));
	hipLaunchKernelGGL(k_oct_lod_count, dim3(div_up(nodes_cap, 256)), dim3(256), 0, st, o->d_nodes, d_lvl + 2 * DEPTH, (uint32_t)nodes_cap, o->lod_cap, d_lcnt);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_lcnt, d_loff, nodes_cap, d_total));
	// the round trip lands in the context's pinned scalars: [0] crowded, [1..2] last level, [3] LOD total, [4..7] cube
	uint32_t *h = ctx->h_scalars;
	float *h_cube = (float *)(h + 4);
	uint32_t *d_crowded = nullptr; // asked here so that a kNN structure sharing this cloud needs no round trip of its own
	TREEE_TRY(ctx->alloc_t(1, &d_crowded));
	TREEE_TRY(long_run_flag_dev(ctx, o->sc->d_keys, n, d_crowded));
	TREEE_HIP(hipMemcpyAsync(h + 0, d_crowded, 4, hipMemcpyDeviceToHost, st));
	TREEE_TRY(cube_fetch_async(ctx, o->sc.get(), h_cube));
	TREEE_HIP(hipMemcpyAsync(h + 1, d_lvl + 2 * DEPTH, 8, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipMemcpyAsync(h + 3, d_total, 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	const uint32_t lod_total = h[3];
	o->n_nodes = (uint64_t)h[1] + h[2];
	o->n_lod = lod_total;
	o->sc->crowded = (int)h[0];
	cube_commit(o->sc.get(), h_cube);
	const uint32_t nn = (uint32_t)o->n_nodes;
	TREEE_TRY(pool_alloc_t(ctx, n, &o->d_leaf));
	TREEE_TRY(pool_alloc_t(ctx, (size_t)(lod_total ? lod_total : 1), &o->d_lod));
	hipLaunchKernelGGL(k_oct_finish, dim3(div_up(nn, 4)), dim3(256), 0, st, o->d_nodes, nn, o->lod_cap, d_loff, o->d_leaf, o->d_lod);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

} // namespace

namespace treee {
std::shared_ptr<SortedCloud> octree_sorted_cloud(const treee_octree *o) { return o->sc; }
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
		if (rc == TREEE_OK) rc = cube_fetch_async(ctx, sc, h_cube);
		if (rc == TREEE_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(TREEE_ERR_HIP, "treee_sort_dev: synchronise");
		if (rc == TREEE_OK) {
			sc->crowded = (int)h[0];
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

int32_t treee_octree_free(treee_octree *o) {
	if (!o) return TREEE_OK;
	(void)hipSetDevice(o->device);
	o->sc.reset();
	pool_release(o->owner, o->d_nodes);
	pool_release(o->owner, o->d_leaf);
	pool_release(o->owner, o->d_lod);
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
