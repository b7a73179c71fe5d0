// knn.hip -- exact k-nearest-neighbour search replacing the `k-nearest` crate
// (k-nearest/src/kd_tree.rs:35-146, best_set.rs:20-131, metric.rs:21-31) and its only
// consumer, the curvature loop of treee/src/calculations.rs:328-373 (+ fast_eigenvalues
// :467-498).
//
// MI355X design (not a KD-tree port): the cloud is Morton-sorted (radix_sort.hip), cut into
// buckets of 64 consecutive points (one wave-load of float4 records), and an implicit
// binary AABB tree is laid over the buckets.  One wave = 64 Morton-adjacent queries walks
// that tree as a packet, bottom-up from its own bucket, staging each visited bucket in LDS
// and broadcasting candidates to all lanes.  A subtree is skipped when no lane's box
// distance is <= that lane's current k-th distance; float rounding is monotone, so the
// skip is exact w.r.t. the reference metric ((0+dx*dx)+dy*dy)+dz*dz, dx = p - q, no FMA.
//
// Semantics kept from the reference: squared distances, strict `d < max_distance`
// (best_set.rs:20-28 initial root), ascending output, count of valid entries, NaN never
// inserted.  Tie rule: the reference resolves equal distances by KD-tree
// visit order; here the total order (distance, sorted position) decides, i.e. among equal
// distances the smaller (morton key, original index) wins.  Distances and counts are
// bit-identical to the reference; indices are identical whenever distances are distinct.
#include "common.h"

using namespace treee;

struct treee_knn {
	std::shared_ptr<SortedCloud> sc = std::make_shared<SortedCloud>();
	uint32_t nb = 0;       // buckets of 64 points
	uint32_t P = 1;        // leaves padded to a power of two; node i: children 2i, 2i+1; leaf b = P + b
	float4 *d_bvh = nullptr; // [2P][2] : (min.xyz,0), (max.xyz,0)
	int device = 0;
};

namespace {

constexpr int BUCKET = 64;

// ---- bucket AABBs and the implicit tree ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_leaf_aabb(const float4 *__restrict__ pts, uint32_t n, uint32_t nb, uint32_t P,
                                                    float4 *__restrict__ bvh) {
	const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (b >= P) return;
	const int lane = threadIdx.x & 63;
	const uint32_t i = b * BUCKET + lane;
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
	if (b < nb && i < n) {
		const float4 p = pts[i];
		mn[0] = mx[0] = p.x;
		mn[1] = mx[1] = p.y;
		mn[2] = mx[2] = p.z;
		// NaN coordinates never produce a valid distance; keep them out of the boxes
#pragma unroll
		for (int a = 0; a < 3; ++a)
			if (mn[a] != mn[a]) {
				mn[a] = INFINITY;
				mx[a] = -INFINITY;
			}
	}
#pragma unroll
	for (int a = 0; a < 3; ++a)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
			mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
		}
	if (lane == 0) {
		bvh[2ull * (P + b) + 0] = make_float4(mn[0], mn[1], mn[2], 0.0f);
		bvh[2ull * (P + b) + 1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
	}
}

__global__ __launch_bounds__(256) void k_inner_aabb(float4 *__restrict__ bvh, uint32_t first, uint32_t count) {
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= count) return;
	const uint32_t i = first + t;
	const float4 a0 = bvh[4ull * i + 0], a1 = bvh[4ull * i + 1], b0 = bvh[4ull * i + 2], b1 = bvh[4ull * i + 3];
	bvh[2ull * i + 0] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), 0.0f);
	bvh[2ull * i + 1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), 0.0f);
}

// ---- per-lane top-k: binary max-heap in LDS, total order (distance bits, sorted position) ------
// Layout [slot][lane] (stride 64 dwords): every lane of a wave hits its own bank.
struct Heap {
	uint32_t *d, *p;
	int k, lane;
	__device__ __forceinline__ uint32_t &D(int s) { return d[s * 64 + lane]; }
	__device__ __forceinline__ uint32_t &Pp(int s) { return p[s * 64 + lane]; }
};

__device__ __forceinline__ bool key_gt(uint32_t ad, uint32_t ap, uint32_t bd, uint32_t bp) {
	return ad > bd || (ad == bd && ap > bp);
}

// place (nd,np) at the root hole and sift down inside [0,size)  (best_set.rs:30-54,77-81)
__device__ __forceinline__ void heap_sift(Heap &h, int size, uint32_t nd, uint32_t np) {
	int i = 0;
	for (;;) {
		const int l = 2 * i + 1;
		if (l >= size) break;
		int c = l;
		uint32_t cd = h.D(l), cp = h.Pp(l);
		if (l + 1 < size) {
			const uint32_t rd = h.D(l + 1), rp = h.Pp(l + 1);
			if (key_gt(rd, rp, cd, cp)) {
				c = l + 1;
				cd = rd;
				cp = rp;
			}
		}
		if (!key_gt(cd, cp, nd, np)) break;
		h.D(i) = cd;
		h.Pp(i) = cp;
		i = c;
	}
	h.D(i) = nd;
	h.Pp(i) = np;
}

// nalgebra Matrix3::determinant (cofactor expansion, first row)
__device__ __forceinline__ float det3(const float m[3][3]) {
	const float a = m[1][1] * m[2][2] - m[2][1] * m[1][2];
	const float b = m[1][0] * m[2][2] - m[2][0] * m[1][2];
	const float c = m[1][0] * m[2][1] - m[2][0] * m[1][1];
	return m[0][0] * a - m[0][1] * b + m[0][2] * c;
}

// treee/src/calculations.rs:467-498
__device__ __forceinline__ void fast_eigenvalues(const float mat[3][3], float out[3]) {
	const float p1 = mat[0][1] * mat[0][1] + mat[0][2] * mat[0][2] + mat[1][2] * mat[1][2];
	if (p1 == 0.0f) {
		out[0] = mat[0][0];
		out[1] = mat[1][1];
		out[2] = mat[2][2];
		return;
	}
	const float q = (mat[0][0] + mat[1][1] + mat[2][2]) / 3.0f;
	const float e0 = mat[0][0] - q, e1 = mat[1][1] - q, e2 = mat[2][2] - q;
	const float p2 = e0 * e0 + e1 * e1 + e2 * e2 + 2.0f * p1;
	const float p = sqrtf(p2 / 6.0f);
	float mb[3][3];
#pragma unroll
	for (int r = 0; r < 3; ++r)
#pragma unroll
		for (int c = 0; c < 3; ++c) mb[r][c] = mat[r][c];
	mb[0][0] -= q;
	mb[1][1] -= q;
	mb[2][2] -= q;
	const float powi_m3 = 1.0f / (p * (p * p));
	const float r = det3(mb) / 2.0f * powi_m3;
	const float PI = 3.14159265358979323846f;
	float phi;
	if (r <= -1.0f)
		phi = PI / 3.0f;
	else if (r >= 1.0f)
		phi = 0.0f;
	else
		phi = acosf(r) / 3.0f;
	const float eig1 = q + 2.0f * p * cosf(phi);
	const float eig3 = q + 2.0f * p * cosf(phi + (2.0f * PI / 3.0f));
	const float eig2 = 3.0f * q - eig1 - eig3;
	out[0] = eig1;
	out[1] = eig2;
	out[2] = eig3;
}

struct KnnArgs {
	const float4 *pts; // sorted records (x,y,z,bitcast(orig))
	const float4 *bvh;
	uint32_t n, nb, P;
	// queries: self mode (q_xyz == nullptr): query j = pts[j], output row = orig index
	const float *q_xyz;          // AoS, original order
	const uint32_t *q_order;     // sorted query slot -> original query index
	const uint32_t *q_start;     // per wave: start bucket
	uint32_t nq;
	uint32_t k, maxbits;
	float *out_dist;
	uint32_t *out_index;
	uint32_t *out_count;
	float *out_curve;
};

// one wave per workgroup; dynamic LDS = k*64*8 (heap) + 64*16 (candidate tile)
__global__ __launch_bounds__(64) void k_knn_heap(KnnArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x;
	float4 *tile = (float4 *)smem;
	Heap h;
	h.d = (uint32_t *)(smem + 64 * 16);
	h.p = h.d + (size_t)a.k * 64;
	h.k = (int)a.k;
	h.lane = lane;

	const uint32_t slot = blockIdx.x * 64u + lane;
	const bool live = slot < a.nq;
	float qx = 0.0f, qy = 0.0f, qz = 0.0f;
	uint32_t row = 0, start;
	if (a.q_xyz) {
		if (live) {
			row = a.q_order[slot];
			qx = a.q_xyz[3ull * row + 0];
			qy = a.q_xyz[3ull * row + 1];
			qz = a.q_xyz[3ull * row + 2];
		}
		start = a.q_start[blockIdx.x];
	} else {
		if (live) {
			const float4 p = a.pts[slot];
			qx = p.x;
			qy = p.y;
			qz = p.z;
			row = __float_as_uint(p.w);
		}
		start = blockIdx.x;
	}
	for (int s = 0; s < h.k; ++s) { // FixedSet::new (best_set.rs:20-28)
		h.D(s) = a.maxbits;
		h.Pp(s) = 0u;
	}
	uint32_t wd = a.maxbits, wp = 0u; // root = current worst

	auto scan_bucket = [&](uint32_t b) {
		const uint32_t base = b * BUCKET;
		if (base >= a.n) return;
		const uint32_t cnt = (a.n - base) < (uint32_t)BUCKET ? (a.n - base) : (uint32_t)BUCKET;
		if ((uint32_t)lane < cnt) tile[lane] = a.pts[base + lane];
		__syncthreads(); // single-wave workgroup: orders the LDS tile write before the broadcasts
		for (uint32_t j = 0; j < cnt; ++j) {
			const float4 c = tile[j]; // same address in every lane: LDS broadcast
			const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
			const float d = ((0.0f + dx * dx) + dy * dy) + dz * dz; // metric.rs:21-26
			const uint32_t db = __float_as_uint(d), pos = base + j;
			if (live && (db < wd || (db == wd && pos < wp))) {
				heap_sift(h, h.k, db, pos); // FixedSet::insert (best_set.rs:77-81)
				wd = h.D(0);
				wp = h.Pp(0);
			}
		}
		__syncthreads();
	};

	auto box_hit = [&](uint32_t node) -> bool {
		const float4 lo = a.bvh[2ull * node + 0], hi = a.bvh[2ull * node + 1];
		const float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.0f);
		const float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.0f);
		const float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.0f);
		const float bd = ((0.0f + dx * dx) + dy * dy) + dz * dz;
		return __any(live && __float_as_uint(bd) <= wd);
	};

	if (start >= a.nb) start = a.nb - 1;
	scan_bucket(start);
	uint32_t node = a.P + start;
	while (node > 1u) { // bottom-up: sibling subtrees, nearest in Morton order first
		const uint32_t root = node ^ 1u;
		uint32_t i = root;
		for (;;) {
			const bool hit = box_hit(i);
			if (hit && i < a.P) {
				i = 2u * i;
				continue;
			}
			if (hit) scan_bucket(i - a.P);
			while (i != root && (i & 1u)) i >>= 1;
			if (i == root) break;
			i += 1u;
		}
		node >>= 1;
	}

	// FixedSet::result (best_set.rs:56-66): in-place heapsort -> ascending
	for (int end = h.k - 1; end > 0; --end) {
		const uint32_t ed = h.D(end), ep = h.Pp(end);
		h.D(end) = h.D(0);
		h.Pp(end) = h.Pp(0);
		heap_sift(h, end, ed, ep);
	}
	uint32_t count = 0;
	for (int s = 0; s < h.k; ++s) count += h.D(s) < a.maxbits ? 1u : 0u;
	if (!live) return;

	if (a.out_count) a.out_count[row] = count;
	if (a.out_dist || a.out_index) {
		for (int s = 0; s < h.k; ++s) {
			const bool ok = (uint32_t)s < count;
			if (a.out_dist) a.out_dist[(size_t)row * a.k + s] = __uint_as_float(ok ? h.D(s) : a.maxbits);
			if (a.out_index) a.out_index[(size_t)row * a.k + s] = ok ? __float_as_uint(a.pts[h.Pp(s)].w) : NONE32;
		}
	}
	if (a.out_curve) { // treee/src/calculations.rs:342-368
		const float len = (float)count;
		float mean[3] = {0.0f, 0.0f, 0.0f};
		for (uint32_t s = 0; s < count; ++s) {
			const float4 p = a.pts[h.Pp(s)];
			mean[0] += p.x;
			mean[1] += p.y;
			mean[2] += p.z;
		}
		mean[0] = mean[0] / len;
		mean[1] = mean[1] / len;
		mean[2] = mean[2] / len;
		float var[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
		for (uint32_t s = 0; s < count; ++s) {
			const float4 p = a.pts[h.Pp(s)];
			const float df[3] = {p.x - mean[0], p.y - mean[1], p.z - mean[2]};
#pragma unroll
			for (int x = 0; x < 3; ++x)
#pragma unroll
				for (int y = 0; y < 3; ++y) var[x][y] += df[x] * df[y];
		}
#pragma unroll
		for (int x = 0; x < 3; ++x)
#pragma unroll
			for (int y = 0; y < 3; ++y) var[x][y] /= len;
		float eig[3];
		fast_eigenvalues(var, eig);
		a.out_curve[row] = (3.0f * eig[2]) / (eig[0] + eig[1] + eig[2]);
	}
}

// start bucket of each query wave: the bucket holding the first query's Morton key
__global__ __launch_bounds__(256) void k_query_start(const uint32_t *__restrict__ tree_keys, uint32_t n, uint32_t nb,
                                                      const uint32_t *__restrict__ q_keys_sorted, uint32_t nq,
                                                      uint32_t nwaves, uint32_t *__restrict__ q_start) {
	const uint32_t w = blockIdx.x * 256u + threadIdx.x;
	if (w >= nwaves) return;
	const uint32_t key = q_keys_sorted[w * 64u < nq ? w * 64u : nq - 1];
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (tree_keys[mid] < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	uint32_t b = lo / BUCKET;
	q_start[w] = b < nb ? b : nb - 1;
}

__global__ __launch_bounds__(256) void k_map_to_u32(const float *__restrict__ in, uint32_t n, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float x = in[i] * 4294967296.0f; // u32::MAX as f32 == 2^32
	uint32_t r;
	if (!(x > 0.0f))
		r = 0u;
	else if (x >= 4294967296.0f)
		r = 0xFFFFFFFFu;
	else
		r = (uint32_t)x;
	out[i] = r;
}

__global__ __launch_bounds__(256) void k_widen_entries(const float *__restrict__ dist, const uint32_t *__restrict__ idx, uint64_t total,
                                                        treee_entry *__restrict__ out) {
	const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (i >= total) return;
	treee_entry e;
	e.distance = dist[i];
	e._pad = 0;
	e.index = idx[i] == NONE32 ? ~0ull : (uint64_t)idx[i];
	out[i] = e;
}

uint32_t max_bits(float max_dist_sq) {
	if (!(max_dist_sq > 0.0f)) return 0u; // NaN, negative, zero: nothing is < max
	uint32_t b;
	memcpy(&b, &max_dist_sq, 4);
	return b;
}

int32_t launch_knn(treee_ctx *ctx, const treee_knn *t, const float *d_q, uint64_t nq, uint32_t k, float max_dist_sq,
                   float *d_dist, uint32_t *d_index, uint32_t *d_counts, float *d_curve) {
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0 (the reference panics at best_set.rs:74)");
	if (nq == 0) return TREEE_OK;
	if (nq >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "nq too large");
	const size_t lds = (size_t)k * 64 * 8 + 64 * 16;
	if (lds > 160 * 1024) return set_err(TREEE_ERR_UNSUPPORTED, "k=%u exceeds the LDS-resident limit (319)", k);
	KnnArgs a;
	a.pts = t->sc->d_sorted;
	a.bvh = t->d_bvh;
	a.n = (uint32_t)t->sc->n;
	a.nb = t->nb;
	a.P = t->P;
	a.q_xyz = nullptr;
	a.q_order = nullptr;
	a.q_start = nullptr;
	a.nq = (uint32_t)nq;
	a.k = k;
	a.maxbits = max_bits(max_dist_sq);
	a.out_dist = d_dist;
	a.out_index = d_index;
	a.out_count = d_counts;
	a.out_curve = d_curve;
	const uint32_t nwaves = div_up(nq, 64);
	if (d_q) { // external queries: Morton-sort them so that a wave is spatially compact
		uint32_t *d_qk = nullptr, *d_qks = nullptr, *d_qo = nullptr, *d_qs = nullptr;
		TREEE_TRY(ctx->alloc_t(nq, &d_qk));
		TREEE_TRY(ctx->alloc_t(nq, &d_qks));
		TREEE_TRY(ctx->alloc_t(nq, &d_qo));
		TREEE_TRY(ctx->alloc_t(nwaves, &d_qs));
		TREEE_TRY(morton_keys_dev(ctx, d_q, nq, t->sc->corner, t->sc->size, d_qk));
		TREEE_TRY(sort_pairs_dev(ctx, d_qk, nullptr, nq, 30, d_qks, d_qo));
		hipLaunchKernelGGL(k_query_start, dim3(div_up(nwaves, 256)), dim3(256), 0, ctx->stream, t->sc->d_keys, a.n, a.nb, d_qks, a.nq, nwaves, d_qs);
		TREEE_LAUNCH_CHECK();
		a.q_xyz = d_q;
		a.q_order = d_qo;
		a.q_start = d_qs;
	}
	if (lds > 64 * 1024)
		TREEE_HIP(hipFuncSetAttribute((const void *)k_knn_heap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(k_knn_heap, dim3(nwaves), dim3(64), lds, ctx->stream, a);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t build_bvh(treee_ctx *ctx, treee_knn *t) {
	const uint64_t n = t->sc->n;
	t->nb = div_up(n, BUCKET);
	t->P = 1;
	while (t->P < t->nb) t->P <<= 1;
	hipError_t e = hipMalloc((void **)&t->d_bvh, (size_t)t->P * 2 * 2 * sizeof(float4));
	if (e != hipSuccess) return set_err(TREEE_ERR_NOMEM, "hipMalloc bvh: %s", hipGetErrorString(e));
	hipLaunchKernelGGL(k_leaf_aabb, dim3(div_up(t->P, 4)), dim3(256), 0, ctx->stream, t->sc->d_sorted, (uint32_t)n, t->nb, t->P, t->d_bvh);
	for (uint32_t lvl = t->P >> 1; lvl >= 1; lvl >>= 1) // nodes [lvl, 2*lvl)
		hipLaunchKernelGGL(k_inner_aabb, dim3(div_up(lvl, 256)), dim3(256), 0, ctx->stream, t->d_bvh, lvl, lvl);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

} // namespace

extern "C" {

int32_t treee_knn_build_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, treee_knn **out) {
	if (!ctx || !out) return set_err(TREEE_ERR_INVALID, "treee_knn_build_dev: null argument");
	*out = nullptr;
	if (n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	if (!d_xyz) return set_err(TREEE_ERR_INVALID, "treee_knn_build_dev: xyz is null");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_knn *t = new treee_knn;
	t->device = ctx->device;
	int32_t rc = build_sorted_cloud(ctx, d_xyz, n, nullptr, 0.0f, t->sc.get());
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	rc = build_bvh(ctx, t);
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}

int32_t treee_knn_from_octree_dev(treee_ctx *ctx, const treee_octree *o, treee_knn **out) {
	if (!ctx || !o || !out) return set_err(TREEE_ERR_INVALID, "treee_knn_from_octree_dev: null argument");
	*out = nullptr;
	std::shared_ptr<SortedCloud> sc = octree_sorted_cloud(o);
	if (sc->n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	TREEE_HIP(hipSetDevice(ctx->device));
	treee_knn *t = new treee_knn;
	t->device = ctx->device;
	t->sc = sc; // one Morton sort, two consumers
	int32_t rc = build_bvh(ctx, t);
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}

int32_t treee_knn_build(const float *xyz, uint64_t n, treee_knn **out) {
	if (!out) return set_err(TREEE_ERR_INVALID, "treee_knn_build: out is null");
	*out = nullptr;
	if (n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	if (!xyz) return set_err(TREEE_ERR_INVALID, "treee_knn_build: xyz is null");
	treee_ctx *ctx = nullptr;
	TREEE_TRY(default_ctx(&ctx));
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_xyz;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
	TREEE_TRY(treee_knn_build_dev(ctx, d_xyz.p, n, out));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_knn_free(treee_knn *t) {
	if (!t) return TREEE_OK;
	(void)hipSetDevice(t->device);
	(void)hipDeviceSynchronize();
	t->sc.reset();
	if (t->d_bvh) (void)hipFree(t->d_bvh);
	delete t;
	return TREEE_OK;
}

int32_t treee_knn_len(const treee_knn *t, uint64_t *n_out) {
	if (!t || !n_out) return set_err(TREEE_ERR_INVALID, "treee_knn_len: null argument");
	*n_out = t->sc->n;
	return TREEE_OK;
}

int32_t treee_knn_query_batch_dev(treee_ctx *ctx, const treee_knn *t, const float *d_q, uint64_t nq, uint32_t k,
                                  float max_dist_sq, float *d_dist, uint32_t *d_index, uint32_t *d_counts) {
	if (!ctx || !t || (nq && !d_q)) return set_err(TREEE_ERR_INVALID, "treee_knn_query_batch_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return launch_knn(ctx, t, d_q, nq, k, max_dist_sq, d_dist, d_index, d_counts, nullptr);
}

int32_t treee_knn_self_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq, float *d_dist,
                           uint32_t *d_index, uint32_t *d_counts) {
	if (!ctx || !t) return set_err(TREEE_ERR_INVALID, "treee_knn_self_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return launch_knn(ctx, t, nullptr, t->sc->n, k, max_dist_sq, d_dist, d_index, d_counts, nullptr);
}

int32_t treee_curvature_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq, float *d_curve) {
	if (!ctx || !t || !d_curve) return set_err(TREEE_ERR_INVALID, "treee_curvature_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return launch_knn(ctx, t, nullptr, t->sc->n, k, max_dist_sq, nullptr, nullptr, nullptr, d_curve);
}

int32_t treee_map_to_u32_dev(treee_ctx *ctx, const float *d_in, uint64_t n, uint32_t *d_out) {
	if (!ctx || (n && (!d_in || !d_out))) return set_err(TREEE_ERR_INVALID, "treee_map_to_u32_dev: null argument");
	if (n == 0) return TREEE_OK;
	TREEE_HIP(hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_map_to_u32, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_in, (uint32_t)n, d_out);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_knn_query_batch(const treee_knn *t, const float *q, uint64_t nq, uint32_t k, float max_dist_sq,
                              treee_entry *out, uint32_t *counts) {
	if (!t || (nq && !q)) return set_err(TREEE_ERR_INVALID, "treee_knn_query_batch: null argument");
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0 (the reference panics at best_set.rs:74)");
	if (nq == 0) return TREEE_OK;
	treee_ctx *ctx = nullptr;
	TREEE_TRY(default_ctx(&ctx));
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_q, d_dist;
	DevBuf<uint32_t> d_idx, d_cnt;
	DevBuf<treee_entry> d_ent;
	const uint64_t total = nq * k;
	TREEE_TRY(d_q.alloc(nq * 3));
	TREEE_TRY(d_dist.alloc(total));
	TREEE_TRY(d_idx.alloc(total));
	TREEE_TRY(d_cnt.alloc(nq));
	TREEE_HIP(hipMemcpyAsync(d_q.p, q, nq * 12, hipMemcpyHostToDevice, ctx->stream));
	TREEE_TRY(treee_knn_query_batch_dev(ctx, t, d_q.p, nq, k, max_dist_sq, d_dist.p, d_idx.p, d_cnt.p));
	if (out) {
		TREEE_TRY(d_ent.alloc(total));
		hipLaunchKernelGGL(k_widen_entries, dim3(div_up(total, 256)), dim3(256), 0, ctx->stream, d_dist.p, d_idx.p, total, d_ent.p);
		TREEE_LAUNCH_CHECK();
		TREEE_HIP(hipMemcpyAsync(out, d_ent.p, total * sizeof(treee_entry), hipMemcpyDeviceToHost, ctx->stream));
	}
	if (counts) TREEE_HIP(hipMemcpyAsync(counts, d_cnt.p, nq * 4, hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_knn_empty(const treee_knn *t, const float *q, uint64_t nq, float max_dist_sq, uint8_t *empty_out) {
	if (!t || (nq && (!q || !empty_out))) return set_err(TREEE_ERR_INVALID, "treee_knn_empty: null argument");
	if (nq == 0) return TREEE_OK;
	// EmptySet (best_set.rs:111-131): empty <=> no point with distance < max  <=> k=1 query finds nothing
	std::vector<uint32_t> counts(nq);
	TREEE_TRY(treee_knn_query_batch(t, q, nq, 1, max_dist_sq, nullptr, counts.data()));
	for (uint64_t i = 0; i < nq; ++i) empty_out[i] = counts[i] == 0 ? 1 : 0;
	return TREEE_OK;
}

int32_t treee_curvature(const float *xyz, uint64_t n, uint32_t k, float max_dist_sq, float *curve_out) {
	if (n == 0) return TREEE_OK;
	if (!xyz || !curve_out) return set_err(TREEE_ERR_INVALID, "treee_curvature: null argument");
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0");
	treee_ctx *ctx = nullptr;
	TREEE_TRY(default_ctx(&ctx));
	treee_knn *t = nullptr;
	TREEE_TRY(treee_knn_build(xyz, n, &t));
	DevBuf<float> d_curve;
	int32_t rc = d_curve.alloc(n);
	if (rc == TREEE_OK) rc = treee_curvature_dev(ctx, t, k, max_dist_sq, d_curve.p);
	if (rc == TREEE_OK) {
		hipError_t e = hipMemcpyAsync(curve_out, d_curve.p, n * 4, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) rc = set_err(TREEE_ERR_HIP, "curvature copy back: %s", hipGetErrorString(e));
	}
	treee_knn_free(t);
	return rc;
}

} // extern "C"

// ---- radius query: KDTree::nearest (kd_tree.rs:86-90) + DynamicSet (best_set.rs:84-109) ---------
namespace {

// one wave per bucket: skip by box distance, then append every point with d < max
__global__ __launch_bounds__(256) void k_radius_collect(const float4 *__restrict__ pts, const float4 *__restrict__ bvh, uint32_t n,
                                                         uint32_t nb, uint32_t P, float qx, float qy, float qz, uint32_t maxbits,
                                                         uint32_t *__restrict__ counter, uint32_t cap,
                                                         uint32_t *__restrict__ out_dbits, uint32_t *__restrict__ out_pos) {
	const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (b >= nb) return;
	const int lane = threadIdx.x & 63;
	const float4 lo = bvh[2ull * (P + b) + 0], hi = bvh[2ull * (P + b) + 1];
	const float bx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.0f);
	const float by = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.0f);
	const float bz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.0f);
	const float bd = ((0.0f + bx * bx) + by * by) + bz * bz;
	if (!(__float_as_uint(bd) < maxbits)) return;
	const uint32_t i = b * BUCKET + lane;
	if (i >= n) return;
	const float4 c = pts[i];
	const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
	const float d = ((0.0f + dx * dx) + dy * dy) + dz * dz;
	if (__float_as_uint(d) < maxbits) {
		const uint32_t o = atomicAdd(counter, 1u);
		if (o < cap) {
			out_dbits[o] = __float_as_uint(d);
			out_pos[o] = i;
		}
	}
}

__global__ __launch_bounds__(256) void k_radius_emit(const uint32_t *__restrict__ dbits, const uint32_t *__restrict__ pos,
                                                      const float4 *__restrict__ pts, uint32_t m, treee_entry *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= m) return;
	treee_entry e;
	e.distance = __uint_as_float(dbits[i]);
	e._pad = 0;
	e.index = (uint64_t)__float_as_uint(pts[pos[i]].w);
	out[i] = e;
}

} // namespace

extern "C" int32_t treee_knn_radius(const treee_knn *t, const float q[3], float max_dist_sq, treee_entry *out, uint64_t cap,
                                    uint64_t *found_out) {
	if (!t || !q || !found_out || (cap && !out)) return set_err(TREEE_ERR_INVALID, "treee_knn_radius: null argument");
	treee_ctx *ctx = nullptr;
	TREEE_TRY(default_ctx(&ctx));
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	const uint32_t n = (uint32_t)t->sc->n;
	uint32_t *d_counter = nullptr, *d_db = nullptr, *d_pos = nullptr;
	TREEE_TRY(ctx->alloc_t(1, &d_counter));
	TREEE_TRY(ctx->alloc_t(n, &d_db));
	TREEE_TRY(ctx->alloc_t(n, &d_pos));
	TREEE_HIP(hipMemsetAsync(d_counter, 0, 4, ctx->stream));
	hipLaunchKernelGGL(k_radius_collect, dim3(div_up(t->nb, 4)), dim3(256), 0, ctx->stream, t->sc->d_sorted, t->d_bvh, n, t->nb, t->P,
	                   q[0], q[1], q[2], max_bits(max_dist_sq), d_counter, n, d_db, d_pos);
	TREEE_LAUNCH_CHECK();
	uint32_t m = 0;
	TREEE_HIP(hipMemcpyAsync(&m, d_counter, 4, hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	*found_out = m;
	if (m == 0 || cap == 0) return TREEE_OK;
	// deterministic order: by sorted position first, then a stable sort by distance bits
	uint32_t *d_pos_s = nullptr, *d_db_s = nullptr, *d_db2 = nullptr, *d_pos2 = nullptr;
	TREEE_TRY(ctx->alloc_t(m, &d_pos_s));
	TREEE_TRY(ctx->alloc_t(m, &d_db_s));
	TREEE_TRY(ctx->alloc_t(m, &d_db2));
	TREEE_TRY(ctx->alloc_t(m, &d_pos2));
	TREEE_TRY(sort_pairs_dev(ctx, d_pos, d_db, m, 32, d_pos_s, d_db_s));
	TREEE_TRY(sort_pairs_dev(ctx, d_db_s, d_pos_s, m, 32, d_db2, d_pos2));
	const uint32_t w = (uint32_t)(cap < m ? cap : m);
	treee_entry *d_ent = nullptr;
	TREEE_TRY(ctx->alloc_t(w, &d_ent));
	hipLaunchKernelGGL(k_radius_emit, dim3(div_up(w, 256)), dim3(256), 0, ctx->stream, d_db2, d_pos2, t->sc->d_sorted, w, d_ent);
	TREEE_LAUNCH_CHECK();
	TREEE_HIP(hipMemcpyAsync(out, d_ent, (size_t)w * sizeof(treee_entry), hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}
