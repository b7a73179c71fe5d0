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
// distances the smaller (cell key, original index) wins; the cell key is the Hilbert index of the
// Morton cell (hilbert30, common.h), the order the structure's buckets are cut in.  Distances and counts are
// bit-identical to the reference; indices are identical whenever distances are distinct.
#include "common.h"
#include "../../include/treee_f32_trig.h"

#include <cstdlib>
#include <type_traits>

using namespace treee;

struct treee_knn {
	std::shared_ptr<SortedCloud> sc = std::make_shared<SortedCloud>();
	uint32_t nb = 0;       // buckets of 64 points
	uint32_t P = 1;        // leaves padded to a power of two; node i: children 2i, 2i+1; leaf b = P + b
	float4 *d_bvh = nullptr; // [2P][2] : (min.xyz, min segment), (max.xyz, max segment); segments are 0 unless `segmented`
	float4 *d_sub = nullptr; // [nb][4][2]: (min.xyz, -), (max.xyz, -) of every quarter (16 consecutive records) of a bucket
	bool segmented = false;  // buckets never straddle a segment; a query only sees its own segment (build_segmented)
	uint64_t n_points = 0;   // points the handle was built from (sc->n counts the padding of a segmented handle too)
	uint32_t *d_order = nullptr; // workgroup -> bucket of the self-query kernel (build_bvh), ~0u = padding
	uint32_t n_order = 0;        // its length (8 * the longest XCD segment)
	int device = 0;
	treee_ctx *owner = nullptr;
	mutable HandleSync sync; // uses of d_bvh / d_order from other streams (the sorted cloud carries its own)
};

namespace {

constexpr int BUCKET = 64;

// ---- bucket AABBs and the implicit tree ---------------------------------------------------------
// 16 lanes per bucket, 4 consecutive records per lane (64 contiguous bytes): a wave covers 4 buckets = 4 KiB, the
// per-lane fold needs no cross-lane traffic and the butterfly has 4 steps instead of 6.
// `up` (P >= 16 only: a workgroup's 16 leaves are then 16 whole siblings): the four tree levels above the workgroup's leaves too --
// two inside the wave by shuffles, two through LDS -- instead of four launches of k_inner_aabb (round 5: the launches, not the
// bytes, were their cost).  `class_key` (or null): the dispatch table's sort key of the bucket (bucket_class_key).
#ifndef TREEE_ORDER_RANGE
#define TREEE_ORDER_RANGE 128
#endif
constexpr uint32_t ORDER_RANGE = TREEE_ORDER_RANGE;
constexpr uint32_t LEAF_UP = 4; // levels k_leaf_aabb builds above the leaves
// five bits of size class: the exponent byte of the squared diagonal, clamped to [111, 142] (diagonals from 4 mm to 180 m; what
// lies outside shares the end classes) -- with the three label bits an 8-bit key, ONE radix pass over the buckets (round 5;
// the 11-bit key of rounds 1-4 took two).  Only the dispatch order depends on it, never a result.
__device__ __forceinline__ uint32_t bucket_class_key(uint32_t b, const float (&lo)[3], const float (&hi)[3]) {
	const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	const float d2 = dx * dx + dy * dy + dz * dz;
	const uint32_t e = (d2 == d2 && d2 > 0.0f) ? ((__float_as_uint(d2) >> 23) & 255u) : 0u;
	const uint32_t c = e < 111u ? 0u : (e > 142u ? 31u : e - 111u);
	return (((b / ORDER_RANGE) & 7u) << 5) | (31u - c);
}
__global__ __launch_bounds__(256) void k_leaf_aabb(const float4 *__restrict__ pts, uint32_t n, uint32_t nb, uint32_t P,
                                                    const uint32_t *__restrict__ bucket_seg /* or null */, float4 *__restrict__ bvh,
                                                    float4 *__restrict__ sub, bool up, uint32_t *__restrict__ class_key /* or null */) {
	__shared__ float s_up[4][8];
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	const uint32_t b = t >> 4; // bucket
	if (b >= P) return;        // whole groups of 16 lanes leave together (the shuffles below stay inside a group; with `up` whole workgroups)
	const uint32_t i0 = b * BUCKET + (t & 15u) * 4u;
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
	if (b < nb) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (i0 + j < n) {
				const float4 p = pts[i0 + j];
				const float c[3] = {p.x, p.y, p.z};
				// NaN coordinates never produce a valid distance; keep them out of the boxes (fminf/fmaxf drop a NaN operand)
#pragma unroll
				for (int a = 0; a < 3; ++a) {
					mn[a] = fminf(mn[a], c[a]);
					mx[a] = fmaxf(mx[a], c[a]);
				}
			}
		}
	}
	// four lanes = one quarter of the bucket (16 consecutive records): its box first (the scans skip a quarter no query needs),
	// then the bucket's (min / max are exact: the order of the folds does not matter)
#pragma unroll
	for (int a = 0; a < 3; ++a)
#pragma unroll
		for (int o = 1; o <= 2; o <<= 1) {
			mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
			mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
		}
	if ((t & 3u) == 0 && b < nb) {
		sub[8ull * b + 2u * ((t & 15u) >> 2) + 0] = make_float4(mn[0], mn[1], mn[2], 0.0f);
		sub[8ull * b + 2u * ((t & 15u) >> 2) + 1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
	}
#pragma unroll
	for (int a = 0; a < 3; ++a)
#pragma unroll
		for (int o = 4; o <= 8; o <<= 1) {
			mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
			mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
		}
	// .w = segment range of the node (uint bits): [s, s] for a bucket of segment s, the empty range [~0, 0]
	// for an unused leaf, [0, 0] everywhere when the cloud is not segmented
	uint32_t slo = 0u, shi = 0u;
	if (bucket_seg) {
		const uint32_t sg = b < nb ? bucket_seg[b] : ~0u;
		slo = sg;
		shi = sg == ~0u ? 0u : sg;
	}
	if ((t & 15u) == 0) {
		bvh[2ull * (P + b) + 0] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(slo));
		bvh[2ull * (P + b) + 1] = make_float4(mx[0], mx[1], mx[2], __uint_as_float(shi));
		if (class_key && b < nb) class_key[b] = bucket_class_key(b, mn, mx);
	}
	if (!up) return; // (uniform)
	// every lane of a 16-lane group holds its bucket's box: node (P + b) >> l is the fold over the 2^l groups that share it
	// (k_inner_aabb's arithmetic: fminf / fmaxf of the boxes, min / max of the segment range)
	uint32_t node = P + b;
#pragma unroll
	for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
			mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
		}
		slo = min(slo, (uint32_t)__shfl_xor((int)slo, o, 64));
		shi = max(shi, (uint32_t)__shfl_xor((int)shi, o, 64));
		node >>= 1;
		if ((t & (2u * o - 1u)) == 0) {
			bvh[2ull * node + 0] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(slo));
			bvh[2ull * node + 1] = make_float4(mx[0], mx[1], mx[2], __uint_as_float(shi));
		}
	}
	const uint32_t w = threadIdx.x >> 6;
	if ((threadIdx.x & 63u) == 0) {
		s_up[w][0] = mn[0], s_up[w][1] = mn[1], s_up[w][2] = mn[2], s_up[w][3] = __uint_as_float(slo);
		s_up[w][4] = mx[0], s_up[w][5] = mx[1], s_up[w][6] = mx[2], s_up[w][7] = __uint_as_float(shi);
	}
	__syncthreads();
	if (threadIdx.x < 2) { // the two nodes over the waves (0, 1) and (2, 3); thread 0 then the one over both
		const uint32_t x = 2u * threadIdx.x, y = x + 1u;
		float lo3[3], hi3[3];
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			lo3[a] = fminf(s_up[x][a], s_up[y][a]);
			hi3[a] = fmaxf(s_up[x][4 + a], s_up[y][4 + a]);
		}
		uint32_t l = min(__float_as_uint(s_up[x][3]), __float_as_uint(s_up[y][3])), h = max(__float_as_uint(s_up[x][7]), __float_as_uint(s_up[y][7]));
		const uint32_t n3 = ((P + (blockIdx.x * 16u)) >> 3) + threadIdx.x;
		bvh[2ull * n3 + 0] = make_float4(lo3[0], lo3[1], lo3[2], __uint_as_float(l));
		bvh[2ull * n3 + 1] = make_float4(hi3[0], hi3[1], hi3[2], __uint_as_float(h));
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], 1, 64));
			hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], 1, 64));
		}
		l = min(l, (uint32_t)__shfl_xor((int)l, 1, 64));
		h = max(h, (uint32_t)__shfl_xor((int)h, 1, 64));
		if (threadIdx.x == 0) {
			const uint32_t n4 = (P + (blockIdx.x * 16u)) >> 4;
			bvh[2ull * n4 + 0] = make_float4(lo3[0], lo3[1], lo3[2], __uint_as_float(l));
			bvh[2ull * n4 + 1] = make_float4(hi3[0], hi3[1], hi3[2], __uint_as_float(h));
		}
	}
}

__global__ __launch_bounds__(256) void k_inner_aabb(float4 *__restrict__ bvh, uint32_t first, uint32_t count) {
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= count) return;
	const uint32_t i = first + t;
	const float4 a0 = bvh[4ull * i + 0], a1 = bvh[4ull * i + 1], b0 = bvh[4ull * i + 2], b1 = bvh[4ull * i + 3];
	const uint32_t slo = min(__float_as_uint(a0.w), __float_as_uint(b0.w)), shi = max(__float_as_uint(a1.w), __float_as_uint(b1.w));
	bvh[2ull * i + 0] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), __uint_as_float(slo));
	bvh[2ull * i + 1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), __uint_as_float(shi));
}

// the top of the tree (levels of at most INNER_TOP nodes) in one workgroup: a level per barrier instead of a launch
constexpr uint32_t INNER_TOP = 1024;
__global__ __launch_bounds__(INNER_TOP) void k_inner_aabb_top(float4 *__restrict__ bvh, uint32_t first_level) {
	for (uint32_t lvl = first_level; lvl >= 1; lvl >>= 1) {
		if (threadIdx.x < lvl) {
			const uint32_t i = lvl + threadIdx.x;
			const float4 a0 = bvh[4ull * i + 0], a1 = bvh[4ull * i + 1], b0 = bvh[4ull * i + 2], b1 = bvh[4ull * i + 3];
			const uint32_t slo = min(__float_as_uint(a0.w), __float_as_uint(b0.w)), shi = max(__float_as_uint(a1.w), __float_as_uint(b1.w));
			bvh[2ull * i + 0] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), __uint_as_float(slo));
			bvh[2ull * i + 1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), __uint_as_float(shi));
		}
		__threadfence_block();
		__syncthreads(); // the next level reads what this one wrote (same workgroup: write-through L1, never read before)
	}
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
		phi = treee_acos_f32(r) / 3.0f; // (the shared last-bit definition of f32::acos / cos: include/treee_f32_trig.h)
	const float eig1 = q + 2.0f * p * treee_cos_f32(phi);
	const float eig3 = q + 2.0f * p * treee_cos_f32(phi + (2.0f * PI / 3.0f));
	const float eig2 = 3.0f * q - eig1 - eig3;
	out[0] = eig1;
	out[1] = eig2;
	out[2] = eig3;
}

struct KnnArgs {
	const float4 *pts; // sorted records (x,y,z,bitcast(orig))
	const float4 *bvh;
	const float4 *sub; // [nb][4][2] quarter boxes of the buckets
	uint32_t n, nb, P;
	// queries: self mode (q_xyz == nullptr): query j = pts[j], output row = orig index
	const float *q_xyz;          // AoS, original order
	const uint32_t *q_order;     // sorted query slot -> original query index
	const uint32_t *q_start;     // per wave: start bucket
	const uint32_t *wave_order;  // self mode: bucket handled by workgroup i (largest boxes first), or null
	uint32_t nq;
	uint32_t q_limit; // self mode: only points with original index < q_limit are queried (the rest are neighbours only)
	uint32_t k, maxbits;
	float *out_dist;
	uint32_t *out_index;
	uint32_t *out_count;
	float *out_curve;
	uint8_t *out_flags; // optional, per query row: bit 0 = two equal distances inside the k, bit 1 = the k-th and the (k+1)-th equally far (needs k < list length)
	uint32_t *stats; // optional [4]: buckets scanned, flushes, nodes tested, waves
	uint32_t *thr_io;  // diagnostic: per-slot final threshold (written when thr_seed == 0, read when 1)
	uint32_t thr_seed;
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
	bool live = slot < a.nq;
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
			live = row < a.q_limit;
		}
		start = blockIdx.x;
		if (!__any(live)) return; // before any barrier: the whole single-wave workgroup leaves
	}
	for (int s = 0; s < h.k; ++s) { // FixedSet::new (best_set.rs:20-28)
		h.D(s) = a.maxbits;
		h.Pp(s) = 0u;
	}
	uint32_t wd = a.maxbits, wp = 0u; // root = current worst

	auto scan_bucket = [&](uint32_t b) __attribute__((always_inline)) {
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

	auto box_hit = [&](uint32_t node) __attribute__((always_inline)) -> bool {
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

// ---- k <= 32: per-lane sorted top-32 in VGPRs, candidate buffer in LDS, bitonic networks ----------
// The LDS heap above serialises on dependent ds_read chains (one sift-down = ~10 LDS round
// trips).  Here each lane keeps its 32 best keys (distance bits << 32 | sorted position) sorted
// in 64 VGPRs.  Candidates that beat the lane's current 32nd key are only appended to a per-lane
// LDS buffer ([slot][lane], conflict-free); when any lane's buffer is half full the whole wave
// sorts its buffers with a 32-input bitonic network, takes min(L[i], B[31-i]) -- the 32 smallest
// of the union, a bitonic sequence -- and re-sorts with a 5-stage bitonic merge.  No data-
// dependent loop, no divergence, no LDS latency chain; ~350 compare-exchanges per flush.
constexpr int SK = 32;          // list size (k <= 32 uses the first k)
#ifndef TREEE_SBUF
#define TREEE_SBUF 16
#endif
#ifndef TREEE_KNN_WAVES
#define TREEE_KNN_WAVES 3
#endif
constexpr int SBUF = TREEE_SBUF;   // buffer slots per lane (power of two <= SK)
#ifndef TREEE_SCHUNK
#define TREEE_SCHUNK (TREEE_SBUF / 4)
#endif
constexpr int SCHUNK = TREEE_SCHUNK; // candidates scanned between flush checks (flush when < SCHUNK slots are free)
#ifndef TREEE_TRANSPOSE_BELOW
#define TREEE_TRANSPOSE_BELOW 12 // (20 until the lane-parallel scans began to skip quarters, round 4: 6 / 8 / 10 / 12 / 16 / 20 / 24 needing lanes: 7.91 / 7.82 / 7.73 / 7.72 / 7.73 / 7.83 / 7.95 ms)
#endif
constexpr int TRANSPOSE_BELOW = TREEE_TRANSPOSE_BELOW; // buckets needed by fewer queries are scanned transposed
// Keys are distance bits << 32 | sorted position.  Distances are >= +0 and at most +inf (0x7F800000), so read as
// an IEEE double every key is a non-negative, non-NaN number (possibly zero or denormal), and for such doubles the
// numeric order equals the unsigned order of the bit patterns.  A compare-exchange is therefore v_min_f64 +
// v_max_f64 (2 VALU instead of a 64-bit compare + 4 selects); min/max return one operand unchanged, so bits
// survive -- PROVIDED f64 denormals are not flushed: the code object keeps hipcc's default
// FLOAT_DENORM_MODE_16_64 = 3 (only -fgpu-flush-denormals-to-zero for f32 exists as a switch), and
// tests/test_gpu_parity.py::test_knn_denormal_distances pins it.  (An earlier version added 0x00100000 to the high
// word to make every key a normal number: 1 VALU per key built or read, 4.5 % of the kernel.)
#ifndef TREEE_KEY_BIAS
#define TREEE_KEY_BIAS 0u
#endif
constexpr uint32_t KEY_BIAS = TREEE_KEY_BIAS;
constexpr unsigned long long KEY_MAX = 0x7FEFFFFFFFFFFFFFull; // largest finite double > every key

__device__ __forceinline__ double key_make(uint32_t db, uint32_t pos) {
	return __longlong_as_double((long long)(((unsigned long long)(db + KEY_BIAS) << 32) | (unsigned long long)pos));
}
__device__ __forceinline__ uint32_t key_db(double k) { return (uint32_t)((unsigned long long)__double_as_longlong(k) >> 32) - KEY_BIAS; }
__device__ __forceinline__ uint32_t key_pos(double k) { return (uint32_t)(unsigned long long)__double_as_longlong(k); }

__device__ __forceinline__ void cmpx(double &a, double &b) { // a <= b afterwards
	double lo, hi;
	asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b));
	a = lo;
	b = hi;
}
__device__ __forceinline__ double key_min(double a, double b) {
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

template <int N> __device__ __forceinline__ void bitonic_sort(double (&v)[N]) {
#pragma unroll
	for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
			for (int i = 0; i < N; ++i) {
				const int l = i ^ j;
				if (l > i) {
					if ((i & k) == 0)
						cmpx(v[i], v[l]);
					else
						cmpx(v[l], v[i]);
				}
			}
		}
	}
}

// Sorting networks with fewer comparators than the bitonic sorter (the kernel is VALU-issue bound and a
// compare-exchange is 2 VALU).  16 inputs: the 60-comparator, 10-layer network (bitonic: 80); verified with the 0-1
// principle over all 2^16 inputs.  32 inputs: Batcher's odd-even merge sort, 191 comparators (bitonic: 240).
__device__ __forceinline__ void sort16_net(double (&v)[16]) {
#define CX(a, b) cmpx(v[a], v[b])
	CX(0, 13); CX(1, 12); CX(2, 15); CX(3, 14); CX(4, 8); CX(5, 6); CX(7, 11); CX(9, 10);
	CX(0, 5); CX(1, 7); CX(2, 9); CX(3, 4); CX(6, 13); CX(8, 14); CX(10, 15); CX(11, 12);
	CX(0, 1); CX(2, 3); CX(4, 5); CX(6, 8); CX(7, 9); CX(10, 11); CX(12, 13); CX(14, 15);
	CX(0, 2); CX(1, 3); CX(4, 10); CX(5, 11); CX(6, 7); CX(8, 9); CX(12, 14); CX(13, 15);
	CX(1, 2); CX(3, 12); CX(4, 6); CX(5, 7); CX(8, 10); CX(9, 11); CX(13, 14);
	CX(1, 4); CX(2, 6); CX(5, 8); CX(7, 10); CX(9, 13); CX(11, 14);
	CX(2, 4); CX(3, 6); CX(9, 12); CX(11, 13);
	CX(3, 5); CX(6, 8); CX(7, 9); CX(10, 12);
	CX(3, 4); CX(5, 6); CX(7, 8); CX(9, 10); CX(11, 12);
	CX(6, 7); CX(8, 9);
#undef CX
}
template <int N> __device__ __forceinline__ void odd_even_merge_sort(double (&v)[N]) {
#pragma unroll
	for (int p = 1; p < N; p <<= 1)
#pragma unroll
		for (int k = p; k >= 1; k >>= 1)
#pragma unroll
			for (int j = k % p; j <= N - 1 - k; j += 2 * k)
#pragma unroll
				for (int i = 0; i < k; ++i)
					if (i + j + k < N && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cmpx(v[i + j], v[i + j + k]);
}
template <int N> __device__ __forceinline__ void sort_net(double (&v)[N]) {
	if constexpr (N == 16)
		sort16_net(v);
	else if constexpr (N == 32)
		odd_even_merge_sort<32>(v);
	else
		bitonic_sort<N>(v);
}

template <int N> __device__ __forceinline__ void bitonic_merge(double (&v)[N]) { // v bitonic -> ascending
#pragma unroll
	for (int j = N >> 1; j > 0; j >>= 1) {
#pragma unroll
		for (int i = 0; i < N; ++i) {
			const int l = i ^ j;
			if (l > i) cmpx(v[i], v[l]);
		}
	}
}

// v <- min or max of (v, the value lane ^ S holds), S < 64: a VALU exchange (DPP within a row of 16 lanes, the gfx950 row /
// half swaps beyond) -- through ds_bpermute the same network was bound by the LDS pipe (428 us per 15 M points)
template <uint32_t S> __device__ __forceinline__ uint32_t lane_xor_minmax(uint32_t v, bool keep_min) {
	if constexpr (S == 16u || S == 32u) {
		uint32_t x, y;
		if constexpr (S == 16u) {
			const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); // (rows 0 0 2 2 | rows 1 1 3 3)
			x = r[0], y = r[1];
		} else {
			const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); // (lower half twice | upper half twice)
			x = r[0], y = r[1];
		}
		return keep_min ? min(x, y) : max(x, y);
	} else {
		uint32_t p;
		if constexpr (S == 1u)
			p = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false); // quad_perm [1, 0, 3, 2]
		else if constexpr (S == 2u)
			p = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false); // quad_perm [2, 3, 0, 1]
		else if constexpr (S == 4u) {
			p = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false); // row_shl:4 into the quads 0 and 2 of a row
			p = (uint32_t)__builtin_amdgcn_update_dpp((int)p, (int)v, 0x114, 0xF, 0xA, false); // row_shr:4 into the quads 1 and 3
		} else {
			static_assert(S == 8u, "lane distance");
			p = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, false); // row_ror:8
		}
		return keep_min ? min(v, p) : max(v, p);
	}
}

// ascending bitonic sort of one u32 per lane across the 64 lanes of a wave (21 exchange stages, no LDS): the listed buckets of a
// group, nearest first (knn_wave)
template <uint32_t K, uint32_t J> __device__ __forceinline__ uint32_t wave_sort_stages(uint32_t v, const uint32_t lane) {
	const bool keep_min = ((lane & J) == 0u) == ((lane & K) == 0u);
	v = lane_xor_minmax<J>(v, keep_min);
	if constexpr (J > 1u)
		return wave_sort_stages<K, J / 2u>(v, lane);
	else if constexpr (K < 64u)
		return wave_sort_stages<2u * K, K>(v, lane);
	else
		return v;
}
// min / max over the 16 lanes of a DPP row, left in EVERY lane of the row: four rotate-and-combine steps with the rotation as a
// modifier of the VALU instruction itself (round 5; __shfl_xor went through ds_bpermute and every fminf / fmaxf around it was
// three instructions: 28 LDS-pipe and ~130 VALU instructions per wave for the four query sub-boxes).  s_nop 1: a DPP operand read
// needs two wait states behind the VALU write of the same register, and the hazard recogniser does not look inside inline asm.
#define TREEE_ROW16(op)                                                                                                                          \
	asm volatile("s_nop 1\n\t" op " %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" op " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t" \
	             "s_nop 1\n\t" op " %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" op " %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"        \
	             : "+v"(v))
__device__ __forceinline__ float row16_min_f32(float v) {
	TREEE_ROW16("v_min_f32_dpp");
	return v;
}
__device__ __forceinline__ float row16_max_f32(float v) {
	TREEE_ROW16("v_max_f32_dpp");
	return v;
}
__device__ __forceinline__ uint32_t row16_max_u32(uint32_t v) {
	TREEE_ROW16("v_max_u32_dpp");
	return v;
}
#undef TREEE_ROW16
// the LDS byte address of an object in shared memory, for hand-written ds_ instructions
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p; }
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// a - b on both halves in ONE instruction.  Written by hand because hipcc packs only one of the scan loop's two candidate
// pairs by itself: for the other it emitted six v_sub_f32 against six further copies of the query (three instructions of a
// trip's 38 and six VGPRs; round 5).  Same IEEE subtraction either way.
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) {
	v2f r;
	asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
// ... with ONE element of a register pair on either side (op_sel / op_sel_hi pick the half a result lane reads): a scalar against
// two values without first copying the scalar into both halves of a pair.  E = 0 / 1: the element of the pair.
template <int E> __device__ __forceinline__ v2f pk_sub_elem_minus(v2f a_pair, v2f b) { // (a[E] - b.x, a[E] - b.y)
	v2f r;
	if constexpr (E == 0)
		asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a_pair), "v"(b));
	else
		asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a_pair), "v"(b));
	return r;
}
template <int E> __device__ __forceinline__ v2f pk_sub_minus_elem(v2f a, v2f b_pair) { // (a.x - b[E], a.y - b[E])
	v2f r;
	if constexpr (E == 0)
		asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b_pair));
	else
		asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b_pair));
	return r;
}

#ifndef TREEE_SCAN_PIPE
#define TREEE_SCAN_PIPE 1
#endif
#ifndef TREEE_NBR
#define TREEE_NBR 4 // Morton neighbours of the own bucket scanned before the tree walk (start-1, start+1, start-2, start+2); measured 0: 16.8 ms, 1: 12.8, 2: 11.6, 4: 11.5
#endif
#ifndef TREEE_NG
#define TREEE_NG 4
#endif
#ifndef TREEE_GROUP
#define TREEE_GROUP 52 // listed buckets whose boxes are in LDS at a time: 52 (+ the 128 bytes of quarter boxes) keeps the workgroup's LDS at 12 800 bytes (below)
#endif
#ifndef TREEE_LIST_CAP
#define TREEE_LIST_CAP 512
#endif
constexpr int LIST_CAP = TREEE_LIST_CAP; // candidate buckets per wave; more -> packet-traversal fallback
#ifndef TREEE_FRONT_CAP
#define TREEE_FRONT_CAP (TREEE_SBUF * 64)
#endif
constexpr int FRONT_CAP = TREEE_FRONT_CAP; // BFS frontier; the default aliases the candidate buffer, which is empty then

// TREEE_AB_DOUBLE (profiles/ab_double.sh; default 0 = the shipped kernel): a bit mask of phases whose work is issued TWICE with the
// same results -- what a phase costs the kernel is read from what doubling it adds (profiles/README.md, "Round 5").
//   1 scan: distance arithmetic    2 scan: append's LDS writes    4 scan: broadcast LDS reads    8 flush: sort + merge networks
//   16 transposed scan: distance + ballot    32 per-lane box tests (find_next, quarters)    64 BFS node tests
#ifndef TREEE_AB_DOUBLE
#define TREEE_AB_DOUBLE 0
#endif
// DIAG = true compiles the diagnostic counters in (TREEE_KNN_STATS); the shipped instantiation has none
#define ST(x)                  \
	do {                       \
		if constexpr (DIAG) {  \
			x;                 \
		}                      \
	} while (0)
// SEG: the cloud is segmented (build_segmented) -- records are padded with (NaN, NaN, NaN, ~0u) so that no bucket
// straddles a segment, tree nodes carry the range of segments below them, and a query only descends into / scans
// nodes and buckets of its own segment: the result is what one KD-tree per segment returns (calculations.rs:329).
struct KnnLds {
	float tile[3][BUCKET];                // SoA candidate tile (x[], y[], z[])
	uint32_t buf[2 * SBUF * 64];          // [pos | distance bits][slot][lane]: conflict-free per slot
	uint32_t s_list[LIST_CAP];            // candidate buckets of the wave
	float4 s_gbox[TREEE_GROUP][2];        // boxes of the current group of listed buckets
	float4 s_sub[4][2];                   // quarter boxes of the bucket about to be scanned by all lanes
};
__device__ __forceinline__ KnnLds &knn_lds() {
	__shared__ __attribute__((aligned(16))) KnnLds s;
	return s;
}
typedef const __attribute__((address_space(4))) struct KnnArgs *KnnArgsInKernarg; // the kernel's own argument block
template <bool SEG, int SKT> __device__ __attribute__((noinline)) void knn_wave_overflowed(KnnArgsInKernarg ka, uint32_t wave_id);

// OVF = false: the kernel proper.  A wave whose candidate list overflows (a huge search radius: 195 of 234 375 waves of the
// 15 M-point forest) starts over as OVF = true -- the same prologue, then a packet traversal of the tree instead of the list --
// through a REAL call (knn_wave_overflowed, noinline).  Inlined (rounds 1-3) the traversal was a third copy of the scan and
// merge code whose register assignment the allocator reconciled with the other two through scratch memory at every phase
// transition: 220 B / lane of spills, 3.5 GB of HBM writes per 15 M-point launch for 60 MB of output.  Behind a call its
// registers are its own affair: the kernel proper needs no scratch, and the call costs the 195 waves that make it a second
// prologue.  (As a second launch behind the kernel the overflowed waves -- ~4 ms each -- formed a tail of 7 ms: they have to
// start early, among the others, which the dispatch table arranges.)
template <bool DIAG, bool SEG, int SKT, bool OVF> __device__ __forceinline__ void knn_wave(const KnnArgs &a, const uint32_t wave_id) {
	constexpr int SK = SKT; // list length of this instantiation: 32 (k <= 32) or 16 (k <= 16: half the list registers and merge work, no spills)
	KnnLds &lds = knn_lds(); // one block of LDS whichever instantiation runs (OVF = true is called from OVF = false)
	auto &tile = lds.tile;
	auto &buf = lds.buf;
	auto &s_list = lds.s_list;
	constexpr uint32_t GROUP = TREEE_GROUP; // listed buckets whose boxes are in LDS at a time
	auto &s_gbox = lds.s_gbox;
	auto &s_sub = lds.s_sub;
	// LDS is handed out in units of 1 280 bytes (160 KB / 128): 12 workgroups per CU -- the 3 waves per SIMD the registers
	// allow -- fit only if each takes at most 10 units.  With 13 056 bytes (GROUP = 64) a CU held 11: measured 8.78 instead
	// of 8.27 ms (12 800 and 12 928 bytes differ by exactly that step).
	static_assert(SKT > 32 || TREEE_FRONT_CAP > TREEE_SBUF * 64 || sizeof(KnnLds) <= 10 * 1280,
	              "the workgroup's LDS must stay within 10 allocation units of 1 280 bytes (12 workgroups per CU)");
	const int lane = threadIdx.x;
	const uint64_t lt_mask = (1ull << lane) - 1ull;

	// self mode: workgroups take the buckets with the largest boxes first (they are the expensive ones:
	// sparse regions and buckets that straddle a jump of the Morton curve), so that no straggler starts
	// near the end of the kernel
	// and the workgroups that share an XCD (blockIdx % 8) walk the same Morton ranges, so its L2 serves the
	// bucket re-reads of neighbouring waves (build_bvh lays the table out that way; ~0u = padding)
	if (wave_id == ~0u) return;
	const uint32_t slot = wave_id * 64u + lane;
	bool live = slot < a.nq;
	float qx = 0.0f, qy = 0.0f, qz = 0.0f;
	uint32_t row = 0, start;
	if (a.q_xyz) {
		if (live) {
			row = a.q_order[slot];
			qx = a.q_xyz[3ull * row + 0];
			qy = a.q_xyz[3ull * row + 1];
			qz = a.q_xyz[3ull * row + 2];
		}
		start = a.q_start[wave_id];
	} else {
		if (live) {
			const float4 p = a.pts[slot];
			qx = p.x;
			qy = p.y;
			qz = p.z;
			row = __float_as_uint(p.w);
			live = row < a.q_limit; // (a padding record of a segmented cloud carries ~0u; ghosts sit behind the own points)
		}
		start = wave_id;
		if (!__any(live)) return; // a bucket of padding or of neighbour-only points
	}
	// dead lanes never pass the filter: NaN query -> NaN distance
	// A dead lane (padding, a neighbour-only point, a slot past the queries) must never need a box nor take a candidate.  Round 5: by
	// DATA, not by a mask ANDed into every ballot (which hipcc sent through a VGPR: select 0 / 1, compare again): its query lies at
	// x = +inf -- every box distance and every candidate distance is then +inf (or NaN) -- and its threshold is held at 0 (below).
	if (!live) qx = INFINITY;
	// detach the scalars from the 16-byte record tuple they were loaded in: otherwise the register
	// allocator spills the whole tuple and re-reads qx from scratch inside the hottest loops
	asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(row));
	const v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};

	double L[SK]; // FixedSet::new (best_set.rs:20-28): (max_distance, invalid)
#pragma unroll
	for (int i = 0; i < SK; ++i) L[i] = key_make(a.maxbits, 0u);
	uint32_t thr = live ? a.maxbits : 0u; // distance bits of the current threshold entry (below); 0 for a dead lane, for good
	// The threshold is the distance of the list's LAST entry only when all SK are asked for.  For k < SK -- the reference's k = 31
	// (calculations.rs:278) in the 32-entry list -- the (SK - 1)-th entry bounds the k-th from above just as well, and the search
	// volume is that of 31 neighbours instead of 32.  Exactness is unchanged: candidates pass with `<=`, so the whole tie group of
	// the k-th distance still arrives, and the entry behind the k-th is still "another point exactly as far, if there is one"
	// (what out_flags bit 1 reports).
#ifdef TREEE_NO_KBELOW // (A/B hook: the threshold of rounds 1-4)
	const bool k_below = false;
#else
	const bool k_below = a.k < (uint32_t)SK; // (wave-uniform)
#endif
	ST(if (a.thr_io && a.thr_seed && live) thr = a.thr_io[slot]); // diagnostic lower bound: start from the final answer
	// byte offset of this lane's next free buffer slot (slot stride 256 B); the buffer is appended
	// to without branches: every candidate is written, the offset only advances if it passed
	// (FASTBUF: the offset is the slot's LDS ADDRESS -- the hand-written append below has no base register to add)
#ifndef TREEE_APPEND_SELECT
	constexpr bool FASTBUF = SBUF == 16 && KEY_BIAS == 0u;
#else
	constexpr bool FASTBUF = false; // A/B hook: rounds 1-5a's branch-free append (write always, count if it passed)
#endif
	const uint32_t bufbase = FASTBUF ? lds_addr(&buf[0]) : 0u; // (wave-uniform)
	const uint32_t wofs0 = bufbase + (uint32_t)lane * 4u;
	uint32_t wofs = wofs0;
	uint32_t wlim = wofs0 + (uint32_t)(SBUF - SCHUNK) * 256u; // a buffer with fewer than SCHUNK free slots is flushed before the next trip
	asm volatile("" : "+v"(wlim)); // (kept in its own register: one compare per trip instead of a subtraction and a compare)
	constexpr uint32_t DB_OFS = SBUF * 64 * 4; // distance-bits plane behind the position plane
	uint32_t st_buckets = 0, st_flush = 0, st_nodes = 0, st_need = 0, st_small = 0;
	unsigned long long st_t0 = 0ull, st_tp = 0ull, st_tf = 0ull;
	uint32_t st_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	ST(st_t0 = st_tp = __builtin_amdgcn_s_memtime());
#define PH(i) ST(const unsigned long long t__ = __builtin_amdgcn_s_memtime(); st_ph[i] += (uint32_t)(t__ - st_tp); st_tp = t__)

	// Every distance slot of the lane's buffer at or behind its fill count holds the high word of KEY_MAX -- above every real key,
	// so it never wins a minimum against a list entry and its position word is never looked at.  Established here, kept by flush()
	// (round 5; before, flush() tested every slot against the fill count: a compare and a select per slot).
	auto buffer_sentinels = [&]() __attribute__((always_inline)) {
		if constexpr (SBUF == 16 && KEY_BIAS == 0u) {
#pragma unroll
			for (int s = 0; s < SBUF; ++s) buf[SBUF * 64 + s * 64 + lane] = (uint32_t)(KEY_MAX >> 32);
		}
	};
	buffer_sentinels();
	// Wave priorities (round 5): a wave in a scan is a chain of dependent steps -- broadcast read, distances, compare, store -- with
	// little to issue at any moment; a wave in a merge has 300 independent VALU instructions.  With the merging waves at the LOWER
	// priority the scanning ones get their few ready instructions issued at once and the merges fill what is left: 6.66 -> 6.53 ms
	// (the other way round: 7.07); the first sort and the epilogue at the merges' priority too: 6.50 -> 6.43.  Which two levels does
	// not matter (1 / 0, 2 / 0, 3 / 0, 3 / 1 within the noise); the BFS above or below the scans costs 0.05 ms either way.
	// -DTREEE_PRIO_SCAN=0 is the kernel without priorities.
#ifndef TREEE_PRIO_SCAN
#define TREEE_PRIO_SCAN 1
#endif
#ifndef TREEE_PRIO_FLUSH
#define TREEE_PRIO_FLUSH 0
#endif
	constexpr int PRIO_SCAN = TREEE_PRIO_SCAN, PRIO_FLUSH = TREEE_PRIO_FLUSH;
	__builtin_amdgcn_s_setprio(PRIO_SCAN);
	auto flush = [&]() __attribute__((always_inline)) {
		ST(st_tf = __builtin_amdgcn_s_memtime());
		__builtin_amdgcn_s_setprio(PRIO_FLUSH);
		double B[SBUF];
		// slots at or behind the lane's fill count carry a key above every real one: distance bits 0x7FEFFFFF (the high word of
		// KEY_MAX) over whatever position the slot last held -- such a key never wins a minimum against a list entry, so its low
		// word is never looked at.  (Round 5: one compare against an immediate and one select per slot; comparing the slot's
		// ADDRESS with the write offset kept sixteen per-lane addresses alive in VGPRs through the whole kernel.)
		const uint32_t fill = (wofs - wofs0) >> 8;
		if constexpr (SBUF == 16 && KEY_BIAS == 0u) {
			// the branch-free append always writes its candidate at the next free slot and only COUNTS it when it passed: that one slot
			// may hold a candidate that did not pass (for a lane outside a deferred scan's mask even one it already holds) -- sentinel
			// (the exec-masked append writes passing candidates only: nothing to repair)
			if (!FASTBUF && fill < (uint32_t)SBUF) *(uint32_t *)((char *)buf + wofs + DB_OFS) = (uint32_t)(KEY_MAX >> 32);
			// every slot as ONE register pair (position, distance bits) = the key itself: ds_read2st64_b32 with the two planes as its
			// two offsets.  Left to itself hipcc pairs two POSITIONS per read and assembles the keys with a v_mov per slot.
			const uint32_t ba = lds_addr(&buf[lane]);
			asm volatile("ds_read2st64_b32 %0, %16 offset0:0 offset1:16\n\tds_read2st64_b32 %1, %16 offset0:1 offset1:17\n\t"
			             "ds_read2st64_b32 %2, %16 offset0:2 offset1:18\n\tds_read2st64_b32 %3, %16 offset0:3 offset1:19\n\t"
			             "ds_read2st64_b32 %4, %16 offset0:4 offset1:20\n\tds_read2st64_b32 %5, %16 offset0:5 offset1:21\n\t"
			             "ds_read2st64_b32 %6, %16 offset0:6 offset1:22\n\tds_read2st64_b32 %7, %16 offset0:7 offset1:23\n\t"
			             "ds_read2st64_b32 %8, %16 offset0:8 offset1:24\n\tds_read2st64_b32 %9, %16 offset0:9 offset1:25\n\t"
			             "ds_read2st64_b32 %10, %16 offset0:10 offset1:26\n\tds_read2st64_b32 %11, %16 offset0:11 offset1:27\n\t"
			             "ds_read2st64_b32 %12, %16 offset0:12 offset1:28\n\tds_read2st64_b32 %13, %16 offset0:13 offset1:29\n\t"
			             "ds_read2st64_b32 %14, %16 offset0:14 offset1:30\n\tds_read2st64_b32 %15, %16 offset0:15 offset1:31\n\t"
			             "s_waitcnt lgkmcnt(0)"
			             : "=&v"(B[0]), "=&v"(B[1]), "=&v"(B[2]), "=&v"(B[3]), "=&v"(B[4]), "=&v"(B[5]), "=&v"(B[6]), "=&v"(B[7]), "=&v"(B[8]), "=&v"(B[9]),
			               "=&v"(B[10]), "=&v"(B[11]), "=&v"(B[12]), "=&v"(B[13]), "=&v"(B[14]), "=&v"(B[15])
			             : "v"(ba)
			             : "memory");
			// (no per-slot validity test: an unfilled slot HOLDS the sentinel -- see buffer_sentinels -- and the buffer is handed
			// back that way: the sixteen stores below cost the LDS pipe what sixteen compares and selects cost the VALU, and the VALU
			// is what this kernel is bound by)
			buffer_sentinels();
		} else {
#pragma unroll
			for (int s = 0; s < SBUF; ++s) {
				const uint32_t pos = buf[s * 64 + lane], db = buf[SBUF * 64 + s * 64 + lane];
				B[s] = key_make((uint32_t)s < fill ? db : (uint32_t)(KEY_MAX >> 32) - KEY_BIAS, pos);
			}
		}
		sort_net<SBUF>(B);
#if TREEE_AB_DOUBLE & 8
		sort_net<SBUF>(B); // (a sorted input leaves every network's output unchanged)
#endif
#pragma unroll
		for (int i = SK - SBUF; i < SK; ++i) L[i] = key_min(L[i], B[SK - 1 - i]); // 32 smallest of the union, bitonic
		bitonic_merge<SK>(L);
#if TREEE_AB_DOUBLE & 8
		bitonic_merge<SK>(L);
#endif
		{
			const uint32_t nt = k_below ? key_db(L[SK - 2]) : key_db(L[SK - 1]);
			thr = nt < thr ? nt : thr;
		}
		ST(st_flush++);
		wofs = wofs0;
		__builtin_amdgcn_s_setprio(PRIO_SCAN);
		ST(st_ph[6] += (uint32_t)(__builtin_amdgcn_s_memtime() - st_tf));
	};

	const float nanv = __uint_as_float(0x7FC00000u);
	auto load_bucket = [&](uint32_t b) __attribute__((always_inline)) -> float4 {
		const uint32_t i = b * BUCKET + lane;
		return i < a.n ? a.pts[i] : make_float4(nanv, nanv, nanv, 0.0f); // padding never passes
	};
	auto stage = [&](const float4 &p) __attribute__((always_inline)) {
		tile[0][lane] = p.x;
		tile[1][lane] = p.y;
		tile[2][lane] = p.z;
		__syncthreads(); // single-wave workgroup: orders the tile write before the broadcasts
	};
	// all 64 candidates of the staged tile against this lane's query; packed f32 math, two
	// candidates per instruction; the append is branch-free (write always, count if it passes)
	// `quarters`: bit q set = candidates [16 q, 16 q + 16) are scanned (wave-uniform; a quarter whose box no lane's search ball
	// reaches is left out: profiles/sim counts 27 % of the quarters of the buckets that come through here).  Runs of adjacent
	// quarters are scanned as one piece, so a bucket needed in full costs what it did.
	auto scan_tile = [&](uint32_t base, uint64_t lanes, int first, uint32_t quarters) __attribute__((always_inline)) {
		const bool act = (lanes >> lane) & 1ull; // lanes outside `lanes` must not append (they saw this bucket already)
		static_assert(SCHUNK == 4, "the pipelined scan reads 4 candidates per trip");
		uint32_t lim1 = act ? thr + 1u : 0u; // a candidate passes iff its distance bits are BELOW this (0: never)
		quarters &= 0xFu << (first >> 4);
#pragma unroll 1
		while (quarters) {
			const int q0 = __builtin_ctz(quarters), qn = __builtin_ctz(~(quarters >> q0)); // first quarter of the run, its length
			quarters &= ~(((1u << qn) - 1u) << q0);
			const int begin = q0 * 16, end = begin + qn * 16;
			// software pipeline: the broadcast reads of the NEXT four candidates are issued as soon as this trip's distances
			// exist (their registers are free again) and are in flight during the append -- the LDS round trip at the head
			// of every trip is what a trip waits for
			v4f nx = *(const v4f *)&tile[0][begin], ny = *(const v4f *)&tile[1][begin], nz = *(const v4f *)&tile[2][begin];
#pragma unroll 1
			for (int c0 = begin; c0 < end; c0 += SCHUNK) {
				if (__any(wofs > wlim)) {
					flush();
					lim1 = act ? thr + 1u : 0u;
				}
				uint32_t db[SCHUNK];
				{
					const v2f cx0 = {nx.x, nx.y}, cx1 = {nx.z, nx.w}, cy0 = {ny.x, ny.y}, cy1 = {ny.z, ny.w}, cz0 = {nz.x, nz.y}, cz1 = {nz.z, nz.w};
					const v2f dx0 = pk_sub(cx0, qx2), dy0 = pk_sub(cy0, qy2), dz0 = pk_sub(cz0, qz2), dx1 = pk_sub(cx1, qx2), dy1 = pk_sub(cy1, qy2), dz1 = pk_sub(cz1, qz2);
					const v2f d0 = (dx0 * dx0 + dy0 * dy0) + dz0 * dz0, d1 = (dx1 * dx1 + dy1 * dy1) + dz1 * dz1; // metric.rs:21-26
					db[0] = __float_as_uint(d0.x);
					db[1] = __float_as_uint(d0.y);
					db[2] = __float_as_uint(d1.x);
					db[3] = __float_as_uint(d1.y);
#if TREEE_AB_DOUBLE & 1
					v2f qxb = qx2, qyb = qy2, qzb = qz2;
					asm volatile("" : "+v"(qxb), "+v"(qyb), "+v"(qzb)); // the same query behind an opaque copy: nothing to share with the first evaluation
					const v2f ex0 = cx0 - qxb, ey0 = cy0 - qyb, ez0 = cz0 - qzb, ex1 = cx1 - qxb, ey1 = cy1 - qyb, ez1 = cz1 - qzb;
					const v2f e0 = (ex0 * ex0 + ey0 * ey0) + ez0 * ez0, e1 = (ex1 * ex1 + ey1 * ey1) + ez1 * ez1;
					db[0] = min(db[0], __float_as_uint(e0.x));
					db[1] = min(db[1], __float_as_uint(e0.y));
					db[2] = min(db[2], __float_as_uint(e1.x));
					db[3] = min(db[3], __float_as_uint(e1.y));
#endif
				}
				{
					const int cn = c0 + SCHUNK < end ? c0 + SCHUNK : c0; // (the last trip re-reads its own candidates: no branch)
#if TREEE_AB_DOUBLE & 4
					{
						v4f tx, ty, tz;
						asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
						             : "=&v"(tx), "=&v"(ty), "=&v"(tz)
						             : "v"(lds_addr(&tile[0][cn])), "v"(lds_addr(&tile[1][cn])), "v"(lds_addr(&tile[2][cn]))
						             : "memory");
					}
#endif
					nx = *(const v4f *)&tile[0][cn];
					ny = *(const v4f *)&tile[1][cn];
					nz = *(const v4f *)&tile[2][cn];
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // keeps the reads above the append (behind it: no gain, 6.68 / 6.68 ms)
				}
				if constexpr (FASTBUF) {
					// The append of the four candidates, by hand (round 5).  Per candidate: v_cmpx narrows EXEC to the lanes it passes
					// in, those lanes store (position, distance bits) at their next free slot and advance, a scalar move restores EXEC --
					// two VALU instructions.  hipcc's branch-free form (store always, compare, select the increment, add) is three, plus
					// a v_mov of the position each; here the four positions come as two register pairs from two v_pk_mov_b32 of
					// scalar pairs.  28 instead of 34 VALU per trip, and a slot never holds a candidate that did not pass.  (Measured: 6.69 ->
					// 6.64 ms per 15 M points, a fifth of what the instruction count promised -- the trip is one dependent chain through the
					// LDS and the compare's EXEC write; skipping store and increment when no lane passed, s_cbranch_execz, gave nothing;
					// the broadcast reads issued by hand so that the trip's head waits for lgkmcnt(4) -- the reads, not the appends behind
					// them -- cost 0.4 ms: twelve registers the compiler cannot see in flight are twelve it copies around; with the run's
					// first reads waited for where they are issued and ONE wait at the trip's head, still +0.1 ms.)
					const uint32_t p0 = base + (uint32_t)c0; // (wave-uniform)
					uint64_t P01, P23, sv;
					asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(P01) : "s"(((uint64_t)(p0 + 1u) << 32) | (uint64_t)p0));
					asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(P23) : "s"(((uint64_t)(p0 + 3u) << 32) | (uint64_t)(p0 + 2u)));
					static_assert(DB_OFS == 16 * 256, "offset1 of the append");
					asm volatile("s_mov_b64 %[sv], exec\n\t"
					             "v_cmpx_gt_u32_e32 %[lim], %[d0]\n\t"
					             "ds_write2st64_b32 %[w], %[q0], %[d0] offset1:16\n\t"
					             "v_add_u32_e32 %[w], 0x100, %[w]\n\t"
					             "s_mov_b64 exec, %[sv]\n\t"
					             "v_cmpx_gt_u32_e32 %[lim], %[d1]\n\t"
					             "ds_write2st64_b32 %[w], %[q1], %[d1] offset1:16\n\t"
					             "v_add_u32_e32 %[w], 0x100, %[w]\n\t"
					             "s_mov_b64 exec, %[sv]\n\t"
					             "v_cmpx_gt_u32_e32 %[lim], %[d2]\n\t"
					             "ds_write2st64_b32 %[w], %[q2], %[d2] offset1:16\n\t"
					             "v_add_u32_e32 %[w], 0x100, %[w]\n\t"
					             "s_mov_b64 exec, %[sv]\n\t"
					             "v_cmpx_gt_u32_e32 %[lim], %[d3]\n\t"
					             "ds_write2st64_b32 %[w], %[q3], %[d3] offset1:16\n\t"
					             "v_add_u32_e32 %[w], 0x100, %[w]\n\t"
					             "s_mov_b64 exec, %[sv]"
					             : [w] "+v"(wofs), [sv] "=&s"(sv)
					             : [lim] "v"(lim1), [d0] "v"(db[0]), [d1] "v"(db[1]), [d2] "v"(db[2]), [d3] "v"(db[3]), [q0] "v"((uint32_t)P01),
					               [q1] "v"((uint32_t)(P01 >> 32)), [q2] "v"((uint32_t)P23), [q3] "v"((uint32_t)(P23 >> 32))
					             : "vcc", "memory");
				} else
#pragma unroll
				for (int j = 0; j < SCHUNK; ++j) { // coarse filter; the merge applies the full (distance, position) order
					char *w = (char *)buf + wofs;
					*(uint32_t *)w = base + c0 + j;
					*(uint32_t *)(w + DB_OFS) = db[j];
#if TREEE_AB_DOUBLE & 2
					static_assert(DB_OFS == 16 * 256, "offset1 of the duplicate append");
					asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:16" ::"v"(lds_addr(w)), "v"(base + c0 + j), "v"(db[j]) : "memory");
#endif
					wofs += (db[j] <= thr && act) ? 256u : 0u;
				}
			}
		}
		__syncthreads();
	};
	auto scan_bucket = [&](uint32_t b) __attribute__((always_inline)) {
		if (b * BUCKET >= a.n) return;
		stage(load_bucket(b));
		scan_tile(b * BUCKET, ~0ull, 0, 0xFu);
	};
	// lanes (queries) whose search ball still reaches the box
	auto lane_box_hit = [&](float lox, float loy, float loz, float hix, float hiy, float hiz) __attribute__((always_inline)) -> uint64_t {
		const float dx = fmaxf(fmaxf(lox - qx, qx - hix), 0.0f);
		const float dy = fmaxf(fmaxf(loy - qy, qy - hiy), 0.0f);
		const float dz = fmaxf(fmaxf(loz - qz, qz - hiz), 0.0f);
		const float bd = (dx * dx + dy * dy) + dz * dz;
#if TREEE_AB_DOUBLE & 32
		float qxb = qx, qyb = qy, qzb = qz;
		asm volatile("" : "+v"(qxb), "+v"(qyb), "+v"(qzb));
		const float ex = fmaxf(fmaxf(lox - qxb, qxb - hix), 0.0f), ey = fmaxf(fmaxf(loy - qyb, qyb - hiy), 0.0f), ez = fmaxf(fmaxf(loz - qzb, qzb - hiz), 0.0f);
		const float be = (ex * ex + ey * ey) + ez * ez;
		return __ballot(live && __float_as_uint(bd) <= thr) & __ballot(live && __float_as_uint(be) <= thr);
#endif
		return __ballot(__float_as_uint(bd) <= thr); // (a dead lane: bd = +inf against a threshold of 0)
	};

	// Quarters of a bucket some lane's search ball still reaches (a wave-uniform 4-bit mask).  Lanes 0 .. 7 fetch one corner
	// each of the bucket's four quarter boxes (k_leaf_aabb) -- together with the bucket's records, so that the test adds no
	// round trip --; through LDS the corners become broadcast operands of the per-lane box test.
	auto sub_load = [&](uint32_t b) __attribute__((always_inline)) -> float4 {
		return lane < 8 ? a.sub[8ull * b + (uint32_t)lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	};
	// (round 5: the four quarter boxes are tested two at a time with packed arithmetic -- 38 VALU instructions instead of 60 per
	// staged bucket.  The corners lie in LDS by coordinate and side, [x lo | x hi | y lo | y hi | z lo | z hi][quarter], so that one
	// broadcast read brings a coordinate of all four quarters; values and operation order are those of lane_box_hit.)
	auto quarters_needed = [&](const float4 &sv) __attribute__((always_inline)) -> uint32_t {
		float *f = (float *)&s_sub[0][0];
		if (lane < 8) { // lane = 2 * quarter + side (k_leaf_aabb's layout of a.sub)
			const int o = (lane & 1) * 4 + (lane >> 1);
			f[o] = sv.x;
			f[8 + o] = sv.y;
			f[16 + o] = sv.z;
		}
		__syncthreads();
		const v4f lox = *(const v4f *)&f[0], hix = *(const v4f *)&f[4], loy = *(const v4f *)&f[8], hiy = *(const v4f *)&f[12], loz = *(const v4f *)&f[16],
		          hiz = *(const v4f *)&f[20]; // same addresses in every lane: LDS broadcasts
		uint32_t m = 0u;
#pragma unroll
		for (int h = 0; h < 2; ++h) { // quarters 2 h and 2 h + 1
			const v2f lx = h ? v2f{lox.z, lox.w} : v2f{lox.x, lox.y}, hx = h ? v2f{hix.z, hix.w} : v2f{hix.x, hix.y};
			const v2f ly = h ? v2f{loy.z, loy.w} : v2f{loy.x, loy.y}, hy = h ? v2f{hiy.z, hiy.w} : v2f{hiy.x, hiy.y};
			const v2f lz = h ? v2f{loz.z, loz.w} : v2f{loz.x, loz.y}, hz = h ? v2f{hiz.z, hiz.w} : v2f{hiz.x, hiz.y};
			const v2f ax = pk_sub(lx, qx2), bx = pk_sub(qx2, hx), ay = pk_sub(ly, qy2), by = pk_sub(qy2, hy), az = pk_sub(lz, qz2), bz = pk_sub(qz2, hz);
			const v2f dx = {fmaxf(fmaxf(ax.x, bx.x), 0.0f), fmaxf(fmaxf(ax.y, bx.y), 0.0f)};
			const v2f dy = {fmaxf(fmaxf(ay.x, by.x), 0.0f), fmaxf(fmaxf(ay.y, by.y), 0.0f)};
			const v2f dz = {fmaxf(fmaxf(az.x, bz.x), 0.0f), fmaxf(fmaxf(az.y, bz.y), 0.0f)};
			const v2f bd = (dx * dx + dy * dy) + dz * dz;
			m |= __ballot(__float_as_uint(bd.x) <= thr) ? 1u << (2 * h) : 0u;
			m |= __ballot(__float_as_uint(bd.y) <= thr) ? 2u << (2 * h) : 0u;
		}
		return m;
	};

	if (start >= a.nb) start = a.nb - 1;
	// buckets start-1, start+1, start-2, ... (TREEE_NBR of them) are scanned right after the own one
	auto prescanned = [&](uint32_t b) __attribute__((always_inline)) -> bool {
		const uint32_t below = start - b, above = b - start; // one of them wraps to a huge value
		return (below >= 1u && below <= (uint32_t)((TREEE_NBR + 1) / 2)) || (above >= 1u && above <= (uint32_t)(TREEE_NBR / 2));
	};
	uint32_t myseg = 0u; // wave-uniform: a bucket belongs to one segment
	if constexpr (SEG) myseg = __float_as_uint(a.bvh[2ull * (a.P + start)].w);
	// Own bucket, then its two neighbours along the curve.  When a bucket straddles a jump of the
	// curve, the lanes on its minority side find their real neighbours in the adjacent bucket; without
	// it their threshold (and with it the wave's candidate list) would be enormous.
#pragma unroll 1
	for (int t = 0; t < 1 + TREEE_NBR; ++t) {
		const uint32_t b = t == 0 ? start : ((t & 1) ? start - (uint32_t)((t + 1) >> 1) : start + (uint32_t)(t >> 1)); // below 0 wraps to a huge value
		if (b >= a.nb) continue;
		if constexpr (SEG) {
			if (t > 0 && __float_as_uint(a.bvh[2ull * (a.P + b)].w) != myseg) continue; // Morton neighbour of another segment
		}
		const float4 sv = sub_load(b);
		stage(load_bucket(b));
		const uint32_t base = b * BUCKET;
		int first = 0;
		uint32_t quarters = 0xFu;
		if (t > 0) quarters = quarters_needed(sv);
		if (t == 0) {
			// against an unbounded threshold everything passes, so the first 32 records go straight into
			// the list (one 240-CE sort instead of eight buffer merges)
#pragma unroll
			for (int j = 0; j < SK; j += 2) {
				const v2f cx = *(const v2f *)&tile[0][j];
				const v2f cy = *(const v2f *)&tile[1][j];
				const v2f cz = *(const v2f *)&tile[2][j];
				const v2f dx = cx - qx2, dy = cy - qy2, dz = cz - qz2;
				const v2f d = (dx * dx + dy * dy) + dz * dz;
				const uint32_t d0 = __float_as_uint(d.x), d1 = __float_as_uint(d.y);
				L[j] = d0 < a.maxbits ? key_make(d0, base + j) : key_make(a.maxbits, 0u); // strict `<` (best_set.rs:20-28)
				L[j + 1] = d1 < a.maxbits ? key_make(d1, base + j + 1) : key_make(a.maxbits, 0u);
			}
			__builtin_amdgcn_s_setprio(PRIO_FLUSH); // (the first sort: 191 compare-exchanges, as dense as a merge)
			sort_net<SKT>(L);
			__builtin_amdgcn_s_setprio(PRIO_SCAN);
			thr = k_below ? key_db(L[SK - 2]) : key_db(L[SK - 1]);
			if (!live) thr = 0u;
			ST(if (a.thr_io && a.thr_seed && live) thr = min(thr, a.thr_io[slot]));
			first = SK;
		}
		scan_tile(base, ~0ull, first, quarters);
		ST(st_buckets++);
	}
	if (__any(wofs != wofs0)) flush(); // threshold = 32nd best inside the own bucket
	PH(0);

	// ---- candidate buckets by a wave-parallel BFS: one lane per tree node -----------------------
	// Query box of the wave and its largest threshold bound every lane's search ball, so a node
	// whose box is farther than that from the query box cannot matter to any lane.  Seeds are the
	// sibling subtrees along the path from the own bucket to the root (an exact cover).
	// Query boxes: one per group of GL consecutive lanes (Morton-adjacent queries), each with the
	// largest threshold of its lanes.  Measured at 15 M points (the kernel is VALU-issue bound, a sub-box costs
	// ~14 VALU per tested node): 16 boxes 12.19 ms, 8: 11.85, 4: 11.53, 2: 11.53, 1: 11.45 -- but a single box lists
	// 58 buckets per wave instead of 50 and overflows the list 1.5x as often, so 4 is the default.  The boxes
	// live in the (idle) tile memory.
	constexpr int NG = TREEE_NG, GL = 64 / NG; // sub-boxes per wave, lanes per sub-box
	float *gbox = &tile[0][0]; // NG == 4: [lo.x | lo.y | lo.z | tau bits | hi.x | hi.y | hi.z][sub-box] (a coordinate of all four per read); else [NG][8]: lo.xyz, tau bits, hi.xyz, pad
	uint32_t list_m = 0;
	if constexpr (!OVF) {
	{
		const float q[3] = {qx, qy, qz};
		float glo[3], ghi[3];
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			float lo = live ? q[c] : INFINITY, hi = live ? q[c] : -INFINITY;
			if constexpr (GL == 16) {
				lo = row16_min_f32(lo);
				hi = row16_max_f32(hi);
			} else {
#pragma unroll
				for (int o = GL / 2; o > 0; o >>= 1) {
					lo = fminf(lo, __shfl_xor(lo, o, 64));
					hi = fmaxf(hi, __shfl_xor(hi, o, 64));
				}
			}
			glo[c] = lo;
			ghi[c] = hi;
		}
		uint32_t gtau = live ? thr : 0u;
		if constexpr (GL == 16) {
			gtau = row16_max_u32(gtau);
		} else {
#pragma unroll
			for (int o = GL / 2; o > 0; o >>= 1) {
				const uint32_t t = __shfl_xor(gtau, o, 64);
				gtau = t > gtau ? t : gtau;
			}
		}
		if ((lane & (GL - 1)) == 0) {
			if constexpr (NG == 4) {
				float *g = gbox + lane / GL;
				g[0] = glo[0];
				g[4] = glo[1];
				g[8] = glo[2];
				g[12] = __uint_as_float(gtau);
				g[16] = ghi[0];
				g[20] = ghi[1];
				g[24] = ghi[2];
			} else {
				float *g = gbox + (lane / GL) * 8;
				g[0] = glo[0];
				g[1] = glo[1];
				g[2] = glo[2];
				g[3] = __uint_as_float(gtau);
				g[4] = ghi[0];
				g[5] = ghi[1];
				g[6] = ghi[2];
				g[7] = 0.0f;
			}
		}
	}
#if TREEE_FRONT_CAP > TREEE_SBUF * 64
	__shared__ uint32_t s_front[2 * FRONT_CAP];
	uint32_t *front = s_front;
#else
	uint32_t *front = (uint32_t *)buf; // two frontiers of FRONT_CAP nodes alias the (empty) buffer
#endif
	uint32_t cur = 0, m = 0;
	for (uint32_t nd = a.P + start; nd > 1u; nd >>= 1) ++m; // tree depth = number of seeds (<= 31)
	if ((uint32_t)lane < m) front[lane] = ((a.P + start) >> lane) ^ 1u;
	__syncthreads();
	bool overflow = false;
	while (m > 0) {
		uint32_t next_m = 0;
		uint32_t *nxt = front + (cur ^ 1u) * FRONT_CAP;
		for (uint32_t base = 0; base < m; base += 64) {
			const uint32_t idx = base + lane;
			const bool valid = idx < m;
			const uint32_t node = valid ? front[cur * FRONT_CAP + idx] : 1u;
			const float4 lo = a.bvh[2ull * node + 0], hi = a.bvh[2ull * node + 1];
			bool hit = false;
			if constexpr (NG == 4) {
				// two sub-boxes per instruction (round 5): the node's corner coordinate is one ELEMENT of the register pair its load
				// left it in, the sub-boxes' coordinates come four to a broadcast read -- 38 VALU instructions per node instead of 60;
				// values and operation order are those of the loop below
				const v2f lxy = {lo.x, lo.y}, lzw = {lo.z, lo.w}, hxy = {hi.x, hi.y}, hzw = {hi.z, hi.w};
				const v4f blx = *(const v4f *)(gbox + 0), bly = *(const v4f *)(gbox + 4), blz = *(const v4f *)(gbox + 8), btau = *(const v4f *)(gbox + 12),
				          bhx = *(const v4f *)(gbox + 16), bhy = *(const v4f *)(gbox + 20), bhz = *(const v4f *)(gbox + 24);
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const v2f glx = h ? v2f{blx.z, blx.w} : v2f{blx.x, blx.y}, gly = h ? v2f{bly.z, bly.w} : v2f{bly.x, bly.y}, glz = h ? v2f{blz.z, blz.w} : v2f{blz.x, blz.y};
					const v2f ghx = h ? v2f{bhx.z, bhx.w} : v2f{bhx.x, bhx.y}, ghy = h ? v2f{bhy.z, bhy.w} : v2f{bhy.x, bhy.y}, ghz = h ? v2f{bhz.z, bhz.w} : v2f{bhz.x, bhz.y};
					const v2f tau = h ? v2f{btau.z, btau.w} : v2f{btau.x, btau.y};
					const v2f ax = pk_sub_elem_minus<0>(lxy, ghx), bx = pk_sub_minus_elem<0>(glx, hxy); // lo.x - bhi.x, blo.x - hi.x
					const v2f ay = pk_sub_elem_minus<1>(lxy, ghy), by = pk_sub_minus_elem<1>(gly, hxy);
					const v2f az = pk_sub_elem_minus<0>(lzw, ghz), bz = pk_sub_minus_elem<0>(glz, hzw);
					const v2f dx = {fmaxf(fmaxf(ax.x, bx.x), 0.0f), fmaxf(fmaxf(ax.y, bx.y), 0.0f)};
					const v2f dy = {fmaxf(fmaxf(ay.x, by.x), 0.0f), fmaxf(fmaxf(ay.y, by.y), 0.0f)};
					const v2f dz = {fmaxf(fmaxf(az.x, bz.x), 0.0f), fmaxf(fmaxf(az.y, bz.y), 0.0f)};
					const v2f bd = (dx * dx + dy * dy) + dz * dz;
					hit = hit || (__float_as_uint(bd.x) <= __float_as_uint(tau.x)) || (__float_as_uint(bd.y) <= __float_as_uint(tau.y));
				}
			} else {
#pragma unroll 2
			for (int g = 0; g < NG; ++g) { // same address in every lane: LDS broadcast (unroll 2: register pressure)
				const float4 blo = *(const float4 *)(gbox + g * 8), bhi = *(const float4 *)(gbox + g * 8 + 4);
				const float dx = fmaxf(fmaxf(lo.x - bhi.x, blo.x - hi.x), 0.0f);
				const float dy = fmaxf(fmaxf(lo.y - bhi.y, blo.y - hi.y), 0.0f);
				const float dz = fmaxf(fmaxf(lo.z - bhi.z, blo.z - hi.z), 0.0f);
				const float bd = (dx * dx + dy * dy) + dz * dz;
				hit = hit || (__float_as_uint(bd) <= __float_as_uint(blo.w));
			}
			}
#if TREEE_AB_DOUBLE & 64
			{
				float lx = lo.x, ly = lo.y, lz = lo.z;
				asm volatile("" : "+v"(lx), "+v"(ly), "+v"(lz));
				bool hit2 = false;
#pragma unroll 2
				for (int g = 0; g < NG; ++g) {
					const float4 blo = NG == 4 ? make_float4(gbox[g], gbox[4 + g], gbox[8 + g], gbox[12 + g]) : *(const float4 *)(gbox + g * 8);
					const float4 bhi = NG == 4 ? make_float4(gbox[16 + g], gbox[20 + g], gbox[24 + g], 0.0f) : *(const float4 *)(gbox + g * 8 + 4);
					const float dx = fmaxf(fmaxf(lx - bhi.x, blo.x - hi.x), 0.0f), dy = fmaxf(fmaxf(ly - bhi.y, blo.y - hi.y), 0.0f), dz = fmaxf(fmaxf(lz - bhi.z, blo.z - hi.z), 0.0f);
					const float bd = (dx * dx + dy * dy) + dz * dz;
					hit2 = hit2 || (__float_as_uint(bd) <= __float_as_uint(blo.w));
				}
				hit = hit && hit2;
			}
#endif
			hit = hit && valid;
			if constexpr (SEG) hit = hit && myseg >= __float_as_uint(lo.w) && myseg <= __float_as_uint(hi.w);
			// the own bucket's Morton neighbours were scanned above (the own bucket is no seed's descendant)
			const bool leafhit = hit && node >= a.P && (node - a.P) < a.nb && !prescanned(node - a.P);
			const bool innerhit = hit && node < a.P;
			// 4-ary step: a hit inner node expands straight to its four grandchildren (contiguous in the
			// implicit layout) when they exist, which halves the number of dependent BFS rounds
			const bool deep = innerhit && 2u * node < a.P;
			const bool shallow = innerhit && !deep;
			const uint64_t bl = __ballot(leafhit), b4 = __ballot(deep), b2 = __ballot(shallow);
			if (leafhit) {
				const uint32_t o = list_m + __popcll(bl & lt_mask);
				if (o < (uint32_t)LIST_CAP) s_list[o] = node - a.P;
			}
			if (innerhit) {
				const uint32_t o = next_m + 4u * __popcll(b4 & lt_mask) + 2u * __popcll(b2 & lt_mask);
				if (deep) {
					if (o + 3u < (uint32_t)FRONT_CAP) {
						nxt[o] = 4u * node;
						nxt[o + 1u] = 4u * node + 1u;
						nxt[o + 2u] = 4u * node + 2u;
						nxt[o + 3u] = 4u * node + 3u;
					}
				} else if (o + 1u < (uint32_t)FRONT_CAP) {
					nxt[o] = 2u * node;
					nxt[o + 1u] = 2u * node + 1u;
				}
			}
			list_m += __popcll(bl);
			next_m += 4u * __popcll(b4) + 2u * __popcll(b2);
			ST(st_nodes++);
		}
		if (list_m > (uint32_t)LIST_CAP || next_m > (uint32_t)FRONT_CAP) {
			overflow = true;
			break;
		}
		cur ^= 1u;
		m = next_m;
		__syncthreads();
	}
	__syncthreads();
#if !(TREEE_FRONT_CAP > TREEE_SBUF * 64)
	buffer_sentinels(); // (the frontiers lay in the buffer's memory)
#endif
	if (overflow) { // wave-uniform; nothing has been written for this wave yet
		if (DIAG && a.stats && lane == 0) atomicAdd(&a.stats[8], 1u);
		// (one callee per list length class: a longer list keeps the same first k.  The callee reads the arguments from the
		// kernel's argument block -- KnnArgs is the kernel's only parameter --: handed over by value they were parked on the
		// stack at kernel entry by every wave, 2 KB of scratch writes each)
		knn_wave_overflowed<SEG, (SKT <= 32 ? 32 : 64)>((KnnArgsInKernarg)__builtin_amdgcn_kernarg_segment_ptr(), wave_id);
		return;
	}
	} // !OVF

	PH(1);
	ST(st_ph[7] = list_m);
	if constexpr (!OVF) {
		// ---- scan the listed buckets; the next bucket's load is in flight during each scan -------
		const float4 own_lo = a.bvh[2ull * (a.P + start) + 0], own_hi = a.bvh[2ull * (a.P + start) + 1]; // (wave-uniform: the box of the wave's queries)
		for (uint32_t g = 0; g < list_m; g += GROUP) {
			const uint32_t gcount = (list_m - g) < GROUP ? (list_m - g) : GROUP;
			// boxes of this group of listed buckets go to LDS (8 VGPRs less to carry through the scans) -- NEAREST FIRST (round 5): the
			// BFS leaves the list in frontier order; scanned in the order of their boxes' distance from the own bucket's box (= the
			// box of the wave's queries) the near buckets tighten the thresholds before the far ones are tested, and fewer of those
			// are still needed when their turn comes (modelled, profiles/sim "listed nearest first": 20.8 instead of 21.7 buckets and
			// 17.2 instead of 18.2 merges per wave; measured 7.31 -> 7.12 ms).  The order is a matter of speed only: the lists are exact
			// in any order.  One key per lane = distance bits with the low six replaced by the lane, a bitonic sort across the wave, the
			// inverse permutation through 64 bytes of LDS (the quarter-box slots, idle here).
			{
				const bool has = (uint32_t)lane < gcount;
				uint32_t my_leaf = 0u;
				float4 blo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), bhi = blo;
				uint32_t comp = 0xFFFFFFC0u | (uint32_t)lane; // (lanes without a bucket sort behind every bucket)
				if (has) {
					my_leaf = s_list[g + lane];
					blo = a.bvh[2ull * (a.P + my_leaf) + 0];
					bhi = a.bvh[2ull * (a.P + my_leaf) + 1];
					const float dx = fmaxf(fmaxf(blo.x - own_hi.x, own_lo.x - bhi.x), 0.0f);
					const float dy = fmaxf(fmaxf(blo.y - own_hi.y, own_lo.y - bhi.y), 0.0f);
					const float dz = fmaxf(fmaxf(blo.z - own_hi.z, own_lo.z - bhi.z), 0.0f);
					comp = (__float_as_uint((dx * dx + dy * dy) + dz * dz) & ~63u) | (uint32_t)lane;
				}
#ifndef TREEE_LIST_BFS_ORDER // (A/B hook: the frontier order of rounds 1-4)
				comp = wave_sort_stages<2u, 1u>(comp, (uint32_t)lane);
#endif
				uint8_t *rank_of = (uint8_t *)&s_sub[0][0];
				rank_of[comp & 63u] = (uint8_t)lane;
				__syncthreads(); // (every lane has read its list entry before any lane rewrites one)
				const uint32_t r = rank_of[lane];
				if (has) {
					s_gbox[r][0] = blo;
					s_gbox[r][1] = bhi;
					s_list[g + r] = my_leaf;
				}
			}
			// the id of listed bucket j is re-read from the list in LDS (one broadcast read) wherever it is needed: kept
			// in a VGPR across the scans it was spilled, and every bucket then began with two scratch round trips --
			// one before its record could be used, one before the next prefetch could be issued
			auto listed = [&](uint32_t jj) __attribute__((always_inline)) -> uint32_t {
				return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_list[g + jj]);
			};
			__syncthreads();
			// next listed bucket (index >= from) that some lane still needs, judged with the current
			// threshold (it only ever shrinks, so a stale decision is merely conservative)
			uint64_t need = 0; // queries that need the bucket find_next returned
			auto find_next = [&](uint32_t from, uint64_t &mask) __attribute__((always_inline)) -> uint32_t {
				for (uint32_t j = from; j < gcount; ++j) {
					const float4 lo = s_gbox[j][0], hi = s_gbox[j][1]; // same address in every lane: LDS broadcast
					mask = lane_box_hit(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
					if (mask) return j;
				}
				return gcount;
			};
			// the record this lane would stage for listed bucket j (clamped address: padding lanes of the
			// last bucket read a valid record and are replaced by NaN after the wait)
			auto bucket_ptr = [&](uint32_t jj) __attribute__((always_inline)) -> const float4 * {
				const uint32_t i = listed(jj) * BUCKET + lane;
				return a.pts + (i < a.n ? i : a.n - 1u);
			};
			PH(5);
			uint32_t j = find_next(0, need);
			PH(2);
			// Prefetch of the next bucket's record.  A plain load would be sunk by hipcc to its first use
			// (below the scan it should overlap); a wavefront-scope release fence costs no instruction
			// but pins the load above it.  The value stays an ordinary compiler-tracked register value:
			// the compiler places the vmcnt wait at the use (an earlier version issued the load from inline
			// asm and waited by hand, which faults as soon as the allocator copies the in-flight registers).
			float4 pf = make_float4(nanv, nanv, nanv, 0.0f), pf_sub = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			auto issue_prefetch = [&](uint32_t jj) __attribute__((always_inline)) {
				pf = *bucket_ptr(jj);
				pf_sub = sub_load(listed(jj));
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			};
			if (j < gcount) issue_prefetch(j);
			while (j < gcount) {
				const uint32_t leaf = listed(j);
				const uint64_t need_cur = need;
				const bool inr = leaf * BUCKET + lane < a.n;
				const float4 cand = make_float4(inr ? pf.x : nanv, inr ? pf.y : nanv, inr ? pf.z : nanv, 0.0f); // this lane's record of bucket `leaf`
				ST(st_need += (uint32_t)__popcll(need_cur));
				const bool small = __popcll(need_cur) < TRANSPOSE_BELOW;
				// order matters: consume this bucket's registers (LDS staging) BEFORE the next load is issued,
				// otherwise the wait for `cand` also drains the prefetch and exposes its latency
				uint32_t quarters = 0xFu;
				if (!small) {
					stage(cand);
					quarters = quarters_needed(pf_sub);
				}
				PH(8);
				j = find_next(j + 1, need);
				if (j < gcount) issue_prefetch(j); // in flight during the scan below
				PH(2);
				uint64_t full = ~0ull; // queries to serve with the lane-parallel scan
				if (small) {
					// ---- transposed scan: few queries still need this bucket, so lanes = candidates (kept in
					// registers, no LDS staging) and the loop runs over the needing queries only.  A query
					// whose buffer lacks room is deferred to the lane-parallel scan below (rare).
					ST(st_small++);
					full = 0ull;
					const uint32_t mypos = leaf * BUCKET + lane;
					uint64_t todo = need_cur;
					while (todo) {
						const int q = __builtin_ctzll(todo);
						todo &= todo - 1ull;
						const float sqx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qx), q));
						const float sqy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qy), q));
						const float sqz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qz), q));
						const uint32_t sthr = (uint32_t)__builtin_amdgcn_readlane((int)thr, q);
						const float dx = cand.x - sqx, dy = cand.y - sqy, dz = cand.z - sqz;
						const float d = (dx * dx + dy * dy) + dz * dz;
						const uint32_t db = __float_as_uint(d);
						const bool pass = db <= sthr; // NaN padding never passes
#if TREEE_AB_DOUBLE & 16
						float cxb = cand.x, cyb = cand.y, czb = cand.z;
						asm volatile("" : "+v"(cxb), "+v"(cyb), "+v"(czb));
						const float sqx2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qx), q ^ 0)), ex = cxb - sqx2, ey = cyb - sqy, ez = czb - sqz;
						const float e = (ex * ex + ey * ey) + ez * ez;
						const uint64_t pm = __ballot(pass) & __ballot(__float_as_uint(e) <= sthr);
#else
						const uint64_t pm = __ballot(pass);
#endif
						if (!pm) continue;
						const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)wofs, q);
						const uint32_t room = (uint32_t)SBUF - ((w - bufbase - (uint32_t)q * 4u) >> 8);
						const uint32_t npass = (uint32_t)__popcll(pm);
						if (npass > room) {
							full |= 1ull << q;
							continue;
						}
						if (pass) {
							const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
							char *wp = (char *)buf + (w - bufbase) + pre * 256u;
							*(uint32_t *)wp = mypos;
							*(uint32_t *)(wp + DB_OFS) = db;
						}
						wofs = lane == q ? w + npass * 256u : wofs;
					}
				}
				PH(3);
				if (full) {
					if (small) stage(cand); // deferred queries of a transposed bucket (rare)
					scan_tile(leaf * BUCKET, full, 0, quarters);
				}
				PH(4);
				ST(st_buckets++);
			}
		}
	} else {
		// ---- the waves whose list overflowed (huge search radius): packet traversal, bottom-up from the own bucket -------
		uint32_t node = a.P + start;
		while (node > 1u) {
			const uint32_t root = node ^ 1u;
			uint32_t i = root;
			for (;;) {
				const float4 lo = a.bvh[2ull * i + 0], hi = a.bvh[2ull * i + 1];
				bool hit = lane_box_hit(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) != 0;
				if constexpr (SEG) hit = hit && myseg >= __float_as_uint(lo.w) && myseg <= __float_as_uint(hi.w);
				ST(st_nodes++);
				if (hit && i < a.P) {
					i = 2u * i;
					continue;
				}
				if (hit && !prescanned(i - a.P)) { // the neighbours were scanned above
					scan_bucket(i - a.P);
					ST(st_buckets++);
				}
				while (i != root && (i & 1u)) i >>= 1;
				if (i == root) break;
				i += 1u;
			}
			node >>= 1;
		}
	}
	if (__any(wofs != wofs0)) flush();
	ST(if (a.thr_io && !a.thr_seed && live) a.thr_io[slot] = thr);
	if (DIAG && a.stats && lane == 0) {
		atomicAdd(&a.stats[0], st_buckets);
		atomicAdd(&a.stats[1], st_flush);
		atomicAdd(&a.stats[2], st_nodes);
		atomicAdd(&a.stats[3], 1u);
		atomicAdd(&a.stats[4], st_need);
		atomicAdd(&a.stats[5], st_small);
		const uint32_t cyc = (uint32_t)((__builtin_amdgcn_s_memtime() - st_t0) >> 6); // 64-cycle units
		atomicAdd(&a.stats[6], cyc >> 4);
		atomicMax(&a.stats[7], cyc);
		for (int i = 0; i < 7; ++i) atomicAdd(&a.stats[40 + i], st_ph[i] >> 10);
		atomicAdd(&a.stats[50], st_ph[8] >> 10);
		atomicAdd(&a.stats[47], st_ph[7]);
		atomicMax(&a.stats[48], st_ph[7]);
		atomicAdd(&a.stats[9 + (cyc >> 10 > 21 ? 21 : cyc >> 10)], 1u); // histogram, 65536-cycle bins
	}
	if (!live) return;

	PH(5);
	__builtin_amdgcn_s_setprio(PRIO_FLUSH); // (the epilogue: straight-line arithmetic)
	const uint32_t k = a.k;
	// valid entries among the first k: the list is ascending and an unused entry carries max_distance, so the valid ones lead --
	// count = min(k, valid entries of the whole list).  (Written as `s < k && valid` per entry, the 32 wave-uniform conditions
	// were hoisted into 32 SGPR pairs that spilled to VGPR lanes: ~250 v_writelane / v_readlane per wave; round 5.)
	uint32_t count = 0;
#pragma unroll
	for (int s = 0; s < SK; ++s) count += key_db(L[s]) < a.maxbits ? 1u : 0u;
	count = count < k ? count : k;
	if (a.out_count) a.out_count[row] = count;
	if (a.out_flags) { // (wave-uniform) rows whose list the reference's KD-tree may order or cut differently: k-nearest/src/kd_tree.rs:114,125 decide
		// equal distances by visit order, this structure by (distance, cell, index) -- SURVEY.md H2
		uint32_t m = 0u; // bit s: entries s and s + 1 are equally far (and both exist)
#pragma unroll
		for (int s = 0; s + 1 < SK && s < 32; ++s) m |= (key_db(L[s]) == key_db(L[s + 1]) && key_db(L[s + 1]) < a.maxbits) ? 1u << s : 0u;
		uint32_t m_hi = 0u; // (the 64-entry list: bits 32 ...)
		if constexpr (SK > 33) {
#pragma unroll
			for (int s = 32; s + 1 < SK; ++s) m_hi |= (key_db(L[s]) == key_db(L[s + 1]) && key_db(L[s + 1]) < a.maxbits) ? 1u << (s - 32) : 0u;
		}
		const unsigned long long mm = ((unsigned long long)m_hi << 32) | m, inside = k >= 2u ? (1ull << (k - 1u)) - 1ull : 0ull; // pairs (s, s + 1) with s + 1 < k
		const uint32_t f = ((mm & inside) ? 1u : 0u) | ((k < (uint32_t)SK && ((mm >> (k - 1u)) & 1ull)) ? 2u : 0u); // ... and the pair (k - 1, k)
		a.out_flags[row] = (uint8_t)f;
	}
	if (a.out_dist || a.out_index) {
#pragma unroll
		for (int s = 0; s < SK; ++s) {
			if ((uint32_t)s < k) {
				const bool ok = (uint32_t)s < count;
				if (a.out_dist) a.out_dist[(size_t)row * k + s] = __uint_as_float(ok ? key_db(L[s]) : a.maxbits);
				if (a.out_index) a.out_index[(size_t)row * k + s] = ok ? __float_as_uint(a.pts[key_pos(L[s])].w) : NONE32;
			}
		}
	}
	if (a.out_curve) { // treee/src/calculations.rs:342-368, neighbours in ascending order
		// The neighbours' coordinates are gathered ONCE and kept in registers (the list is dead from here on: 3 x SK values);
		// a slot beyond the lane's count holds position 0 -- a valid address -- and is left out of the sums below.  Nearly
		// every wave has count == k in all its lanes (unbounded radius): then no sum carries a per-lane condition.
		const float len = (float)count;
		float mean[3] = {0.0f, 0.0f, 0.0f};
		float var[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
		// GENERAL form: two gathers, a lane sums its first `count` entries (one per-lane compare per entry; an unused entry names
		// position 0, a valid address, and is never summed).  It serves the 64-entry list (64 x 3 coordinates do not fit beside the
		// list) and the waves in which some lane holds fewer than k neighbours (a bounded radius, a tiny cloud).
		auto general = [&]() __attribute__((always_inline)) {
#pragma unroll
			for (int s = 0; s < SK; ++s) {
				if ((uint32_t)s < count) {
					const float4 p = a.pts[key_pos(L[s])];
					mean[0] += p.x;
					mean[1] += p.y;
					mean[2] += p.z;
				}
			}
			mean[0] = mean[0] / len;
			mean[1] = mean[1] / len;
			mean[2] = mean[2] / len;
#pragma unroll
			for (int s = 0; s < SK; ++s) {
				if ((uint32_t)s < count) {
					const float4 p = a.pts[key_pos(L[s])];
					const float df[3] = {p.x - mean[0], p.y - mean[1], p.z - mean[2]};
#pragma unroll
					for (int x = 0; x < 3; ++x)
#pragma unroll
						for (int y = 0; y < 3; ++y) var[x][y] += df[x] * df[y];
				}
			}
		};
		if constexpr (SKT > 32) {
			general();
		} else {
			// FAST form (round 5): the reference's k (31 in the 32-entry list; a full 16-entry list) with every lane of the wave holding
			// k neighbours -- nearly every wave under the unbounded radius: the coordinates gathered once, straight-line code over a
			// compile-time number of entries, no condition anywhere.  (Written with wave-uniform `s < k` conditions per entry, the 32
			// of them were hoisted into SGPR pairs that spilled to VGPR lanes: ~250 v_writelane / v_readlane per wave.)
			constexpr int FASTK = SK == 32 ? 31 : SK;
			if (k == (uint32_t)FASTK && __all(count == k)) {
				float nx[FASTK], ny[FASTK], nz[FASTK];
#pragma unroll
				for (int s = 0; s < FASTK; ++s) {
					const float4 p = a.pts[key_pos(L[s])];
					nx[s] = p.x;
					ny[s] = p.y;
					nz[s] = p.z;
				}
#pragma unroll
				for (int s = 0; s < FASTK; ++s) {
					mean[0] += nx[s];
					mean[1] += ny[s];
					mean[2] += nz[s];
				}
				mean[0] = mean[0] / len;
				mean[1] = mean[1] / len;
				mean[2] = mean[2] / len;
#pragma unroll
				for (int s = 0; s < FASTK; ++s) {
					const float df[3] = {nx[s] - mean[0], ny[s] - mean[1], nz[s] - mean[2]};
#pragma unroll
					for (int x = 0; x < 3; ++x)
#pragma unroll
						for (int y = 0; y < 3; ++y) var[x][y] += df[x] * df[y];
				}
			} else {
				general();
			}
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

template <bool DIAG, bool SEG, int SKT = 32> __global__ __launch_bounds__(64, SKT <= 32 ? TREEE_KNN_WAVES : 2) void k_knn_sorted32(KnnArgs a) {
	knn_wave<DIAG, SEG, SKT, false>(a, (!a.q_xyz && a.wave_order) ? a.wave_order[blockIdx.x] : blockIdx.x);
}
template <bool SEG, int SKT> __device__ __attribute__((noinline)) void knn_wave_overflowed(KnnArgsInKernarg ka, uint32_t wave_id) {
	__syncthreads(); // the prologue below reuses the LDS the caller's BFS was reading
	const KnnArgs a = *(const KnnArgs *)ka;
	knn_wave<false, SEG, SKT, true>(a, wave_id);
}

// start bucket of each query wave: the bucket holding the first query's key (the order of the tree: Hilbert, or Morton with TREEE_KNN_MORTON)
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

int32_t launch_knn_on(treee_ctx *ctx, const treee_knn *t, const float *d_q, uint64_t nq, uint32_t k, float max_dist_sq,
                      float *d_dist, uint32_t *d_index, uint32_t *d_counts, float *d_curve, uint64_t q_limit, uint8_t *d_flags);

// every query path comes through here: a context other than the handle's owner is ordered behind the build and its
// use is remembered for treee_knn_free (HandleSync in common.h)
int32_t launch_knn(treee_ctx *ctx, const treee_knn *t, const float *d_q, uint64_t nq, uint32_t k, float max_dist_sq,
                   float *d_dist, uint32_t *d_index, uint32_t *d_counts, float *d_curve, uint64_t q_limit = ~0ull, uint8_t *d_flags = nullptr) {
	TREEE_TRY(use_begin(ctx, t->owner, &t->sync));
	if (t->sc->owner != t->owner) TREEE_TRY(use_begin(ctx, t->sc->owner, &t->sc->sync));
	const int32_t rc = launch_knn_on(ctx, t, d_q, nq, k, max_dist_sq, d_dist, d_index, d_counts, d_curve, q_limit, d_flags);
	const int32_t e1 = use_end(ctx, t->owner, &t->sync), e2 = use_end(ctx, t->sc->owner, &t->sc->sync);
	return rc != TREEE_OK ? rc : (e1 != TREEE_OK ? e1 : e2);
}

int32_t launch_knn_on(treee_ctx *ctx, const treee_knn *t, const float *d_q, uint64_t nq, uint32_t k, float max_dist_sq,
                      float *d_dist, uint32_t *d_index, uint32_t *d_counts, float *d_curve, uint64_t q_limit, uint8_t *d_flags) {
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0 (the reference panics at best_set.rs:74)");
	if (nq == 0) return TREEE_OK;
	// tie flags need the entry BEHIND the k-th: a list one class longer when k is exactly a list length (16 -> 32, 32 -> 64)
	if (d_flags && (k >= 64u || (t->segmented && k >= 32u)))
		return set_err(TREEE_ERR_UNSUPPORTED, "tie flags need k <= 63 (k <= 31 on a segmented handle): the entry behind the k-th must fit the register-resident list");
	if (nq >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "nq too large");
	if (t->segmented && d_q) return set_err(TREEE_ERR_UNSUPPORTED, "segmented handle: only self queries (a position has no segment)");
	const size_t lds = (size_t)k * 64 * 8 + 64 * 16;
	if (lds > 160 * 1024) return set_err(TREEE_ERR_UNSUPPORTED, "k=%u exceeds the LDS-resident limit (319)", k);
	KnnArgs a;
	a.pts = t->sc->d_sorted;
	a.bvh = t->d_bvh;
	a.sub = t->d_sub;
	a.n = (uint32_t)t->sc->n;
	a.nb = t->nb;
	a.P = t->P;
	a.q_xyz = nullptr;
	a.q_order = nullptr;
	a.q_start = nullptr;
	static const bool no_order = getenv("TREEE_KNN_NO_ORDER") != nullptr; // A/B hook
	a.wave_order = no_order ? nullptr : t->d_order;
	a.nq = (uint32_t)nq;
	a.k = k;
	a.maxbits = max_bits(max_dist_sq);
	a.out_dist = d_dist;
	a.out_index = d_index;
	a.out_count = d_counts;
	a.out_curve = d_curve;
	a.out_flags = d_flags;
	a.stats = nullptr;
	a.thr_io = nullptr;
	a.thr_seed = 0;
	a.q_limit = q_limit >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)q_limit;
	static const bool want_stats = getenv("TREEE_KNN_STATS") != nullptr; // diagnostic counters
	if (want_stats && !t->segmented) {
		TREEE_TRY(ctx->alloc_t(64, &a.stats));
		TREEE_HIP(hipMemsetAsync(a.stats, 0, 256, ctx->stream));
	}
	uint32_t nwaves = div_up(nq, 64);
	if (d_q) { // external queries: Morton-sort them so that a wave is spatially compact
		uint32_t *d_qk = nullptr, *d_qks = nullptr, *d_qo = nullptr, *d_qs = nullptr;
		TREEE_TRY(ctx->alloc_t(nq, &d_qk));
		TREEE_TRY(ctx->alloc_t(nq, &d_qks));
		TREEE_TRY(ctx->alloc_t(nq, &d_qo));
		TREEE_TRY(ctx->alloc_t(nwaves, &d_qs));
		TREEE_TRY(morton_keys_dev(ctx, d_q, nq, t->sc->corner, t->sc->size, d_qk));
		if (t->sc->hilbert) { // the tree is in Hilbert order: so are the queries (in place: one element per thread)
			TREEE_TRY(hilbert_keys_dev(ctx, d_qk, nullptr, nq, d_qk));
		}
		TREEE_TRY(sort_pairs_dev(ctx, d_qk, nullptr, nq, 30, d_qks, d_qo));
		hipLaunchKernelGGL(k_query_start, dim3(div_up(nwaves, 256)), dim3(256), 0, ctx->stream, t->sc->d_keys, a.n, a.nb, d_qks, a.nq, nwaves, d_qs);
		TREEE_LAUNCH_CHECK();
		a.q_xyz = d_q;
		a.q_order = d_qo;
		a.q_start = d_qs;
	}
	static const bool force_heap = getenv("TREEE_KNN_FORCE_HEAP") != nullptr; // A/B and test hook
	static const unsigned pad_lds = getenv("TREEE_KNN_PAD_LDS") ? (unsigned)atoi(getenv("TREEE_KNN_PAD_LDS")) : 0u; // A/B hook: unused dynamic LDS lowers the occupancy
	static const bool no64 = getenv("TREEE_KNN_NO_K64") != nullptr; // A/B and test hook
	if (((k > (uint32_t)SK && k <= 64u) || (d_flags && k == (uint32_t)SK)) && !force_heap && !no64 && !t->segmented) {
		if (a.wave_order && !d_q) nwaves = t->n_order;
		hipLaunchKernelGGL((k_knn_sorted32<false, false, 64>), dim3(nwaves), dim3(64), 0, ctx->stream, a); // 64-entry list, 2 waves/SIMD
	} else if (k <= (uint32_t)SK && !force_heap) {
		if (a.wave_order && !d_q) nwaves = t->n_order; // padded dispatch table
		static const bool no16 = getenv("TREEE_KNN_NO_K16") != nullptr; // A/B and test hook
		if (k <= 16u && !no16 && !a.stats && !(d_flags && k == 16u)) {
			if (t->segmented)
				hipLaunchKernelGGL((k_knn_sorted32<false, true, 16>), dim3(nwaves), dim3(64), 0, ctx->stream, a);
			else
				hipLaunchKernelGGL((k_knn_sorted32<false, false, 16>), dim3(nwaves), dim3(64), 0, ctx->stream, a);
		} else if (t->segmented)
			hipLaunchKernelGGL((k_knn_sorted32<false, true>), dim3(nwaves), dim3(64), 0, ctx->stream, a);
		else if (a.stats)
			hipLaunchKernelGGL((k_knn_sorted32<true, false>), dim3(nwaves), dim3(64), 0, ctx->stream, a);
		else
			hipLaunchKernelGGL((k_knn_sorted32<false, false>), dim3(nwaves), dim3(64), pad_lds, ctx->stream, a);
	} else {
		if (t->segmented) return set_err(TREEE_ERR_UNSUPPORTED, "segmented handle: k=%u > 32 is not supported", k);
		if (d_flags) return set_err(TREEE_ERR_UNSUPPORTED, "tie flags are written by the register-list kernels only (k <= 63)");
		a.wave_order = nullptr;
		if (lds > 64 * 1024)
			TREEE_HIP(hipFuncSetAttribute((const void *)k_knn_heap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL(k_knn_heap, dim3(nwaves), dim3(64), lds, ctx->stream, a);
	}
	TREEE_LAUNCH_CHECK();
	if (a.stats) {
		uint32_t h[64];
		TREEE_HIP(hipMemcpyAsync(h, a.stats, 256, hipMemcpyDeviceToHost, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		fprintf(stderr, "[treee knn phases] kcycles/wave: own+nbr=%.0f bfs=%.0f find_next=%.0f transposed=%.0f fullscan=%.0f gbox=%.0f bucket-wait=%.0f | of which flushes=%.0f | listed buckets mean=%.1f max=%u\n", h[40] * 1.024 / h[3],
		        h[41] * 1.024 / h[3], h[42] * 1.024 / h[3], h[43] * 1.024 / h[3], h[44] * 1.024 / h[3], h[45] * 1.024 / h[3], h[50] * 1.024 / h[3], h[46] * 1.024 / h[3], h[47] / (double)h[3], h[48]);
		fprintf(stderr, "[treee knn cycles] mean=%.0f max=%.0f overflow-waves=%u hist(64k-cycle bins):", h[6] * 1024.0 / h[3], h[7] * 64.0, h[8]);
		for (int i = 0; i < 22; ++i) fprintf(stderr, " %u", h[9 + i]);
		fprintf(stderr, "\n");
		fprintf(stderr, "[treee knn stats] waves=%u buckets/wave=%.2f flushes/wave=%.2f nodes/wave=%.2f need/bucket=%.2f small-buckets/wave=%.2f\n", h[3], h[0] / (double)h[3],
		        h[1] / (double)h[3], h[2] / (double)h[3], h[4] / (double)(h[0] - h[3] ? h[0] - h[3] : 1), h[5] / (double)h[3]);
		static const bool want_ideal = getenv("TREEE_KNN_IDEAL") != nullptr;
		if (want_ideal && k <= (uint32_t)SK && !d_q) { // lower bound: rerun with every threshold seeded by its final value
			TREEE_TRY(ctx->alloc_t(nq, &a.thr_io));
			hipLaunchKernelGGL((k_knn_sorted32<true, false>), dim3(nwaves), dim3(64), 0, ctx->stream, a);
			a.thr_seed = 1;
			TREEE_HIP(hipMemsetAsync(a.stats, 0, 256, ctx->stream));
			hipLaunchKernelGGL((k_knn_sorted32<true, false>), dim3(nwaves), dim3(64), 0, ctx->stream, a);
			TREEE_HIP(hipMemcpyAsync(h, a.stats, 256, hipMemcpyDeviceToHost, ctx->stream));
			TREEE_HIP(hipStreamSynchronize(ctx->stream));
			fprintf(stderr, "[treee knn IDEAL] waves=%u buckets/wave=%.2f flushes/wave=%.2f nodes/wave=%.2f\n", h[3], h[0] / (double)h[3], h[1] / (double)h[3], h[2] / (double)h[3]);
		}
	}
	return TREEE_OK;
}

// dispatch key of a bucket: (XCD label of its Morton range) << 5 | size class, where the size class falls with the
// exponent byte of its squared box diagonal (small key = large box) and ranges of ORDER_RANGE buckets
// are dealt to the 8 XCD labels in turn
// number of buckets with XCD label x, and of those with a smaller label
__device__ __host__ inline uint32_t order_count(uint32_t nb, uint32_t x) {
	const uint32_t nr = nb / ORDER_RANGE, rem = nb % ORDER_RANGE;
	return (nr / 8u + (x < nr % 8u ? 1u : 0u)) * ORDER_RANGE + (x == nr % 8u ? rem : 0u);
}

// table[slot * 8 + x] = slot-th bucket of label x in (class, Morton) order: the hardware deals consecutive
// workgroups to the XCDs in turn, so the workgroups of one XCD walk one label
__global__ __launch_bounds__(256) void k_order_interleave(const uint32_t *__restrict__ sorted, uint32_t nb, uint32_t len, uint32_t *__restrict__ table) {
	const uint32_t p = blockIdx.x * 256u + threadIdx.x;
	if (p >= len) return;
	const uint32_t x = p & 7u, slot = p >> 3;
	uint32_t start = 0;
	for (uint32_t i = 0; i < x; ++i) start += order_count(nb, i);
	table[p] = slot < order_count(nb, x) ? sorted[start + slot] : ~0u;
}


// ---- dense cells: a finer order inside Morton cells that hold many points -------------------------------------
// The shared sort uses 10 bits per axis.  A cloud whose density varies by orders of magnitude (a scanner standing
// next to a wall, a blob of returns) puts thousands of points into one cell; inside a cell the order is the
// original index, so every bucket cut from it spans the whole cell and a query there has to scan them all.  When
// a cell holds more than LONG_RUN points the kNN structure therefore re-sorts its own copy of the cloud by a
// 20-bit-per-axis key (low 10 bits first, then stable by the existing key).  The octree keeps the 30-bit order.
__device__ __forceinline__ uint32_t quant_lo10(float p, float c, float size) { // bits 9..0 of the 20-bit cell index
	const float t = ((p - c) / size) * 1048576.0f; // same quotient as the 10-bit key, scaled by a power of two
	if (!(t >= 0.0f)) return 0u;
	if (t >= 1048576.0f) return 1023u;
	return (uint32_t)t & 1023u;
}
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
	v &= 0x3FFu;
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
// `hil`: the low key as the continuation of the cell's Hilbert index (hilbert_low_of_morton_lut) instead of the low Morton key
__global__ __launch_bounds__(256) void k_fine_keys(const float4 *__restrict__ rec, uint32_t n, float cx, float cy, float cz, float size, bool hil,
                                                    uint32_t *__restrict__ lo) {
	__shared__ uint16_t s_lut[24 * 8];
	if (threadIdx.x < 24 * 8) {
		constexpr uint16_t LUT[24 * 8] = {TREEE_HILBERT_LUT_INIT};
		s_lut[threadIdx.x] = LUT[threadIdx.x];
	}
	__syncthreads();
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float4 p = rec[i];
	const uint32_t m = spread10(quant_lo10(p.x, cx, size)) | (spread10(quant_lo10(p.y, cy, size)) << 1) | (spread10(quant_lo10(p.z, cz, size)) << 2);
	lo[i] = hil ? hilbert_low_of_morton_lut(morton30(p.x, p.y, p.z, cx, cy, cz, size), m, s_lut) : m;
}
__global__ __launch_bounds__(256) void k_fine_keys_aos(const float *__restrict__ xyz, uint32_t n, float cx, float cy, float cz, float size, bool hil,
                                                        uint32_t *__restrict__ lo) {
	__shared__ uint16_t s_lut[24 * 8];
	if (threadIdx.x < 24 * 8) {
		constexpr uint16_t LUT[24 * 8] = {TREEE_HILBERT_LUT_INIT};
		s_lut[threadIdx.x] = LUT[threadIdx.x];
	}
	__syncthreads();
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float x = xyz[3ull * i + 0], y = xyz[3ull * i + 1], z = xyz[3ull * i + 2];
	const uint32_t m = spread10(quant_lo10(x, cx, size)) | (spread10(quant_lo10(y, cy, size)) << 1) | (spread10(quant_lo10(z, cz, size)) << 2);
	lo[i] = hil ? hilbert_low_of_morton_lut(morton30(x, y, z, cx, cy, cz, size), m, s_lut) : m;
}
__global__ __launch_bounds__(256) void k_gather_u32(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ dst) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_gather_rec(const float4 *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, float4 *__restrict__ dst,
                                                     uint32_t *__restrict__ perm) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float4 r = src[idx[i]];
	dst[i] = r;
	perm[i] = __float_as_uint(r.w);
}

// records of a shard's kNN structure: position p of the concatenation [the rank's own Morton-sorted records | its ghosts in
// input order] -> (x, y, z, LOCAL index): an own record carries index_base + input position, ghost j is point n_own + j
__global__ __launch_bounds__(256) void k_gather_own_ghost(const float4 *__restrict__ own, uint32_t n_own, uint32_t index_base, const float *__restrict__ ghost_xyz,
                                                           const uint32_t *__restrict__ pos, uint32_t m, float4 *__restrict__ dst, uint32_t *__restrict__ perm) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= m) return;
	const uint32_t p = pos[i];
	float4 r;
	if (p < n_own) {
		r = own[p];
		r.w = __uint_as_float(__float_as_uint(r.w) - index_base);
	} else {
		const uint32_t j = p - n_own;
		r = make_float4(ghost_xyz[3ull * j + 0], ghost_xyz[3ull * j + 1], ghost_xyz[3ull * j + 2], __uint_as_float(p));
	}
	dst[i] = r;
	perm[i] = __float_as_uint(r.w);
}

// ---- Hilbert order of a Morton-sorted cloud by ordering CELLS instead of points ------------------------------------------------
// The points of one 2^-10 cell share their Morton key and their Hilbert index, are contiguous in the Morton order and keep
// their relative order (both sorts are stable by original index).  So the Hilbert order of the cloud is: the occupied cells
// in the order of their Hilbert index, each with its points as they stand.  A forest scan puts ~10 points into an occupied
// cell, so sorting the K cells is a tenth of sorting the n points; the rest is streaming: head flags + scan (the cell of
// every point), one key per cell, the cells' sizes scanned in Hilbert order, and one pass that moves every record to
// cell destination + position inside the cell -- which also replaces the record gather (runs of a cell's records are
// WRITTEN contiguously instead of single records being read from all over the cloud).  Same result as sort_pairs_dev on the
// points' Hilbert indices, bit for bit; taken when a sample of the keys promises K <= 0.4 n and the exact count confirms
// K <= n / 2 (the 15 M / 200 m benchmark cloud has K ~ 0.8 n -- crowns are volumes -- and keeps the point sort; the 150 M /
// 632 m one, whose cells are 3 times larger, takes this path).
__global__ __launch_bounds__(256) void k_cell_heads(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ flag) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_cell_emit(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ cid_ex,
                                                    uint32_t n, uint32_t n_cells, uint32_t *__restrict__ cell_start, uint32_t *__restrict__ cell_hkey) {
	__shared__ uint16_t s_lut[24 * 8];
	if (threadIdx.x < 24 * 8) {
		constexpr uint16_t LUT[24 * 8] = {TREEE_HILBERT_LUT_INIT};
		s_lut[threadIdx.x] = LUT[threadIdx.x];
	}
	__syncthreads();
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i == 0) cell_start[n_cells] = n;
	if (i >= n || !flag[i]) return;
	const uint32_t c = cid_ex[i];
	cell_start[c] = i;
	cell_hkey[c] = hilbert_of_morton_lut(keys[i], s_lut);
}
__global__ __launch_bounds__(256) void k_cell_len(const uint32_t *__restrict__ order, const uint32_t *__restrict__ cell_start, uint32_t n_cells,
                                                   uint32_t *__restrict__ len_sorted) {
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= n_cells) return;
	const uint32_t c = order[j];
	len_sorted[j] = cell_start[c + 1u] - cell_start[c];
}
__global__ __launch_bounds__(256) void k_cell_dest(const uint32_t *__restrict__ order, const uint32_t *__restrict__ dest_sorted, uint32_t n_cells,
                                                    uint32_t *__restrict__ cell_dest) {
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j < n_cells) cell_dest[order[j]] = dest_sorted[j];
}
// every record to its cell's destination + its position inside the cell
__global__ __launch_bounds__(256) void k_cell_expand(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ cid_ex,
                                                      const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ cell_dest,
                                                      const uint32_t *__restrict__ cell_hkey, const float4 *__restrict__ rec, uint32_t n,
                                                      uint32_t *__restrict__ out_keys, uint32_t *__restrict__ out_perm, float4 *__restrict__ out_rec) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t c = cid_ex[i] + flag[i] - 1u;
	const uint32_t d = cell_dest[c] + (i - cell_start[c]);
	const float4 r = rec[i];
	out_keys[d] = cell_hkey[c];
	out_perm[d] = __float_as_uint(r.w);
	out_rec[d] = r;
}

// *taken = false (and nothing written) when the exact count shows more than n / 2 occupied cells after all
int32_t hilbert_order_by_cells(treee_ctx *ctx, const SortedCloud *sc, SortedCloud *fine, bool *taken) {
	const uint32_t n = (uint32_t)sc->n;
	hipStream_t st = ctx->stream;
	uint32_t *d_flag = nullptr, *d_cid = nullptr, *d_start = nullptr, *d_hkey = nullptr, *d_hsorted = nullptr, *d_order = nullptr, *d_len = nullptr,
	         *d_dsorted = nullptr, *d_dest = nullptr, *d_total = nullptr;
	*taken = false;
	TREEE_TRY(ctx->alloc_t(n, &d_flag));
	TREEE_TRY(ctx->alloc_t(n, &d_cid));
	TREEE_TRY(ctx->alloc_t(1, &d_total));
	hipLaunchKernelGGL(k_cell_heads, dim3(div_up(n, 256)), dim3(256), 0, st, sc->d_keys, n, d_flag);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_flag, d_cid, n, d_total));
	// the number of cells sizes their sort: one host round trip of this path's own (it saves milliseconds where it is taken)
	uint32_t *h = ctx->h_scalars;
	TREEE_HIP(hipMemcpyAsync(h + 12, d_total, 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	const uint32_t n_cells = h[12];
	if (n_cells == 0 || (uint64_t)n_cells * 2u > n) return TREEE_OK;
	*taken = true;
	TREEE_TRY(ctx->alloc_t((size_t)n_cells + 1, &d_start));
	TREEE_TRY(ctx->alloc_t(n_cells, &d_hkey));
	TREEE_TRY(ctx->alloc_t(n_cells, &d_hsorted));
	TREEE_TRY(ctx->alloc_t(n_cells, &d_order));
	TREEE_TRY(ctx->alloc_t(n_cells, &d_len));
	TREEE_TRY(ctx->alloc_t(n_cells, &d_dsorted));
	TREEE_TRY(ctx->alloc_t(n_cells, &d_dest));
	hipLaunchKernelGGL(k_cell_emit, dim3(div_up(n, 256)), dim3(256), 0, st, sc->d_keys, d_flag, d_cid, n, n_cells, d_start, d_hkey);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(sort_pairs_dev(ctx, d_hkey, nullptr, n_cells, 30, d_hsorted, d_order)); // (distinct keys: no ties to keep stable)
	hipLaunchKernelGGL(k_cell_len, dim3(div_up(n_cells, 256)), dim3(256), 0, st, d_order, d_start, n_cells, d_len);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_len, d_dsorted, n_cells, nullptr));
	hipLaunchKernelGGL(k_cell_dest, dim3(div_up(n_cells, 256)), dim3(256), 0, st, d_order, d_dsorted, n_cells, d_dest);
	hipLaunchKernelGGL(k_cell_expand, dim3(div_up(n, 256)), dim3(256), 0, st, d_flag, d_cid, d_start, d_dest, d_hkey, sc->d_sorted, n, fine->d_keys,
	                   fine->d_perm, fine->d_sorted);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

// ---- Hilbert order of a Morton-sorted cloud by sorting TILES OF WHOLE BLOCKS in LDS (round 4) -----------------------------------
// The Hilbert curve and the Z curve visit the same octants at every level, only in another child order: the first 3 j bits of
// a point's Hilbert index are a function of the first 3 j bits of its Morton key.  So the points of one 2^-7 block (keys
// sharing their top 21 bits) are contiguous in BOTH orders, and re-sorting a Morton-sorted cloud by Hilbert index is
//   (1) a permutation of the blocks -- where block m starts in the Hilbert order is the number of points in the blocks with a
//       smaller 21-bit Hilbert prefix: a scan over a dense table of 2^21 block sizes, read in Hilbert order --, and
//   (2) inside every block a stable sort by the low 9 bits of the Hilbert index (three levels).
// A workgroup takes the whole blocks that start inside its tile of BLK_TILE positions (blocks hold fewer than BLOCK7_MAX points
// -- the build's existing round trip asked, bit 1 of long_run_flag_dev --, so at most BLK_CAP), sorts the composite keys
// (block offset in the tile, low Hilbert bits, position in the tile) -- all distinct -- with a bitonic network in LDS and
// writes keys, permutation and records to block start + rank.  Bytes per point: 4 (key) + 16 (record) read, 24 written, against
// 8 (Hilbert conversion) + 4 x 16 + 4 (four radix passes and their histograms) + 16 + 24 (record gather) of the point sort
// it replaces; same arrays, bit for bit (test_knn_structure_from_an_octree_orders_cells_like_the_point_sort).  15 M points /
// 200 m: structure stage 1.58 -> see profiles/README.md.  Clouds with a fuller block keep the other paths.
constexpr uint32_t BLK_BITS = 21, BLK_CELLS = 1u << BLK_BITS, BLK_TILE = 1536, BLK_CAP = 2048, BLK_CAP_BIG = 4096;
// (tiles whose blocks do not fit BLK_CAP -- a block of more than 512 points at their edge: 25 of 157 000 blocks of the 15 M-point
// forest -- are listed and sorted by a second launch with twice the LDS; a block of BLOCK7_MAX points fits neither)
static_assert(BLK_TILE + BLOCK7_MAX <= BLK_CAP_BIG && BLOCK7_MAX <= 2048, "a tile's whole blocks must fit the larger LDS sort, a position in a block 11 bits");

// start7[m] / end7[m]: the sorted positions [start, end) of the points whose key >> 9 == m (both tables zeroed before)
// (four consecutive keys per thread, one 16-byte load: with one key per thread the launch was bound by its 58 000 workgroups,
// 47 us for 60 MB)
constexpr uint32_t BOUNDS_PER_THREAD = 4;
__global__ __launch_bounds__(256) void k_block_bounds(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ start7, uint32_t *__restrict__ end7) {
	const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * BOUNDS_PER_THREAD;
	if (i0 >= n) return;
	uint32_t k[BOUNDS_PER_THREAD + 2]; // [0] = the key before, [5] = the key behind
	if (i0 + BOUNDS_PER_THREAD <= n && ((uintptr_t)keys & 15u) == 0u) { // (pool blocks are 256-byte aligned, i0 is a multiple of four)
		const uint4 v = *(const uint4 *)(keys + i0);
		k[1] = v.x, k[2] = v.y, k[3] = v.z, k[4] = v.w;
	} else {
#pragma unroll
		for (uint32_t j = 0; j < BOUNDS_PER_THREAD; ++j) k[1 + j] = i0 + j < n ? keys[i0 + j] : 0xFFFFFFFFu;
	}
	k[0] = i0 ? keys[i0 - 1u] : 0xFFFFFFFFu;
	k[BOUNDS_PER_THREAD + 1] = i0 + BOUNDS_PER_THREAD < n ? keys[i0 + BOUNDS_PER_THREAD] : 0xFFFFFFFFu;
#pragma unroll
	for (uint32_t j = 0; j < BOUNDS_PER_THREAD; ++j) {
		const uint32_t i = i0 + j;
		if (i >= n) break;
		const uint32_t m = k[1 + j] >> 9;
		if (i == 0 || (k[j] >> 9) != m) start7[m] = i;
		if (i + 1 == n || (k[2 + j] >> 9) != m) end7[m] = i + 1u;
	}
}
// block sizes in the order of the blocks' 21-bit Hilbert prefixes
// (and zeroes the counter of the tiles left to the second sort launch: a memset of four bytes is a launch of its own)
__global__ __launch_bounds__(256) void k_block_sizes_hilbert(const uint32_t *__restrict__ start7, const uint32_t *__restrict__ end7, uint32_t *__restrict__ size_h,
                                                              uint32_t *__restrict__ big_count) {
	__shared__ uint16_t s_lut[24 * 8];
	if (threadIdx.x < 24 * 8) {
		constexpr uint16_t LUT[24 * 8] = {TREEE_HILBERT_LUT_INIT};
		s_lut[threadIdx.x] = LUT[threadIdx.x];
	}
	__syncthreads();
	const uint32_t m = blockIdx.x * 256u + threadIdx.x; // (the grid covers the table exactly)
	if (m == 0) *big_count = 0u;
	size_h[hilbert_of_morton_lut(m << 9, s_lut) >> 9] = end7[m] - start7[m];
}
// stage (K, J) of the bitonic network over CAP keys, EPT consecutive ones per thread, and all the stages behind it (the
// recursion is the compile-time loop nest: every stage has its own code)
template <uint32_t CAP, uint32_t EPT, uint32_t K, uint32_t J> __device__ __forceinline__ void bitonic_stages(uint32_t (&v)[EPT], const uint32_t t, uint32_t *s_x) {
	if constexpr (J < EPT) { // inside the thread
#pragma unroll
		for (uint32_t i = 0; i < EPT; ++i) {
			const uint32_t l = i ^ J;
			if (l > i) {
				const bool up = K >= EPT ? ((t * EPT) & K) == 0u : (i & K) == 0u;
				const uint32_t x = v[i], y = v[l], mn = min(x, y), mx = max(x, y);
				v[i] = up ? mn : mx;
				v[l] = up ? mx : mn;
			}
		}
	} else { // a whole register set against the one of thread t ^ (J / EPT)
		constexpr uint32_t S = J / EPT;
		const bool keep_min = (((t & S) == 0u) == (((t * EPT) & K) == 0u));
		if constexpr (S < 64u) {
#pragma unroll
			for (uint32_t i = 0; i < EPT; ++i) v[i] = lane_xor_minmax<S>(v[i], keep_min);
		} else {
			__syncthreads();
#pragma unroll
			for (uint32_t i = 0; i < EPT; ++i) s_x[i * 256u + t] = v[i];
			__syncthreads();
#pragma unroll
			for (uint32_t i = 0; i < EPT; ++i) {
				const uint32_t p = s_x[i * 256u + (t ^ S)];
				v[i] = keep_min ? min(v[i], p) : max(v[i], p);
			}
		}
	}
	if constexpr (J > 1u)
		bitonic_stages<CAP, EPT, K, J / 2u>(v, t, s_x);
	else if constexpr (K < CAP)
		bitonic_stages<CAP, EPT, 2u * K, K>(v, t, s_x);
}

// CAP = BLK_CAP: every tile, one workgroup each; a tile that does not fit is appended to big_list ([0] = count) instead.
// CAP = BLK_CAP_BIG: the listed tiles, a workgroup taking every gridDim.x-th of them.
// The bitonic network runs on EPT = CAP / 256 consecutive keys per thread held in registers: strides below EPT are compare-
// exchanges inside a thread, strides up to 32 EPT exchange whole register sets between lanes of a wave (xor shuffles), and only
// the last strides (two or three stages of the whole network) go through LDS and a barrier.  (The first version -- every stage
// through LDS, 66 barriers -- took 357 us per 15 M points; the radix passes it replaces take 500.)
template <uint32_t CAP> __global__ __launch_bounds__(256) void k_block_sort(const uint32_t *__restrict__ keys, const float4 *__restrict__ rec, uint32_t n,
                                                     const uint32_t *__restrict__ start7, const uint32_t *__restrict__ end7, const uint32_t *__restrict__ dest_h,
                                                     uint32_t *__restrict__ out_keys, uint32_t *__restrict__ out_perm, float4 *__restrict__ out_rec,
                                                     uint32_t *__restrict__ big_list) {
	constexpr uint32_t EPT = CAP / 256u;
	__shared__ uint16_t s_lut[24 * 8];
	__shared__ uint32_t s_x[CAP];     // exchange buffer of the widest strides, [register][thread]
	__shared__ uint32_t s_hil[CAP];   // Hilbert index of the element at a position in the tile
	__shared__ uint32_t s_base[CAP];  // destination of its block minus the block's offset in the tile
	if (threadIdx.x < 24 * 8) {
		constexpr uint16_t LUT[24 * 8] = {TREEE_HILBERT_LUT_INIT};
		s_lut[threadIdx.x] = LUT[threadIdx.x];
	}
	const uint32_t t = threadIdx.x;
	for (uint32_t turn = blockIdx.x;; turn += gridDim.x) {
		uint32_t tile = turn;
		if constexpr (CAP != BLK_CAP) {
			if (turn >= big_list[0]) return; // (uniform)
			tile = big_list[1u + turn];
		}
		// the whole blocks that START in [a, b): a block that straddles a belongs to the tile before, one that straddles b to this one
		const uint32_t a = tile * BLK_TILE, b = min(n, a + BLK_TILE);
		uint32_t lo = a, hi = b;
		{
			const uint32_t m = keys[a] >> 9;
			if (start7[m] != a) lo = end7[m];
			if (b < n) {
				const uint32_t mb = keys[b] >> 9;
				if (start7[mb] != b) hi = end7[mb];
			}
		}
		const uint32_t n_loc = hi > lo ? hi - lo : 0u; // (<= BLK_CAP_BIG: every block is shorter than BLOCK7_MAX)
		if constexpr (CAP == BLK_CAP) {
			if (n_loc > CAP) { // (uniform) left to the second launch
				if (t == 0) big_list[1u + atomicAdd(&big_list[0], 1u)] = tile;
				return;
			}
		}
		__syncthreads(); // the table is there; the arrays of the tile before are free
		// composite keys, all distinct: block's offset in the tile << 20 | low 9 Hilbert bits << 11 | position in the block (< 2^11).
		// Global memory is touched lane-contiguously (element i * 256 + t); the network wants EPT consecutive keys per thread:
		// the keys pass through s_x on the way in and on the way out.
#pragma unroll
		for (uint32_t i = 0; i < EPT; ++i) {
			const uint32_t e = i * 256u + t;
			uint32_t sk = 0xFFFFFFFFu;
			if (e < n_loc) {
				const uint32_t key = keys[lo + e], m = key >> 9;
				const uint32_t h = hilbert_of_morton_lut(key, s_lut), brel = start7[m] - lo;
				s_hil[e] = h;
				s_base[e] = dest_h[h >> 9] - brel;
				sk = (brel << 20) | ((h & 511u) << 11) | (e - brel);
			}
			s_x[e] = sk;
		}
		__syncthreads();
		uint32_t v[EPT];
#pragma unroll
		for (uint32_t i = 0; i < EPT; ++i) v[i] = s_x[t * EPT + i];
		bitonic_stages<CAP, EPT, 2u, 1u>(v, t, s_x);
		__syncthreads();
#pragma unroll
		for (uint32_t i = 0; i < EPT; ++i) s_x[t * EPT + i] = v[i];
		__syncthreads();
		// slot sl of the sorted tile: 64 consecutive slots per wave instruction -- their records lie within one or two blocks of
		// the input (a window of a few KiB), their destinations are consecutive
#pragma unroll
		for (uint32_t i = 0; i < EPT; ++i) {
			const uint32_t sl = i * 256u + t;
			if (sl < n_loc) {
				const uint32_t sk = s_x[sl], e = (sk >> 20) + (sk & 2047u);
				const uint32_t g = s_base[e] + sl; // block destination + (slot - block offset): the block's slots are [offset, offset + size)
				const float4 r = rec[lo + e];
				out_keys[g] = s_hil[e];
				out_perm[g] = __float_as_uint(r.w);
				out_rec[g] = r;
			}
		}
		if constexpr (CAP == BLK_CAP) return;
	}
}

// the three tables the tile sort reads -- where a 2^-7 block starts and ends in the Morton order, where it starts in the
// Hilbert order -- from the sorted KEYS alone (so they can be made beside the record gather: treee_sort_build_knn_dev)
static bool blocks_pay(uint64_t n) { return n >= 4u * BLK_CELLS / 8u; } // (below ~1 M points the 2^21-entry tables cost more than the radix passes)
int32_t block_tables_enqueue(treee_ctx *ctx, const SortedCloud *sc, BlockTables *bt) {
	const uint32_t n = (uint32_t)sc->n;
	hipStream_t st = ctx->stream;
	uint32_t *d_size_h = nullptr;
	TREEE_TRY(ctx->alloc_t(BLK_CELLS, &bt->d_start));
	TREEE_TRY(ctx->alloc_t(BLK_CELLS, &bt->d_end));
	TREEE_TRY(ctx->alloc_t(BLK_CELLS, &d_size_h));
	TREEE_TRY(ctx->alloc_t(BLK_CELLS, &bt->d_dest_h));
	TREEE_HIP(hipMemsetAsync(bt->d_start, 0, (size_t)BLK_CELLS * 4, st));
	TREEE_HIP(hipMemsetAsync(bt->d_end, 0, (size_t)BLK_CELLS * 4, st));
	hipLaunchKernelGGL(k_block_bounds, dim3(div_up(n, 256 * BOUNDS_PER_THREAD)), dim3(256), 0, st, sc->d_keys, n, bt->d_start, bt->d_end);
	TREEE_TRY(ctx->alloc_t((size_t)div_up(n, BLK_TILE) + 1, &bt->d_big)); // [0] = number of tiles left to the second launch, then their indices
	hipLaunchKernelGGL(k_block_sizes_hilbert, dim3(BLK_CELLS / 256), dim3(256), 0, st, bt->d_start, bt->d_end, d_size_h, bt->d_big);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_size_h, bt->d_dest_h, BLK_CELLS, nullptr));
	bt->made = true;
	return TREEE_OK;
}

// *taken = false (nothing written) when the cloud is too small for the tables to pay; `pre`: tables made already
int32_t hilbert_order_by_blocks(treee_ctx *ctx, const SortedCloud *sc, SortedCloud *fine, bool *taken, const BlockTables *pre) {
	const uint32_t n = (uint32_t)sc->n;
	*taken = false;
	if (!blocks_pay(n)) return TREEE_OK;
	hipStream_t st = ctx->stream;
	BlockTables own;
	if (!pre || !pre->made) {
		TREEE_TRY(block_tables_enqueue(ctx, sc, &own));
		pre = &own;
	}
	uint32_t *const d_start = pre->d_start, *const d_end = pre->d_end, *const d_dest_h = pre->d_dest_h;
	const uint32_t n_tiles = div_up(n, BLK_TILE);
	uint32_t *const d_big = pre->d_big;
	hipLaunchKernelGGL(k_block_sort<BLK_CAP>, dim3(n_tiles), dim3(256), 0, st, sc->d_keys, sc->d_sorted, n, d_start, d_end, d_dest_h, fine->d_keys, fine->d_perm,
	                   fine->d_sorted, d_big);
	hipLaunchKernelGGL(k_block_sort<BLK_CAP_BIG>, dim3(std::min(n_tiles, 512u)), dim3(256), 0, st, sc->d_keys, sc->d_sorted, n, d_start, d_end, d_dest_h, fine->d_keys,
	                   fine->d_perm, fine->d_sorted, d_big);
	TREEE_LAUNCH_CHECK();
	*taken = true;
	return TREEE_OK;
}

// The kNN structure's own order.  The shared sort is by Morton key (it is also the octree's); the kNN structure re-sorts a
// private copy of the cloud by the HILBERT index of the same 10-bit cells: a bucket of 64 consecutive points is then one
// contiguous blob instead of, every so often, two pieces on either side of a jump of the Z curve, and its queries share
// their neighbourhoods -- 22.4 instead of 33.7 buckets scanned per wave, 8.7 instead of 11.5 ms per 15 M points; the second
// sort costs ~0.6 ms.  Where a cell is crowded (> LONG_RUN points: see above) the copy is ordered by the 20-bit-per-axis
// low bits inside the cell as before.  Results do not depend on the order (exact search); only the tie rule names it.
// A cloud sorted for this structure alone (treee_knn_build_dev) is sorted by the Hilbert index in the first place
// (build_sorted_cloud's hilbert_order: one sort), and comes through here only for the crowded-cell case.
int32_t refine_dense_cells(treee_ctx *ctx, const std::shared_ptr<SortedCloud> &sc, std::shared_ptr<SortedCloud> *out, const BlockTables *pre = nullptr) {
	*out = sc;
	const uint64_t n = sc->n;
	static const bool off = getenv("TREEE_KNN_NO_REFINE") != nullptr; // A/B and test hook
	if (off) return cube_to_host(ctx, sc.get());
	hipStream_t st = ctx->stream;
	uint32_t flag = 0;
	if (n <= LONG_RUN) { // no cell can be crowded; the structure still gets ITS order (the tie rule names the Hilbert index:
		TREEE_TRY(cube_to_host(ctx, sc.get())); // a small cloud must not answer ties differently when it comes through an octree)
	} else if (sc->crowded >= 0) { // the octree build already asked (its own host round trip carried the answer)
		flag = (uint32_t)sc->crowded;
	} else {
		uint32_t *d_flag = nullptr;
		TREEE_TRY(ctx->alloc_t(1, &d_flag));
		TREEE_TRY(long_run_flag_dev(ctx, sc->d_keys, n, d_flag));
		uint32_t *h = ctx->h_scalars; // pinned, owned by the context: an early return cannot leave a copy aimed at a dead frame
		float *h_cube = (float *)(h + 4);
		TREEE_HIP(hipMemcpyAsync(h, d_flag, 4, hipMemcpyDeviceToHost, st));
		TREEE_TRY(cube_fetch_async(ctx, sc.get(), h_cube));
		TREEE_HIP(hipStreamSynchronize(st));
		flag = h[0] & 1u;
		sc->crowded = (int)flag;
		sc->block7_big = (int)((h[0] >> 1) & 1u);
		cube_commit(sc.get(), h_cube);
	}
	static const bool morton_order = getenv("TREEE_KNN_MORTON") != nullptr; // A/B hook: keep the shared Morton order
	if (!flag && (morton_order || sc->hilbert)) return TREEE_OK; // (a cloud sorted for this structure is in its order already)
	auto fine = std::make_shared<SortedCloud>();
	fine->n = n;
	fine->device = sc->device;
	fine->owner = ctx;
	fine->size = sc->size;
	fine->hilbert = !morton_order || sc->hilbert;
	for (int a = 0; a < 3; ++a) fine->corner[a] = sc->corner[a];
	TREEE_TRY(pool_alloc_t(ctx, n, &fine->d_keys));
	TREEE_TRY(pool_alloc_t(ctx, n, &fine->d_perm));
	TREEE_TRY(pool_alloc_t(ctx, n, &fine->d_sorted));
	static const bool no_cells = getenv("TREEE_KNN_NO_CELLSORT") != nullptr; // A/B and test hook: always sort the points
	if (!flag && !morton_order && !sc->hilbert && !no_cells && sc->cell_share >= 0.0f && sc->cell_share <= 0.4f) {
		bool taken = false;
		TREEE_TRY(hilbert_order_by_cells(ctx, sc.get(), fine.get(), &taken));
		if (taken) {
			*out = fine;
			return TREEE_OK;
		}
	}
	static const bool no_blocks = getenv("TREEE_KNN_NO_BLOCKSORT") != nullptr; // A/B and test hook: always the radix passes
	static const bool trace = getenv("TREEE_KNN_TRACE") != nullptr;
	if (trace) fprintf(stderr, "[treee knn order] n=%llu crowded=%d block7_big=%d cell_share=%.3f hilbert=%d\n", (unsigned long long)n, sc->crowded, sc->block7_big, sc->cell_share, (int)sc->hilbert);
	if (!flag && !morton_order && !sc->hilbert && !no_blocks && sc->block7_big == 0) {
		bool taken = false;
		TREEE_TRY(hilbert_order_by_blocks(ctx, sc.get(), fine.get(), &taken, pre));
		if (taken) {
			*out = fine;
			return TREEE_OK;
		}
	}
	uint32_t *d_lo = nullptr, *d_lo_s = nullptr, *d_pos1 = nullptr, *d_hi = nullptr, *d_pos2 = nullptr;
	TREEE_TRY(ctx->alloc_t(n, &d_hi));
	TREEE_TRY(ctx->alloc_t(n, &d_pos2));
	const uint32_t *d_first = nullptr; // order inside a cell: the low bits where cells are crowded, else the shared order
	if (flag) {
		TREEE_TRY(ctx->alloc_t(n, &d_lo));
		TREEE_TRY(ctx->alloc_t(n, &d_lo_s));
		TREEE_TRY(ctx->alloc_t(n, &d_pos1));
		static const bool fine_morton = getenv("TREEE_KNN_FINE_MORTON") != nullptr; // A/B hook: the low Morton key of rounds 1-4
		hipLaunchKernelGGL(k_fine_keys, dim3(div_up(n, 256)), dim3(256), 0, st, sc->d_sorted, (uint32_t)n, sc->corner[0], sc->corner[1], sc->corner[2],
		                   sc->size, !fine_morton && !morton_order, d_lo);
		TREEE_LAUNCH_CHECK();
		TREEE_TRY(sort_pairs_dev(ctx, d_lo, nullptr, n, 30, d_lo_s, d_pos1));
		d_first = d_pos1;
	}
	if (morton_order || sc->hilbert) { // the keys are what this structure sorts by: only carried along
		hipLaunchKernelGGL(k_gather_u32, dim3(div_up(n, 256)), dim3(256), 0, st, sc->d_keys, d_pos1, (uint32_t)n, d_hi);
		TREEE_LAUNCH_CHECK();
	} else {
		TREEE_TRY(hilbert_keys_dev(ctx, sc->d_keys, d_first, n, d_hi));
	}
	TREEE_TRY(sort_pairs_dev(ctx, d_hi, d_first, n, 30, fine->d_keys, d_pos2)); // stable: keeps the order inside a cell
	hipLaunchKernelGGL(k_gather_rec, dim3(div_up(n, 256)), dim3(256), 0, st, sc->d_sorted, d_pos2, (uint32_t)n, fine->d_sorted, fine->d_perm);
	TREEE_LAUNCH_CHECK();
	*out = fine;
	return TREEE_OK;
}

// ---- segmented clouds: one logical KD-tree per segment (calculations.rs:97-118,329) in one structure ----------
// segment id of every point in Morton order; ids outside [0, nseg) raise the error flag
__global__ __launch_bounds__(256) void k_seg_gather(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ perm, uint32_t n, uint32_t nseg,
                                                     uint32_t *__restrict__ out, uint32_t *__restrict__ err) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	uint32_t sg = seg[perm[i]];
	if (sg >= nseg) {
		atomicOr(err, 1u);
		sg = nseg - 1u;
	}
	out[i] = sg;
}

// start[s] = first position of segment s in the (segment, Morton)-sorted order, start[nseg] = n;
// nb[s] = its number of 64-point buckets
__global__ __launch_bounds__(256) void k_seg_starts(const uint32_t *__restrict__ sorted_seg, uint32_t n, uint32_t nseg, uint32_t *__restrict__ start) {
	const uint32_t sg = blockIdx.x * 256u + threadIdx.x;
	if (sg > nseg) return;
	uint32_t lo = 0, hi = n;
	while (lo < hi) { // first position with segment >= sg
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (sorted_seg[mid] < sg)
			lo = mid + 1;
		else
			hi = mid;
	}
	start[sg] = lo;
}

__global__ __launch_bounds__(256) void k_seg_buckets(const uint32_t *__restrict__ start, uint32_t nseg, uint32_t *__restrict__ nb) {
	const uint32_t sg = blockIdx.x * 256u + threadIdx.x;
	if (sg >= nseg) return;
	nb[sg] = (start[sg + 1] - start[sg] + 63u) / 64u;
}

// records into the padded layout: segment s owns buckets [base[s], base[s] + nb[s]); unused slots keep the
// (NaN, NaN, NaN, ~0u) fill
__global__ __launch_bounds__(256) void k_seg_scatter(const float *__restrict__ xyz, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ sorted_seg,
                                                      const uint32_t *__restrict__ start, const uint32_t *__restrict__ base, uint32_t n,
                                                      float4 *__restrict__ rec, uint32_t *__restrict__ bucket_seg) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t sg = sorted_seg[i], o = perm[i];
	const uint32_t dst = base[sg] * 64u + (i - start[sg]);
	rec[dst] = make_float4(xyz[3ull * o + 0], xyz[3ull * o + 1], xyz[3ull * o + 2], __uint_as_float(o));
	if ((dst & 63u) == 0u) bucket_seg[dst >> 6] = sg;
}

int32_t build_bvh(treee_ctx *ctx, treee_knn *t, const uint32_t *d_bucket_seg = nullptr) {
	const uint64_t n = t->sc->n;
	t->nb = div_up(n, BUCKET);
	t->P = 1;
	while (t->P < t->nb) t->P <<= 1;
	t->owner = ctx;
	TREEE_TRY(pool_alloc_t(ctx, (size_t)t->P * 2 * 2, &t->d_bvh));
	TREEE_TRY(pool_alloc_t(ctx, (size_t)t->nb * 8, &t->d_sub));
	// dispatch table of the self-query kernel: a stable radix sort of the buckets by (XCD label, size class) -- an 8-bit key,
	// one pass --, then the 8 label segments interleaved.  (On the context's auxiliary stream beside the inner levels it was
	// measured 15 us SLOWER per 15 M-point build than behind them: the fork / join events cost more than the two short chains overlap.)
	uint32_t *d_key = nullptr, *d_key_sorted = nullptr, *d_sorted = nullptr;
	TREEE_TRY(ctx->alloc_t(t->nb, &d_key));
	TREEE_TRY(ctx->alloc_t(t->nb, &d_key_sorted));
	TREEE_TRY(ctx->alloc_t(t->nb, &d_sorted));
	static const bool no_up = getenv("TREEE_KNN_NO_LEAF_UP") != nullptr; // A/B and test hook: every inner level its own launch
	const bool up = t->P >= 16u && !no_up;
	hipLaunchKernelGGL(k_leaf_aabb, dim3(div_up(t->P, 16)), dim3(256), 0, ctx->stream, t->sc->d_sorted, (uint32_t)n, t->nb, t->P, d_bucket_seg, t->d_bvh, t->d_sub, up,
	                   d_key);
	TREEE_LAUNCH_CHECK();
	uint32_t lvl = t->P >> (up ? 1u + LEAF_UP : 1u);
	for (; lvl > INNER_TOP; lvl >>= 1) // nodes [lvl, 2*lvl)
		hipLaunchKernelGGL(k_inner_aabb, dim3(div_up(lvl, 256)), dim3(256), 0, ctx->stream, t->d_bvh, lvl, lvl);
	if (lvl >= 1) hipLaunchKernelGGL(k_inner_aabb_top, dim3(1), dim3(INNER_TOP), 0, ctx->stream, t->d_bvh, lvl); // the levels that fit one workgroup
	TREEE_LAUNCH_CHECK();
	uint32_t longest = 0;
	for (uint32_t x = 0; x < 8; ++x) longest = std::max(longest, order_count(t->nb, x));
	t->n_order = 8u * longest;
	TREEE_TRY(pool_alloc_t(ctx, t->n_order, &t->d_order));
	TREEE_TRY(sort_pairs_dev(ctx, d_key, nullptr, t->nb, 8, d_key_sorted, d_sorted));
	hipLaunchKernelGGL(k_order_interleave, dim3(div_up(t->n_order, 256)), dim3(256), 0, ctx->stream, d_sorted, t->nb, t->n_order, t->d_order);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}


// Sorted by (segment, cell key, original index): a 30-bit sort by the Hilbert index of the Morton cell followed by a stable
// sort on the segment id.
int32_t build_segmented(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_seg, uint64_t n, uint32_t nseg, treee_knn *t, const float *cube = nullptr) {
	if (n >= (1ull << 31) - 64ull * nseg) return set_err(TREEE_ERR_UNSUPPORTED, "n=%llu exceeds 2^31-1 points per shard", (unsigned long long)n);
	SortedCloud *sc = t->sc.get();
	sc->device = ctx->device;
	sc->owner = ctx;
	t->owner = ctx;
	t->segmented = true;
	t->n_points = n;
	if (cube && cube[3] > 0.0f) { // a cube handed in (the whole cloud's, when this is one device's share of the segments)
		for (int a = 0; a < 3; ++a) sc->corner[a] = cube[a];
		sc->size = cube[3];
	} else {
		TREEE_TRY(bbox_dev(ctx, d_xyz, n, sc->corner, &sc->size));
	}
	const uint64_t nb_max = div_up(n, 64) + nseg; // sum over segments of ceil(count / 64) <= n / 64 + nseg
	sc->n = nb_max * 64;
	TREEE_TRY(pool_alloc_t(ctx, sc->n, &sc->d_sorted));
	uint32_t *d_keys_raw = nullptr, *d_keys = nullptr, *d_perm = nullptr, *d_sk = nullptr, *d_sks = nullptr, *d_perm2 = nullptr;
	uint32_t *d_start = nullptr, *d_nb = nullptr, *d_base = nullptr, *d_bseg = nullptr, *d_err = nullptr;
	TREEE_TRY(ctx->alloc_t(n, &d_keys_raw));
	TREEE_TRY(ctx->alloc_t(n, &d_keys));
	TREEE_TRY(ctx->alloc_t(n, &d_perm));
	TREEE_TRY(ctx->alloc_t(n, &d_sk));
	TREEE_TRY(ctx->alloc_t(n, &d_sks));
	TREEE_TRY(ctx->alloc_t(n, &d_perm2));
	TREEE_TRY(ctx->alloc_t((size_t)nseg + 1, &d_start));
	TREEE_TRY(ctx->alloc_t(nseg, &d_nb));
	TREEE_TRY(ctx->alloc_t(nseg, &d_base));
	TREEE_TRY(ctx->alloc_t(nb_max, &d_bseg));
	TREEE_TRY(ctx->alloc_t(1, &d_err));
	hipStream_t st = ctx->stream;
	TREEE_HIP(hipMemsetAsync(d_err, 0, 4, st));
	TREEE_HIP(hipMemsetAsync(sc->d_sorted, 0xFF, sc->n * sizeof(float4), st)); // (NaN, NaN, NaN, ~0u)
	TREEE_HIP(hipMemsetAsync(d_bseg, 0xFF, nb_max * 4, st));
	TREEE_TRY(morton_keys_dev(ctx, d_xyz, n, sc->corner, sc->size, d_keys_raw));
	static const bool morton_order = getenv("TREEE_KNN_MORTON") != nullptr; // A/B hook
	if (!morton_order && n) { // Hilbert order inside every segment (see refine_dense_cells)
		TREEE_TRY(hilbert_keys_dev(ctx, d_keys_raw, nullptr, n, d_keys_raw));
		sc->hilbert = true;
	}
	TREEE_TRY(sort_pairs_dev(ctx, d_keys_raw, nullptr, n, 30, d_keys, d_perm));
	{ // crowded Morton cells: order by the 20-bit-per-axis key instead (see refine_dense_cells)
		uint32_t flag = 0;
		static const bool off = getenv("TREEE_KNN_NO_REFINE") != nullptr;
		if (n > LONG_RUN && !off) {
			TREEE_TRY(long_run_flag_dev(ctx, d_keys, n, d_err));
			TREEE_HIP(hipMemcpyAsync(&flag, d_err, 4, hipMemcpyDeviceToHost, st));
			TREEE_HIP(hipStreamSynchronize(st));
			TREEE_HIP(hipMemsetAsync(d_err, 0, 4, st));
			flag &= 1u; // (bit 1 is about 2^-7 blocks of MORTON-sorted keys: not this structure's business)
		}
		if (flag) {
			uint32_t *d_lo = d_sk, *d_lo_s = d_sks, *d_pos1 = d_perm2, *d_hi = d_keys; // buffers that are free at this point
			static const bool fine_morton = getenv("TREEE_KNN_FINE_MORTON") != nullptr; // A/B hook
			hipLaunchKernelGGL(k_fine_keys_aos, dim3(div_up(n, 256)), dim3(256), 0, st, d_xyz, (uint32_t)n, sc->corner[0], sc->corner[1], sc->corner[2],
			                   sc->size, !fine_morton && !morton_order, d_lo);
			TREEE_LAUNCH_CHECK();
			TREEE_TRY(sort_pairs_dev(ctx, d_lo, nullptr, n, 30, d_lo_s, d_pos1));
			hipLaunchKernelGGL(k_gather_u32, dim3(div_up(n, 256)), dim3(256), 0, st, d_keys_raw, d_pos1, (uint32_t)n, d_hi);
			TREEE_LAUNCH_CHECK();
			TREEE_TRY(sort_pairs_dev(ctx, d_hi, d_pos1, n, 30, d_lo, d_perm)); // d_lo: sorted keys, not needed again
		}
	}
	hipLaunchKernelGGL(k_seg_gather, dim3(div_up(n, 256)), dim3(256), 0, st, d_seg, d_perm, (uint32_t)n, nseg, d_sk, d_err);
	TREEE_LAUNCH_CHECK();
	int bits = 1;
	while (bits < 32 && (1ull << bits) < nseg) ++bits;
	TREEE_TRY(sort_pairs_dev(ctx, d_sk, d_perm, n, bits, d_sks, d_perm2));
	hipLaunchKernelGGL(k_seg_starts, dim3(div_up((uint64_t)nseg + 1, 256)), dim3(256), 0, st, d_sks, (uint32_t)n, nseg, d_start);
	hipLaunchKernelGGL(k_seg_buckets, dim3(div_up(nseg, 256)), dim3(256), 0, st, d_start, nseg, d_nb);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_nb, d_base, nseg, nullptr));
	hipLaunchKernelGGL(k_seg_scatter, dim3(div_up(n, 256)), dim3(256), 0, st, d_xyz, d_perm2, d_sks, d_start, d_base, (uint32_t)n, sc->d_sorted, d_bseg);
	TREEE_LAUNCH_CHECK();
	TREEE_TRY(build_bvh(ctx, t, d_bseg));
	uint32_t err = 0;
	TREEE_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	if (err) return set_err(TREEE_ERR_INVALID, "segment id >= n_segments (%u)", nseg);
	return TREEE_OK;
}

} // namespace

extern "C" {

int32_t treee_knn_build_segments_dev(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_segment, uint64_t n, uint32_t n_segments, treee_knn **out) {
	if (!ctx || !out) return set_err(TREEE_ERR_INVALID, "treee_knn_build_segments_dev: null argument");
	*out = nullptr;
	if (n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	if (!d_xyz || !d_segment) return set_err(TREEE_ERR_INVALID, "treee_knn_build_segments_dev: null array");
	if (n_segments == 0 || n_segments > (1u << 24)) return set_err(TREEE_ERR_INVALID, "n_segments %u outside [1, 2^24]", n_segments);
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_knn *t = new treee_knn;
	t->device = ctx->device;
	const int32_t rc = build_segmented(ctx, d_xyz, d_segment, n, n_segments, t);
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}

} // extern "C"
namespace treee {
std::shared_ptr<SortedCloud> knn_sorted_cloud(const treee_knn *t, uint64_t *n_points, bool *segmented) {
	*n_points = t->n_points;
	*segmented = t->segmented;
	return t->sc;
}
int32_t knn_build_segments_cube(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_segment, uint64_t n, uint32_t n_segments, const float cube[4],
                                treee_knn **out) {
	*out = nullptr;
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_knn *t = new treee_knn;
	t->device = ctx->device;
	const int32_t rc = build_segmented(ctx, d_xyz, d_segment, n, n_segments, t, cube);
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}
// the structure over a cloud that is sorted already (one Morton sort, two consumers): its own order (refine_dense_cells) and
// the box tree, on ctx's stream; the caller has reset the context and ordered the stream behind the sort
int32_t knn_from_sorted_cloud(treee_ctx *ctx, const std::shared_ptr<SortedCloud> &sc, treee_knn **out, const BlockTables *pre) {
	*out = nullptr;
	if (sc->n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	treee_knn *t = new (std::nothrow) treee_knn;
	if (!t) return set_err(TREEE_ERR_HIP, "kNN structure: out of host memory");
	t->device = ctx->device;
	t->sc = sc;
	t->n_points = sc->n;
	int32_t rc = refine_dense_cells(ctx, sc, &t->sc, pre); // (a private, finer-sorted copy if a Morton cell is crowded)
	if (rc == TREEE_OK) rc = build_bvh(ctx, t);
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}
// (speculative: the caller does not know yet whether the structure will order by blocks -- the flags are in flight; the tables
// are 60 MB of key reads)
int32_t knn_block_tables_enqueue(treee_ctx *ctx, const SortedCloud *sc, BlockTables *bt) {
	if (!blocks_pay(sc->n) || sc->hilbert) return TREEE_OK;
	if (sc->n > (64ull << 20)) return TREEE_OK; // (a cloud this large rarely orders by blocks -- some 2^-7 block holds 2 048 points --: no guess, the tables are made if the flags ask)
	static const bool off = getenv("TREEE_KNN_NO_BLOCKSORT") != nullptr || getenv("TREEE_KNN_NO_REFINE") != nullptr || getenv("TREEE_KNN_MORTON") != nullptr;
	if (off) return TREEE_OK;
	return block_tables_enqueue(ctx, sc, bt);
}
} // namespace treee
extern "C" {

int32_t treee_curvature_segments(const float *xyz, const uint32_t *segment, uint64_t n, uint32_t n_segments, uint32_t k, float max_dist_sq,
                                 float *curve_out) {
	if (n == 0) return TREEE_OK;
	if (!xyz || !segment || !curve_out) return set_err(TREEE_ERR_INVALID, "treee_curvature_segments: null argument");
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_xyz, d_curve;
	DevBuf<uint32_t> d_seg;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_TRY(d_seg.alloc(n));
	TREEE_TRY(d_curve.alloc(n));
	TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
	TREEE_HIP(hipMemcpyAsync(d_seg.p, segment, n * 4, hipMemcpyHostToDevice, ctx->stream));
	treee_knn *t = nullptr;
	TREEE_TRY(treee_knn_build_segments_dev(ctx, d_xyz.p, d_seg.p, n, n_segments, &t));
	int32_t rc = treee_curvature_dev(ctx, t, k, max_dist_sq, d_curve.p);
	if (rc == TREEE_OK) {
		hipError_t e = hipMemcpyAsync(curve_out, d_curve.p, n * 4, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) rc = set_err(TREEE_ERR_HIP, "curvature copy back: %s", hipGetErrorString(e));
	}
	treee_knn_free(t);
	return rc;
}

int32_t treee_knn_build_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, treee_knn **out) {
	if (!ctx || !out) return set_err(TREEE_ERR_INVALID, "treee_knn_build_dev: null argument");
	*out = nullptr;
	if (n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	if (!d_xyz) return set_err(TREEE_ERR_INVALID, "treee_knn_build_dev: xyz is null");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	treee_knn *t = new treee_knn; // (after reset(): nothing to leak if that fails)
	t->device = ctx->device;
	t->n_points = n;
	static const bool morton_order = getenv("TREEE_KNN_MORTON") != nullptr; // A/B hook (refine_dense_cells)
	int32_t rc = build_sorted_cloud(ctx, d_xyz, n, nullptr, 0.0f, t->sc.get(), false, /*hilbert_order=*/!morton_order); // one sort, in this structure's order
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	if (rc == TREEE_OK) rc = refine_dense_cells(ctx, t->sc, &t->sc); // (crowded cells only)
	if (rc == TREEE_OK) rc = build_bvh(ctx, t);
	if (rc != TREEE_OK) {
		treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}

int32_t treee_knn_from_octree_dev(treee_ctx *ctx, const treee_octree *o, treee_knn **out) {
	if (!ctx || !o || !out) return set_err(TREEE_ERR_INVALID, "treee_knn_from_octree_dev: null argument");
	if (octree_is_shard(o)) return set_err(TREEE_ERR_UNSUPPORTED, "treee_knn_from_octree_dev: a sharded octree's indices are global; build the kNN structure from the rank's own + ghost points");
	*out = nullptr;
	std::shared_ptr<SortedCloud> sc = octree_sorted_cloud(o);
	if (sc->n == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	TREEE_TRY(use_begin(ctx, sc->owner, &sc->sync)); // the octree may have been built on another stream
	treee_knn *t = nullptr;
	int32_t rc = knn_from_sorted_cloud(ctx, sc, &t);
	const int32_t e = use_end(ctx, sc->owner, &sc->sync);
	if (rc == TREEE_OK) rc = e;
	if (rc != TREEE_OK) {
		if (t) treee_knn_free(t);
		return rc;
	}
	*out = t;
	return TREEE_OK;
}

// The kNN structure of one rank of a sharded cloud -- own points followed by ghost points (treee_shard_exchange_ghosts_dev's
// layout: the handle's point i is own point i for i < n_own, ghost i - n_own behind them) -- WITHOUT sorting the tile a
// second time: the sharded octree build has the own points in Morton order of the common cube already, the structure's
// order is (Hilbert index of the cell, index), and a stable sort by Hilbert index of [own in (Morton cell, index) order |
// ghosts in index order] is exactly that order, because inside one cell the own points stand in index order and every
// ghost index is larger.  So: Hilbert indices of the sorted own keys (a table walk per key), Morton + Hilbert of the g
// ghosts, ONE stable sort of n_own + g pairs, one gather.  Same arrays as treee_knn_build_dev over the concatenated cloud
// would give with the common cube; only the cube (and with it the order inside distance ties) differs from that call's.
int32_t treee_knn_from_shard_dev(treee_ctx *ctx, const treee_octree *o, const float *d_ghost_xyz, uint64_t g, treee_knn **out) {
	if (!ctx || !o || !out || (g && !d_ghost_xyz)) return set_err(TREEE_ERR_INVALID, "treee_knn_from_shard_dev: null argument");
	*out = nullptr;
	uint32_t base = 0;
	std::shared_ptr<SortedCloud> loc = octree_shard_local(o, &base);
	if (!loc)
		return set_err(TREEE_ERR_INVALID, octree_is_shard(o) ? "treee_knn_from_shard_dev: the rank's own sorted copy was released (treee_shard_octree_release_local)"
		                                                     : "treee_knn_from_shard_dev: not a sharded octree (treee_knn_from_octree_dev serves the others)");
	const uint64_t n = loc->n, m = n + g;
	if (m == 0) return set_err(TREEE_ERR_EMPTY, "KDTree::new on an empty slice (the reference panics, kd_tree.rs:53,60)");
	if (m >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "n=%llu exceeds 2^31-1 points per shard", (unsigned long long)m);
	TREEE_HIP(hipSetDevice(ctx->device));
	treee_knn *t = new treee_knn;
	t->device = ctx->device;
	t->n_points = m;
	auto fine = std::make_shared<SortedCloud>();
	fine->n = m;
	fine->device = ctx->device;
	fine->owner = ctx;
	fine->size = loc->size;
	fine->hilbert = true;
	for (int a = 0; a < 3; ++a) fine->corner[a] = loc->corner[a];
	t->sc = fine;
	hipStream_t st = ctx->stream;
	uint32_t *d_hk = nullptr, *d_pos = nullptr;
	int32_t rc = ctx->reset();
	bool begun = false; // (a use that has begun is ended on EVERY path: the owner's stream must learn of it -- ADVICE r4)
	if (rc == TREEE_OK) {
		rc = use_begin(ctx, loc->owner, &loc->sync);
		begun = rc == TREEE_OK;
	}
	if (rc == TREEE_OK) rc = pool_alloc_t(ctx, m, &fine->d_keys);
	if (rc == TREEE_OK) rc = pool_alloc_t(ctx, m, &fine->d_perm);
	if (rc == TREEE_OK) rc = pool_alloc_t(ctx, m, &fine->d_sorted);
	if (rc == TREEE_OK) rc = ctx->alloc_t(m, &d_hk);
	if (rc == TREEE_OK) rc = ctx->alloc_t(m, &d_pos);
	if (rc == TREEE_OK && n) rc = hilbert_keys_dev(ctx, loc->d_keys, nullptr, n, d_hk);
	if (rc == TREEE_OK && g) rc = morton_keys_dev(ctx, d_ghost_xyz, g, loc->corner, loc->size, d_hk + n);
	if (rc == TREEE_OK && g) rc = hilbert_keys_dev(ctx, d_hk + n, nullptr, g, d_hk + n);
	if (rc == TREEE_OK) rc = sort_pairs_dev(ctx, d_hk, nullptr, m, 30, fine->d_keys, d_pos);
	if (rc == TREEE_OK) {
		hipLaunchKernelGGL(k_gather_own_ghost, dim3(div_up(m, 256)), dim3(256), 0, st, loc->d_sorted, (uint32_t)n, base, d_ghost_xyz, d_pos, (uint32_t)m, fine->d_sorted,
		                   fine->d_perm);
		if (hipGetLastError() != hipSuccess) rc = set_err(TREEE_ERR_HIP, "treee_knn_from_shard_dev: gather launch");
	}
	if (begun) {
		const int32_t e = use_end(ctx, loc->owner, &loc->sync);
		if (rc == TREEE_OK) rc = e;
	}
	if (rc == TREEE_OK) rc = refine_dense_cells(ctx, fine, &t->sc); // (crowded cells only: the cloud is in this structure's order already)
	if (rc == TREEE_OK) rc = build_bvh(ctx, t);
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
	TREEE_DEFAULT_CTX(ctx);
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
	// uses from other streams: the owners' streams wait for them before the arrays can be handed out again.  The sorted
	// cloud may still be shared with an octree: its own events (and `ready`, which lets a foreign stream wait for the build
	// only) are retired by SortedCloud::release when the LAST reference goes; here its owner is merely ordered behind this
	// handle's users.
	if (t->sc && t->sc.use_count() > 1) wait_users(t->sc->owner, &t->sc->sync);
	retire(t->owner, &t->sync);
	t->sc.reset();
	pool_release(t->owner, t->d_bvh);
	pool_release(t->owner, t->d_sub);
	pool_release(t->owner, t->d_order);
	delete t;
	return TREEE_OK;
}

int32_t treee_knn_len(const treee_knn *t, uint64_t *n_out) {
	if (!t || !n_out) return set_err(TREEE_ERR_INVALID, "treee_knn_len: null argument");
	*n_out = t->n_points;
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

int32_t treee_knn_self_ties_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq, float *d_dist, uint32_t *d_index, uint32_t *d_counts,
                                uint8_t *d_tie_flags) {
	if (!ctx || !t) return set_err(TREEE_ERR_INVALID, "treee_knn_self_ties_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return launch_knn(ctx, t, nullptr, t->sc->n, k, max_dist_sq, d_dist, d_index, d_counts, nullptr, ~0ull, d_tie_flags);
}

int32_t treee_curvature_ties_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq, float *d_curve, uint8_t *d_tie_flags) {
	if (!ctx || !t || !d_curve) return set_err(TREEE_ERR_INVALID, "treee_curvature_ties_dev: null argument");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return launch_knn(ctx, t, nullptr, t->sc->n, k, max_dist_sq, nullptr, nullptr, nullptr, d_curve, ~0ull, d_tie_flags);
}

int32_t treee_curvature_own_dev(treee_ctx *ctx, const treee_knn *t, uint64_t n_own, uint32_t k, float max_dist_sq, float *d_curve) {
	if (!ctx || !t || (n_own && !d_curve)) return set_err(TREEE_ERR_INVALID, "treee_curvature_own_dev: null argument");
	if (n_own == 0) return TREEE_OK; // (a rank that owns nothing: its ghosts are neighbours of nobody here)
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	return launch_knn(ctx, t, nullptr, t->sc->n, k, max_dist_sq, nullptr, nullptr, nullptr, d_curve, n_own);
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
	TREEE_DEFAULT_CTX(ctx);
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
	HostBuf<uint32_t> counts;
	TREEE_TRY(counts.alloc(nq));
	TREEE_TRY(treee_knn_query_batch(t, q, nq, 1, max_dist_sq, nullptr, counts.data()));
	for (uint64_t i = 0; i < nq; ++i) empty_out[i] = counts[i] == 0 ? 1 : 0;
	return TREEE_OK;
}

static int32_t curvature_host(const char *what, const float *xyz, uint64_t n, uint32_t k, float max_dist_sq, float *curve_out, uint8_t *flags_out) {
	if (n == 0) return TREEE_OK;
	if (!xyz || !curve_out) return set_err(TREEE_ERR_INVALID, "%s: null argument", what);
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	treee_knn *t = nullptr;
	TREEE_TRY(treee_knn_build(xyz, n, &t));
	DevBuf<float> d_curve;
	DevBuf<uint8_t> d_flags;
	int32_t rc = d_curve.alloc(n);
	if (rc == TREEE_OK && flags_out) rc = d_flags.alloc(n);
	if (rc == TREEE_OK) rc = flags_out ? treee_curvature_ties_dev(ctx, t, k, max_dist_sq, d_curve.p, d_flags.p) : treee_curvature_dev(ctx, t, k, max_dist_sq, d_curve.p);
	if (rc == TREEE_OK) {
		hipError_t e = hipMemcpyAsync(curve_out, d_curve.p, n * 4, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess && flags_out) e = hipMemcpyAsync(flags_out, d_flags.p, n, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) rc = set_err(TREEE_ERR_HIP, "%s copy back: %s", what, hipGetErrorString(e));
	}
	treee_knn_free(t);
	return rc;
}

int32_t treee_curvature(const float *xyz, uint64_t n, uint32_t k, float max_dist_sq, float *curve_out) {
	return curvature_host("treee_curvature", xyz, n, k, max_dist_sq, curve_out, nullptr);
}

int32_t treee_curvature_ties(const float *xyz, uint64_t n, uint32_t k, float max_dist_sq, float *curve_out, uint8_t *tie_flags_out) {
	if (n && !tie_flags_out) return set_err(TREEE_ERR_INVALID, "treee_curvature_ties: null argument");
	return curvature_host("treee_curvature_ties", xyz, n, k, max_dist_sq, curve_out, tie_flags_out);
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
	if (t->segmented) return set_err(TREEE_ERR_UNSUPPORTED, "segmented handle: only self queries (a position has no segment)");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
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
