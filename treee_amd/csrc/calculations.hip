// calculations.hip -- the Calculations phase around the kNN consumer, for every segment
// of a segmented cloud in one call:
//   SegmentData::new                  treee/src/calculations.rs:133-175   per-segment min / max, classification
//   SegmentInformation::new           treee/src/calculations.rs:210-276   trunk / crown heights from slice areas
//   get_size_areas                    treee/src/calculations.rs:429-452   0.1 m slices, one convex hull each
//   Tree::new / insert / statistics   treee/src/segmenting.rs:258-269,346-396,320-323 and centroid :526-545
//   SegmentInformation::update        treee/src/calculations.rs:278-327   expansion, height (curve: knn.hip)
//   DisplayData::new                  treee/src/interactive/mod.rs:1002-1039  the four per-point `value` arrays
//
// The reference runs one rayon task per segment (calculations.rs:97) and inside it inserts the points of a slice into
// that slice's hull one after the other.  Here the unit of parallelism is the (segment, slice) pair -- ~200 slices
// for each of thousands of trees -- and inside a pair a whole wave works on 64 points at a time: every lane tests its
// point against the current hull, the FIRST point (in slice order) that changes the hull is applied, the later lanes
// test again.  A point that lies inside the hull at its turn leaves no trace, so this is exactly the sequential
// insertion order of the reference, and the arithmetic (edge normal = normalize((dir.y, -dir.x)), dot, strict `> 0`)
// is evaluated in the same order with correctly rounded sqrt / divide and no FMA: hull vertices and areas are
// bit-identical to the CPU restatement (oracle/calc_oracle.cpp).
//
// Order of the points inside a slice = their order in the input arrays (the order of the reference's
// `segments: HashMap<u32, Vec<Point3<f32>>>` vectors): two stable radix sorts (slice, then segment) keep it.
#include "common.h"

using namespace treee;

namespace {

constexpr float SLICE_WIDTH = 0.1f; // calculations.rs:16
constexpr uint32_t SLICE_BITS = 16;  // slice index inside a segment: up to 65 535 slices = 6.5 km of height

__device__ __forceinline__ uint32_t as_usize_u32(float v) { // Rust `as usize` (saturating, NaN -> 0), clamped to u32
	if (!(v > 0.0f)) return 0u;
	if (v >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)v;
}

// order-preserving u32 image of a float (-inf .. +inf; NaNs are never encoded)
__device__ __forceinline__ uint32_t f2ord(float f) {
	const uint32_t b = __float_as_uint(f);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }
// f32::total_cmp as a signed integer order
__device__ __forceinline__ int32_t f2total(float f) {
	int32_t x = (int32_t)__float_as_uint(f);
	x ^= (int32_t)((uint32_t)(x >> 31) >> 1);
	return x;
}
__device__ __forceinline__ float total2f(int32_t x) {
	x ^= (int32_t)((uint32_t)(x >> 31) >> 1);
	return __uint_as_float((uint32_t)x);
}

__global__ __launch_bounds__(256) void k_check_segments(const uint32_t *__restrict__ seg, uint32_t n, uint32_t nseg, uint32_t *__restrict__ err) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n && seg[i] >= nseg) atomicOr(err, 1u);
}

// start[s] = first position of segment s in the segment-sorted order, start[nseg] = n
// ---- SegmentData::new, min / max (calculations.rs:135-147) --------------------------------------------------
// f32::min / f32::max drop a NaN operand, so the result is the extreme of the non-NaN values: an order-independent
// reduction over ordered integer images of the floats.  (Rounds 2-3: the points sorted by segment, a tile's extremes per
// wave through a 12-byte gather, a wave per segment over its tiles -- 1.15 ms per 15 M points; plain atomics from every
// wave had piled up on a ground segment's six words: 2.5 ms.)
constexpr uint32_t AABB_ROW = 8; // words per segment in the table of extremes
__global__ __launch_bounds__(256) void k_init_aabb(uint32_t *__restrict__ aabb, uint32_t nseg) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < nseg * AABB_ROW) aabb[i] = (i % AABB_ROW) < 3u ? 0xFFFFFFFFu : 0u; // [s][0..2] = min (ordered image), [3..5] = max, [6..7] pad
}
#ifndef TREEE_AABB_STRIDES // (A/B hook: the sampling ladder)
#define TREEE_AABB_STRIDES 50021u, 3001u, 251u, 16u, 1u
#define TREEE_AABB_RUNS 1u, 1u, 1u, 64u, 1u
#endif
// The extremes straight from the cloud as it lies, no sort.  A value is sent (atomicMin / atomicMax on the ordered image)
// only if it beats what the segment's word already holds, and the kernel runs as a ladder of samples -- every 50 021st point,
// every 3 001st, every 251st, a sixteenth of the cloud in runs of 64, every point -- so that by the time the half million threads of a full launch read those words they hold a sample's extremes
// and almost nothing is left to send (a segment's running extreme over randomly ordered points improves ~ln(n) times; without
// the samples every thread resident at launch sees the initial word and the ground's six words serialise them: the 2.5 ms
// above).  `present` marks the segments that have a point (an empty segment is (0, 0, 0), calculations.rs:135-136).
__global__ __launch_bounds__(256) void k_seg_aabb_direct(const float *__restrict__ xyz, const uint32_t *__restrict__ seg, uint32_t n, uint32_t nseg,
                                                          uint32_t stride, uint32_t run, uint32_t *aabb, uint32_t *present) {
	// (the sample: runs of `run` consecutive points, one run every `stride` runs -- single points for the sparse levels, runs of
	// 64 for the dense one, whose reads then come in whole lines)
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	const uint64_t i64 = (uint64_t)(t / run) * run * stride + t % run;
	if (i64 >= n) return;
	const uint32_t i = (uint32_t)i64, sg = seg[i];
	if (sg >= nseg) return; // (reported by k_check_segments; nothing is indexed with it)
	if (present[sg] == 0u) present[sg] = 1u;
	// plain loads: a copy in the CU's L1 is at worst as old as this launch, i.e. holds the previous launch's sample -- an older
	// value only means an atomic that was not needed (through the L2 -- volatile -- the six reads per point were 1.0 ms)
	const uint4 *row = (const uint4 *)(aabb + (size_t)sg * AABB_ROW); // 32-byte rows: two 16-byte gathers per point, not three of 8
	const uint4 r0 = row[0], r1 = row[1];
	const uint32_t cur[6] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y};
	uint32_t o[3], better = 0; // bit d: below the minimum read, bit 3 + d: above the maximum
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const float v = xyz[3ull * i + d];
		o[d] = f2ord(v);
		if (v == v) better |= (o[d] < cur[d] ? 1u << d : 0u) | (o[d] > cur[3 + d] ? 8u << d : 0u); // (f32::min / f32::max drop a NaN operand)
	}
	if (better == 0u) return; // (nearly every thread of the full launch: one branch, not six)
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		if (better & (1u << d)) atomicMin(&aabb[(size_t)sg * AABB_ROW + d], o[d]);
		if (better & (8u << d)) atomicMax(&aabb[(size_t)sg * AABB_ROW + 3u + d], o[d]);
	}
}
// The dense levels of the ladder with the table of extremes in LDS (round 5).  The kernel above spends its time in the three
// per-lane gathers of a point's row (64 lanes, 64 segments, a table that does not fit the L1: 340 us for the full pass over 15 M
// points, 0.7 TB/s).  Here a workgroup copies the table as the samples left it (24 bytes per segment: up to AABB_LDS_SEGMENTS of
// them), walks its share of the points against that copy -- LDS gathers --, and what beats the copy goes to the copy (LDS atomics: the
// workgroup's later points see it) and to the global table (the same atomicMin / atomicMax on the same ordered images as above, so
// the result is the same extremes whatever the order).  `present` is marked the first time a workgroup meets a segment.
constexpr uint32_t AABB_LDS_SEGMENTS = 2688, AABB_LDS_THREADS = 1024; // 2688 x 24 B = 63 KiB + 336 B of seen bits: two workgroups per CU
__global__ __launch_bounds__(AABB_LDS_THREADS) void k_seg_aabb_lds(const float *__restrict__ xyz, const uint32_t *__restrict__ seg, uint32_t n, uint32_t nseg,
                                                                    uint32_t stride, uint32_t run, uint32_t *aabb, uint32_t *present) {
	__shared__ uint32_t s_row[AABB_LDS_SEGMENTS * 6];
	__shared__ uint32_t s_seen[(AABB_LDS_SEGMENTS + 31) / 32];
	for (uint32_t i = threadIdx.x; i < nseg * 6u; i += AABB_LDS_THREADS) s_row[i] = aabb[(size_t)(i / 6u) * AABB_ROW + i % 6u];
	for (uint32_t i = threadIdx.x; i < (nseg + 31u) / 32u; i += AABB_LDS_THREADS) s_seen[i] = 0u;
	__syncthreads();
	const uint64_t total = ((uint64_t)n + (uint64_t)stride * run - 1u) / ((uint64_t)stride * run) * run; // threads' worth of sample points (an upper bound)
	for (uint64_t t = (uint64_t)blockIdx.x * AABB_LDS_THREADS + threadIdx.x; t < total; t += (uint64_t)gridDim.x * AABB_LDS_THREADS) {
		const uint64_t i64 = (t / run) * run * stride + t % run; // (the sample of k_seg_aabb_direct)
		if (i64 >= n) continue;
		const uint32_t i = (uint32_t)i64, sg = seg[i];
		if (sg >= nseg) continue; // (reported by k_check_segments)
		if (!((s_seen[sg >> 5] >> (sg & 31u)) & 1u)) {
			if (!((atomicOr(&s_seen[sg >> 5], 1u << (sg & 31u)) >> (sg & 31u)) & 1u)) present[sg] = 1u;
		}
		const uint32_t *row = &s_row[sg * 6u];
		uint32_t o[3], better = 0;
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			const float v = xyz[3ull * i + d];
			o[d] = f2ord(v);
			if (v == v) better |= (o[d] < row[d] ? 1u << d : 0u) | (o[d] > row[3 + d] ? 8u << d : 0u); // (f32::min / f32::max drop a NaN operand)
		}
		if (better == 0u) continue;
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if (better & (1u << d)) {
				atomicMin(&s_row[sg * 6u + d], o[d]);
				atomicMin(&aabb[(size_t)sg * AABB_ROW + d], o[d]);
			}
			if (better & (8u << d)) {
				atomicMax(&s_row[sg * 6u + 3u + d], o[d]);
				atomicMax(&aabb[(size_t)sg * AABB_ROW + 3u + d], o[d]);
			}
		}
	}
}
// decode; an empty segment is (0, 0, 0) (calculations.rs:135-136), a segment whose every value along an axis is NaN keeps NaN
// (min(NaN, NaN)); slice count of the segment (calculations.rs:217, 435)
__global__ __launch_bounds__(256) void k_seg_finish_aabb(const uint32_t *__restrict__ aabb, const uint32_t *__restrict__ present, uint32_t nseg,
                                                          float *__restrict__ seg_min, float *__restrict__ seg_max, uint32_t *__restrict__ n_slices,
                                                          uint32_t *__restrict__ err) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= nseg) return;
	const bool empty = present[s] == 0u;
	float mn[3], mx[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const uint32_t a = aabb[s * AABB_ROW + d], b = aabb[s * AABB_ROW + 3u + d];
		mn[d] = empty ? 0.0f : (a == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : ord2f(a));
		mx[d] = empty ? 0.0f : (b == 0u ? __uint_as_float(0x7FC00000u) : ord2f(b));
		seg_min[s * 3u + d] = mn[d];
		seg_max[s * 3u + d] = mx[d];
	}
	const float height = mx[1] - mn[1];
	const uint32_t raw = as_usize_u32(height / SLICE_WIDTH);
	if (raw >= (1u << SLICE_BITS) - 1u) atomicOr(err, 2u);
	const uint32_t ns = (raw >= (1u << SLICE_BITS) - 1u ? (1u << SLICE_BITS) - 2u : raw) + 1u;
	n_slices[s] = ns;
	atomicMax(err + 1, ns); // (the width of the slice field in the sort key below)
}

// slice of every point, original order: ((pos.y - min) / SLICE_WIDTH) as usize (calculations.rs:442)
__global__ __launch_bounds__(256) void k_slice_index(const float *__restrict__ xyz, const uint32_t *__restrict__ seg, uint32_t n,
                                                      const float *__restrict__ seg_min, const uint32_t *__restrict__ n_slices, uint32_t *__restrict__ slice) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t sg = seg[i];
	const uint32_t idx = as_usize_u32((xyz[3ull * i + 1] - seg_min[sg * 3u + 1]) / SLICE_WIDTH);
	const uint32_t ns = n_slices[sg];
	slice[i] = idx < ns ? idx : ns - 1u; // (min <= y <= max: always inside; the clamp only guards a capped segment)
}
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ dst) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}
// task = run of equal (segment, slice) in the final order; flag its first element
__global__ __launch_bounds__(256) void k_join_keys(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ slice, uint32_t n, uint32_t slbits,
                                                    uint32_t *__restrict__ key) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) key[i] = (seg[i] << slbits) | slice[i]; // (slice < its segment's count <= the largest count < 2^slbits)
}
__global__ __launch_bounds__(256) void k_split_keys(const uint32_t *__restrict__ key, uint32_t n, uint32_t slbits, uint32_t *__restrict__ seg,
                                                     uint32_t *__restrict__ slice) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t k = key[i];
	seg[i] = slbits < 32u ? k >> slbits : 0u;
	slice[i] = slbits < 32u ? k & ((1u << slbits) - 1u) : k;
}
__global__ __launch_bounds__(256) void k_task_flags(const uint32_t *__restrict__ sorted_seg, const uint32_t *__restrict__ sorted_slice, uint32_t n,
                                                     uint32_t *__restrict__ flag) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	flag[i] = (i == 0 || sorted_seg[i] != sorted_seg[i - 1] || sorted_slice[i] != sorted_slice[i - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_task_starts(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ task_of, uint32_t n,
                                                      uint32_t *__restrict__ task_start) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	if (flag[i]) task_start[task_of[i]] = i;
}

// ---- Tree::new / insert / statistics for one (segment, slice) per wave -----------------------------------------
constexpr uint32_t HULL_LDS = 128; // hull vertices kept in LDS per wave; a larger hull moves to the task's slot in HBM
constexpr uint32_t EDGE_MODE_POINTS = 64; // the first points of a slice are inserted one by one, an edge per lane (slice_hull_wave)

struct P2 {
	float x, y;
};

// segmenting.rs:347-353: the outward unit normal of the edge a -> b, normalize((dir.y, -dir.x)) = v / sqrt(x*x + y*y).
// It depends on the edge only, so it is computed when the hull changes and kept beside the vertex (the square root and
// the two divisions are ~100 instructions; the per-point test that is left is 2 sub, 2 mul, 1 add, 1 compare).
__device__ __forceinline__ P2 edge_normal(P2 a, P2 b) {
	const float dx = b.x - a.x, dy = b.y - a.y;
	const float ox = dy, oy = -dx;
	const float norm = sqrtf(ox * ox + oy * oy);
	return P2{ox / norm, oy / norm};
}
__device__ __forceinline__ bool unit_length(P2 n) { // (false for NaN, infinite and zero "normals": see slice_hull_block)
	const float l2 = n.x * n.x + n.y * n.y;
	return l2 > 0.5f && l2 < 2.0f;
}
__device__ __forceinline__ bool edge_outside(P2 a, P2 nrm, P2 p) {
	const float fx = p.x - a.x, fy = p.y - a.y;
	const float dist = nrm.x * fx + nrm.y * fy;
	return dist > 0.0f;
}

// Slices of BIG_TASK points or more (the ground of a plot holds hundreds of thousands per slice) get a workgroup of 16
// waves instead of one wave: 1024 points are tested per round, everything else is the same algorithm.
#ifndef TREEE_BIG_ONE_BY_ONE
#define TREEE_BIG_ONE_BY_ONE 512
#endif
constexpr uint32_t BIG_TASK = 2048, BIG_THREADS = 1024, BIG_LDS = 2048, BIG_ONE_BY_ONE = TREEE_BIG_ONE_BY_ONE;
__global__ __launch_bounds__(256) void k_big_tasks(const uint32_t *__restrict__ task_start, uint32_t n_tasks, uint32_t n, uint32_t *__restrict__ big_list,
                                                    uint32_t *__restrict__ big_count) {
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= n_tasks) return;
	const uint32_t count = (t + 1u < n_tasks ? task_start[t + 1u] : n) - task_start[t];
	if (count >= BIG_TASK) big_list[atomicAdd(big_count, 1u)] = t; // (any order: the tasks are independent)
}

// The crown pass (calculations.rs:299-301: the slice hulls of the points classified Crown) needs a hull for ONE slice per segment.
// A point is Crown iff neither y < ground_sep nor y < crown_sep (k_classify), i.e. y >= sep = max of the two, and the slice
// index as_usize((y - min) / SLICE_WIDTH) never decreases with y; so every slice above sep's own holds Crown points only --
// the same points in the same order as in the first pass: the same hull, the same area, bit for bit -- and every slice below
// it holds none (area 0.0).  Only sep's slice is mixed (and slice 0, where a NaN height lands as Crown).
struct CrownPass {
	const float *seg_min, *seg_max, *seg_info; // null: not the crown pass
	const uint32_t *n_slices;
	const float *areas_all; // the first pass's areas
};
// true: the slice that starts at `begin` is settled (its area copied by `writer`, or left 0.0)
__device__ __forceinline__ bool crown_slice_settled(const CrownPass &cp, uint32_t begin, const uint32_t *__restrict__ sorted_seg,
                                                    const uint32_t *__restrict__ sorted_slice, const uint32_t *__restrict__ slice_off,
                                                    float *__restrict__ areas, bool writer) {
	if (!cp.areas_all) return false;
	const uint32_t sg = sorted_seg[begin], sl = sorted_slice[begin];
	const float crown_sep = cp.seg_max[sg * 3u + 1] - cp.seg_info[sg * 4u + 1]; // (k_classify's expressions)
	const float ground_sep = crown_sep - cp.seg_info[sg * 4u + 0];
	const float sep = fmaxf(ground_sep, crown_sep); // (drops a NaN operand; NaN if both are: every point is Crown, as_usize(NaN) = 0)
	uint32_t s_sep = as_usize_u32((sep - cp.seg_min[sg * 3u + 1]) / SLICE_WIDTH);
	const uint32_t ns = cp.n_slices[sg];
	if (s_sep >= ns) s_sep = ns - 1u; // (k_slice_index's clamp)
	if (sl == 0u || sl == s_sep) return false;
	if (sl > s_sep && writer) areas[slice_off[sg] + sl] = cp.areas_all[slice_off[sg] + sl];
	return true;
}

// The first points of a slice nearly all change the hull (a tree's slice holds ~20 points and ~10 of them do; a ground slice of
// 300 000 makes most of its ~300 changes in its first few thousand), and in the batch loops below every change costs a scan of
// the whole hull by every lane.  So the first points go the other way round while the hull has at most 64 edges: ONE EDGE PER
// LANE of one wave, the points one after the other in slice order -- one edge test per point, and the scan of
// segmenting.rs:356-370 (`start` = the last edge that is outside after an edge that is not, `end` = the last that is not
// after one that is, the last edge preceding edge 0) is bit arithmetic on the ballot.  Same hull, vertex for vertex: the order of
// the insertions is the slice's either way.  Takes the pending points of one 64-point batch (lane = position in the batch);
// returns with `pending` still set for the points it did not get to (the hull outgrew 64 edges).
template <class Splice, class NewTree>
__device__ __forceinline__ void insert_one_by_one(P2 *&H, P2 *&N, uint32_t &m, const P2 p, bool &pending, const uint32_t lane, Splice &&splice,
                                                  NewTree &&new_tree) {
	uint64_t pm = __ballot(pending);
	P2 hv = {0.0f, 0.0f}, nv = {0.0f, 0.0f};
	bool fresh = false; // hv / nv hold the current hull's vertex `lane` and the normal of the edge that starts there
	while (pm != 0ull && m <= 64u) {
		const int f = __builtin_ctzll(pm);
		pm &= pm - 1ull;
		const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), f)), py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), f));
		pending = pending && (int)lane > f;
		if (m == 0) { // Tree::new (segmenting.rs:258-269)
			new_tree(px, py);
			continue;
		}
		if (!fresh) {
			if (lane < m) {
				hv = H[lane];
				nv = N[lane];
			}
			fresh = true;
		}
		const uint64_t outside = __ballot(lane < m && edge_outside(hv, nv, P2{px, py}));
		const uint64_t all = m == 64u ? ~0ull : (1ull << m) - 1ull;
		const uint64_t before = ((outside << 1) | (outside >> (m - 1u))) & all; // bit j = edge j - 1 (edge m - 1 for j = 0)
		const uint64_t starts = outside & ~before, ends = ~outside & before;
		if (starts == 0ull || ends == 0ull) continue; // inside (or outside every edge): Tree::insert returns without a change
		// (more than one run of outside edges: the reference's splice may then cut a piece of the hull off -- slice_hull_block counts on being told)
		splice(63u - (uint32_t)__builtin_clzll(starts), 63u - (uint32_t)__builtin_clzll(ends), px, py, __popcll(starts) > 1);
		fresh = false;
	}
}

// FOUR SHORT SLICES PER WAVE, one per row of 16 lanes.  A tree's slice holds ~20 points of which ~12 change the hull, and a hull of
// such a slice has at most 16 vertices 97 % of the time: a whole wave per slice (insert_one_by_one) leaves three quarters of its
// lanes idle through the ~100 instructions a change costs (the reference's normalisation is a square root and two divisions).
// Here a row holds its slice's hull in REGISTERS (lane j = vertex j and the normal of the edge that starts there), its points wait
// in LDS, and the four rows step through their own points in lockstep: one edge test per lane, the reference's scan
// (segmenting.rs:356-370) as bit arithmetic on the row's 16 bits of the ballot, the splice (:374-380) as one lane permutation.
// Same insertions in the same order with the same arithmetic: the same hull, vertex for vertex.  Returns a ballot with bit 16 r
// set for every row whose slice is left to the whole wave (slice_hull_wave): more than 64 points, or a hull that outgrew 16.
__device__ __forceinline__ uint64_t slice_hulls_rows(P2 *pts /* this wave's 4 x 64 entries */, uint32_t t0, const float *__restrict__ xyz,
                                                     const uint32_t *__restrict__ order, const uint32_t *__restrict__ sorted_seg,
                                                     const uint32_t *__restrict__ sorted_slice, const uint32_t *__restrict__ task_start,
                                                     uint32_t n_tasks, uint32_t n, const uint32_t *__restrict__ cls, uint32_t want,
                                                     const uint32_t *__restrict__ slice_off, float *__restrict__ areas, const CrownPass &cp) {
	const uint32_t lane = threadIdx.x & 63u, row = lane >> 4, j = lane & 15u, row_base = lane & 48u;
	const uint32_t t = t0 + row;
	uint32_t begin = 0, count = 0;
	if (t < n_tasks) {
		begin = task_start[t];
		count = (t + 1u < n_tasks ? task_start[t + 1u] : n) - begin;
	}
	bool redo = t < n_tasks && count > 64u && count < BIG_TASK; // (a big slice has its own workgroup)
	bool packed = t < n_tasks && count <= 64u;
	if (packed && crown_slice_settled(cp, begin, sorted_seg, sorted_slice, slice_off, areas, j == 0u)) packed = false;
	// the row's points, in slice order, those of another class left out
	uint32_t cnt = 0;
#pragma unroll
	for (uint32_t q = 0; q < 4u; ++q) {
		const uint32_t i = q * 16u + j;
		bool keep = packed && i < count;
		P2 p = {0.0f, 0.0f};
		if (keep) {
			const uint32_t o = order[begin + i];
			p.x = xyz[3ull * o + 0];
			p.y = xyz[3ull * o + 2];
			if (cls) keep = cls[o] == want;
		}
		const uint32_t km = (uint32_t)(__ballot(keep) >> row_base) & 0xFFFFu;
		if (keep) pts[row * 64u + cnt + (uint32_t)__popc(km & ((1u << j) - 1u))] = p;
		cnt += (uint32_t)__popc(km);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	uint32_t k = 0, m = 0; // next point, hull size (0 = no Tree yet): the same in all lanes of a row
	P2 hv = {0.0f, 0.0f}, nv = {0.0f, 0.0f};
	for (;;) {
		const bool live = k < cnt;
		if (__ballot(live) == 0ull) break;
		const P2 p = pts[row * 64u + (live ? k : 0u)];
		bool changed = false;
		// Tree::insert, scan part: this lane's edge; then the row's 16 bits
		const bool test = live && m != 0u;
		const uint32_t O = (uint32_t)(__ballot(test && j < m && edge_outside(hv, nv, p)) >> row_base) & 0xFFFFu;
		const uint32_t all = (1u << m) - 1u;
		const uint32_t before = ((O << 1) | (O >> (m ? m - 1u : 0u))) & all; // bit e = edge e - 1 (edge m - 1 for e = 0)
		const uint32_t starts = O & ~before, ends = ~O & before;
		const bool mod = test && starts != 0u && ends != 0u;
		const uint32_t s0 = 31u - (uint32_t)__clz((int)(starts | 1u)), e0 = 31u - (uint32_t)__clz((int)(ends | 1u));
		const bool wraps = e0 < s0;                                                     // H' = H[e0 ..= s0] + [point]
		const uint32_t len = wraps ? s0 - e0 + 1u : m - e0, dst = wraps ? 0u : s0 + 2u; // else H[0 ..= s0] + [point] + H[e0 .. m)
		const uint32_t new_m = wraps ? len + 1u : dst + len;
		uint32_t src = j;
		bool is_point = false;
		if (wraps) {
			if (j < len) src = e0 + j;
			is_point = j == len;
		} else {
			is_point = j == s0 + 1u;
			if (j >= dst) src = e0 + (j - dst);
		}
		// (all 64 lanes take part in the permutation: no branch around it)
		const float sx = __shfl(hv.x, (int)(row_base + (src & 15u)), 64), sy = __shfl(hv.y, (int)(row_base + (src & 15u)), 64);
		if (mod) {
			if (new_m > 16u) { // the hull outgrows the row: the whole wave does this slice again
				redo = true;
				cnt = 0;
			} else {
				hv = is_point ? p : P2{sx, sy};
				m = new_m;
				changed = true;
			}
		} else if (live && m == 0u) { // Tree::new (segmenting.rs:258-269)
			hv = j == 0u ? p : (j == 1u ? P2{p.x + 0.1f, p.y} : P2{p.x, p.y + 0.1f});
			m = 3;
			changed = true;
		}
		if (__ballot(changed) != 0ull) { // the normals of the new hull (every edge again, as the reference's every test does)
			const uint32_t nxt = j + 1u < m ? j + 1u : 0u;
			const float qx = __shfl(hv.x, (int)(row_base + nxt), 64), qy = __shfl(hv.y, (int)(row_base + nxt), 64);
			const P2 nn = edge_normal(hv, P2{qx, qy});
			if (changed) nv = nn;
		}
		k += live ? 1u : 0u;
	}
	// Tree::statistics().area: the triangle fan around vertex 0, summed in order (segmenting.rs:526-545)
	const float ax = __shfl(hv.x, (int)row_base, 64), ay = __shfl(hv.y, (int)row_base, 64);
	float area = 0.0f;
	for (uint32_t i = 1; i + 1u < 16u; ++i) {
		const float bx = __shfl(hv.x, (int)(row_base + i), 64) - ax, by = __shfl(hv.y, (int)(row_base + i), 64) - ay;
		const float cx = __shfl(hv.x, (int)(row_base + i + 1u), 64) - ax, cy = __shfl(hv.y, (int)(row_base + i + 1u), 64) - ay;
		if (i + 1u < m) area += bx * cy - by * cx;
	}
	if (packed && !redo && j == 0u && m >= 3u) areas[slice_off[sorted_seg[begin]] + sorted_slice[begin]] = area / 2.0f;
	return __ballot(redo && j == 0u);
}

// one wave per task.  `cls` (optional): only points with cls[orig] == want take part (calculations.rs:299-301).
__device__ __forceinline__ void slice_hull_wave(P2 *lds /* this wave's 2 * HULL_LDS entries */, uint32_t t, const float *__restrict__ xyz,
                                                const uint32_t *__restrict__ order /* final order -> original index */,
                                                const uint32_t *__restrict__ sorted_seg, const uint32_t *__restrict__ sorted_slice,
                                                const uint32_t *__restrict__ task_start, uint32_t n_tasks, uint32_t n,
                                                const uint32_t *__restrict__ cls, uint32_t want, const uint32_t *__restrict__ slice_off,
                                                P2 *__restrict__ spill /* [2 * n + 4 * n_tasks] */, float *__restrict__ areas, const CrownPass &cp) {
	const uint32_t lane = threadIdx.x & 63u;
	if (t >= n_tasks) return; // (whole waves leave: no block-wide barrier below)
	const uint32_t begin = task_start[t], end = t + 1u < n_tasks ? task_start[t + 1u] : n;
	const uint32_t count = end - begin;
	if (count >= BIG_TASK) return; // a whole workgroup takes it (slice_hull_block)
	if (crown_slice_settled(cp, begin, sorted_seg, sorted_slice, slice_off, areas, lane == 0u)) return;
	P2 *H = lds, *N = lds + HULL_LDS; // vertices, then the outward normal of the edge that starts at each vertex; generic
	                                  // pointers: LDS first, HBM after an overflow
	P2 *const big = spill + 2ull * begin + 4ull * t; // room for count + 2 vertices and as many normals
	uint32_t cap = HULL_LDS, m = 0;                // m = hull size, 0 = no Tree yet
	auto sync = [&]() __attribute__((always_inline)) { // the wave's own writes before its next reads
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
	};
	auto normals = [&]() __attribute__((always_inline)) { // all of them again: a splice renumbers the edges
		for (uint32_t c = 0; c < m; c += 64u) {
			const uint32_t j = c + lane;
			if (j < m) N[j] = edge_normal(H[j], H[j + 1u < m ? j + 1u : 0u]);
		}
		sync();
	};

	// splice (segmenting.rs:374-380) of the point (px, py) between the edges s0 and e0, then the normals of the new hull
	auto splice = [&](uint32_t s0, uint32_t e0, float px, float py, bool /*two_runs*/ = false) __attribute__((always_inline)) {
		// splice (segmenting.rs:374-380), cooperative.  Lanes move in lockstep: every lane reads its element before any
		// lane writes, and the chunks run in the direction that never overwrites an unread source.
		// (new size first: a hull that outgrows its LDS copy moves to the task's HBM slot BEFORE anything is written --
		// the append case e0 == 0, s0 == m - 1 of the first branch grows the hull by one too)
		const uint32_t new_m = e0 < s0 ? s0 - e0 + 2u : s0 + 2u + (m - e0);
		if (new_m > cap) {
			for (uint32_t c = 0; c < m; c += 64u)
				if (c + lane < m) big[c + lane] = H[c + lane];
			H = big;
			N = big + (count + 2u);
			cap = 0xFFFFFFFFu;
			sync();
		}
		if (e0 < s0) { // H' = H[e0 ..= s0] + [point]
			const uint32_t keep = s0 - e0 + 1u;
			if (e0 > 0u)
				for (uint32_t c = 0; c < keep; c += 64u) {
					const uint32_t k = c + lane;
					P2 v = {0.0f, 0.0f};
					if (k < keep) v = H[e0 + k];
					__builtin_amdgcn_wave_barrier();
					if (k < keep) H[k] = v;
					sync();
				}
			if (lane == 0) H[keep] = P2{px, py};
		} else { // H' = H[0 ..= s0] + [point] + H[e0 .. m)
			const uint32_t tail = m - e0, dst = s0 + 2u;
			if (dst < e0) { // shrink: shift the tail left, ascending
				for (uint32_t c = 0; c < tail; c += 64u) {
					const uint32_t k = c + lane;
					P2 v = {0.0f, 0.0f};
					if (k < tail) v = H[e0 + k];
					__builtin_amdgcn_wave_barrier();
					if (k < tail) H[dst + k] = v;
					sync();
				}
			} else if (dst > e0) { // grow by one: shift the tail right, descending
				for (uint32_t c = (tail + 63u) / 64u; c-- > 0u;) {
					const uint32_t k = c * 64u + lane;
					P2 v = {0.0f, 0.0f};
					if (k < tail) v = H[e0 + k];
					__builtin_amdgcn_wave_barrier();
					if (k < tail) H[dst + k] = v;
					sync();
				}
			}
			if (lane == 0) H[s0 + 1u] = P2{px, py};
		}
		m = new_m;
		sync();
		normals();
	};

	for (uint32_t base = 0; base < count; base += 64u) {
		const uint32_t i = base + lane;
		bool pending = i < count;
		P2 p = {0.0f, 0.0f};
		if (pending) {
			const uint32_t o = order[begin + i];
			p.x = xyz[3ull * o + 0];
			p.y = xyz[3ull * o + 2];
			if (cls) pending = cls[o] == want;
		}
		// (the first batch: an edge per lane, see insert_one_by_one)
		if (base < EDGE_MODE_POINTS)
			insert_one_by_one(
			    H, N, m, p, pending, lane,
			    [&](uint32_t s0, uint32_t e0, float px, float py, bool) __attribute__((always_inline)) {
				    if (cap != HULL_LDS) return splice(s0, e0, px, py); // (the hull once outgrew its LDS copy: the general code)
				    // m <= 64 here and the new hull has at most 65 vertices: one chunk, every lane reads before any lane writes
				    const bool wraps = e0 < s0;                                                     // H' = H[e0 ..= s0] + [point]
				    const uint32_t len = wraps ? s0 - e0 + 1u : m - e0, dst = wraps ? 0u : s0 + 2u; // else H[0 ..= s0] + [point] + H[e0 .. m)
				    P2 v = {0.0f, 0.0f};
				    if (lane < len) v = H[e0 + lane];
				    __builtin_amdgcn_wave_barrier();
				    if (lane < len && dst != e0) H[dst + lane] = v;
				    if (lane == 0) H[wraps ? len : s0 + 1u] = P2{px, py};
				    m = wraps ? len + 1u : dst + len;
				    sync();
				    normals();
			    },
			    [&](float px, float py) __attribute__((always_inline)) {
				if (lane == 0) {
					H[0] = P2{px, py};
					H[1] = P2{px + 0.1f, py};
					H[2] = P2{px, py + 0.1f};
				}
				m = 3;
				sync();
				normals();
			});
		for (;;) {
			uint64_t pm = __ballot(pending);
			if (!pm) break;
			if (m == 0) { // Tree::new (segmenting.rs:258-269) from the first point of the slice
				const int f = __builtin_ctzll(pm);
				const float px = __shfl(p.x, f, 64), py = __shfl(p.y, f, 64);
				if (lane == 0) {
					H[0] = P2{px, py};
					H[1] = P2{px + 0.1f, py};
					H[2] = P2{px, py + 0.1f};
				}
				m = 3;
				pending = pending && (int)lane > f;
				sync();
				normals();
				continue;
			}
			// Tree::insert, scan part (segmenting.rs:356-370): every pending lane against the current hull
			bool has_s = false, has_e = false;
			uint32_t st = 0, en = 0;
			{
				bool last = edge_outside(H[m - 1u], N[m - 1u], p);
				uint32_t j = 0;
				for (; j + 4u <= m; j += 4u) { // four edges per trip: their eight reads are in flight together
					P2 hv[4], nv[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						hv[u] = H[j + u];
						nv[u] = N[j + u];
					}
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const bool out = edge_outside(hv[u], nv[u], p);
						if (!last && out) {
							st = j + u;
							has_s = true;
						} else if (last && !out) {
							en = j + u;
							has_e = true;
						}
						last = out;
					}
				}
				for (; j < m; ++j) {
					const bool out = edge_outside(H[j], N[j], p);
					if (!last && out) {
						st = j;
						has_s = true;
					} else if (last && !out) {
						en = j;
						has_e = true;
					}
					last = out;
				}
			}
			const bool modifies = pending && has_s && has_e;
			const uint64_t mm = __ballot(modifies);
			if (!mm) break; // nobody of this batch changes the hull any more
			const int f = __builtin_ctzll(mm); // the first one in slice order: everything before it was a no-op
			const uint32_t s0 = (uint32_t)__shfl((int)st, f, 64), e0 = (uint32_t)__shfl((int)en, f, 64);
			const float px = __shfl(p.x, f, 64), py = __shfl(p.y, f, 64);
			splice(s0, e0, px, py);
			pending = pending && (int)lane > f;
		}
	}
	// Tree::statistics().area = centroid(points).1 (segmenting.rs:526-545): triangle fan around points[0], summed in order
	if (lane == 0 && m >= 3u) {
		float area = 0.0f;
		const P2 a = H[0];
		for (uint32_t i = 1; i + 1u < m; ++i) {
			const P2 b = H[i], c = H[i + 1u];
			const float bx = b.x - a.x, by = b.y - a.y, cx = c.x - a.x, cy = c.y - a.y;
			area += bx * cy - by * cx;
		}
		areas[slice_off[sorted_seg[begin]] + sorted_slice[begin]] = area / 2.0f;
	}
}

// The same for one big slice per workgroup (see BIG_TASK).  A ground slice holds 300 000 points and its hull ~40 edges; one
// workgroup = one CU tests every point against every edge, and that alone was 1.2 ms of a 3 ms pass that had the other 255 CUs
// idle at its end.  Nearly all of those points lie deep inside the hull that the slice's first points built, so:
//   * FILTER.  COARSE vertices of the current hull, evenly spaced by index, are a convex polygon inside it.  A point inside
//     that polygon by a margin (2^-14 of the coordinates' magnitude: ~600 x the rounding error of the reference's edge test)
//     is inside every later hull by the same margin less a few roundings -- hulls only grow, see below -- so Tree::insert
//     would return without a change: it is dropped after COARSE x 3 instructions.  The others (a few per cent) are
//     compacted, in slice order, into an LDS buffer, and only they go through the reference's scan.
//   * "HULLS ONLY GROW" holds while every insertion replaces ONE run of visible edges: the new hull then contains the old one
//     up to slivers as thin as the test's rounding error.  A point that is (numerically) collinear with a far edge can show
//     TWO runs, and the reference's splice (last start, last end: segmenting.rs:356-380) then cuts a piece of the hull off;
//     points dropped earlier could matter again.  Every insertion therefore counts its runs, and a slice that sees such an
//     insertion is done AGAIN from its first point with the filter off (bit for bit the reference either way).
//   * ... and while every edge has a unit normal.  An edge whose normal is NaN or zero -- a vertex at infinity, two equal
//     consecutive vertices -- never calls a point outside: the polygon is no convex polygon any more, and a later insertion
//     beside such an edge (ONE run, nothing to count) can turn a new edge's outside towards the old interior.  Found by the
//     soak (tests/soak_calculations_seeds.py, seed 531: a point at -inf in a 20 000-point slice): a hull that gets such an edge
//     sends its slice to the second attempt too.
#ifndef TREEE_FILL_PPT
#define TREEE_FILL_PPT 2
#endif
constexpr uint32_t COARSE = 12, BUF_CAP = 2048, BUF_GO = 1024, FILL_PPT = TREEE_FILL_PPT;
constexpr float COARSE_MARGIN = 6.103515625e-05f; // 2^-14
__device__ __forceinline__ void slice_hull_block(P2 *s_hull /* 2 * BIG_LDS + BUF_CAP entries */, const float *__restrict__ xyz,
                                                 const uint32_t *__restrict__ order, const uint32_t *__restrict__ sorted_seg,
                                                 const uint32_t *__restrict__ sorted_slice, const uint32_t *__restrict__ task_start, uint32_t n_tasks,
                                                 uint32_t n, const uint32_t *__restrict__ cls, uint32_t want, const uint32_t *__restrict__ slice_off,
                                                 const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ big_count,
                                                 P2 *__restrict__ spill, float *__restrict__ areas, const CrownPass &cp) {
	__shared__ uint32_t s_first, s_se[3], s_m, s_done, s_wcnt[FILL_PPT][BIG_THREADS / 64u], s_two_runs;
	__shared__ P2 s_p;
	__shared__ float s_coarse[COARSE][3]; // outward unit normal of a coarse edge and the bound its points' n . p must stay under
	if (blockIdx.x >= *big_count) return;
	const uint32_t tid = threadIdx.x, t = big_list[blockIdx.x];
	const uint32_t begin = task_start[t], end = t + 1u < n_tasks ? task_start[t + 1u] : n;
	const uint32_t count = end - begin;
	if (crown_slice_settled(cp, begin, sorted_seg, sorted_slice, slice_off, areas, tid == 0u)) return; // (uniform: before any barrier)
	P2 *const s_buf = s_hull + 2u * BIG_LDS;
	P2 *const big = spill + 2ull * begin + 4ull * t;
	constexpr uint32_t NONE = 0xFFFFFFFFu;
	uint32_t m = 0;
	P2 *H = s_hull, *N = s_hull + BIG_LDS; // generic pointers: LDS first, the task's HBM slot once the hull outgrows BIG_LDS
	for (int attempt = 0; attempt < 2; ++attempt) {
		const bool filter_on = attempt == 0;
		H = s_hull;
		N = s_hull + BIG_LDS;
		uint32_t cap = BIG_LDS;
		m = 0;
#ifdef TREEE_TEST_FORCE_SECOND_ATTEMPT // (validation build: every big slice is done twice, the second time without filter or drops)
		if (tid == 0) s_two_runs = 1u;
#else
		if (tid == 0) s_two_runs = 0u;
#endif
		// (a normal that is not a unit vector -- NaN or zero: a vertex at infinity, two equal consecutive vertices, an edge too long
		// for its squared length -- is an edge that never calls a point outside: the hull is then no convex polygon any more, an insertion next to such an edge can turn an edge's outside
		// towards the old interior, and points dropped before would matter again: the slice is done again without drops)
		auto normals = [&]() __attribute__((always_inline)) {
			for (uint32_t c = 0; c < m; c += BIG_THREADS) {
				const uint32_t j = c + tid;
				if (j < m) {
					const P2 nn = edge_normal(H[j], H[j + 1u < m ? j + 1u : 0u]);
					N[j] = nn;
					if (!unit_length(nn)) s_two_runs = 1u;
				}
			}
			__syncthreads();
		};
		// move H[src .. src + len) to H[dst ..): every thread reads its element before any thread writes, chunks in the
		// direction that never overwrites an unread source
		auto move = [&](uint32_t src, uint32_t dst, uint32_t len) __attribute__((always_inline)) {
			if (src == dst || len == 0u) return;
			const uint32_t chunks = (len + BIG_THREADS - 1u) / BIG_THREADS;
			for (uint32_t q = 0; q < chunks; ++q) {
				const uint32_t c = dst < src ? q : chunks - 1u - q;
				const uint32_t k = c * BIG_THREADS + tid;
				P2 v = {0.0f, 0.0f};
				if (k < len) v = H[src + k];
				__syncthreads();
				if (k < len) H[dst + k] = v;
				__syncthreads();
			}
		};
		// the coarse polygon of the current hull (any later hull contains it); m < COARSE or a degenerate edge (NaN bound)
		// simply lets every point through
		auto build_coarse = [&]() __attribute__((always_inline)) {
			if (tid < COARSE) {
				float nx = 0.0f, ny = 0.0f, bound = -INFINITY; // (nothing is under -inf: no point is dropped)
				if (filter_on && m >= COARSE) {
					const P2 a = H[(uint32_t)((uint64_t)tid * m / COARSE)], b = H[(uint32_t)((uint64_t)(tid + 1u) * m / COARSE) % m];
					const P2 nrm = edge_normal(a, b);
					nx = nrm.x;
					ny = nrm.y;
					bound = (nx * a.x + ny * a.y) - COARSE_MARGIN * (fabsf(a.x) + fabsf(a.y));
				}
				s_coarse[tid][0] = nx;
				s_coarse[tid][1] = ny;
				s_coarse[tid][2] = bound;
			}
			__syncthreads();
		};
		// The first BIG_ONE_BY_ONE points: wave 0 alone, an edge per lane (insert_one_by_one) -- most of a long slice's changes
		// happen there, and a change in the rounds below costs the whole workgroup a scan of the hull and seven barriers
		__syncthreads(); // (s_two_runs; and, on the second attempt, everybody is done with the first)
		// wave 0's tools for inserting points one by one (insert_one_by_one): hull in LDS, at most 64 edges before an insertion
		bool spliced = false;
		auto wave_sync = [&]() __attribute__((always_inline)) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
		};
		auto wave_normals = [&]() __attribute__((always_inline)) {
			for (uint32_t c = 0; c < m; c += 64u) {
				const uint32_t j = c + tid;
				if (j < m) {
					const P2 nn = edge_normal(H[j], H[j + 1u < m ? j + 1u : 0u]);
					N[j] = nn;
					if (!unit_length(nn)) s_two_runs = 1u; // (see `normals`)
				}
			}
			wave_sync();
		};
		auto wave_splice = [&](uint32_t s0, uint32_t e0, float px, float py, bool two_runs) __attribute__((always_inline)) { // one chunk
			const bool wraps = e0 < s0;                                                     // H' = H[e0 ..= s0] + [point]
			const uint32_t len = wraps ? s0 - e0 + 1u : m - e0, dst = wraps ? 0u : s0 + 2u; // else H[0 ..= s0] + [point] + H[e0 .. m)
			P2 v = {0.0f, 0.0f};
			if (tid < len) v = H[e0 + tid];
			__builtin_amdgcn_wave_barrier();
			if (tid < len && dst != e0) H[dst + tid] = v;
			if (tid == 0) H[wraps ? len : s0 + 1u] = P2{px, py};
			if (tid == 0 && two_runs) s_two_runs = 1u;
			m = wraps ? len + 1u : dst + len;
			spliced = true;
			wave_sync();
			wave_normals();
		};
		auto wave_new_tree = [&](float px, float py) __attribute__((always_inline)) {
			if (tid == 0) {
				H[0] = P2{px, py};
				H[1] = P2{px + 0.1f, py};
				H[2] = P2{px, py + 0.1f};
			}
			m = 3;
			spliced = true;
			wave_sync();
			wave_normals();
		};
		if (tid < 64u) {
			uint32_t done = 0;
			for (uint32_t base = 0; base < count && base < BIG_ONE_BY_ONE && m <= 64u; base += 64u) {
				const uint32_t i = base + tid;
				bool pending = i < count;
				P2 p = {0.0f, 0.0f};
				if (pending) {
					const uint32_t o = order[begin + i];
					p.x = xyz[3ull * o + 0];
					p.y = xyz[3ull * o + 2];
					if (cls) pending = cls[o] == want;
				}
				insert_one_by_one(H, N, m, p, pending, tid, wave_splice, wave_new_tree);
				const uint64_t left = __ballot(pending); // the hull outgrew 64 edges inside this batch: the rounds below take over there
				done = left ? base + (uint32_t)__builtin_ctzll(left) : (base + 64u < count ? base + 64u : count);
				if (left) break;
			}
			if (tid == 0) {
				s_m = m;
				s_done = done;
			}
		}
		__syncthreads();
		m = s_m;
		uint32_t next = s_done; // the first point of the slice that has not been looked at
		bool wide = false;      // the next fill round takes FILL_PPT points per thread
		build_coarse();
		while (next < count) {
			// ---- fill: rounds of FILL_PPT x 1024 points through the filter, the survivors behind one another in the buffer.  A
			// round is two dependent gathers (order -> x, z) and was ~4 us of latency per 1 024 points: four points per thread have
			// their loads in flight together.  A round whose survivors would not fit is not committed and done again point by
			// point per thread (the filter is off, the hull still small, or the slice just left the coarse polygon)
			uint32_t nbuf = 0;
			while (next < count && nbuf < BUF_GO) {
				const uint32_t ppt = wide ? FILL_PPT : 1u;
				bool keep[FILL_PPT];
				P2 p[FILL_PPT];
#pragma unroll
				for (uint32_t q = 0; q < FILL_PPT; ++q) {
					const uint32_t i = next + q * BIG_THREADS + tid;
					keep[q] = q < ppt && i < count;
					p[q] = P2{0.0f, 0.0f};
					if (keep[q]) {
						const uint32_t o = order[begin + i];
						p[q].x = xyz[3ull * o + 0];
						p[q].y = xyz[3ull * o + 2];
						if (cls) keep[q] = cls[o] == want;
					}
				}
				uint64_t km[FILL_PPT];
#pragma unroll
				for (uint32_t q = 0; q < FILL_PPT; ++q) {
					if (keep[q]) {
						const float mag = COARSE_MARGIN * (fabsf(p[q].x) + fabsf(p[q].y));
						bool inside = true;
#pragma unroll
						for (uint32_t j = 0; j < COARSE; ++j) inside = inside && (s_coarse[j][0] * p[q].x + s_coarse[j][1] * p[q].y) + mag <= s_coarse[j][2];
						keep[q] = !inside; // (NaN anywhere compares false: the point stays)
					}
					km[q] = __ballot(keep[q]);
					if ((tid & 63u) == 0u) s_wcnt[q][tid >> 6] = (uint32_t)__popcll(km[q]);
				}
				__syncthreads();
				uint32_t before[FILL_PPT], total = 0; // slice order inside a round: row q (1 024 points) after row q - 1, waves in order
#pragma unroll
				for (uint32_t q = 0; q < FILL_PPT; ++q) {
					before[q] = nbuf + total;
#pragma unroll
					for (uint32_t w = 0; w < BIG_THREADS / 64u; ++w) {
						const uint32_t c = s_wcnt[q][w];
						if (w < (tid >> 6)) before[q] += c;
						total += c;
					}
				}
				const bool fits = nbuf + total <= BUF_CAP; // (uniform; always true with one point per thread: nbuf < BUF_GO)
				if (fits) {
#pragma unroll
					for (uint32_t q = 0; q < FILL_PPT; ++q)
						if (keep[q]) s_buf[before[q] + (uint32_t)__popcll(km[q] & ((1ull << (tid & 63u)) - 1ull))] = p[q];
					nbuf += total;
					next += ppt * BIG_THREADS;
					wide = total * 4u <= BIG_THREADS * ppt; // few survivors: the next round can be a wide one
				} else {
					wide = false;
				}
				__syncthreads(); // (s_wcnt is written again by the next round; s_buf is read below)
			}
			// ---- the reference's insertion over the buffered points, 1024 per batch ------------------------------------------
			bool changed = false;
			for (uint32_t sb = 0; sb < nbuf; sb += BIG_THREADS) {
				bool pending = sb + tid < nbuf;
				P2 p = pending ? s_buf[sb + tid] : P2{0.0f, 0.0f};
				// A change of the hull costs the rounds below a scan by every thread and seven barriers (~6 us), and a ground slice
				// makes ~150 of them after its first points.  While the hull has at most 64 edges the batch is therefore tested ONCE,
				// in parallel, against the real hull with the filter's margin -- a point inside by the margin is inside every later
				// hull (see above) and is dropped --, and the few that remain are inserted one by one by wave 0, an edge per lane.
				// (Not on the second attempt: there a hull has been cut, and nothing may be dropped.)
				if (filter_on && m != 0u && m <= 64u && cap == BIG_LDS) {
					bool rest = pending;
					if (pending) {
						const float mag = fabsf(p.x) + fabsf(p.y);
						bool inside = true;
						for (uint32_t e = 0; e < m; ++e) {
							const P2 hv = H[e], nv = N[e];
							const float fx = p.x - hv.x, fy = p.y - hv.y;
							inside = inside && nv.x * fx + nv.y * fy <= -COARSE_MARGIN * (mag + fabsf(hv.x) + fabsf(hv.y));
						}
						rest = !inside; // (NaN anywhere compares false: the point stays)
#ifdef TREEE_TEST_RESIDUE_KEEPS_ALL // (validation build: the one-by-one path without its drops; how seed 531's cause was found)
						rest = true;
#endif
					}
					const uint64_t km = __ballot(rest);
					if ((tid & 63u) == 0u) s_wcnt[0][tid >> 6] = (uint32_t)__popcll(km);
					__syncthreads(); // (every thread holds its s_buf[sb + tid] by now: the compaction below is in place)
					uint32_t before = 0, total = 0;
#pragma unroll
					for (uint32_t w = 0; w < BIG_THREADS / 64u; ++w) {
						const uint32_t c = s_wcnt[0][w];
						if (w < (tid >> 6)) before += c;
						total += c;
					}
					if (rest) s_buf[sb + before + (uint32_t)__popcll(km & ((1ull << (tid & 63u)) - 1ull))] = p;
					__syncthreads();
					if (tid < 64u) {
						uint32_t done = 0;
						spliced = false;
						for (uint32_t base = 0; base < total && m <= 64u; base += 64u) {
							bool pend = base + tid < total;
							const P2 q = pend ? s_buf[sb + base + tid] : P2{0.0f, 0.0f};
							insert_one_by_one(H, N, m, q, pend, tid, wave_splice, wave_new_tree);
							const uint64_t left = __ballot(pend); // the hull outgrew 64 edges: the rounds below take the others
							done = left ? base + (uint32_t)__builtin_ctzll(left) : (base + 64u < total ? base + 64u : total);
							if (left) break;
						}
						if (tid == 0) {
							s_m = m;
							s_done = done;
							s_se[0] = spliced ? 1u : 0u;
						}
					}
					__syncthreads();
					m = s_m;
					changed = changed || s_se[0] != 0u;
					pending = tid >= s_done && tid < total; // (nobody, unless the hull outgrew 64 edges on the way)
					p = pending ? s_buf[sb + tid] : P2{0.0f, 0.0f};
				}
				for (;;) {
					if (tid == 0) s_first = NONE;
					__syncthreads();
					bool has_s = false, has_e = false;
					uint32_t st = 0, en = 0, runs = 0;
					if (m == 0) { // Tree::new from the first point of the slice
						if (pending) atomicMin(&s_first, tid);
					} else if (pending) { // (a thread with nothing to test skips the scan; the barriers below are outside)
						bool last = edge_outside(H[m - 1u], N[m - 1u], p);
						auto edge = [&](uint32_t e, P2 hv, P2 nv) __attribute__((always_inline)) {
							const bool out = edge_outside(hv, nv, p);
							if (!last && out) {
								st = e;
								has_s = true;
								++runs;
							} else if (last && !out) {
								en = e;
								has_e = true;
							}
							last = out;
						};
						uint32_t e = 0;
						for (; e + 4u <= m; e += 4u) { // four edges per trip: their eight reads are in flight together
							P2 hv[4], nv[4];
#pragma unroll
							for (int u = 0; u < 4; ++u) {
								hv[u] = H[e + u];
								nv[u] = N[e + u];
							}
#pragma unroll
							for (int u = 0; u < 4; ++u) edge(e + u, hv[u], nv[u]);
						}
						for (; e < m; ++e) edge(e, H[e], N[e]);
						if (has_s && has_e) atomicMin(&s_first, tid);
					}
					__syncthreads();
					const uint32_t f = s_first; // the first one in slice order (uniform)
					if (f == NONE) break;       // the whole batch is settled
					if (tid == f) {
						s_se[0] = st;
						s_se[1] = en;
						s_se[2] = runs;
						s_p = p;
					}
					__syncthreads();
					const uint32_t s0 = s_se[0], e0 = s_se[1];
					const P2 np = s_p;
					if (tid == 0 && m != 0 && s_se[2] > 1u) s_two_runs = 1u;
					// (new size first, as in slice_hull_wave: the move to HBM happens before any write of either branch)
					if (m != 0 && (e0 < s0 ? s0 - e0 + 2u : s0 + 2u + (m - e0)) > cap) {
						for (uint32_t c = 0; c < m; c += BIG_THREADS)
							if (c + tid < m) big[c + tid] = H[c + tid];
						__syncthreads();
						H = big;
						N = big + (count + 2u);
						cap = NONE;
					}
					if (m == 0) {
						if (tid == 0) {
							H[0] = np;
							H[1] = P2{np.x + 0.1f, np.y};
							H[2] = P2{np.x, np.y + 0.1f};
						}
						m = 3;
					} else if (e0 < s0) { // H' = H[e0 ..= s0] + [point]
						const uint32_t keep = s0 - e0 + 1u;
						move(e0, 0u, keep);
						if (tid == 0) H[keep] = np;
						m = keep + 1u;
					} else { // H' = H[0 ..= s0] + [point] + H[e0 .. m)
						const uint32_t tail = m - e0, dst = s0 + 2u;
						move(e0, dst, tail);
						if (tid == 0) H[s0 + 1u] = np;
						m = dst + tail;
					}
					pending = pending && tid > f;
					changed = true;
					__syncthreads();
					normals();
				}
			}
			if (changed && next < count) build_coarse(); // (uniform: every thread saw the same s_first values)
		}
		__syncthreads();
		const uint32_t two_runs = s_two_runs;
		__syncthreads(); // (thread 0 clears the word at the top of the next attempt)
#ifdef TREEE_TEST_NO_SECOND_ATTEMPT // (validation build: shows that the tests notice a filtered slice whose hull was cut)
		break;
#endif
		if (!filter_on || two_runs == 0u) break; // (uniform)
	}
	if (tid == 0 && m >= 3u) {
		float area = 0.0f;
		const P2 a = H[0];
		for (uint32_t i = 1; i + 1u < m; ++i) {
			const P2 b = H[i], c = H[i + 1u];
			const float bx = b.x - a.x, by = b.y - a.y, cx = c.x - a.x, cy = c.y - a.y;
			area += bx * cy - by * cx;
		}
		areas[slice_off[sorted_seg[begin]] + sorted_slice[begin]] = area / 2.0f;
	}
}

// One launch for both kinds of slice: workgroups [0, big_max) take the big slices (those past *big_count leave at once;
// being first in the grid, the long ones start first), the others take 16 ordinary slices each, one per wave.  As two
// launches the ~20 workgroups of a plot's ground slices kept 236 CUs waiting for 3 ms.
constexpr uint32_t WAVE_LDS = (2 * BIG_LDS + BUF_CAP) / 16; // P2 entries of the workgroup's LDS per wave in the short-slice role
static_assert(WAVE_LDS >= 4 * 64 && WAVE_LDS >= 2 * HULL_LDS, "four rows of 64 points, or one hull of HULL_LDS vertices with its normals");
// (8 waves per SIMD = two workgroups per CU: without the attribute the block role's 106 SGPRs allow seven, i.e. ONE workgroup)
__global__ __launch_bounds__(BIG_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_slice_hulls(const float *__restrict__ xyz, const uint32_t *__restrict__ order,
                                                             const uint32_t *__restrict__ sorted_seg, const uint32_t *__restrict__ sorted_slice,
                                                             const uint32_t *__restrict__ task_start, uint32_t n_tasks, uint32_t n,
                                                             const uint32_t *__restrict__ cls, uint32_t want, const uint32_t *__restrict__ slice_off,
                                                             const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ big_count, uint32_t big_max,
                                                             P2 *__restrict__ spill, float *__restrict__ areas, const CrownPass cp) {
	__shared__ P2 s_mem[2 * BIG_LDS + BUF_CAP];
	if (blockIdx.x < big_max) {
		slice_hull_block(s_mem, xyz, order, sorted_seg, sorted_slice, task_start, n_tasks, n, cls, want, slice_off, big_list, big_count, spill, areas, cp);
	} else {
		const uint32_t w = threadIdx.x >> 6;
		P2 *const mine = s_mem + w * WAVE_LDS; // the wave's share: four rows' points first, a long slice's hull and normals after
		const uint32_t t0 = ((blockIdx.x - big_max) * 16u + w) * 4u;
		const uint64_t left = slice_hulls_rows(mine, t0, xyz, order, sorted_seg, sorted_slice, task_start, n_tasks, n, cls, want, slice_off, areas, cp);
		for (uint32_t r = 0; r < 4u; ++r)
			if ((left >> (16u * r)) & 1ull) slice_hull_wave(mine, t0 + r, xyz, order, sorted_seg, sorted_slice, task_start, n_tasks, n, cls, want, slice_off, spill, areas, cp);
	}
}

// ---- SegmentInformation::new (calculations.rs:210-276), one thread per segment --------------------------------------
__global__ __launch_bounds__(256) void k_seg_info(const float *__restrict__ areas, const uint32_t *__restrict__ slice_off, const uint32_t *__restrict__ n_slices,
                                                   const float *__restrict__ seg_min, const float *__restrict__ seg_max, uint32_t nseg, float *__restrict__ info) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= nseg) return;
	const float *a = areas + slice_off[s];
	const uint32_t slices = n_slices[s];
	const float mn = seg_min[s * 3u + 1], mx = seg_max[s * 3u + 1];
	const float ground_max_search_height = 1.0f, ground_min_area_scale = 1.5f, min_crown_diameter = 2.0f;
	const uint32_t skip = as_usize_u32(1.0f / SLICE_WIDTH), take = as_usize_u32(10.0f / SLICE_WIDTH);
	bool have = false;
	float min_area = 0.5f;
	for (uint32_t i = skip; i < slices && i < skip + take; ++i) // min_by(total_cmp): the first minimum
		if (!have || f2total(a[i]) < f2total(min_area)) {
			min_area = a[i];
			have = true;
		}
	if (!have) min_area = 0.5f;
	min_area = fmaxf(min_area, 0.5f);
	const float thresh = min_area * ground_min_area_scale;
	bool has_ground = false;
	uint32_t ground = 0;
	const uint32_t gtake = as_usize_u32(ground_max_search_height / SLICE_WIDTH);
	for (uint32_t i = 0; i < slices && i < gtake; ++i)
		if (a[i] > thresh) {
			ground = i;
			has_ground = true;
			break;
		}
	uint32_t ground_sep = 0;
	if (has_ground)
		for (uint32_t i = ground; i < slices && i < slices / 2u; ++i)
			if (a[i] < thresh) {
				ground_sep = i;
				break;
			}
	const float pi = 3.14159265358979323846f, half = min_crown_diameter / 2.0f;
	const float min_crown_area = pi * (half * half);
	uint32_t crown_sep = 0;
	for (uint32_t i = ground_sep; i < slices; ++i)
		if (a[i] > min_crown_area) {
			crown_sep = i;
			break;
		}
	const float ground_sep_f = mn + (float)ground_sep * SLICE_WIDTH;
	const float crown_sep_f = mn + (float)crown_sep * SLICE_WIDTH;
	info[s * 4u + 0] = crown_sep_f - ground_sep_f; // trunk_height
	info[s * 4u + 1] = mx - crown_sep_f;           // crown_height
	info[s * 4u + 2] = ground_sep_f;
	info[s * 4u + 3] = crown_sep_f;
}

// SegmentData::new, classification (calculations.rs:149-163): 0 Ground, 1 Trunk, 2 Crown
__global__ __launch_bounds__(256) void k_classify(const float *__restrict__ xyz, const uint32_t *__restrict__ seg, uint32_t n, const float *__restrict__ seg_max,
                                                   const float *__restrict__ info, uint32_t *__restrict__ cls) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t sg = seg[i];
	const float crown_sep = seg_max[sg * 3u + 1] - info[sg * 4u + 1];
	const float ground_sep = crown_sep - info[sg * 4u + 0];
	const float y = xyz[3ull * i + 1];
	cls[i] = y < ground_sep ? 0u : (y < crown_sep ? 1u : 2u);
}

__device__ __forceinline__ float approximate_diameter(float area) { return 2.0f * sqrtf(area / 3.14159265358979323846f); } // calculations.rs:460-462

// SegmentInformation::update (calculations.rs:303-313): slices[i] = diameter(area_i) / diameter(largest crown slice)
__global__ __launch_bounds__(256) void k_seg_expansion(float *__restrict__ areas, const uint32_t *__restrict__ slice_off, const uint32_t *__restrict__ n_slices,
                                                        const float *__restrict__ seg_min, const float *__restrict__ info, uint32_t nseg) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= nseg) return;
	float *a = areas + slice_off[s];
	const uint32_t slices = n_slices[s];
	const uint32_t skip = as_usize_u32((info[s * 4u + 3] - seg_min[s * 3u + 1]) / SLICE_WIDTH);
	bool have = false;
	float crown_area = 0.0f;
	for (uint32_t i = skip; i < slices; ++i) // max_by(total_cmp): the last maximum
		if (!have || f2total(a[i]) >= f2total(crown_area)) {
			crown_area = a[i];
			have = true;
		}
	if (!have) crown_area = 0.0f;
	const float crown_diameter = approximate_diameter(crown_area);
	for (uint32_t i = 0; i < slices; ++i) a[i] = approximate_diameter(a[i]) / crown_diameter;
}
__global__ __launch_bounds__(256) void k_point_properties(const float *__restrict__ xyz, const uint32_t *__restrict__ seg, const uint32_t *__restrict__ slice,
                                                           uint32_t n, const float *__restrict__ ratios, const uint32_t *__restrict__ slice_off,
                                                           const float *__restrict__ seg_min, const float *__restrict__ seg_max,
                                                           float *__restrict__ expansion, float *__restrict__ height) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t sg = seg[i];
	const float mn = seg_min[sg * 3u + 1], mx = seg_max[sg * 3u + 1];
	if (expansion) expansion[i] = ratios[slice_off[sg] + slice[i]];
	if (height) height[i] = (xyz[3ull * i + 1] - mn) / (mx - mn);
}

// ---- DisplayData::new (interactive/mod.rs:1002-1039) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_i32(int32_t *__restrict__ p, uint32_t n, int32_t v) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void k_seg_max_expansion(const float *__restrict__ expansion, const uint32_t *__restrict__ seg, uint32_t n,
                                                            uint32_t stride, int32_t *best /* read plainly: an L1 copy is at worst the previous launch's sample */) {
	const uint64_t i64 = (uint64_t)(blockIdx.x * 256u + threadIdx.x) * stride;
	const uint32_t i = i64 < n ? (uint32_t)i64 : n;
	// max_by(total_cmp): equal keys are identical bits.  The stored maximum only grows, so a value that does not beat even a
	// stale copy of it has nothing to add -- but the half million threads resident when the kernel starts all read the
	// initial word, and their atomics on a ground segment's ONE word are served one after the other (1.17 ms of a 15 M-point
	// cloud's 1.3, 5.4 ms for a cloud that is one segment).  Two remedies: a wave whose lanes share the segment reduces first and
	// sends one atomic (mixed waves -- segment borders, unsorted clouds -- keep one per lane that still beats the stored value);
	// and the kernel runs twice, first over every `stride`-th point: a segment's expansions are its slices' (few distinct
	// values), so the sample nearly always holds a large segment's maximum and the full pass finds nothing left to send.
	const bool live = i < n;
	const uint32_t sg = live ? seg[i] : 0xFFFFFFFFu;
	int32_t key = live ? f2total(expansion[i]) : INT32_MIN;
	const uint32_t sg0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sg);
	if (__ballot(live && sg != sg0) == 0ull && sg0 != 0xFFFFFFFFu) { // (lane 0 is live whenever any lane is: i ascends with the lane)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o, 64));
		if ((threadIdx.x & 63u) == 0u && key > best[sg0]) atomicMax(&best[sg0], key);
		return;
	}
	if (!live || key <= best[sg]) return;
	atomicMax(&best[sg], key);
}
__device__ __forceinline__ uint32_t map_to_u32(float v) { // calculations.rs:455-457 (saturating cast, NaN -> 0)
	const float x = v * 4294967296.0f;
	if (!(x > 0.0f)) return 0u;
	if (x >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)x;
}
__global__ __launch_bounds__(256) void k_display_values(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ cls, const float *__restrict__ curve,
                                                         const float *__restrict__ expansion, const float *__restrict__ height, uint32_t n,
                                                         const int32_t *__restrict__ best, uint32_t *__restrict__ cls_val, uint32_t *__restrict__ curve_val,
                                                         uint32_t *__restrict__ exp_val, uint32_t *__restrict__ height_val) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	if (cls_val && cls) {
		const uint32_t c = cls[i];
		cls_val[i] = c == 0u ? 0xFFFFFFFFu / 8u : (c == 1u ? 0xFFFFFFFFu / 8u * 3u : 0xFFFFFFFFu / 8u * 6u);
	}
	if (curve_val && curve) curve_val[i] = map_to_u32(curve[i]);
	if (height_val && height) height_val[i] = map_to_u32(height[i]);
	if (exp_val && expansion) exp_val[i] = map_to_u32(expansion[i] / total2f(best[seg[i]]));
}

int bits_for(uint32_t count) {
	int bits = 1;
	while (bits < 32 && (1ull << bits) < count) ++bits;
	return bits;
}

} // namespace

extern "C" {

int32_t treee_segments_calculate_dev(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_seg, uint64_t n64, uint32_t nseg, float *d_seg_min,
                                     float *d_seg_max, float *d_seg_info, uint32_t *d_cls, float *d_expansion, float *d_height) {
	if (!ctx) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate_dev: ctx is null");
	if (nseg == 0 || nseg > (1u << 24)) return set_err(TREEE_ERR_INVALID, "n_segments %u outside [1, 2^24]", nseg);
	if (n64 >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "n=%llu exceeds 2^31-1 points per shard", (unsigned long long)n64);
	if (n64 && (!d_xyz || !d_seg)) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate_dev: null array");
	if (!d_seg_min || !d_seg_max || !d_seg_info) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate_dev: null per-segment output");
	if ((d_expansion || d_height) && !d_cls) return set_err(TREEE_ERR_INVALID, "expansion / height need the classification array");
	const uint32_t n = (uint32_t)n64;
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	hipStream_t st = ctx->stream;
	const uint32_t gn = div_up(n ? n : 1, 256), gs = div_up(nseg, 256);
	uint32_t *d_err = nullptr, *d_aabb = nullptr, *d_nsl = nullptr, *d_sloff = nullptr;
	uint32_t *d_total = nullptr;
	const size_t cnt = n ? n : 1;
	TREEE_TRY(ctx->alloc_t(4, &d_err));
	TREEE_TRY(ctx->alloc_t((size_t)nseg * AABB_ROW, &d_aabb));
	TREEE_TRY(ctx->alloc_t(nseg, &d_nsl));
	TREEE_TRY(ctx->alloc_t(nseg, &d_sloff));
	TREEE_TRY(ctx->alloc_t(4, &d_total));
	TREEE_HIP(hipMemsetAsync(d_err, 0, 16, st));
	const int sbits = bits_for(nseg);
	// 1. min / max of every segment (SegmentData::new)
	if (n) hipLaunchKernelGGL(k_check_segments, dim3(gn), dim3(256), 0, st, d_seg, n, nseg, d_err);
	hipLaunchKernelGGL(k_init_aabb, dim3(div_up((uint64_t)nseg * AABB_ROW, 256)), dim3(256), 0, st, d_aabb, nseg);
	TREEE_LAUNCH_CHECK();
	uint32_t *d_present = nullptr; // [s] != 0: the segment has a point
	TREEE_TRY(ctx->alloc_t((size_t)nseg + 1, &d_present));
	// straight from the cloud as it lies: two samples, then every point (k_seg_aabb_direct)
	TREEE_HIP(hipMemsetAsync(d_present, 0, ((size_t)nseg + 1) * 4, st));
	{
		const uint32_t strides[] = {TREEE_AABB_STRIDES}, runs[] = {TREEE_AABB_RUNS};
		static_assert(sizeof(strides) == sizeof(runs), "one run length per level");
		for (size_t l = 0; l < sizeof(strides) / sizeof(strides[0]); ++l) {
			const uint32_t stride = strides[l], run = runs[l];
			if (!(n && (stride == 1u || n >= 64ull * stride * run))) continue;
			static const bool no_lds = getenv("TREEE_AABB_NO_LDS") != nullptr; // A/B and test hook: every level through the table in global memory
			const uint64_t sample = div_up((uint64_t)n, (uint64_t)stride);
			if (!no_lds && nseg <= AABB_LDS_SEGMENTS && sample >= (1ull << 20)) { // (the dense levels: the copy of the table has to pay)
				const uint32_t grid = (uint32_t)std::min<uint64_t>(2ull * (uint64_t)ctx->num_cus, div_up(sample, (uint64_t)AABB_LDS_THREADS));
				hipLaunchKernelGGL(k_seg_aabb_lds, dim3(grid), dim3(AABB_LDS_THREADS), 0, st, d_xyz, d_seg, n, nseg, stride, run, d_aabb, d_present);
			} else {
				hipLaunchKernelGGL(k_seg_aabb_direct, dim3(div_up(div_up(n, stride) + run, 256)), dim3(256), 0, st, d_xyz, d_seg, n, nseg, stride, run, d_aabb, d_present);
			}
		}
	}
	hipLaunchKernelGGL(k_seg_finish_aabb, dim3(gs), dim3(256), 0, st, d_aabb, d_present, nseg, d_seg_min, d_seg_max, d_nsl, d_err);
	TREEE_LAUNCH_CHECK();
	{ // first host round trip: ids beyond the range would index past the per-segment arrays (stop here if any); the largest
	  // slice count of a segment sizes the sort key
		TREEE_HIP(hipMemcpyAsync(ctx->h_scalars, d_err, 8, hipMemcpyDeviceToHost, st));
		TREEE_HIP(hipStreamSynchronize(st));
		if (ctx->h_scalars[0] & 1u) return set_err(TREEE_ERR_INVALID, "segment id >= n_segments (%u)", nseg);
		if (ctx->h_scalars[0] & 2u)
			return set_err(TREEE_ERR_UNSUPPORTED, "a segment is taller than %u slices of %.1f m", (1u << SLICE_BITS) - 2u, (double)SLICE_WIDTH);
	}
	const int slbits = bits_for(ctx->h_scalars[1]); // slice < the largest count < 2^slbits
	TREEE_TRY(exclusive_scan_u32_dev(ctx, d_nsl, d_sloff, nseg, d_total));
	// 2. final order = (segment, slice, input order): stable sort by slice, then by segment
	uint32_t *d_slice = nullptr, *d_slice_s = nullptr, *d_permA = nullptr, *d_segA = nullptr, *d_segF = nullptr, *d_order = nullptr, *d_sliceF = nullptr;
	uint32_t *d_flag = nullptr, *d_task_of = nullptr, *d_task_start = nullptr;
	TREEE_TRY(ctx->alloc_t(cnt, &d_slice));
	TREEE_TRY(ctx->alloc_t(cnt, &d_slice_s));
	TREEE_TRY(ctx->alloc_t(cnt, &d_permA));
	TREEE_TRY(ctx->alloc_t(cnt, &d_segA));
	TREEE_TRY(ctx->alloc_t(cnt, &d_segF));
	TREEE_TRY(ctx->alloc_t(cnt, &d_order));
	TREEE_TRY(ctx->alloc_t(cnt, &d_sliceF));
	TREEE_TRY(ctx->alloc_t(cnt, &d_flag));
	TREEE_TRY(ctx->alloc_t(cnt, &d_task_of));
	TREEE_TRY(ctx->alloc_t(cnt, &d_task_start));
	uint32_t n_tasks = 0, total_slices = 0;
	if (n) {
		hipLaunchKernelGGL(k_slice_index, dim3(gn), dim3(256), 0, st, d_xyz, d_seg, n, d_seg_min, d_nsl, d_slice);
		TREEE_LAUNCH_CHECK();
		if (sbits + slbits <= 32) {
			// one stable sort by (segment << slbits | slice) is the same permutation as a stable sort by slice followed by a
			// stable sort by segment -- and leaves both keys in final order without the two gathers (0.5 ms of a 15 M-point cloud)
			hipLaunchKernelGGL(k_join_keys, dim3(gn), dim3(256), 0, st, d_seg, d_slice, n, (uint32_t)slbits, d_segA);
			TREEE_LAUNCH_CHECK();
			TREEE_TRY(sort_pairs_dev(ctx, d_segA, nullptr, n, sbits + slbits, d_slice_s, d_order));
			hipLaunchKernelGGL(k_split_keys, dim3(gn), dim3(256), 0, st, d_slice_s, n, (uint32_t)slbits, d_segF, d_sliceF);
		} else { // (millions of segments AND thousands of slices: two sorts)
			TREEE_TRY(sort_pairs_dev(ctx, d_slice, nullptr, n, slbits, d_slice_s, d_permA));
			hipLaunchKernelGGL(k_gather, dim3(gn), dim3(256), 0, st, d_seg, d_permA, n, d_segA);
			TREEE_LAUNCH_CHECK();
			TREEE_TRY(sort_pairs_dev(ctx, d_segA, d_permA, n, sbits, d_segF, d_order));
			hipLaunchKernelGGL(k_gather, dim3(gn), dim3(256), 0, st, d_slice, d_order, n, d_sliceF);
		}
		hipLaunchKernelGGL(k_task_flags, dim3(gn), dim3(256), 0, st, d_segF, d_sliceF, n, d_flag);
		TREEE_LAUNCH_CHECK();
		TREEE_TRY(exclusive_scan_u32_dev(ctx, d_flag, d_task_of, n, d_total + 1));
		hipLaunchKernelGGL(k_task_starts, dim3(gn), dim3(256), 0, st, d_flag, d_task_of, n, d_task_start);
		TREEE_LAUNCH_CHECK();
	}
	// the only other host round trip: how many slices and tasks there are (they size the next allocations)
	TREEE_HIP(hipMemcpyAsync(ctx->h_scalars, d_total, 8, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipMemcpyAsync(ctx->h_scalars + 2, d_err, 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	total_slices = ctx->h_scalars[0];
	n_tasks = n ? ctx->h_scalars[1] : 0u;
	if (ctx->h_scalars[2] & 2u)
		return set_err(TREEE_ERR_UNSUPPORTED, "a segment is taller than %u slices of %.1f m", (1u << SLICE_BITS) - 2u, (double)SLICE_WIDTH);
	float *d_areas = nullptr;
	P2 *d_spill = nullptr;
	uint32_t *d_big = nullptr, *d_nbig = nullptr;
	const uint32_t big_max = n / BIG_TASK + 1u; // no more slices than that can hold BIG_TASK points
	TREEE_TRY(ctx->alloc_t(big_max, &d_big));
	TREEE_TRY(ctx->alloc_t(4, &d_nbig));
	TREEE_HIP(hipMemsetAsync(d_nbig, 0, 4, st));
	if (n_tasks) hipLaunchKernelGGL(k_big_tasks, dim3(div_up(n_tasks, 256)), dim3(256), 0, st, d_task_start, n_tasks, n, d_big, d_nbig);
	const uint32_t big_grid = n_tasks < big_max ? n_tasks : big_max;
	TREEE_TRY(ctx->alloc_t((size_t)total_slices + 1, &d_areas));
	TREEE_TRY(ctx->alloc_t(2 * (size_t)n + 4ull * n_tasks + 1, &d_spill));
	// 3. get_size_areas over all points -> SegmentInformation::new -> classification
	TREEE_HIP(hipMemsetAsync(d_areas, 0, ((size_t)total_slices + 1) * 4, st)); // a slice without points has area 0.0
	if (n_tasks)
		hipLaunchKernelGGL(k_slice_hulls, dim3(big_grid + div_up(n_tasks, 64)), dim3(BIG_THREADS), 0, st, d_xyz, d_order, d_segF, d_sliceF, d_task_start, n_tasks, n,
		                   nullptr, 0u, d_sloff, d_big, d_nbig, big_grid, d_spill, d_areas, CrownPass{nullptr, nullptr, nullptr, nullptr, nullptr});
	hipLaunchKernelGGL(k_seg_info, dim3(gs), dim3(256), 0, st, d_areas, d_sloff, d_nsl, d_seg_min, d_seg_max, nseg, d_seg_info);
	if (n && d_cls) hipLaunchKernelGGL(k_classify, dim3(gn), dim3(256), 0, st, d_xyz, d_seg, n, d_seg_max, d_seg_info, d_cls);
	TREEE_LAUNCH_CHECK();
	// 4. SegmentInformation::update: crown-only slice areas -> expansion; height
	if (n && (d_expansion || d_height)) {
		float *d_crown = d_areas; // (unused when there is no expansion to write)
		if (d_expansion) {
			// the crown points' slice areas: one slice per segment needs its hull built again, the others are copies or 0.0 (CrownPass)
			TREEE_TRY(ctx->alloc_t((size_t)total_slices + 1, &d_crown));
			TREEE_HIP(hipMemsetAsync(d_crown, 0, ((size_t)total_slices + 1) * 4, st));
			hipLaunchKernelGGL(k_slice_hulls, dim3(big_grid + div_up(n_tasks, 64)), dim3(BIG_THREADS), 0, st, d_xyz, d_order, d_segF, d_sliceF, d_task_start, n_tasks, n,
			                   d_cls, 2u, d_sloff, d_big, d_nbig, big_grid, d_spill, d_crown, CrownPass{d_seg_min, d_seg_max, d_seg_info, d_nsl, d_areas});
			hipLaunchKernelGGL(k_seg_expansion, dim3(gs), dim3(256), 0, st, d_crown, d_sloff, d_nsl, d_seg_min, d_seg_info, nseg);
		}
		hipLaunchKernelGGL(k_point_properties, dim3(gn), dim3(256), 0, st, d_xyz, d_seg, d_slice, n, d_crown, d_sloff, d_seg_min, d_seg_max, d_expansion, d_height);
		TREEE_LAUNCH_CHECK();
	}
	return TREEE_OK;
}

int32_t treee_display_values_dev(treee_ctx *ctx, const uint32_t *d_seg, uint64_t n64, uint32_t nseg, const uint32_t *d_cls, const float *d_curve,
                                 const float *d_expansion, const float *d_height, uint32_t *d_cls_val, uint32_t *d_curve_val, uint32_t *d_exp_val,
                                 uint32_t *d_height_val) {
	if (!ctx) return set_err(TREEE_ERR_INVALID, "treee_display_values_dev: ctx is null");
	if (n64 == 0) return TREEE_OK;
	if (n64 >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large");
	if (nseg == 0 || nseg > (1u << 24)) return set_err(TREEE_ERR_INVALID, "n_segments %u outside [1, 2^24]", nseg);
	if (d_exp_val && (!d_expansion || !d_seg)) return set_err(TREEE_ERR_INVALID, "expansion values need the expansion and segment arrays");
	const uint32_t n = (uint32_t)n64;
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	int32_t *d_best = nullptr;
	TREEE_TRY(ctx->alloc_t(nseg, &d_best));
	if (d_exp_val) {
		// max_by(total_cmp).unwrap_or_default(): every queried segment has at least the querying point, so the start value
		// (the smallest key of the total order) never survives where it is read
		hipLaunchKernelGGL(k_init_i32, dim3(div_up(nseg, 256)), dim3(256), 0, ctx->stream, d_best, nseg, (int32_t)0x80000000);
		for (uint32_t stride : {50021u, 3001u, 61u, 1u}) // (a ladder of samples, as for the segments' extremes: k_seg_aabb_direct)
			if (stride == 1u || n >= 64ull * stride)
				hipLaunchKernelGGL(k_seg_max_expansion, dim3(div_up(div_up(n, stride), 256)), dim3(256), 0, ctx->stream, d_expansion, d_seg, n, stride, d_best);
	}
	hipLaunchKernelGGL(k_display_values, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_seg, d_cls, d_curve, d_expansion, d_height, n, d_best, d_cls_val,
	                   d_curve_val, d_exp_val, d_height_val);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_segments_calculate(const float *xyz, const uint32_t *segment, uint64_t n, uint32_t n_segments, float *seg_min_out, float *seg_max_out,
                                 float *seg_info_out, uint32_t *classification_out, float *expansion_out, float *height_out) {
	if (n && (!xyz || !segment)) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate: null argument");
	if (!seg_min_out || !seg_max_out || !seg_info_out) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate: null per-segment output");
	if (n_segments == 0 || n_segments > (1u << 24)) return set_err(TREEE_ERR_INVALID, "n_segments %u outside [1, 2^24]", n_segments);
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_xyz, d_min, d_max, d_info, d_exp, d_hgt;
	DevBuf<uint32_t> d_seg, d_cls;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_TRY(d_seg.alloc(n));
	TREEE_TRY(d_cls.alloc(n));
	TREEE_TRY(d_min.alloc((size_t)n_segments * 3));
	TREEE_TRY(d_max.alloc((size_t)n_segments * 3));
	TREEE_TRY(d_info.alloc((size_t)n_segments * 4));
	if (expansion_out) TREEE_TRY(d_exp.alloc(n));
	if (height_out) TREEE_TRY(d_hgt.alloc(n));
	if (n) {
		TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
		TREEE_HIP(hipMemcpyAsync(d_seg.p, segment, n * 4, hipMemcpyHostToDevice, ctx->stream));
	}
	TREEE_TRY(treee_segments_calculate_dev(ctx, d_xyz.p, d_seg.p, n, n_segments, d_min.p, d_max.p, d_info.p, d_cls.p, expansion_out ? d_exp.p : nullptr,
	                                       height_out ? d_hgt.p : nullptr));
	hipStream_t st = ctx->stream;
	TREEE_HIP(hipMemcpyAsync(seg_min_out, d_min.p, (size_t)n_segments * 12, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipMemcpyAsync(seg_max_out, d_max.p, (size_t)n_segments * 12, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipMemcpyAsync(seg_info_out, d_info.p, (size_t)n_segments * 16, hipMemcpyDeviceToHost, st));
	if (n && classification_out) TREEE_HIP(hipMemcpyAsync(classification_out, d_cls.p, n * 4, hipMemcpyDeviceToHost, st));
	if (n && expansion_out) TREEE_HIP(hipMemcpyAsync(expansion_out, d_exp.p, n * 4, hipMemcpyDeviceToHost, st));
	if (n && height_out) TREEE_HIP(hipMemcpyAsync(height_out, d_hgt.p, n * 4, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st));
	return TREEE_OK;
}

} // extern "C"
