// common.h -- internals shared by the HIP translation units of libtreee_hip.so.
// gfx950 (CDNA4) only: wave = 64 lanes, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/treee_hip.h"

namespace treee {

constexpr int WAVE = 64;
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

// ---- errors ---------------------------------------------------------------------------
int32_t set_err(int32_t code, const char *fmt, ...);
#define TREEE_HIP(expr)                                                                          \
	do {                                                                                         \
		hipError_t e__ = (expr);                                                                 \
		if (e__ != hipSuccess)                                                                   \
			return ::treee::set_err(TREEE_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
			                        hipGetErrorString(e__), __FILE__, __LINE__);                \
	} while (0)
#define TREEE_TRY(expr)                                                                          \
	do {                                                                                         \
		int32_t rc__ = (expr);                                                                   \
		if (rc__ != TREEE_OK) return rc__;                                                       \
	} while (0)
#define TREEE_LAUNCH_CHECK() TREEE_HIP(hipGetLastError())

inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
// host threads worth starting for the library's host-side phases (segmenting, LAZ): the hardware's count, or the CPU quota
// of the cgroup when that is smaller (a container that shows 256 CPUs and grants 16 only gains jitter from 256 threads)
unsigned host_threads();

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 30-bit Hilbert index of three 10-bit cell coordinates (Skilling's "transpose" form: J. Skilling, Programming the Hilbert
// curve, AIP Conf. Proc. 707, 2004).  Like the Morton key it is a prefix code of the octree cells -- the top 3 l bits name
// the level-l cell -- but consecutive indices are always face neighbours, so a run of consecutive points of the sorted order
// is one spatially contiguous blob.  The kNN structure orders its buckets by it (knn.hip); the octree keeps Morton.
__device__ __host__ inline uint32_t hilbert30(uint32_t x0, uint32_t x1, uint32_t x2) {
	uint32_t X[3] = {x0, x1, x2};
	for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
		const uint32_t P = Q - 1u;
		for (int i = 0; i < 3; ++i) {
			if (X[i] & Q)
				X[0] ^= P;
			else {
				const uint32_t t = (X[0] ^ X[i]) & P;
				X[0] ^= t;
				X[i] ^= t;
			}
		}
	}
	X[1] ^= X[0];
	X[2] ^= X[1];
	uint32_t t = 0;
	for (uint32_t Q = 512u; Q > 1u; Q >>= 1)
		if (X[2] & Q) t ^= Q - 1u;
	X[0] ^= t;
	X[1] ^= t;
	X[2] ^= t;
	uint32_t h = 0; // transpose -> index: bit b of X[0] is the most significant of the b-th triple
	for (int b = 9; b >= 0; --b) h = (h << 3) | (((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u);
	return h;
}
// every third bit of a 30-bit Morton key -> 10 bits (inverse of the key's spread)
__device__ __host__ inline uint32_t compact3(uint32_t v) {
	v &= 0x09249249u;
	v = (v | (v >> 2)) & 0x030C30C3u;
	v = (v | (v >> 4)) & 0x0300F00Fu;
	v = (v | (v >> 8)) & 0x030000FFu;
	v = (v | (v >> 16)) & 0x000003FFu;
	return v;
}
#define TREEE_HILBERT_LUT_INIT \
	    64, 135, 195, 260, 321, 390, 2, 5, \
	    448, 771, 705, 66, 1095, 324, 1222, 69, \
	    580, 519, 133, 710, 387, 1152, 130, 1217, \
	    518, 711, 197, 964, 1153, 1216, 194, 3, \
	    704, 449, 963, 258, 1223, 1094, 4, 261, \
	    256, 835, 903, 68, 1, 322, 1286, 325, \
	    644, 199, 131, 1024, 389, 6, 386, 1281, \
	    0, 257, 1287, 902, 1475, 450, 708, 453, \
	    198, 7, 1025, 1280, 517, 1476, 514, 707, \
	    578, 961, 581, 1414, 643, 192, 132, 1031, \
	    642, 1473, 579, 512, 645, 1350, 388, 1159, \
	    320, 391, 65, 134, 515, 452, 706, 709, \
	    966, 773, 1409, 770, 263, 836, 896, 67, \
	    1478, 837, 455, 772, 1345, 834, 1088, 323, \
	    1412, 901, 1479, 454, 1283, 898, 1344, 1089, \
	    774, 577, 965, 962, 71, 128, 196, 259, \
	    1026, 1411, 513, 1472, 1029, 1284, 1158, 1351, \
	    1348, 1093, 1219, 1090, 967, 262, 1408, 897, \
	    1154, 1347, 1157, 1220, 193, 960, 1030, 1415, \
	    1156, 1091, 1221, 1218, 775, 576, 70, 129, \
	    1028, 899, 839, 640, 1285, 1282, 326, 385, \
	    1346, 1349, 1155, 1092, 833, 646, 768, 583, \
	    1410, 1413, 769, 582, 1027, 900, 832, 647, \
	    838, 641, 327, 384, 1477, 1474, 516, 451
// The same index, three bits per step: the curve Skilling's algorithm draws is a 24-state machine over the octant digits of
// the Morton key (derived from hilbert30 by enumerating the octant -> index-digit maps of all cell prefixes: 24 distinct maps,
// deterministic transitions, checked on 200 000 random cells; tests/cpp/hilbert_host_test.hip pins both forms to the oracle).
// Entry [state][octant] = (8 * next state) << 3 | index digit.  ~6 operations and one table read per level instead of ~25.
__device__ __host__ inline uint32_t hilbert_of_morton(uint32_t key) {
	constexpr uint16_t LUT[24 * 8] = {TREEE_HILBERT_LUT_INIT};
	uint32_t s = 0, h = 0; // s = 8 * state
#pragma unroll
	for (int sh = 27; sh >= 0; sh -= 3) {
		const uint32_t e = LUT[s + ((key >> sh) & 7u)];
		h = (h << 3) | (e & 7u);
		s = e >> 3;
	}
	return h;
}
// (device kernels that convert many keys keep the 384-byte table in LDS: a read of it is then an LDS access, not a cached
// global load per level)
__device__ __forceinline__ uint32_t hilbert_of_morton_lut(uint32_t key, const uint16_t *lut) {
	uint32_t s = 0, h = 0;
#pragma unroll
	for (int sh = 27; sh >= 0; sh -= 3) {
		const uint32_t e = lut[s + ((key >> sh) & 7u)];
		h = (h << 3) | (e & 7u);
		s = e >> 3;
	}
	return h;
}

// Ten further levels of the same curve: the index digits of a point's LOW Morton key (the 2^-20 sub-cells inside its 2^-10
// cell) continued from the state the cell's own ten levels leave the machine in -- the low 30 bits of the 60-bit Hilbert index
// whose high 30 bits hilbert_of_morton gives.  Ordering a crowded cell's points by it keeps a run of consecutive points one
// compact blob at every scale, where the low Morton key itself (rounds 1-4) brings the Z curve's jumps back inside the cell.
__device__ __forceinline__ uint32_t hilbert_low_of_morton_lut(uint32_t key_hi, uint32_t key_lo, const uint16_t *lut) {
	uint32_t s = 0, h = 0;
#pragma unroll
	for (int sh = 27; sh >= 0; sh -= 3) s = lut[s + ((key_hi >> sh) & 7u)] >> 3;
#pragma unroll
	for (int sh = 27; sh >= 0; sh -= 3) {
		const uint32_t e = lut[s + ((key_lo >> sh) & 7u)];
		h = (h << 3) | (e & 7u);
		s = e >> 3;
	}
	return h;
}

// Morton key pieces (definition: oracle/octree_oracle.cpp): q = clamp(floor(((p - c) / size) * 1024), 0, 1023), NaN -> 0
__device__ __forceinline__ uint32_t quant10(float p, float c, float size) {
	float t = ((p - c) / size) * 1024.0f; // IEEE divide (hipcc default is correctly rounded)
	if (!(t >= 0.0f)) return 0u;          // negative or NaN
	if (t >= 1024.0f) return 1023u;
	return (uint32_t)t;
}
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
	v &= 0x3FFu;
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
__device__ __forceinline__ uint32_t morton30(float x, float y, float z, float cx, float cy, float cz, float size) {
	return spread3(quant10(x, cx, size)) | (spread3(quant10(y, cy, size)) << 1) | (spread3(quant10(z, cz, size)) << 2);
}

// inclusive wave scan (64 lanes) of u32
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
	const int lane = lane_id();
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		uint32_t t = __shfl_up(v, o, 64);
		if (lane >= o) v += t;
	}
	return v;
}

// exclusive block scan for blockDim.x == 256 (4 waves); `part` = 4 u32 of LDS scratch.
// Returns exclusive prefix; *total (optional) receives the block sum.  Contains barriers.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *part, uint32_t *total) {
	const int lane = lane_id(), w = threadIdx.x >> 6;
	uint32_t inc = wave_incl_scan(v);
	if (lane == 63) part[w] = inc;
	__syncthreads();
	uint32_t base = 0, sum = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		uint32_t p = part[i];
		if (i < w) base += p;
		sum += p;
	}
	__syncthreads();
	if (total) *total = sum;
	return base + inc - v;
}

} // namespace treee

// ---- context -----------------------------------------------------------------------------
struct treee_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	// workspace arena: bump allocation inside blocks; reset at the start of every API call.
	struct Block {
		char *p;
		size_t cap;
	};
	std::vector<Block> blocks;
	size_t cur = 0; // block index
	size_t off = 0; // offset in blocks[cur]
	int num_cus = 256;
	// pinned landing area for the few scalars a build brings back (node counts, flags, the cube): an asynchronous
	// device-to-host copy must never target a stack frame that an early error return can unwind before the copy runs
	uint32_t *h_scalars = nullptr; // [64]

	// pool of persistent blocks (the arrays owned by kNN / octree handles).  Reuse is ordered by
	// this context's stream, so a block can be handed out again without synchronising.
	struct PoolBlock {
		char *p;
		size_t cap;
	};
	std::vector<PoolBlock> pool_idle;
	// idle bytes kept before the oldest blocks go back to the driver: 8 GiB, or a quarter of the device's memory when that is
	// more (72 GB of an MI355X) -- the arrays of a 1 G-point cloud are 4 - 16 GB EACH, and a block that comes back from
	// hipMalloc fresh is slow at its first touch (kNN over 1.005 G points: 2.7 s on fresh arrays, 0.52 s on reused ones)
	size_t pool_idle_limit = size_t(8) << 30;
	std::unordered_map<void *, size_t> pool_live;
	std::mutex pool_mu;

	// Fork / join inside one API call: independent work (the octree's ~50 small launches next to the record gather) runs
	// on an auxiliary stream of the context.  fork(): aux waits for what `stream` holds so far.  on_aux(true/false):
	// launches of the internal stage functions go to aux / back to `stream`.  join(): `stream` waits for aux.  Every
	// build joins before it returns, so outside an API call the context has exactly one stream, as documented.
	hipStream_t aux = nullptr, main_stream = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr; // (ev_mid: a point on aux the HOST waits for while aux goes on)
	int32_t fork();
	void on_aux(bool yes);
	int32_t join();

	int32_t reset();                         // start of an API call: coalesce + rewind
	int32_t alloc(size_t bytes, void **out); // 256-B aligned
	template <class T> int32_t alloc_t(size_t count, T **out) {
		void *p = nullptr;
		int32_t rc = alloc(count * sizeof(T), &p);
		*out = (T *)p;
		return rc;
	}
};

namespace treee {

// ---- handles used from a stream other than their owner's ----------------------------------------------
// A handle's arrays come from its owner context's pool and go back to it on free, where the next pool_alloc hands
// them out again WITHOUT synchronising: reuse is ordered by the owner's stream.  That only covers work the owner's
// stream can see, but every _dev entry point accepts any context.  Stream contract (treee_hip.h, "Streams"):
//   use_begin  -- the using stream waits for everything the owner's stream has been given so far (the build)
//   use_end    -- an event on the using stream remembers the last foreign use
//   retire     -- before the arrays go back to the pool the owner's stream waits for those events
// Same stream (or the same context): all three are no-ops.
struct HandleSync {
	std::mutex mu;
	std::vector<std::pair<hipStream_t, hipEvent_t>> users; // last use per foreign stream
	hipEvent_t ready = nullptr; // optional: "the arrays are complete", recorded on the owner's stream by mark_ready.  With it
	                            // a foreign stream waits for the build only, not for what the owner enqueued afterwards
	HandleSync() = default;
	HandleSync(const HandleSync &) = delete;
	HandleSync &operator=(const HandleSync &) = delete;
	~HandleSync();
};
bool ctx_is_live(treee_ctx *ctx);
int32_t mark_ready(treee_ctx *owner, HandleSync *hs);
int32_t use_begin(treee_ctx *ctx, treee_ctx *owner, HandleSync *hs);
int32_t use_end(treee_ctx *ctx, treee_ctx *owner, HandleSync *hs);
void retire(treee_ctx *owner, HandleSync *hs);
void wait_users(treee_ctx *owner, HandleSync *hs); // order the owner behind the uses so far without ending the handle's events

// ---- internal stage entry points (device pointers, enqueue on ctx->stream) ---------------
// radix_sort.hip
int32_t morton_keys_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float corner[3],
                        float size, uint32_t *d_keys);
// Morton keys -> Hilbert indices of the same cells (hilbert_of_morton), optionally gathered through d_idx; d_out may be d_morton
int32_t hilbert_keys_dev(treee_ctx *ctx, const uint32_t *d_morton, const uint32_t *d_idx, uint64_t n, uint32_t *d_out);
// sorts (key,val) pairs by the low `bits` bits of key, stable.  d_vals_in == nullptr means
// val = index.  Results land in d_keys_out / d_vals_out; tmp buffers come from the arena.
int32_t sort_pairs_dev(treee_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in,
                       uint64_t n, int bits, uint32_t *d_keys_out, uint32_t *d_vals_out);
int32_t bbox_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, float corner_out[3],
                 float *size_out); // synchronises
int32_t gather_sorted_dev(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_perm, uint64_t n,
                          float4 *d_sorted);
// d_live (device, optional): number of leading elements that matter -- the rest of d_in is taken as zero and the
// rest of d_out is left unwritten (capacity-sized arrays of the octree levels)
int32_t exclusive_scan_u32_dev(treee_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n,
                               uint32_t *d_total /* device, optional */, const uint32_t *d_live = nullptr);

// persistent device blocks for handles: taken from / returned to the owning context's pool.
// pool_release tolerates a context that has already been destroyed (falls back to hipFree).
int32_t pool_alloc(treee_ctx *ctx, size_t bytes, void **out);
void pool_release(treee_ctx *ctx, void *p);
template <class T> int32_t pool_alloc_t(treee_ctx *ctx, size_t count, T **out) {
	void *p = nullptr;
	int32_t rc = pool_alloc(ctx, count * sizeof(T), &p);
	*out = (T *)p;
	return rc;
}

// common per-cloud preparation shared by octree and knn: cube, keys, sort, gather
struct SortedCloud {
	uint64_t n = 0;
	float corner[3] = {0, 0, 0};
	float size = 0;
	uint32_t *d_keys = nullptr;  // sorted
	uint32_t *d_perm = nullptr;  // sorted position -> original index
	float4 *d_sorted = nullptr;  // (x,y,z,bitcast(orig)) sorted
	int device = 0;
	bool hilbert = false; // d_keys are Hilbert indices (the kNN structure's private copy), not Morton keys
	int crowded = -1; // some Morton cell holds more than LONG_RUN points: -1 unknown, else 0 / 1 (long_run_flag_dev)
	int block7_big = -1; // some 2^-7 block may hold BLOCK7_MAX points or more: -1 unknown, else 0 / 1 (the same flag word, bit 1)
	float cell_share = -1.0f; // occupied 2^-10 cells per point, estimated from a sample of the sorted keys: < 0 unknown (distinct_keys_dev; asked in the build's own round trip)
	float *d_cube = nullptr;   // (corner.xyz, size) on the device when the cube was computed from the data
	bool cube_pending = false; // corner / size above are not yet valid on the host (cube_fetch_async + cube_commit)
	treee_ctx *owner = nullptr; // context whose pool the arrays came from
	mutable HandleSync sync;    // uses of the arrays from other streams (see HandleSync)
	SortedCloud() = default;
	SortedCloud(const SortedCloud &) = delete;
	SortedCloud &operator=(const SortedCloud &) = delete;
	~SortedCloud() { release(); }
	void release();
};
// the sorted cloud of an octree can be shared with a kNN structure (one sort, two consumers)
std::shared_ptr<SortedCloud> octree_sorted_cloud(const treee_octree *o);
// the records a kNN structure holds in ITS order (Hilbert index of the cell, index), the number of points it was built from,
// and whether it is a segmented structure (records padded with (NaN, NaN, NaN, ~0u)) -- knn.hip
std::shared_ptr<SortedCloud> knn_sorted_cloud(const treee_knn *t, uint64_t *n_points, bool *segmented);
// the kNN structure over an already sorted cloud, enqueued on ctx's stream (knn.hip; treee_knn_from_octree_dev and
// treee_sort_build_knn_dev are its callers)
// (tables of the structure's order by blocks, made from the sorted keys alone: knn_block_tables_enqueue may run them on another
// stream beside the record gather; scratch of the API call that made them)
struct BlockTables {
	uint32_t *d_start = nullptr, *d_end = nullptr, *d_dest_h = nullptr, *d_big = nullptr;
	bool made = false;
};
int32_t knn_block_tables_enqueue(treee_ctx *ctx, const SortedCloud *sc, BlockTables *bt);
int32_t knn_from_sorted_cloud(treee_ctx *ctx, const std::shared_ptr<SortedCloud> &sc, treee_knn **out, const BlockTables *pre = nullptr);
// hilbert_order: sort by the Hilbert index of the cell instead of its Morton key (a kNN structure's own cloud; out->hilbert)
int32_t build_sorted_cloud(treee_ctx *ctx, const float *d_xyz, uint64_t n, const float *corner,
                           float size, SortedCloud *out, bool fork_after_sort = false, bool hilbert_order = false);
// *d_flag (device, zeroed here) = 1 iff sorted keys hold a run of more than LONG_RUN equal keys
#ifndef TREEE_LONG_RUN
#define TREEE_LONG_RUN 256
#endif
constexpr uint32_t LONG_RUN = TREEE_LONG_RUN;
int32_t long_run_flag_dev(treee_ctx *ctx, const uint32_t *d_sorted_keys, uint64_t n, uint32_t *d_flag);
// (bit 1 of that flag: some 2^-7 block -- keys sharing their top 21 bits -- may hold BLOCK7_MAX points or more)
constexpr uint32_t BLOCK7_MAX = 2048, BLOCK7_PROBE = BLOCK7_MAX - 64;
// d_out[0] / d_out[1] (device) = distinct values / keys in a sample of windows of the sorted keys
int32_t distinct_keys_dev(treee_ctx *ctx, const uint32_t *d_sorted_keys, uint64_t n, uint32_t *d_out);
// The cube of a cloud built from its data stays on the device (keys are generated from there); the host copy is
// brought over by whichever host round trip of the build comes next: enqueue with cube_fetch_async, and after that
// synchronisation call cube_commit.  Both are no-ops when the host copy is already valid.
int32_t cube_fetch_async(treee_ctx *ctx, SortedCloud *sc, float h_cube[4]);
void cube_commit(SortedCloud *sc, const float h_cube[4]);
int32_t cube_to_host(treee_ctx *ctx, SortedCloud *sc); // both, with a synchronisation of its own (paths without another one)

// is this octree handle a rank's piece of a sharded octree (octree.hip)?  Its indices are global: no kNN structure over it
bool octree_is_shard(const treee_octree *o);
// a sharded octree's copy of the rank's own input in Morton order of the common cube (records carry *index_base + input
// position); null for an ordinary octree
std::shared_ptr<SortedCloud> octree_shard_local(const treee_octree *o, uint32_t *index_base);
// per-block partial boxes of a cloud: d_partial[grid][6] = (min.xyz, max.xyz) (radix_sort.hip)
int32_t bbox_partials_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, float **d_partial_out, uint32_t *grid_out);

// treee_knn_build_segments_dev with the cube handed in (knn.hip): one device's share of a cloud's segments keeps the whole
// cloud's cell keys, hence its tie order
int32_t knn_build_segments_cube(treee_ctx *ctx, const float *d_xyz, const uint32_t *d_segment, uint64_t n, uint32_t n_segments, const float cube[4],
                                treee_knn **out);

// default context for the host-pointer API (created lazily, one per process)
int32_t default_ctx(treee_ctx **out);
// The host-pointer API shares that context (one stream, one workspace): calls from several host threads
// are serialised by this recursive lock (host entry points call each other).
std::recursive_mutex &default_ctx_mutex();
#define TREEE_DEFAULT_CTX(ctx)                                                     \
	std::lock_guard<std::recursive_mutex> default_ctx_lock__(default_ctx_mutex()); \
	TREEE_TRY(default_ctx(&ctx))

// Host scratch whose size depends on the caller's arguments, inside extern "C" bodies: nothing may unwind across the C ABI, so
// no std::vector there (its allocation throws) -- a failed allocation is a status like any other.
template <class T> struct HostBuf {
	T *p = nullptr;
	size_t n = 0;
	HostBuf() = default;
	HostBuf(const HostBuf &) = delete;
	HostBuf &operator=(const HostBuf &) = delete;
	~HostBuf() { delete[] p; }
	int32_t alloc(size_t count) {
		p = new (std::nothrow) T[count ? count : 1]();
		if (!p) return set_err(TREEE_ERR_NOMEM, "host scratch of %zu bytes", count * sizeof(T));
		n = count;
		return TREEE_OK;
	}
	T *data() { return p; }
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; }
};

template <class T> struct DevBuf { // RAII for hipMalloc'd temporaries in host wrappers
	T *p = nullptr;
	~DevBuf() {
		if (p) (void)hipFree(p);
	}
	int32_t alloc(size_t count) {
		if (count == 0) count = 1;
		hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
		if (e != hipSuccess) return set_err(TREEE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", count * sizeof(T), hipGetErrorString(e));
		return TREEE_OK;
	}
};

} // namespace treee
