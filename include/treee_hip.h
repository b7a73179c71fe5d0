/*
 * treee_hip.h -- C ABI of the MI355X-native treee hot path (libtreee_hip.so).
 *
 * The reference (antonWetzel/treee @ 2024_10_08) has no FFI / plugin boundary: every
 * hot-path call is a monomorphised Rust call.  This header cuts the boundary at the three
 * narrowest seams of the reference and states, per entry point, the Rust interface it
 * replaces (file:line relative to the reference root).  INTEGRATION.md shows the Rust
 * `extern "C"` block + safe wrappers a maintainer would add.
 *
 * Conventions
 *   - every function returns int32_t status (TREEE_OK == 0); nothing unwinds across the ABI;
 *     treee_last_error() returns a thread-local message for the last failure
 *   - plain pointers and sizes only; no torch / C++ types
 *   - functions without suffix take HOST pointers (drop-in for the Rust slices they replace:
 *     they stage through HBM and copy results back)
 *   - functions with suffix _dev take DEVICE pointers (HBM resident) and a treee_ctx that
 *     owns the HIP stream and a reusable workspace; they enqueue work on that stream and do
 *     not synchronise unless documented
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with TREEE_ERR_HIP
 *   - indices on the device are u32 and a cloud (one shard) holds at most 2^31 - 1 points: every build
 *     refuses more with TREEE_ERR_UNSUPPORTED (position arithmetic inside the kernels is 32-bit; the
 *     largest cloud run is 1.005 G points, tests/test_gpu_parity.py::test_full_size_1g_whole_cloud_properties);
 *     the host structs widen to u64 like Rust's usize
 *
 * Streams
 *   - a handle (treee_knn, treee_octree) is OWNED by the context that built it: its arrays come from
 *     that context's block pool and return to it on free, where later builds on the owner's stream
 *     reuse them without synchronising
 *   - any _dev entry point may be given a DIFFERENT context than the handle's owner: the library then
 *     makes that context's stream wait for the work the owner's stream holds (the build), remembers
 *     the use with an event, and treee_*_free makes the owner's stream wait for those events before the
 *     arrays can be reused.  Freeing a handle right after enqueueing a query on another stream is safe;
 *     neither call blocks the host
 *   - pointers obtained from treee_octree_dev_ptr are the caller's to order: they are valid until
 *     treee_octree_free, and work the CALLER enqueues on them from other streams must be complete (or
 *     ordered before the owner's stream) when the handle is freed
 *   - a context is used by one host thread at a time; different contexts may be used concurrently
 *   - INSIDE a build (treee_sort_build_dev, treee_sort_build_knn_dev) a context may run an independent chain of small
 *     launches on an auxiliary stream of its own (created on first use at the highest stream priority, so that a small
 *     launch does not queue behind the workgroups of a kernel that fills the card); every such call joins it into the
 *     context's stream before it returns: outside an API call a context has exactly one stream, the caller's
 */
#ifndef TREEE_HIP_H
#define TREEE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TREEE_ABI_VERSION 1

enum {
	TREEE_OK = 0,
	TREEE_ERR_INVALID = 1,     /* bad argument (null pointer, k == 0, ...) */
	TREEE_ERR_HIP = 2,         /* HIP runtime error / no device */
	TREEE_ERR_EMPTY = 3,       /* KDTree::new on an empty slice panics in the reference */
	TREEE_ERR_UNSUPPORTED = 4, /* e.g. projection matrix that mixes x/y axes */
	TREEE_ERR_NOMEM = 5
};

/* repr(C) mirror of k_nearest::Entry<f32> (k-nearest/src/kd_tree.rs:8-12; the Rust struct is
 * not repr(C), so the shim converts).  index == UINT64_MAX marks an unused slot. */
typedef struct {
	float distance; /* SQUARED euclidean distance (k-nearest/readme.md:90-92) */
	uint32_t _pad;
	uint64_t index;
} treee_entry;

typedef struct treee_ctx treee_ctx;       /* stream + workspace */
typedef struct treee_knn treee_knn;       /* replaces k_nearest::KDTree<3,f32,..> */
typedef struct treee_octree treee_octree; /* Morton order + octree + LOD lists */

int32_t treee_abi_version(void);
const char *treee_last_error(void);
int32_t treee_device_count(int32_t *count_out);

/* ---- context ------------------------------------------------------------------------- */
/* hip_stream: the hipStream_t all work of this context is enqueued on (e.g. torch's current
 * stream); NULL is HIP's legacy default stream, as everywhere in HIP. */
int32_t treee_ctx_create(int32_t device, void *hip_stream, treee_ctx **out);
int32_t treee_ctx_destroy(treee_ctx *ctx);
int32_t treee_ctx_sync(treee_ctx *ctx);
/* pre-size the workspace so later calls never allocate (benchmarks, graph capture) */
int32_t treee_ctx_reserve(treee_ctx *ctx, uint64_t bytes);

/* ---- B1: kNN seam -------------------------------------------------------------------- */
/* KDTree::new(&[Point]) (k-nearest/src/kd_tree.rs:35-42) via NeighborsTree::new
 * (treee/src/calculations.rs:396-407).  Copies the coordinates; n == 0 -> TREEE_ERR_EMPTY. */
int32_t treee_knn_build(const float *xyz_aos, uint64_t n, treee_knn **out);
int32_t treee_knn_build_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n, treee_knn **out);
/* kNN structure over the cloud an octree has already Morton-sorted (shares the sorted arrays
 * by reference count: one sort feeds both; either handle may be freed first) */
int32_t treee_knn_from_octree_dev(treee_ctx *ctx, const treee_octree *o, treee_knn **out);
/* The same for one rank of a sharded cloud: `o` is the rank's sharded octree (treee_shard_octree_build_dev), which holds
 * the rank's own points in Morton order of the common cube; the structure is laid over those points followed by the
 * `g` ghost points (own point i is the handle's point i, ghost j its point n_own + j: the layout of
 * treee_shard_exchange_ghosts_dev's buffer), without sorting the tile a second time.  Cell keys -- and with them the
 * order inside distance ties -- are those of the whole cloud's cube.  Serves treee_curvature_own_dev / treee_knn_self_dev
 * and position queries.  No reference counterpart (the reference is one process). */
int32_t treee_knn_from_shard_dev(treee_ctx *ctx, const treee_octree *o, const float *d_ghost_xyz_aos, uint64_t n_ghosts, treee_knn **out);
/* A sharded octree whose points changed hands keeps the rank's OWN input in Morton order (24 B per point) for the call above.
 * Once the kNN structure is built -- or if none will be -- give it back; treee_knn_from_shard_dev then fails with
 * TREEE_ERR_INVALID for this handle.  Everything else about the octree is unaffected. */
int32_t treee_shard_octree_release_local(treee_octree *o);
/* One structure for a cloud that is cut into segments the way the reference holds it after segmentation
 * (`segment: Vec<u32>` beside the points, treee/src/calculations.rs:97-118): every segment behaves as its own
 * KDTree::new(&segment_points) (calculations.rs:329, 392-403) -- a self query only ever returns points of the
 * query's segment.  d_segment[i] in [0, n_segments), n_segments <= 2^24.  The handle serves
 * treee_knn_self_dev and treee_curvature_dev (k <= 32); position queries are refused (TREEE_ERR_UNSUPPORTED). */
int32_t treee_knn_build_segments_dev(treee_ctx *ctx, const float *d_xyz_aos, const uint32_t *d_segment,
                                     uint64_t n, uint32_t n_segments, treee_knn **out);
int32_t treee_knn_free(treee_knn *t);
int32_t treee_knn_len(const treee_knn *t, uint64_t *n_out);

/* KDTree::k_nearest / nearest_to_position (kd_tree.rs:82-84,92-96) for nq positions.
 * out[nq*k] is fully overwritten ("initial values are never read"); counts[nq] receives the
 * number of valid entries; out[i*k .. i*k+counts[i]) ascending by distance, all distance <
 * max_dist_sq (strict, squared).  Tie rule: distances and counts are
 * bit-identical to the reference; among equal distances the entry with the smaller
 * (cell key, index) wins -- the key is the 30-bit Hilbert index of the point's cell in the
 * structure's cube (the order of the structure's buckets).  k == 0 -> TREEE_ERR_INVALID (reference panics).
 * k <= 16, <= 32 and <= 64 run register-resident lists of that length (the reference uses 31);
 * 64 < k <= 319 runs an LDS-heap kernel; larger k -> TREEE_ERR_UNSUPPORTED. */
int32_t treee_knn_query_batch(const treee_knn *t, const float *q_xyz, uint64_t nq, uint32_t k,
                              float max_dist_sq, treee_entry *out, uint32_t *counts);
/* device form: SoA outputs d_dist[nq*k] f32, d_index[nq*k] u32 (0xFFFFFFFF unused),
 * d_counts[nq] u32.  Any output pointer may be NULL. */
int32_t treee_knn_query_batch_dev(treee_ctx *ctx, const treee_knn *t, const float *d_q_xyz,
                                  uint64_t nq, uint32_t k, float max_dist_sq, float *d_dist,
                                  uint32_t *d_index, uint32_t *d_counts);
/* every tree point queries itself (the loop of treee/src/calculations.rs:332-340); row i
 * belongs to original point i */
int32_t treee_knn_self_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq,
                           float *d_dist, uint32_t *d_index, uint32_t *d_counts);
/* The same, and WHICH ROWS A DISTANCE TIE TOUCHES (d_tie_flags[n], one byte per original point; may be NULL):
 *   bit 0  two of the row's k entries are equally far -- the reference's list holds the same points, possibly in another
 *          order (best_set.rs:56-66 orders equal keys by heap mechanics), so sums over the list may differ in their last bits;
 *   bit 1  the k-th entry and the next candidate behind it are equally far -- the reference keeps whichever its KD-tree
 *          visited first (strict `<` at kd_tree.rs:114,125), this library the smaller (cell, index): the two lists may hold
 *          DIFFERENT points there, and everything derived from the list (calculations.rs:332-368) may differ.
 * A row with flag 0 has pairwise distinct distances up to and including the (k + 1)-th candidate: its list is the
 * reference's, index for index.  A host that needs the reference's exact lists re-runs its own KDTree on the flagged rows
 * only (<= 0.03 % of a 1 mm-quantised forest: INTEGRATION.md).  The entry behind the k-th must fit the register-resident
 * list: k <= 63 (k <= 31 on a segmented handle), else TREEE_ERR_UNSUPPORTED; k = 16 / 32 run the next longer list. */
int32_t treee_knn_self_ties_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq,
                                float *d_dist, uint32_t *d_index, uint32_t *d_counts, uint8_t *d_tie_flags);
/* KDTree::nearest (kd_tree.rs:86-90): all points with distance < max_dist_sq, ascending.
 * Writes min(found, cap) entries, *found_out = total. */
int32_t treee_knn_radius(const treee_knn *t, const float q_xyz[3], float max_dist_sq,
                         treee_entry *out, uint64_t cap, uint64_t *found_out);
/* KDTree::empty (kd_tree.rs:98-102) for nq positions: empty_out[i] = 1 iff no point has
 * distance < max_dist_sq */
int32_t treee_knn_empty(const treee_knn *t, const float *q_xyz, uint64_t nq, float max_dist_sq,
                        uint8_t *empty_out);

/* kNN consumer: SegmentInformation::update curve branch (treee/src/calculations.rs:328-373,
 * fast_eigenvalues :467-498): curve_out[i] = 3*eig3/(eig1+eig2+eig3) of the covariance of the
 * k nearest neighbours (reference: k = 31, max_dist_sq = f32::MAX).  acos / cos are the one
 * definition of include/treee_f32_trig.h (f64 evaluation rounded once): `curve` is bit for bit what
 * the reference's loop gives on the same neighbour list with a correctly rounding libm, and within
 * a few 1e-7 of any libm's result; lists differ from the reference's only inside distance ties (H2). */
int32_t treee_curvature(const float *xyz_aos, uint64_t n, uint32_t k, float max_dist_sq,
                        float *curve_out);
int32_t treee_curvature_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq,
                            float *d_curve_out);
/* curve and the tie flags of treee_knn_self_ties_dev in one pass: a row whose flag is 0 has the reference's curve wherever
 * the two libms agree; bit 0 rows may differ by the f32 summation order (<= 1e-6 measured), bit 1 rows by another neighbour */
int32_t treee_curvature_ties_dev(treee_ctx *ctx, const treee_knn *t, uint32_t k, float max_dist_sq,
                                 float *d_curve_out, uint8_t *d_tie_flags);
/* host slices in and out (the form a host without device buffers binds: build, query, copy back) */
int32_t treee_curvature_ties(const float *xyz_aos, uint64_t n, uint32_t k, float max_dist_sq,
                             float *curve_out, uint8_t *tie_flags_out);
/* the same, but only points whose original index is < n_own are queried; the rest (a shard's ghost points,
 * appended behind its own points) only serve as neighbours.  d_curve_out needs n_own entries. */
int32_t treee_curvature_own_dev(treee_ctx *ctx, const treee_knn *t, uint64_t n_own, uint32_t k,
                                float max_dist_sq, float *d_curve_out);
/* the same loop run for every segment of a segmented cloud in one pass: curve_out[i] is what
 * SegmentInformation::update(calc_curve = true) computes for point i inside its own segment
 * (calculations.rs:328-373 called per segment, interactive/mod.rs:224-231) */
int32_t treee_curvature_segments(const float *xyz_aos, const uint32_t *segment, uint64_t n,
                                 uint32_t n_segments, uint32_t k, float max_dist_sq, float *curve_out);
/* map_to_u32 (treee/src/calculations.rs:455-457) over a device array */
int32_t treee_map_to_u32_dev(treee_ctx *ctx, const float *d_in, uint64_t n, uint32_t *d_out);

/* ---- F1: the slice segmentation that produces `segment: Vec<u32>` (host code, as in the reference) ---- */
/* segmentation (treee/src/segmenting.rs:110-235) with TreeSet::new / add_point / filter_trees / tree_positions (:419-524),
 * Tree (:242-396) and centroid (:526-545): 1 m height slices top-down, per slice the convex areas grown from its points,
 * matched to the centroids handed down from the slice above, Voronoi cells of the centroids clipped to the box
 * [bbox_min_xz, bbox_max_xz] (the loaded cloud's min / max in x and z; both NULL: taken from the points), every point assigned to the first cell that
 * contains it with 0.1 slack (move-to-front order).  segment_out[i] = id of point i, 0 = none (DELETED_INDEX);
 * *n_segments_out = largest id + 1.  The reference's run is random (thread_rng shuffle, RandomState map order, random
 * ids) and its Voronoi cells come from a third-party crate absent from the snapshot: this is the deterministic variant
 * SURVEY.md F1 prescribes -- seeded shuffle, ordered maps, ids 1, 2, 3 ... in creation order, cells restated -- with
 * every departure marked in csrc/segmenting.cpp.  It runs on the host: the algorithm is serial per slice (each point
 * is tested against the hulls the earlier points built), the reference keeps it on CPU workers too.
 * max_distance: the reference's slider value (default 0.75, segmenting.rs:16). */
int32_t treee_segment_slices(const float *xyz_aos, uint64_t n, const float bbox_min_xz[2], const float bbox_max_xz[2],
                             float max_distance, uint64_t seed, uint32_t *segment_out, uint32_t *n_segments_out);

/* ---- F4: the Calculations phase around the kNN consumer, all segments in one call ------ */
/* SegmentData::new for every segment of a segmented cloud (treee/src/calculations.rs:133-175, run per segment by
 * Calculations::new :97-118): seg_min / seg_max [n_segments][3]; SegmentInformation::new (:210-276) ->
 * seg_info [n_segments][4] = {trunk_height, crown_height, ground_sep, crown_sep}, from the areas of one convex hull
 * per 0.1 m slice (get_size_areas :429-452, Tree::new / insert / statistics treee/src/segmenting.rs:258-269,346-396,
 * 320-323); classification [n] = 0 Ground / 1 Trunk / 2 Crown (:149-163; optional).  Points are inserted into a
 * slice's hull in their order in the input arrays, as the reference inserts them in the order of the segment's
 * Vec.  A segment without points gets min = max = 0 (:135-136).
 * expansion / height [n] (optional, need classification): SegmentInformation::update(calc_curve = false)
 * (:278-327) with min / max = the segment's min.y / max.y (treee/src/interactive/mod.rs:224-231).
 * Bit-exact against the CPU restatement (hull vertices, areas, every output): only + - * / sqrt, no FMA.
 * A segment taller than 65 534 slices (6.5 km) -> TREEE_ERR_UNSUPPORTED; segment id >= n_segments -> TREEE_ERR_INVALID. */
int32_t treee_segments_calculate(const float *xyz_aos, const uint32_t *segment, uint64_t n, uint32_t n_segments,
                                 float *seg_min_out, float *seg_max_out, float *seg_info_out,
                                 uint32_t *classification_out, float *expansion_out, float *height_out);
int32_t treee_segments_calculate_dev(treee_ctx *ctx, const float *d_xyz_aos, const uint32_t *d_segment, uint64_t n,
                                     uint32_t n_segments, float *d_seg_min, float *d_seg_max, float *d_seg_info,
                                     uint32_t *d_classification, float *d_expansion, float *d_height);
/* DisplayData::new (treee/src/interactive/mod.rs:1002-1039) for every segment at once: the four per-point `value`
 * arrays the splat colours by -- classification u32::MAX/8 * {1, 3, 6}, map_to_u32(curve), map_to_u32(expansion /
 * max expansion of the point's segment), map_to_u32(height).  Any input/output pair may be NULL. */
int32_t treee_display_values_dev(treee_ctx *ctx, const uint32_t *d_segment, uint64_t n, uint32_t n_segments,
                                 const uint32_t *d_classification, const float *d_curve, const float *d_expansion,
                                 const float *d_height, uint32_t *d_classification_value, uint32_t *d_curve_value,
                                 uint32_t *d_expansion_value, uint32_t *d_height_value);

/* ---- A0: Morton sort + octree + LOD (build-defined; no reference counterpart) -------- */
/* corner/size: the cube (render/src/camera_3d.rs:20 parametrisation); size <= 0 -> the cube is
 * computed from the data (min corner, max extent).  leaf_cap / lod_cap 0 -> 256. */
int32_t treee_sort_build(const float *xyz_aos, uint64_t n, const float corner[3], float size,
                         uint32_t leaf_cap, uint32_t lod_cap, treee_octree **out);
int32_t treee_sort_build_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n,
                             const float corner[3], float size, uint32_t leaf_cap,
                             uint32_t lod_cap, treee_octree **out);
/* The two halves of treee_sort_build_dev as separate calls.  treee_sort_dev: the Morton sort alone -- a handle whose
 * sorted arrays (KEYS, PERM, SORTED_XYZ) exist and whose node table does not yet (n_nodes = 0); it synchronises once (the
 * cube when it is computed from the data, and the crowded-cell flag a kNN structure needs) and
 * treee_knn_from_octree_dev accepts it.  treee_octree_complete_dev builds the node table, leaf ids and LOD lists later,
 * on ANY context of the same device: on a context bound to a second stream the octree's ~50 small launches run under
 * the kNN kernel of the first (the library orders the streams, see "Streams").  One handle, freed once. */
int32_t treee_sort_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n, const float corner[3], float size,
                       treee_octree **out);
int32_t treee_octree_complete_dev(treee_ctx *ctx, treee_octree *o, uint32_t leaf_cap, uint32_t lod_cap);
/* The octree AND the kNN structure (treee_knn_from_octree_dev's) of one cloud from ONE call: one Morton sort, and the host's
 * launches in the order that keeps the GPU fed -- the octree's small dependent launches on the context's auxiliary stream
 * beside the record gather and the kNN structure's own work; the host waits for the crowded-cell flags (done long before
 * the gather) and later for the node count, never with an empty queue.  Same arrays as the separate calls, bit for bit.
 * Both handles are the caller's (treee_octree_free, treee_knn_free); n == 0 -> TREEE_ERR_EMPTY as KDTree::new. */
int32_t treee_sort_build_knn_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n, const float corner[3], float size,
                                 uint32_t leaf_cap, uint32_t lod_cap, treee_octree **octree_out, treee_knn **knn_out);
int32_t treee_octree_free(treee_octree *o);
typedef struct {
	uint64_t n_points, n_nodes, n_lod;
	float corner[3], size;
	uint32_t leaf_cap, lod_cap;
} treee_octree_info;
int32_t treee_octree_info_get(const treee_octree *o, treee_octree_info *info);
/* node record, identical layout on host and device */
typedef struct {
	uint32_t key_prefix, level, first, count, first_child, child_mask, parent, lod_first;
} treee_octree_node;
enum {
	TREEE_OCT_KEYS = 0,        /* u32[n]  sorted morton keys */
	TREEE_OCT_PERM = 1,        /* u32[n]  perm[i] = original index of sorted position i */
	TREEE_OCT_NODES = 2,       /* treee_octree_node[n_nodes], BFS order */
	TREEE_OCT_LEAF = 3,        /* u32[n]  leaf node index per sorted position */
	TREEE_OCT_LOD = 4,         /* u32[n_lod] sorted positions sampled per inner node */
	TREEE_OCT_SORTED_XYZ = 5   /* float4[n] (x,y,z,bitcast(original index)) in sorted order */
};
/* copy an array to host memory (synchronises) */
int32_t treee_octree_copy(const treee_octree *o, int32_t what, void *dst_host, uint64_t dst_bytes);
/* raw device pointer of an array (valid until treee_octree_free) */
int32_t treee_octree_dev_ptr(const treee_octree *o, int32_t what, void **d_ptr_out);

/* ---- B2: splat seam ------------------------------------------------------------------ */
/* render::PointCloud::new + PointCloudProperty::new + PointCloud::render
 * (render/src/point_cloud.rs:136-173,183-193) with Camera3DGPU::new (render/src/camera_3d.rs:59)
 * and PointCloudEnvironment::new (point_cloud.rs:214); shader render/src/point_cloud.wgsl.
 * view/proj column-major f32[16].  depth_out[w*h] (1.0 = empty), id_out[w*h] = `value` of
 * the winning point (0 where empty).  proj must not mix x/y (TREEE_ERR_UNSUPPORTED). */
int32_t treee_splat(const float *xyz_aos, const uint32_t *value, uint64_t n, const float view[16],
                    const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t w,
                    uint32_t h, float *depth_out, uint32_t *id_out);
/* d_value may be NULL when the environment culls nothing (vmin = 0, vmax = UINT32_MAX) and no id buffer is
 * requested: the packed frame then depends on positions only and can be drawn -- and, across GPUs, reduced --
 * before the values exist (treee_splat_resolve_dev looks them up afterwards).
 * d_packed (optional, u64[w*h]) receives (depth_bits << 32 | point index) for cross-GPU
 * min-reduction; when accumulate != 0 the packed buffer is NOT cleared first and
 * index_base is added to the point index (multi-chunk / multi-GPU draws). */
int32_t treee_splat_dev(treee_ctx *ctx, const float *d_xyz_aos, const uint32_t *d_value, uint64_t n,
                        const float view[16], const float proj[16], float scale, uint32_t vmin,
                        uint32_t vmax, uint32_t w, uint32_t h, uint64_t *d_packed,
                        int32_t accumulate, uint32_t index_base, float *d_depth_out,
                        uint32_t *d_id_out);
/* The same draw from the Morton-sorted records an octree handle already holds (TREEE_OCT_SORTED_XYZ) instead of the
 * caller's cloud: one 16-byte load per point, and points that are neighbours in space are drawn by neighbouring lanes.
 * The frame is bit for bit the one treee_splat_dev gives for the cloud the octree was built from: the packed word carries
 * the point's ORIGINAL index, so equal depths still resolve to the smallest original index ("first drawn wins",
 * render/src/point_cloud.rs:76-82) and d_value / the id buffer are indexed as before.  For a rank's piece of a sharded
 * octree the indices are global: pass index_base = 0, and a d_value indexed globally (or NULL). */
int32_t treee_splat_octree_dev(treee_ctx *ctx, const treee_octree *o, const uint32_t *d_value, const float view[16],
                               const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h,
                               uint64_t *d_packed, int32_t accumulate, uint32_t index_base, float *d_depth, uint32_t *d_id_out);
/* ... and from the records a kNN structure holds in ITS order (the Hilbert order of the cells: 1 024 consecutive records are a more
 * compact blob on the screen than along the Z curve -- fewer tiles that do not fit the LDS, 0.156 -> 0.145 ms per 15 M points).
 * Same frame, bit for bit.  A segmented structure's padding records are skipped.  No reference counterpart (the reference draws
 * the chunks' own vertex buffers, render/src/point_cloud.rs:148-171). */
int32_t treee_splat_knn_dev(treee_ctx *ctx, const treee_knn *t, const uint32_t *d_value, const float view[16],
                            const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t w, uint32_t h,
                            uint64_t *d_packed, int32_t accumulate, uint32_t index_base, float *d_depth, uint32_t *d_id_out);
/* resolve a packed buffer (after a cross-GPU min) into depth + id; d_value indexes by the
 * low 32 bits */
int32_t treee_splat_resolve_dev(treee_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_value,
                                uint64_t n_values, uint32_t w, uint32_t h, float *d_depth_out,
                                uint32_t *d_id_out);
/* nalgebra Perspective3::new(aspect,fovy,near,far).to_homogeneous() (camera_3d.rs:16-18) */
int32_t treee_perspective(float aspect, float fovy, float znear, float zfar, float out[16]);
/* eye-dome pass (render/src/eye_dome.wgsl:37-67) over a depth buffer; strength is the mapped
 * uniform (render/src/eye_dome.rs:226-230) */
int32_t treee_eye_dome_dev(treee_ctx *ctx, const float *d_depth, uint32_t w, uint32_t h,
                           float strength, float *d_alpha_out);

/* ---- B3: ingest seam ----------------------------------------------------------------- */
/* Chunk::read / next_point (treee/src/laz.rs:193-213): n records of `stride` bytes whose
 * first 12 bytes are i32 LE x,y,z -> centred f32 (x, z, -y) */
int32_t treee_decode_las_records(const uint8_t *rec, uint64_t n, uint32_t stride,
                                 const double scale[3], const double offset[3],
                                 const double center[3], float *xyz_out);
int32_t treee_decode_las_records_dev(treee_ctx *ctx, const uint8_t *d_rec, uint64_t n,
                                     uint32_t stride, const double scale[3],
                                     const double offset[3], const double center[3],
                                     float *d_xyz_out);
/* LAZ sources (laz.rs:77-94,146-160).  The reference hands these to the crate `laz` 0.9.2 (not in the snapshot); this
 * library restates the published LASzip format for LAS 1.0-1.3 files: pointwise chunked compressor, items POINT10 /
 * GPSTIME11 / RGB12 / BYTE of version 2.  Host code on host threads, as in the reference.  PARITY UNPINNED: no LAZ file
 * or tool exists where this was built (csrc/laz_codec.cpp).
 * treee_laz_chunk_table: `laz::laszip::ChunkTable::read_from` -- byte and point count of every chunk (the first chunk
 * starts at offset_to_point_data + 8); chunk_size = the LAZ record's (0xFFFFFFFF: variable); point counts are clipped to
 * total_points like laz.rs:88-89.  Call with byte_counts_out == NULL to learn *n_chunks_out. */
int32_t treee_laz_chunk_table(const uint8_t *file, uint64_t file_len, uint64_t offset_to_point_data, uint32_t chunk_size,
                              uint64_t total_points, uint64_t *byte_counts_out, uint64_t *point_counts_out, uint64_t cap,
                              uint64_t *n_chunks_out);
/* `decompress_many` of every chunk (laz.rs:146-160), one chunk per task on `threads` host threads (0 = all): chunk c
 * (file[chunk_starts[c] ..+ chunk_bytes[c]), chunk_points[c] records) lands behind the chunks before it in dst; a record
 * is the concatenation of the items (item_types / item_sizes / item_versions as in the LAZ record, n_items of them). */
int32_t treee_laz_decompress_chunks(const uint8_t *file, uint64_t file_len, const uint16_t *item_types, const uint16_t *item_sizes,
                                    const uint16_t *item_versions, uint32_t n_items, const uint64_t *chunk_starts,
                                    const uint64_t *chunk_bytes, const uint64_t *chunk_points, uint64_t n_chunks, uint32_t threads,
                                    uint8_t *dst, uint64_t dst_len);
/* loading.rs:99: key = floor(p.y) as isize */
int32_t treee_slice_keys(const float *xyz_aos, uint64_t n, int64_t *key_out);
int32_t treee_slice_keys_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n, int64_t *d_key_out);
/* the y-slice map of loading.rs:97-101 densified as segmenting.rs:121-133: order_out[n] =
 * point indices grouped by slice (arrival order inside a slice), offsets_out[layers+1].
 * Call with offsets_out == NULL to learn *layers_out / *min_key_out first. */
int32_t treee_slice_bin(const float *xyz_aos, uint64_t n, int64_t *min_key_out,
                        uint64_t *layers_out, uint32_t *order_out, uint64_t *offsets_out,
                        uint64_t offsets_cap);
/* device form: d_order_out[n] u32 and d_offsets_out[offsets_cap] u64 are device arrays; the
 * layer count is returned through one stream synchronisation (it sizes the offsets array) */
int32_t treee_slice_bin_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n, int64_t *min_key_out,
                            uint64_t *layers_out, uint32_t *d_order_out, uint64_t *d_offsets_out,
                            uint64_t offsets_cap);

/* ---- E: spatial shards (multi-GPU; no reference counterpart -- the reference is one process) ---- */
/* Stable compaction of the points whose (x, z) is inside keep_box = {x0, x1, z0, z1} (half-open) and NOT
 * inside drop_box.  Boundary points of a tile: keep = everything, drop = the tile shrunk by the halo on
 * its shared faces; ghosts from peers: keep = the dilated tile, drop = an empty box.  NaN never passes.
 * Writes min(count, cap) points in input order; *d_count_out (device u32) = count. */
int32_t treee_select_xz_dev(treee_ctx *ctx, const float *d_xyz_aos, uint64_t n, const float keep_box[4],
                            const float drop_box[4], float *d_out_xyz, uint64_t cap, uint32_t *d_count_out);


/* ---- E: many GPUs -- communicator ----------------------------------------------------------------------------------
 * One process per GPU.  The entry points below that take a treee_comm run the same code on every rank and talk to the
 * other ranks through it.  Two transports:
 *   - RCCL (treee_comm_create_rccl / treee_comm_wrap_rccl): collectives are enqueued on the context's stream, payloads
 *     stay in HBM and travel over xGMI.  librccl.so is opened at run time (dlopen; the copy the process already holds, else
 *     TREEE_RCCL_LIB, else /opt/rocm/lib).  UNVERIFIED ON MULTI-GPU HARDWARE: the development pool has one-GPU boxes, the
 *     RCCL transport is exercised with a one-rank communicator only.
 *   - host callbacks (treee_comm_create_host): the caller moves host buffers with whatever it has (MPI, sockets, a test's
 *     torch.distributed/gloo group); the library stages device data through host memory.  This is what the two-rank
 *     rehearsals on one GPU and hosts without RCCL use.
 * No reference counterpart (the reference is one process); SURVEY.md section 8 row E. */
typedef struct treee_comm treee_comm;
typedef struct {
	void *user;
	/* every rank contributes `bytes` bytes; recv holds world * bytes in rank order */
	int32_t (*all_gather)(void *user, const void *send, void *recv, uint64_t bytes);
	/* rank r receives send[send_off[r] ..+ send_bytes[r]) of every rank s into recv[recv_off[s] ..+ recv_bytes[s]) */
	int32_t (*all_to_all_v)(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_bytes, void *recv,
	                        const uint64_t *recv_off, const uint64_t *recv_bytes);
	/* element-wise minimum over all ranks, in place */
	int32_t (*all_reduce_min_u64)(void *user, uint64_t *buf, uint64_t count);
} treee_comm_host_ops;
int32_t treee_comm_create_host(const treee_comm_host_ops *ops, int32_t rank, int32_t world, treee_comm **out);
/* TREEE_OK iff librccl can be opened and holds every entry point the RCCL transport uses.  Local and non-collective: each
 * rank asks for itself, the ranks agree on the answer (e.g. a MIN all-reduce of a flag over whatever carries the unique
 * id), and only then does rank 0 create the id and everybody enter treee_comm_create_rccl -- a rank that finds no RCCL
 * must never leave the others waiting inside ncclCommInitRank. */
int32_t treee_rccl_available(void);
/* ncclGetUniqueId on one rank; the caller brings the 128 bytes to the others */
int32_t treee_rccl_unique_id(uint8_t id_out[128]);
/* ncclCommInitRank on `device` (collective: every rank calls it) */
int32_t treee_comm_create_rccl(const uint8_t unique_id[128], int32_t rank, int32_t world, int32_t device, treee_comm **out);
/* an ncclComm_t the caller already owns (not destroyed by treee_comm_destroy) */
int32_t treee_comm_wrap_rccl(void *nccl_comm, int32_t rank, int32_t world, treee_comm **out);
int32_t treee_comm_destroy(treee_comm *c);
int32_t treee_comm_info(const treee_comm *c, int32_t *rank_out, int32_t *world_out, int32_t *is_rccl_out);

/* ---- E: the sharded octree -------------------------------------------------------------------------------------------
 * The octree of the WHOLE cloud (the concatenation of every rank's input, rank order; global index of a point = points on
 * lower ranks + its position) held in pieces: node for node, field for field the table treee_sort_build gives for that
 * cloud with the same cube (tests assemble the pieces and compare bit for bit).
 *   - all ranks use one cube (given, or the union's cube by the rule of treee_sort_build);
 *   - the cube's level-`top_level` Morton cells (8^top_level of them) are dealt to the ranks in key order, balanced by
 *     point count: rank r owns one contiguous key range = one contiguous range of the global sorted order;
 *   - one all-to-all moves every point to the owner of its cell (nothing moves when the input already is so tiled);
 *   - levels < top_level (the "top") are the same on every rank, computed from the all-gathered cell counts; a node of
 *     level >= top_level lives on the rank that owns its cell, with GLOBAL first / parent / first_child / lod_first;
 *     global index of the j-th own node of level l = own_node_first[l] + j.
 * Collectives: all-gather of cell counts (4 * 8^top_level B per rank), all-to-all of 16-B records, all-gather of 22 words. */
#define TREEE_SHARD_LEVELS 11
typedef struct treee_shard_plan treee_shard_plan;
typedef struct {
	uint64_t n_total;       /* points of the whole cloud */
	uint32_t world, top_level;
	uint32_t n_top_nodes;   /* nodes of levels < top_level (replicated) */
	uint32_t n_top_lod;     /* their LOD entries */
	uint32_t n_level_nodes; /* nodes of level top_level, all ranks */
	uint32_t leaf_cap, lod_cap;
} treee_shard_plan_info;
typedef struct {
	uint32_t cell_begin, cell_end; /* owned level-top_level cells */
	uint64_t first, count;         /* owned range of the global sorted order */
	uint64_t index_base;           /* global index of the rank's input point 0 */
	uint32_t level_node_first, level_node_count; /* owned nodes of level top_level: position inside that level, count */
} treee_shard_plan_rank;
typedef struct {
	uint64_t n_nodes_total, n_lod_total;
	uint32_t node_begin[TREEE_SHARD_LEVELS];     /* global index of the level's first node */
	uint32_t own_node_first[TREEE_SHARD_LEVELS]; /* global index of the rank's first node of the level */
	uint32_t own_node_count[TREEE_SHARD_LEVELS];
	uint32_t lod_begin[TREEE_SHARD_LEVELS], own_lod_first[TREEE_SHARD_LEVELS], own_lod_count[TREEE_SHARD_LEVELS];
} treee_shard_levels;
/* Pure host arithmetic (no GPU, no communication; every rank computes the same plan from the same counts).
 * cell_counts[world][8^top_level]: points of every rank's INPUT per cell.  top_level in [1, 6]. */
int32_t treee_shard_plan_create(const uint32_t *cell_counts, int32_t world, uint32_t top_level, uint32_t leaf_cap, uint32_t lod_cap,
                                treee_shard_plan **out);
int32_t treee_shard_plan_free(treee_shard_plan *p);
int32_t treee_shard_plan_info_get(const treee_shard_plan *p, treee_shard_plan_info *info);
int32_t treee_shard_plan_rank_get(const treee_shard_plan *p, int32_t rank, treee_shard_plan_rank *out);
/* the replicated top: nodes_out[n_top_nodes] (global fields, lod_first set) */
int32_t treee_shard_plan_top_nodes(const treee_shard_plan *p, treee_octree_node *nodes_out);
/* the rank's nodes of level top_level as the roots of its local forest: `first` relative to the rank's range, parent
 * global (a top node), first_child / child_mask / lod_first still unset */
int32_t treee_shard_plan_level_nodes(const treee_shard_plan *p, int32_t rank, treee_octree_node *nodes_out);
/* what rank `rank` sends: positions [send_first[r], send_first[r] + send_count[r]) of its locally sorted input go to rank
 * r; recv_count[s] points arrive from rank s */
int32_t treee_shard_plan_exchange(const treee_shard_plan *p, int32_t rank, uint64_t *send_first, uint64_t *send_count, uint64_t *recv_count);
/* after the local builds: level_counts / lod_counts [world][TREEE_SHARD_LEVELS] = every rank's own nodes / LOD entries
 * per level (all-gathered) -> the global numbering */
int32_t treee_shard_plan_levels(const treee_shard_plan *p, const uint32_t *level_counts, const uint32_t *lod_counts, int32_t rank,
                                treee_shard_levels *out);

/* The cube of the union of all ranks' points by the rule of treee_sort_build (min corner, largest extent). */
int32_t treee_shard_cube_dev(treee_ctx *ctx, treee_comm *comm, const float *d_xyz_aos, uint64_t n, float corner_out[3], float *size_out);
/* Collective.  size <= 0: the cube is computed with treee_shard_cube_dev.  top_level 0 -> 5.  The handle is a
 * treee_octree whose arrays are the rank's piece: KEYS / PERM (global indices) / SORTED_XYZ (.w = global index) / LEAF
 * (global node index) over the owned range, NODES = own nodes of levels >= top_level in level order, LOD = their
 * entries (global sorted positions), TOP_NODES / TOP_LOD = the replicated top.  treee_octree_info_get reports the own
 * counts; treee_knn_from_octree_dev refuses such a handle. */
int32_t treee_shard_octree_build_dev(treee_ctx *ctx, treee_comm *comm, const float *d_xyz_aos, uint64_t n, const float corner[3],
                                     float size, uint32_t leaf_cap, uint32_t lod_cap, uint32_t top_level, treee_octree **out);
int32_t treee_shard_octree_info_get(const treee_octree *o, treee_shard_plan_info *info, treee_shard_plan_rank *rank,
                                    treee_shard_levels *levels);
enum { TREEE_OCT_TOP_NODES = 6, TREEE_OCT_TOP_LOD = 7 };

/* ---- E: ghost points and the frame across ranks ---------------------------------------------------------------------
 * x/z tiles (treee_select_xz_dev above).  tile_box = {x0, x1, z0, z1} of this rank; shared_faces bit 0..3 = the x-, x+,
 * z-, z+ face touches another tile.  d_cloud holds the rank's own n_own points first and has room for n_own + ghost_cap
 * points: the ghosts (points of other ranks inside the tile dilated by halo) are written behind the own points in
 * (rank, input order).  ONE all-gather of fixed-capacity blocks (send_cap points per rank; row 0 of a block carries the
 * sender's count).  *n_ghost_out = ghosts found, *most_sent_out = the largest contribution: when it exceeds send_cap or
 * n_ghost exceeds ghost_cap every rank sees the same numbers and the caller repeats with larger capacities (status
 * TREEE_OK either way).  Synchronises once (the two counts). */
int32_t treee_shard_exchange_ghosts_dev(treee_ctx *ctx, treee_comm *comm, float *d_cloud, uint64_t n_own, const float tile_box[4],
                                        uint32_t shared_faces, float halo, uint64_t send_cap, uint64_t ghost_cap, uint64_t *n_ghost_out,
                                        uint64_t *most_sent_out);
/* element-wise MIN of the packed (depth_bits << 32 | index) frames of all ranks, in place (treee_splat_dev with
 * index_base = points on lower ranks): what one GPU drawing everything would hold */
int32_t treee_frame_reduce_min_dev(treee_ctx *ctx, treee_comm *comm, uint64_t *d_packed, uint64_t count);

/* ---- E: the curvature loop over the segments of a cloud on several GPUs of one process ---------------------------------
 * How the reference parallelises it: whole segments are independent units (treee/src/calculations.rs:97-118, one KD-tree
 * per segment :329), so no data is exchanged at all.  Segments are dealt to the devices by point count (longest first to
 * the least loaded device; ties: lower segment id, lower list position), each device runs treee_curvature_segments on
 * its share from its own host thread, results land in curve_out[n] in input order -- bit for bit what one device gives.
 * devices[n_devices]: HIP device ordinals (an ordinal may repeat: two contexts on one GPU). */
int32_t treee_curvature_segments_multi(const int32_t *devices, uint32_t n_devices, const float *xyz_aos, const uint32_t *segment,
                                       uint64_t n, uint32_t n_segments, uint32_t k, float max_dist_sq, float *curve_out);
int32_t treee_segments_calculate_multi(const int32_t *devices, uint32_t n_devices, const float *xyz_aos, const uint32_t *segment,
                                       uint64_t n, uint32_t n_segments, float *seg_min_out, float *seg_max_out, float *seg_info_out,
                                       uint32_t *classification_out, float *expansion_out, float *height_out);

#ifdef __cplusplus
}
#endif
#endif
