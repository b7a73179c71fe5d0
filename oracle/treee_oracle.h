/*
 * treee_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * Plain C++17 restatement of the antonWetzel/treee hot path, function by
 * function, each entry citing the reference file:line it follows.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library; the shipped library (treee_amd/csrc) never does.
 *
 * PARITY STATUS: "parity unpinned".  The reference is Rust + WGSL, has zero
 * tests / golden vectors and no Rust toolchain exists in
 * this image, so the oracle is pinned only by (i) the prose invariants of
 * k-nearest/readme.md:26-34,90-92, (ii) hand-traced KATs in tests/golden/,
 * (iii) brute-force cross checks.  The octree/LOD part has no reference
 * counterpart at all and is build-defined.
 *
 * Compile with: -O2 -ffp-contract=off -fno-fast-math (Rust never contracts
 * a*b+c).
 */
#ifndef TREEE_ORACLE_H
#define TREEE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* repr(C) mirror of k_nearest::Entry<f32> (k-nearest/src/kd_tree.rs:8-12). */
typedef struct {
	float distance;
	uint32_t _pad;
	uint64_t index; /* usize::MAX == invalid */
} orc_entry;

/* ---- k-nearest crate ---------------------------------------------------- */
typedef struct orc_kdtree orc_kdtree;

/* KDTree::new (kd_tree.rs:35-42).  n == 0 -> returns NULL (reference panics). */
orc_kdtree *orc_kdtree_new(const float *xyz, uint64_t n);
void orc_kdtree_free(orc_kdtree *t);
uint64_t orc_kdtree_len(const orc_kdtree *t);
/* copies the implicit-tree array (kd_tree.rs:24): xyz_out[3n], idx_out[n] */
void orc_kdtree_dump(const orc_kdtree *t, float *xyz_out, uint64_t *idx_out);
/* KDTree::nearest_to_position (kd_tree.rs:92-96): returns count */
uint64_t orc_kdtree_k_nearest(const orc_kdtree *t, const float q[3], orc_entry *out, uint64_t k,
                              float max_distance);
/* batch of the above; out[nq*k], counts[nq]; threads<=1 -> sequential */
void orc_kdtree_k_nearest_batch(const orc_kdtree *t, const float *q, uint64_t nq, orc_entry *out,
                                uint32_t *counts, uint64_t k, float max_distance, int threads);
/* KDTree::nearest (kd_tree.rs:86-90): radius query; returns number found,
 * writes at most cap entries (stable sorted by distance). */
uint64_t orc_kdtree_nearest(const orc_kdtree *t, const float q[3], float max_distance,
                            orc_entry *out, uint64_t cap);
/* KDTree::empty (kd_tree.rs:98-102) */
int orc_kdtree_empty(const orc_kdtree *t, const float q[3], float max_distance);

/* brute force: the k smallest (distance, index) pairs in lexicographic
 * order, distance computed as metric.rs:21-26.  Returns count. */
uint64_t orc_brute_k_nearest(const float *xyz, uint64_t n, const float q[3], orc_entry *out,
                             uint64_t k, float max_distance);

/* ---- curvature consumer (treee/src/calculations.rs:328-373,467-498) ------ */
void orc_fast_eigenvalues(const float m[9] /* row-major 3x3 */, float out[3]);
/* curve from an explicit neighbour list (indices into xyz) */
float orc_curve_from_neighbors(const float *xyz, const orc_entry *nb, uint64_t count);
/* the same for nq given lists (u32 index[nq][k], counts[nq] or NULL = k), summed in the order given */
void orc_curve_from_lists(const float *xyz, const uint32_t *index, const uint32_t *counts, uint64_t nq, uint64_t k,
                          float *curve_out, int threads);
// the same with acos / cos of this host's libm (libm != 0) instead of include/treee_f32_trig.h: what a Rust build of the
// reference on this machine computes (f32::acos / f32::cos forward to acosf / cosf)
void orc_curve_from_lists_trig(const float *xyz, const uint32_t *index, const uint32_t *counts, uint64_t nq, uint64_t k,
                               float *curve_out, int threads, int libm);
/* include/treee_f32_trig.h (the shared acos / cos) beside this host's libm */
void orc_trig_f32(const float *x, uint64_t n, float *acos_out, float *cos_out, float *libm_acos_out, float *libm_cos_out);
/* full loop: tree over xyz[n], k neighbours (k=31 in the reference), radius */
void orc_curvature(const float *xyz, uint64_t n, uint64_t k, float max_distance, float *curve_out,
                   int threads);
/* calculations.rs:455-457 */
uint32_t orc_map_to_u32(float v);

/* ---- Calculations phase around the kNN consumer (calc_oracle.cpp) -------- */
/* Tree::new + Tree::insert over xz[n][2] in order (treee/src/segmenting.rs:258-269,346-396); writes the hull
 * vertices (at most cap) and Tree::statistics().area (segmenting.rs:320-323,526-545); returns the vertex count */
uint64_t orc_hull_insert_all(const float *xz, uint64_t n, float *hull_out, uint64_t cap, float *area_out);
/* get_size_areas (treee/src/calculations.rs:429-452); cls == NULL: every point, else only points with
 * cls[i] == want_cls.  Returns the slice count, writes at most cap areas. */
uint64_t orc_size_areas(const float *xyz, uint64_t n, const uint8_t *cls, int32_t want_cls, float min, float height,
                        float *areas_out, uint64_t cap);
/* SegmentData::new (calculations.rs:133-175) for ONE segment's points: min/max, SegmentInformation::new
 * (:210-276) -> info = {trunk_height, crown_height, ground_sep, crown_sep}, classification 0/1/2 = Ground/Trunk/Crown */
void orc_segment_data(const float *xyz, uint64_t n, float min_out[3], float max_out[3], float info_out[4], uint8_t *class_out);
/* SegmentInformation::update (calculations.rs:278-327) without the curve branch: expansion[n], height[n];
 * min / max = the segment's min.y / max.y (interactive/mod.rs:224-231) */
void orc_segment_update(const float *xyz, const uint8_t *cls, uint64_t n, float min, float max, const float info[4],
                        float *expansion_out, float *height_out);
/* DisplayData::new (treee/src/interactive/mod.rs:1002-1039) for one segment: the four per-point `value` arrays */
void orc_display_values(const uint8_t *cls, const float *curve, const float *expansion, const float *height, uint64_t n,
                        uint32_t *cls_val, uint32_t *curve_val, uint32_t *exp_val, uint32_t *height_val);

/* ---- ingest (treee/src/laz.rs, treee/src/loading.rs) --------------------- */
/* Chunk::next_point (laz.rs:202-213) over n records of `stride` bytes */
void orc_decode_las_records(const uint8_t *rec, uint64_t n, uint32_t stride, const double scale[3],
                            const double offset[3], const double center[3], float *xyz_out);
/* Header::new + Laz::new arithmetic (laz.rs:33-74,228-287).  Returns 0 ok,
 * -1 on bad signature / short buffer. */
typedef struct {
	uint64_t number_of_point_records;
	uint32_t offset_to_point_data;
	uint16_t header_size;
	uint16_t point_data_record_length;
	uint8_t point_data_record_format;
	uint8_t version_major, version_minor;
	double scale[3], offset[3];
	double min[3], max[3]; /* swizzled (x, z, -y) like laz.rs:72-73 */
	double center[3];      /* na::center(min,max) laz.rs:74 */
} orc_las_header;
int orc_las_header_parse(const uint8_t *bytes, uint64_t len, orc_las_header *out);
/* loading.rs:99: floor(p.y) as isize */
void orc_slice_keys(const float *xyz, uint64_t n, int64_t *key_out);
/* loading.rs:97-101 + segmenting.rs:121-133: dense slices indexed key-min, points in
 * arrival order.  order_out[n] = point indices grouped by slice; offsets_out[layers+1].
 * returns layers (0 if n==0); *min_key_out receives min key. */
uint64_t orc_slice_bin(const float *xyz, uint64_t n, int64_t *min_key_out, uint64_t *order_out,
                       uint64_t *offsets_out, uint64_t offsets_cap);

/* ---- splat (render/src/point_cloud.wgsl + fixed function state) ---------- */
/* nalgebra Perspective3::new(aspect,fovy,near,far).to_homogeneous(), column-major
 * (render/src/camera_3d.rs:16-18) */
void orc_perspective(float aspect, float fovy, float znear, float zfar, float out[16]);
/* software rasteriser: depth_out[w*h] (cleared 1.0), id_out[w*h] (cleared 0),
 * index_out (optional, winning point index or 0xFFFFFFFF). */
void orc_splat(const float *xyz, const uint32_t *value, uint64_t n, const float view[16],
               const float proj[16], float scale, uint32_t vmin, uint32_t vmax, uint32_t w,
               uint32_t h, float *depth_out, uint32_t *id_out, uint32_t *index_out);
/* render/src/eye_dome.wgsl:37-67 on a depth buffer; alpha_out[w*h] */
void orc_eye_dome(const float *depth, uint32_t w, uint32_t h, float strength, float *alpha_out);

/* ---- octree / LOD (BUILD-DEFINED, no reference counterpart) -------------- */
typedef struct {
	uint32_t key_prefix; /* morton prefix of the cell at `level` */
	uint32_t level;      /* 0 = root */
	uint32_t first;      /* first sorted position */
	uint32_t count;
	uint32_t first_child; /* node index of first child, 0xFFFFFFFF for leaf */
	uint32_t child_mask;  /* bit c set <=> octant c non-empty */
	uint32_t parent;      /* 0xFFFFFFFF for root */
	uint32_t lod_first;   /* offset into lod list (inner nodes), else 0xFFFFFFFF */
} orc_octree_node;

void orc_morton_keys(const float *xyz, uint64_t n, const float corner[3], float size,
                     uint32_t *keys_out);
/* Hilbert index (Skilling 2004) of the same 10-bit cell coordinates: the order of the kNN structure's buckets and the
 * middle term of the library's tie rule (distance, key, original index) */
void orc_hilbert_keys(const float *xyz, uint64_t n, const float corner[3], float size,
                      uint32_t *keys_out);
/* stable sort by key: perm_out[i] = original index of sorted position i */
void orc_sort_keys(const uint32_t *keys, uint64_t n, uint32_t *sorted_keys_out, uint32_t *perm_out);
/* nodes in BFS order; returns node count (call with nodes==NULL to size).
 * leaf_of_sorted[n] optional; lod list optional (lod_cap samples per inner node). */
uint64_t orc_octree_build(const uint32_t *sorted_keys, uint64_t n, uint32_t leaf_cap,
                          uint32_t lod_cap, orc_octree_node *nodes, uint64_t nodes_cap,
                          uint32_t *leaf_of_sorted, uint32_t *lod_list, uint64_t lod_cap_total,
                          uint64_t *lod_total_out);

/* the same build continued from a given front of nodes of one level (local indices; see octree_oracle.cpp) */
uint64_t orc_octree_build_from(const uint32_t *sorted_keys, uint64_t n, uint32_t leaf_cap, uint32_t lod_cap,
                               const orc_octree_node *roots, uint64_t n_roots, orc_octree_node *nodes, uint64_t nodes_cap,
                               uint32_t *leaf_of_sorted, uint32_t *lod_list, uint64_t lod_cap_total, uint64_t *lod_total_out);

#ifdef __cplusplus
}
#endif
#endif
