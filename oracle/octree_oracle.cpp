// octree_oracle.cpp -- CPU ORACLE (test infrastructure) for the Morton sort / octree / LOD
// stage.  BUILD-DEFINED: the reference snapshot contains no octree, LOD or Morton code at all
// (the only vestige is the cube (corner,size) frustum test at
// render/src/camera_3d.rs:20-51, whose cube parametrisation is reused here).  "Parity
// unpinned": this file is the definition the HIP path is held bit-exact to.
//
// Definition
//   q_a   = clamp(floor(((p_a - corner_a) / size) * 1024), 0, 1023)   (f32, no FMA, NaN -> 0)
//   key   = 30-bit Morton code, bit 3j+0 = x_j, 3j+1 = y_j, 3j+2 = z_j
//   sort  = stable by key (ties keep original index order)
//   nodes = BFS order; root = all points; a node at level l < 10 with count > leaf_cap is
//           split into its non-empty octants (child c = 3 key bits at level l+1), children
//           appended in octant order; otherwise it is a leaf
//   LOD   = for every inner node, in node order, m = min(count, lod_cap) sorted positions
//           first + floor(j * count / m), j = 0..m-1
#include "treee_oracle.h"

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

constexpr uint32_t DEPTH = 10;
constexpr uint32_t NONE = 0xFFFFFFFFu;

inline uint32_t quant(float p, float c, float size) {
	float t = ((p - c) / size) * 1024.0f;
	if (!(t >= 0.0f)) return 0; // negative or NaN
	if (t >= 1024.0f) return 1023;
	return (uint32_t)t;
}

inline uint32_t spread3(uint32_t v) { // 10 bits -> every third bit
	v &= 0x3FF;
	v = (v | (v << 16)) & 0x030000FF;
	v = (v | (v << 8)) & 0x0300F00F;
	v = (v | (v << 4)) & 0x030C30C3;
	v = (v | (v << 2)) & 0x09249249;
	return v;
}

} // namespace

extern "C" {

void orc_morton_keys(const float *xyz, uint64_t n, const float corner[3], float size,
                     uint32_t *keys_out) {
	for (uint64_t i = 0; i < n; ++i) {
		uint32_t qx = quant(xyz[3 * i + 0], corner[0], size);
		uint32_t qy = quant(xyz[3 * i + 1], corner[1], size);
		uint32_t qz = quant(xyz[3 * i + 2], corner[2], size);
		keys_out[i] = spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2);
	}
}

// 30-bit Hilbert index of the same three 10-bit cell coordinates (J. Skilling, "Programming the Hilbert curve", AIP Conf.
// Proc. 707, 2004: AxesToTranspose, then the transposed bits read most significant first).  BUILD-DEFINED like the Morton
// key: the kNN structure orders its buckets by it, and the library's tie rule is (distance, this key, original index).
void orc_hilbert_keys(const float *xyz, uint64_t n, const float corner[3], float size, uint32_t *keys_out) {
	for (uint64_t i = 0; i < n; ++i) {
		uint32_t X[3] = {quant(xyz[3 * i + 0], corner[0], size), quant(xyz[3 * i + 1], corner[1], size), quant(xyz[3 * i + 2], corner[2], size)};
		const uint32_t M = 1u << 9;
		for (uint32_t Q = M; Q > 1; Q >>= 1) { // inverse undo
			const uint32_t P = Q - 1;
			for (int a = 0; a < 3; ++a) {
				if (X[a] & Q) {
					X[0] ^= P; // invert
				} else { // exchange
					const uint32_t t = (X[0] ^ X[a]) & P;
					X[0] ^= t;
					X[a] ^= t;
				}
			}
		}
		for (int a = 1; a < 3; ++a) X[a] ^= X[a - 1]; // Gray encode
		uint32_t t = 0;
		for (uint32_t Q = M; Q > 1; Q >>= 1)
			if (X[2] & Q) t ^= Q - 1;
		for (int a = 0; a < 3; ++a) X[a] ^= t;
		uint32_t h = 0;
		for (int b = 9; b >= 0; --b)
			for (int a = 0; a < 3; ++a) h = (h << 1) | ((X[a] >> b) & 1u);
		keys_out[i] = h;
	}
}

void orc_sort_keys(const uint32_t *keys, uint64_t n, uint32_t *sorted_keys_out, uint32_t *perm_out) {
	std::vector<uint32_t> perm(n);
	std::iota(perm.begin(), perm.end(), 0u);
	std::stable_sort(perm.begin(), perm.end(),
	                 [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
	for (uint64_t i = 0; i < n; ++i) {
		if (perm_out) perm_out[i] = perm[i];
		if (sorted_keys_out) sorted_keys_out[i] = keys[perm[i]];
	}
}

// The build from a given front: `roots` are nodes of ONE level whose ranges lie inside keys[0..n) (first relative to keys);
// everything below them is built breadth first exactly as from the single root.  This is what one rank of a sharded
// build runs on its owned key range (roots = its nodes of the shard level; include/treee_hip.h "the sharded octree");
// the whole-cloud build is the same code with the root as the only front node.  Indices in the result are LOCAL (node
// index = position in nodes_out, roots first; parents of the roots are left as given).
uint64_t orc_octree_build_from(const uint32_t *keys, uint64_t n, uint32_t leaf_cap, uint32_t lod_cap, const orc_octree_node *roots,
                               uint64_t n_roots, orc_octree_node *nodes_out, uint64_t nodes_cap, uint32_t *leaf_of_sorted,
                               uint32_t *lod_list, uint64_t lod_cap_total, uint64_t *lod_total_out) {
	(void)n;
	std::vector<orc_octree_node> nodes(roots, roots + n_roots);
	for (size_t i = 0; i < nodes.size(); ++i) {
		orc_octree_node nd = nodes[i];
		if (!(nd.count > leaf_cap && nd.level < DEPTH)) continue;
		uint32_t shift = 3 * (DEPTH - (nd.level + 1));
		const uint32_t *lo = keys + nd.first, *hi = lo + nd.count;
		uint32_t mask = 0, first_child = NONE;
		for (uint32_t c = 0; c < 8; ++c) {
			uint32_t pre = nd.key_prefix * 8 + c;
			auto cmp = [shift](uint32_t k, uint32_t p) { return (k >> shift) < p; };
			const uint32_t *a = std::lower_bound(lo, hi, pre, cmp);
			const uint32_t *b = std::lower_bound(lo, hi, pre + 1, cmp);
			if (a == b) continue;
			if (first_child == NONE) first_child = (uint32_t)nodes.size();
			mask |= 1u << c;
			nodes.push_back(orc_octree_node{pre, nd.level + 1, (uint32_t)(a - keys),
			                                (uint32_t)(b - a), NONE, 0, (uint32_t)i, NONE});
		}
		nodes[i].first_child = first_child;
		nodes[i].child_mask = mask;
	}
	uint64_t lod_total = 0;
	for (size_t i = 0; i < nodes.size(); ++i) {
		orc_octree_node &nd = nodes[i];
		if (nd.first_child == NONE) {
			if (leaf_of_sorted)
				for (uint32_t j = 0; j < nd.count; ++j) leaf_of_sorted[nd.first + j] = (uint32_t)i;
			continue;
		}
		uint32_t m = std::min(nd.count, lod_cap);
		nd.lod_first = (uint32_t)lod_total;
		for (uint32_t j = 0; j < m; ++j) {
			uint64_t pos = nd.first + (uint64_t)j * nd.count / m;
			if (lod_list && lod_total + j < lod_cap_total) lod_list[lod_total + j] = (uint32_t)pos;
		}
		lod_total += m;
	}
	if (lod_total_out) *lod_total_out = lod_total;
	if (nodes_out)
		for (size_t i = 0; i < nodes.size() && i < nodes_cap; ++i) nodes_out[i] = nodes[i];
	return nodes.size();
}

uint64_t orc_octree_build(const uint32_t *keys, uint64_t n, uint32_t leaf_cap, uint32_t lod_cap,
                          orc_octree_node *nodes_out, uint64_t nodes_cap, uint32_t *leaf_of_sorted,
                          uint32_t *lod_list, uint64_t lod_cap_total, uint64_t *lod_total_out) {
	const orc_octree_node root{0, 0, 0, (uint32_t)n, NONE, 0, NONE, NONE};
	return orc_octree_build_from(keys, n, leaf_cap, lod_cap, &root, n > 0 ? 1 : 0, nodes_out, nodes_cap, leaf_of_sorted, lod_list,
	                             lod_cap_total, lod_total_out);
}

} // extern "C"
