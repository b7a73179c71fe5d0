// shard_plan.cpp -- the host arithmetic of the sharded octree: who owns which Morton cells, what
// travels where, the replicated top of the tree, and the global numbering of the nodes the ranks build locally.  No GPU and
// no communication in here: every rank evaluates the same functions on the same all-gathered counts and gets the same plan.
// BUILD-DEFINED like the octree itself (no reference counterpart; the reference is one process and has no octree) -- the
// definition it must reproduce is the single-GPU table of octree.hip / oracle/octree_oracle.cpp for the whole cloud.
#include "shard_plan.h"

using namespace treee;

namespace {
constexpr uint32_t DEPTH = 10;
}

extern "C" {

int32_t treee_shard_plan_create(const uint32_t *cell_counts, int32_t world, uint32_t top_level, uint32_t leaf_cap, uint32_t lod_cap,
                                treee_shard_plan **out) {
	if (!cell_counts || !out || world < 1) return set_err(TREEE_ERR_INVALID, "treee_shard_plan_create: bad argument");
	if (top_level < 1 || top_level > 6) return set_err(TREEE_ERR_INVALID, "treee_shard_plan_create: top_level %u outside [1, 6]", top_level);
	*out = nullptr;
	try {
		std::unique_ptr<treee_shard_plan> p(new treee_shard_plan);
		p->world = world;
		p->L = top_level;
		p->leaf_cap = leaf_cap ? leaf_cap : 256;
		p->lod_cap = lod_cap ? lod_cap : 256;
		const uint32_t nc = p->ncell = 1u << (3u * top_level);
		p->counts.assign(cell_counts, cell_counts + (size_t)world * nc);
		p->prefix.assign((size_t)nc + 1, 0);
		for (uint32_t c = 0; c < nc; ++c) {
			uint64_t t = 0;
			for (int r = 0; r < world; ++r) t += cell_counts[(size_t)r * nc + c];
			p->prefix[c + 1] = p->prefix[c] + t;
		}
		const uint64_t N = p->prefix[nc];
		if (N >= 0xFFFFFFFFull) return set_err(TREEE_ERR_UNSUPPORTED, "sharded octree: %llu points exceed the 32-bit global index", (unsigned long long)N);
		p->index_base.assign((size_t)world + 1, 0);
		for (int r = 0; r < world; ++r) {
			uint64_t t = 0;
			for (uint32_t c = 0; c < nc; ++c) t += cell_counts[(size_t)r * nc + c];
			p->index_base[r + 1] = p->index_base[r] + t;
		}
		// ownership: rank r starts at the first cell boundary at or behind r * N / world points
		p->cell_begin.assign((size_t)world + 1, nc);
		p->cell_begin[0] = 0;
		uint32_t c = 0;
		for (int r = 1; r < world; ++r) {
			const uint64_t target = (uint64_t)(((unsigned __int128)N * (unsigned)r) / (unsigned)world);
			while (c < nc && p->prefix[c] < target) ++c;
			p->cell_begin[r] = c;
		}
		// the top of the tree, breadth first, from the cell totals: a node of level l < L with key prefix q spans the cells
		// [q << 3(L-l), (q+1) << 3(L-l))
		std::vector<treee_octree_node> all;
		if (N > 0) all.push_back(treee_octree_node{0u, 0u, 0u, (uint32_t)N, NONE32, 0u, NONE32, NONE32});
		for (size_t i = 0; i < all.size(); ++i) {
			const treee_octree_node nd = all[i];
			if (nd.level >= top_level) continue; // expanded by its owner
			if (!(nd.count > p->leaf_cap && nd.level < DEPTH)) continue;
			const uint32_t sh = 3u * (top_level - (nd.level + 1u));
			uint32_t mask = 0, first_child = NONE32;
			for (uint32_t o = 0; o < 8; ++o) {
				const uint32_t pre = nd.key_prefix * 8u + o;
				const uint64_t a = p->prefix[(size_t)pre << sh], b = p->prefix[((size_t)pre + 1u) << sh];
				if (a == b) continue;
				if (first_child == NONE32) first_child = (uint32_t)all.size();
				mask |= 1u << o;
				all.push_back(treee_octree_node{pre, nd.level + 1u, (uint32_t)a, (uint32_t)(b - a), NONE32, 0u, (uint32_t)i, NONE32});
			}
			all[i].first_child = first_child;
			all[i].child_mask = mask;
		}
		size_t n_top = 0;
		while (n_top < all.size() && all[n_top].level < top_level) ++n_top;
		p->top.assign(all.begin(), all.begin() + n_top);
		p->level.assign(all.begin() + n_top, all.end());
		uint32_t lod = 0;
		for (auto &nd : p->top)
			if (nd.first_child != NONE32) {
				nd.lod_first = lod;
				lod += nd.count < p->lod_cap ? nd.count : p->lod_cap;
			}
		p->n_top_lod = lod;
		for (uint32_t l = 0; l < TREEE_SHARD_LEVELS; ++l) p->top_begin[l] = (uint32_t)n_top;
		for (size_t i = n_top; i-- > 0;) p->top_begin[p->top[i].level] = (uint32_t)i;
		for (uint32_t l = top_level; l-- > 0;) // an absent level starts where the next one does
			if (p->top_begin[l] > p->top_begin[l + 1]) p->top_begin[l] = p->top_begin[l + 1];
		// nodes of level L by owner (they are in key order, owners hold key ranges)
		p->level_first.assign((size_t)world + 1, (uint32_t)p->level.size());
		p->level_first[0] = 0;
		size_t j = 0;
		for (int r = 0; r < world; ++r) {
			while (j < p->level.size() && p->level[j].key_prefix < p->cell_begin[r]) ++j;
			p->level_first[r] = (uint32_t)j;
		}
		*out = p.release();
		return TREEE_OK;
	} catch (const std::exception &e) {
		return set_err(TREEE_ERR_NOMEM, "treee_shard_plan_create: %s", e.what());
	}
}

int32_t treee_shard_plan_free(treee_shard_plan *p) {
	delete p;
	return TREEE_OK;
}

int32_t treee_shard_plan_info_get(const treee_shard_plan *p, treee_shard_plan_info *info) {
	if (!p || !info) return set_err(TREEE_ERR_INVALID, "treee_shard_plan_info_get: null argument");
	info->n_total = p->prefix[p->ncell];
	info->world = (uint32_t)p->world;
	info->top_level = p->L;
	info->n_top_nodes = (uint32_t)p->top.size();
	info->n_top_lod = p->n_top_lod;
	info->n_level_nodes = (uint32_t)p->level.size();
	info->leaf_cap = p->leaf_cap;
	info->lod_cap = p->lod_cap;
	return TREEE_OK;
}

int32_t treee_shard_plan_rank_get(const treee_shard_plan *p, int32_t rank, treee_shard_plan_rank *out) {
	if (!p || !out || rank < 0 || rank >= p->world) return set_err(TREEE_ERR_INVALID, "treee_shard_plan_rank_get: bad argument");
	out->cell_begin = p->cell_begin[rank];
	out->cell_end = p->cell_begin[rank + 1];
	out->first = p->prefix[out->cell_begin];
	out->count = p->prefix[out->cell_end] - out->first;
	out->index_base = p->index_base[rank];
	out->level_node_first = p->level_first[rank];
	out->level_node_count = p->level_first[rank + 1] - p->level_first[rank];
	return TREEE_OK;
}

int32_t treee_shard_plan_top_nodes(const treee_shard_plan *p, treee_octree_node *nodes_out) {
	if (!p || (!nodes_out && !p->top.empty())) return set_err(TREEE_ERR_INVALID, "treee_shard_plan_top_nodes: null argument");
	// children of level L - 1 nodes are level nodes: their index in the whole table is n_top + position, which is what the
	// breadth-first construction above already stored
	for (size_t i = 0; i < p->top.size(); ++i) nodes_out[i] = p->top[i];
	return TREEE_OK;
}

int32_t treee_shard_plan_level_nodes(const treee_shard_plan *p, int32_t rank, treee_octree_node *nodes_out) {
	if (!p || rank < 0 || rank >= p->world) return set_err(TREEE_ERR_INVALID, "treee_shard_plan_level_nodes: bad argument");
	const uint64_t g0 = p->prefix[p->cell_begin[rank]];
	for (uint32_t j = p->level_first[rank]; j < p->level_first[rank + 1]; ++j) {
		treee_octree_node nd = p->level[j];
		nd.first = (uint32_t)(nd.first - g0);
		nodes_out[j - p->level_first[rank]] = nd;
	}
	return TREEE_OK;
}

int32_t treee_shard_plan_exchange(const treee_shard_plan *p, int32_t rank, uint64_t *send_first, uint64_t *send_count, uint64_t *recv_count) {
	if (!p || !send_first || !send_count || !recv_count || rank < 0 || rank >= p->world)
		return set_err(TREEE_ERR_INVALID, "treee_shard_plan_exchange: bad argument");
	const uint32_t nc = p->ncell;
	auto range = [&](int from, int to) { // points of rank `from`'s input inside the cells rank `to` owns, and how many precede them
		const uint32_t *c = p->counts.data() + (size_t)from * nc;
		uint64_t before = 0, inside = 0;
		for (uint32_t i = 0; i < p->cell_begin[to]; ++i) before += c[i];
		for (uint32_t i = p->cell_begin[to]; i < p->cell_begin[to + 1]; ++i) inside += c[i];
		return std::make_pair(before, inside);
	};
	for (int r = 0; r < p->world; ++r) {
		const auto s = range(rank, r);
		send_first[r] = s.first;
		send_count[r] = s.second;
		recv_count[r] = range(r, rank).second;
	}
	return TREEE_OK;
}

int32_t treee_shard_plan_levels(const treee_shard_plan *p, const uint32_t *level_counts, const uint32_t *lod_counts, int32_t rank,
                                treee_shard_levels *out) {
	if (!p || !level_counts || !lod_counts || !out || rank < 0 || rank >= p->world)
		return set_err(TREEE_ERR_INVALID, "treee_shard_plan_levels: bad argument");
	const uint32_t L = p->L;
	std::memset(out, 0, sizeof(*out));
	for (uint32_t l = 0; l < L; ++l) {
		out->node_begin[l] = out->own_node_first[l] = p->top_begin[l];
		out->lod_begin[l] = out->own_lod_first[l] = 0; // (per node: treee_shard_plan_top_nodes)
	}
	uint64_t node = p->top.size(), lod = p->n_top_lod;
	for (uint32_t l = L; l < TREEE_SHARD_LEVELS; ++l) {
		out->node_begin[l] = (uint32_t)node;
		out->lod_begin[l] = (uint32_t)lod;
		uint64_t nb = 0, lb = 0;
		for (int r = 0; r < p->world; ++r) {
			const uint32_t cn = level_counts[(size_t)r * TREEE_SHARD_LEVELS + l], cl = lod_counts[(size_t)r * TREEE_SHARD_LEVELS + l];
			if (r == rank) {
				out->own_node_first[l] = (uint32_t)(node + nb);
				out->own_node_count[l] = cn;
				out->own_lod_first[l] = (uint32_t)(lod + lb);
				out->own_lod_count[l] = cl;
			}
			nb += cn;
			lb += cl;
		}
		if (l == L && nb != p->level.size())
			return set_err(TREEE_ERR_INVALID, "treee_shard_plan_levels: the ranks report %llu nodes of level %u, the plan holds %zu",
			               (unsigned long long)nb, L, p->level.size());
		node += nb;
		lod += lb;
	}
	if (node >= 0xFFFFFFFFull || lod >= 0xFFFFFFFFull) return set_err(TREEE_ERR_UNSUPPORTED, "sharded octree: node or LOD index exceeds 32 bits");
	out->n_nodes_total = node;
	out->n_lod_total = lod;
	return TREEE_OK;
}

} // extern "C"
