// knn_oracle.cpp -- CPU ORACLE (test infrastructure): restatement of the
// `k-nearest` crate for N=3, f32, EuclideanDistanceSquared, the only
// instantiation treee uses (treee/src/calculations.rs:393,398-404).
// Follows k-nearest/src/kd_tree.rs, best_set.rs, metric.rs line by line.
// Parity unpinned (see treee_oracle.h).
#include "treee_oracle.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t INVALID = ~uint64_t(0); // usize::MAX

// kd_tree.rs:24  tree: Vec<([Value; N], usize)>
struct Rec {
	float p[3];
	uint64_t idx;
};

// metric.rs:21-26  fold(0, |sum, v| sum + v) over (l-r)^2, no FMA
inline float distance(const float *l, const float *r) {
	float sum = 0.0f;
	for (int d = 0; d < 3; ++d) {
		float v = l[d] - r[d];
		v = v * v;
		sum = sum + v;
	}
	return sum;
}

// metric.rs:28-31
inline float distance_plane(const float *position, float plane, size_t dim) {
	float diff = position[dim] - plane;
	return diff * diff;
}

// kd_tree.rs:58-80  quickselect, Lomuto partition, pivot = middle of [lower,upper]
void partition(size_t target, size_t dim, Rec *pos, size_t len) {
	size_t lower = 0;
	size_t upper = len - 1;
	for (;;) {
		size_t pivot_index = (upper - lower) / 2 + lower;
		float pivot = pos[pivot_index].p[dim];
		std::swap(pos[pivot_index], pos[upper]);
		size_t store = lower;
		for (size_t i = lower; i < upper; ++i) {
			if (pos[i].p[dim] < pivot) {
				std::swap(pos[store], pos[i]);
				store += 1;
			}
		}
		std::swap(pos[store], pos[upper]);
		if (target == store)
			break;
		else if (target < store)
			upper = store - 1;
		else
			lower = store + 1;
	}
}

// kd_tree.rs:46-56
void create_tree(size_t dim, Rec *tree, size_t len) {
	size_t middle = len / 2;
	partition(middle, dim, tree, len);
	size_t next_dim = (dim + 1) % 3;
	if (0 < middle) create_tree(next_dim, tree, middle);
	if (middle < len - 1) create_tree(next_dim, tree + middle + 1, len - middle - 1);
}

// best_set.rs:9-82
struct FixedSet {
	orc_entry *values;
	size_t len;

	FixedSet(float max_distance, orc_entry *v, size_t k) : values(v), len(k) { // :20-28
		for (size_t i = 0; i < k; ++i) {
			values[i].distance = max_distance;
			values[i]._pad = 0;
			values[i].index = INVALID;
		}
	}
	void fix_down(size_t index, size_t max_size) { // :30-54
		for (;;) {
			size_t swap_index = index;
			float swap_value = values[index].distance;
			size_t child_left = index * 2 + 1;
			if (child_left < max_size && values[child_left].distance > swap_value) {
				swap_index = child_left;
				swap_value = values[child_left].distance;
			}
			size_t child_right = index * 2 + 2;
			if (child_right < max_size && values[child_right].distance > swap_value) {
				swap_index = child_right;
			}
			if (swap_index == index) break;
			std::swap(values[index], values[swap_index]);
			index = swap_index;
		}
	}
	size_t result() { // :56-66
		size_t size = len;
		for (size_t end = len; end-- > 0;) {
			if (values[0].index == INVALID) size = end;
			std::swap(values[0], values[end]);
			fix_down(0, end);
		}
		return size;
	}
	float dist() const { return values[0].distance; } // :73-75
	void insert(float d, uint64_t idx) {             // :77-81
		values[0].distance = d;
		values[0].index = idx;
		fix_down(0, len);
	}
};

// best_set.rs:84-109
struct DynamicSet {
	float distance;
	std::vector<orc_entry> values;
	float dist() const { return distance; }
	void insert(float d, uint64_t idx) { values.push_back(orc_entry{d, 0, idx}); }
};

// best_set.rs:111-131
struct EmptySet {
	bool some;
	float v;
	float dist() const { return some ? v : 0.0f; } // unwrap_or_default
	void insert(float, uint64_t) { some = false; }
};

// kd_tree.rs:104-146
template <class Set>
void search_nearest(const Rec *tree, size_t len, const float *position, size_t dim, Set &best) {
	if (len < 32) { // :111-119 linear scan in array order
		for (size_t i = 0; i < len; ++i) {
			float diff = distance(tree[i].p, position);
			if (diff < best.dist()) best.insert(diff, tree[i].idx);
		}
		return;
	}
	size_t middle = len / 2; // :122-127
	const Rec &m = tree[middle];
	float diff = distance(m.p, position);
	if (diff < best.dist()) best.insert(diff, m.idx);

	size_t next_dim = (dim + 1) % 3; // :130-136
	bool is_left = position[dim] < m.p[dim];
	if (is_left)
		search_nearest(tree, middle, position, next_dim, best);
	else
		search_nearest(tree + middle + 1, len - middle - 1, position, next_dim, best);

	if (distance_plane(position, m.p[dim], dim) < best.dist()) { // :139-145
		if (is_left)
			search_nearest(tree + middle + 1, len - middle - 1, position, next_dim, best);
		else
			search_nearest(tree, middle, position, next_dim, best);
	}
}

} // namespace

struct orc_kdtree {
	std::vector<Rec> tree;
};

extern "C" {

orc_kdtree *orc_kdtree_new(const float *xyz, uint64_t n) {
	if (n == 0) return nullptr; // reference: len()-1 underflow panic (kd_tree.rs:53,60)
	orc_kdtree *t = new orc_kdtree;
	t->tree.resize(n);
	for (uint64_t i = 0; i < n; ++i) { // kd_tree.rs:37-39
		t->tree[i].p[0] = xyz[3 * i + 0];
		t->tree[i].p[1] = xyz[3 * i + 1];
		t->tree[i].p[2] = xyz[3 * i + 2];
		t->tree[i].idx = i;
	}
	create_tree(0, t->tree.data(), n);
	return t;
}

void orc_kdtree_free(orc_kdtree *t) { delete t; }

uint64_t orc_kdtree_len(const orc_kdtree *t) { return t ? t->tree.size() : 0; }

void orc_kdtree_dump(const orc_kdtree *t, float *xyz_out, uint64_t *idx_out) {
	for (size_t i = 0; i < t->tree.size(); ++i) {
		if (xyz_out) std::memcpy(xyz_out + 3 * i, t->tree[i].p, 12);
		if (idx_out) idx_out[i] = t->tree[i].idx;
	}
}

uint64_t orc_kdtree_k_nearest(const orc_kdtree *t, const float q[3], orc_entry *out, uint64_t k,
                              float max_distance) {
	if (k == 0) return 0; // reference panics at values[0] (best_set.rs:74)
	FixedSet set(max_distance, out, k);
	search_nearest(t->tree.data(), t->tree.size(), q, 0, set);
	return set.result();
}

void orc_kdtree_k_nearest_batch(const orc_kdtree *t, const float *q, uint64_t nq, orc_entry *out,
                                uint32_t *counts, uint64_t k, float max_distance, int threads) {
	auto work = [&](uint64_t lo, uint64_t hi) {
		for (uint64_t i = lo; i < hi; ++i)
			counts[i] = (uint32_t)orc_kdtree_k_nearest(t, q + 3 * i, out + i * k, k, max_distance);
	};
	if (threads <= 1) {
		work(0, nq);
		return;
	}
	std::vector<std::thread> pool;
	uint64_t per = (nq + threads - 1) / threads;
	for (int th = 0; th < threads; ++th) {
		uint64_t lo = std::min<uint64_t>(nq, per * th), hi = std::min<uint64_t>(nq, lo + per);
		if (lo < hi) pool.emplace_back(work, lo, hi);
	}
	for (auto &th : pool) th.join();
}

uint64_t orc_kdtree_nearest(const orc_kdtree *t, const float q[3], float max_distance,
                            orc_entry *out, uint64_t cap) {
	DynamicSet set{max_distance, {}};
	search_nearest(t->tree.data(), t->tree.size(), q, 0, set);
	// best_set.rs:94-98: sort_by partial_cmp (stable)
	std::stable_sort(set.values.begin(), set.values.end(),
	                 [](const orc_entry &a, const orc_entry &b) { return a.distance < b.distance; });
	uint64_t n = set.values.size();
	for (uint64_t i = 0; i < n && i < cap; ++i) out[i] = set.values[i];
	return n;
}

int orc_kdtree_empty(const orc_kdtree *t, const float q[3], float max_distance) {
	EmptySet set{true, max_distance};
	search_nearest(t->tree.data(), t->tree.size(), q, 0, set);
	return set.some ? 1 : 0;
}

uint64_t orc_brute_k_nearest(const float *xyz, uint64_t n, const float q[3], orc_entry *out,
                             uint64_t k, float max_distance) {
	std::vector<orc_entry> all;
	all.reserve(n);
	for (uint64_t i = 0; i < n; ++i) {
		float d = distance(xyz + 3 * i, q);
		if (d < max_distance) all.push_back(orc_entry{d, 0, i});
	}
	auto less = [](const orc_entry &a, const orc_entry &b) {
		return a.distance < b.distance || (a.distance == b.distance && a.index < b.index);
	};
	uint64_t m = std::min<uint64_t>(k, all.size());
	std::partial_sort(all.begin(), all.begin() + m, all.end(), less);
	for (uint64_t i = 0; i < m; ++i) out[i] = all[i];
	for (uint64_t i = m; i < k; ++i) out[i] = orc_entry{max_distance, 0, INVALID};
	return m;
}

} // extern "C"
