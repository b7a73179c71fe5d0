// segmenting.cpp -- the slice segmentation of the reference, HOST code like the reference's:
//   segmentation       treee/src/segmenting.rs:110-235   slices top-down, centroid hand-off, Voronoi cells, assignment
//   Tree               treee/src/segmenting.rs:242-396   convex area: new, from_points, distance, outside_bounds, contains, insert
//   TreeSet            treee/src/segmenting.rs:419-524   new, add_point, filter_trees, tree_positions
//   centroid           treee/src/segmenting.rs:526-545
// It produces the `segment: Vec<u32>` that the Calculations phase, the per-segment kNN and the splat consume.  The
// reference's run is not reproducible even against itself -- `points.shuffle(&mut thread_rng())` (:425), iteration over a
// RandomState HashMap (:488, :166), `rand::random()` ids (:516-519) -- so this is the DETERMINISTIC VARIANT SURVEY.md F1
// prescribes, every departure marked (D1..D4) below:
//   D1  the shuffle is Fisher-Yates driven by SplitMix64(seed, slice key) instead of rand's ThreadRng
//   D2  HashMap<u32, Centroid> is iterated in ascending key order
//   D3  a new tree's id is the next integer (1, 2, 3 ... in creation order, top slice first; 0 stays DELETED_INDEX)
//   D4  the Voronoi cells of `voronator` 0.2.1 (Cargo.lock:2920-2921; source absent from the snapshot) are restated from
//       its published behaviour: the Delaunay dual of the sites plus four helper sites one box width / height beyond the
//       middle of each side, every cell clipped to the box -- computed here as the intersection of the box with the
//       bisector half-planes of all other sites (the same polygons; vertex order starts at the box's min corner, counter-
//       clockwise), in f64 and stored as f32 like `Centroid::from_xy` (:410-412).  Parity with the crate: unpinned.
// Slices run on a thread pool as the reference's run on rayon workers, the centroid hand-off chain between them is
// sequential in both; the result does not depend on the number of threads.  The work is inherently serial per slice -- every point is tested against the hulls the earlier points built --
// which is why the reference keeps it on the CPU and so does this library: there is no GPU path to fall back from.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <deque>
#include <functional>
#include <map>
#include <thread>

using namespace treee;

namespace {

struct P2 {
	float x, y;
};
inline P2 operator-(P2 a, P2 b) { return {a.x - b.x, a.y - b.y}; }

// segmenting.rs:347-353 and its copies in distance / contains: signed distance of `point` to the edge a -> b along the
// outward unit normal normalize((dir.y, -dir.x)) (nalgebra: v / sqrt(x*x + y*y), dot = x*x' + y*y'; no FMA)
inline float edge_dist(P2 a, P2 b, P2 point) {
	const float dx = b.x - a.x, dy = b.y - a.y;
	const float ox = dy, oy = -dx;
	const float norm = std::sqrt(ox * ox + oy * oy);
	const float nx = ox / norm, ny = oy / norm;
	const float fx = point.x - a.x, fy = point.y - a.y;
	return nx * fx + ny * fy;
}

// segmenting.rs:526-545
inline void centroid(const std::vector<P2> &points, P2 *center_out, float *area_out) {
	float cx = 0.0f, cy = 0.0f, area = 0.0f;
	const P2 a = points[0];
	for (size_t i = 1; i + 1 < points.size(); ++i) {
		const P2 b = points[i] - a, c = points[i + 1] - a;
		const float tx = (b.x + c.x) / 3.0f, ty = (b.y + c.y) / 3.0f;
		const float t_area = b.x * c.y - b.y * c.x;
		cx += tx * t_area;
		cy += ty * t_area;
		area += t_area;
	}
	*center_out = {a.x + cx / area, a.y + cy / area};
	*area_out = area / 2.0f;
}

struct Tree { // segmenting.rs:242-247
	std::vector<P2> points;
	P2 min, max;

	static Tree make(P2 p) { // Tree::new :258-269
		Tree t;
		t.points = {p, {p.x + 0.1f, p.y}, {p.x, p.y + 0.1f}};
		t.min = p;
		t.max = {p.x + 0.1f, p.y + 0.1f};
		return t;
	}
	static Tree from_points(std::vector<P2> points) { // :271-296
		Tree t;
		if (points.empty()) {
			t.min = {3.402823466e38f, 3.402823466e38f};
			t.max = {-3.402823466e38f, -3.402823466e38f};
			return t;
		}
		if (points.size() == 1) {
			points.push_back({points[0].x + 0.1f, points[0].y + 0.0f});
			points.push_back({points[0].x + 0.0f, points[0].y + 0.1f});
		} else if (points.size() == 2) {
			const P2 diff = points[1] - points[0];
			const float vx = -diff.y, vy = diff.x;
			const float norm = std::sqrt(vx * vx + vy * vy);
			points.push_back({points[0].x + (vx / norm) * 0.1f, points[0].y + (vy / norm) * 0.1f});
		}
		t.min = t.max = points[0];
		for (size_t i = 1; i < points.size(); ++i) {
			t.min = {std::fmin(t.min.x, points[i].x), std::fmin(t.min.y, points[i].y)};
			t.max = {std::fmax(t.max.x, points[i].x), std::fmax(t.max.y, points[i].y)};
		}
		t.points = std::move(points);
		return t;
	}
	bool outside_bounds(P2 p, float md) const { // :318-323
		return p.x + md < min.x || max.x + md <= p.x || p.y + md < min.y || max.y + md <= p.y;
	}
	float distance(P2 p, float md) const { // :298-316
		if (outside_bounds(p, md)) return 3.402823466e38f;
		float best = -3.402823466e38f;
		for (size_t i = 0; i < points.size(); ++i) {
			const float d = edge_dist(points[i], points[(i + 1) % points.size()], p);
			if (d > md) return 3.402823466e38f;
			best = std::fmax(best, d);
		}
		return best;
	}
	bool contains(P2 p, float md) const { // :330-345
		if (outside_bounds(p, md)) return false;
		for (size_t i = 0; i < points.size(); ++i)
			if (edge_dist(points[i], points[(i + 1) % points.size()], p) > md) return false;
		return true;
	}
	void insert(P2 point) { // :346-396
		const size_t len = points.size();
		bool last = edge_dist(points[len - 1], points[0], point) > 0.0f;
		bool has_s = false, has_e = false;
		size_t start = 0, end = 0;
		for (size_t i = 0; i < len; ++i) {
			const bool out = edge_dist(points[i], points[(i + 1) % len], point) > 0.0f;
			if (!last && out) {
				start = i;
				has_s = true;
			} else if (last && !out) {
				end = i;
				has_e = true;
			}
			last = out;
		}
		if (!has_s || !has_e) return;
		if (end < start) {
			points.erase(points.begin() + (start + 1), points.end());
			points.push_back(point);
			points.erase(points.begin(), points.begin() + end);
		} else {
			points.erase(points.begin() + (start + 1), points.begin() + end);
			points.insert(points.begin() + (start + 1), point);
		}
		min = {std::fmin(min.x, point.x), std::fmin(min.y, point.y)};
		max = {std::fmax(max.x, point.x), std::fmax(max.y, point.y)};
	}
};

struct TreeSet { // segmenting.rs:414-524
	std::vector<Tree> trees;

	void add_point(P2 p, float md) { // :437-467
		std::vector<size_t> near;
		for (size_t i = 0; i < trees.size(); ++i) {
			const float dist = trees[i].distance(p, md);
			if (dist <= 0.0f) return;
			if (dist <= md) near.push_back(i);
		}
		if (near.empty()) {
			trees.push_back(Tree::make(p));
		} else if (near.size() == 1) {
			trees[near[0]].insert(p);
		} else {
			const size_t target = near[0];
			for (size_t k = near.size(); k-- > 1;) { // near[1..].iter().rev()
				Tree o = std::move(trees[near[k]]);
				trees.erase(trees.begin() + near[k]);
				for (const P2 q : o.points) trees[target].insert(q);
			}
			trees[target].insert(p);
		}
	}
	void filter_trees(float md) { // :469-486
		for (size_t i = trees.size(); i-- > 0;) {
			P2 center;
			float area;
			centroid(trees[i].points, &center, &area);
			if (area < (md * md) / 4.0f) {
				trees.erase(trees.begin() + i);
				continue;
			}
			for (size_t j = 0; j < i; ++j)
				if (trees[j].contains(center, 0.1f)) {
					trees.erase(trees.begin() + i);
					break;
				}
		}
	}
	// :488-524; D2: `prev` in ascending key order, D3: new ids from *next_id
	std::map<uint32_t, P2> tree_positions(const std::map<uint32_t, P2> &prev, float md, uint32_t *next_id) const {
		std::map<uint32_t, P2> res;
		std::vector<P2> centroids;
		for (const Tree &t : trees) {
			P2 c;
			float area;
			centroid(t.points, &c, &area);
			centroids.push_back(c);
		}
		for (const auto &kv : prev) {
			bool found = false;
			size_t nearest = 0;
			float nearest_dist = md * 2.0f;
			for (size_t i = 0; i < centroids.size(); ++i) {
				const float dx = kv.second.x - centroids[i].x, dy = kv.second.y - centroids[i].y;
				const float d = std::sqrt(dx * dx + dy * dy);
				if (d < nearest_dist) {
					nearest = i;
					nearest_dist = d;
					found = true;
				}
			}
			if (found) {
				res[kv.first] = centroids[nearest]; // swap_remove
				centroids[nearest] = centroids.back();
				centroids.pop_back();
			} else {
				res[kv.first] = kv.second;
			}
		}
		for (const P2 c : centroids) {
			while (*next_id == 0u || res.count(*next_id)) ++*next_id; // (never loops: ids only grow)
			res[(*next_id)++] = c;
		}
		return res;
	}
};

// ---- D4: Voronoi cells clipped to the box ------------------------------------------------------------------------
struct D2 {
	double x, y;
};
// keep the part of the convex polygon with a * x + b * y <= c (Sutherland-Hodgman against one line, f64)
void clip_half_plane(std::vector<D2> &poly, double a, double b, double c) {
	const size_t n = poly.size();
	if (n == 0) return;
	bool any_out = false;
	for (const D2 &p : poly)
		if (a * p.x + b * p.y > c) {
			any_out = true;
			break;
		}
	if (!any_out) return;
	std::vector<D2> out;
	out.reserve(n + 2);
	for (size_t i = 0; i < n; ++i) {
		const D2 p = poly[i], q = poly[(i + 1) % n];
		const double sp = a * p.x + b * p.y - c, sq = a * q.x + b * q.y - c;
		if (sp <= 0.0) out.push_back(p);
		if ((sp <= 0.0) != (sq <= 0.0)) {
			const double t = sp / (sp - sq);
			out.push_back({p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)});
		}
	}
	poly.swap(out);
}
std::vector<std::vector<P2>> voronoi_cells(P2 bmin, P2 bmax, const std::vector<P2> &sites) {
	const double x0 = bmin.x, y0 = bmin.y, x1 = bmax.x, y1 = bmax.y, w = x1 - x0, h = y1 - y0;
	std::vector<D2> all;
	for (const P2 s : sites) all.push_back({(double)s.x, (double)s.y});
	const size_t real = all.size();
	all.push_back({x0 - w, y0 + h / 2.0}); // the helper sites that close the outer cells
	all.push_back({x1 + w, y0 + h / 2.0});
	all.push_back({x0 + w / 2.0, y0 - h});
	all.push_back({x0 + w / 2.0, y1 + h});
	std::vector<std::vector<P2>> cells(real);
	for (size_t i = 0; i < real; ++i) {
		std::vector<D2> poly = {{x0, y0}, {x1, y0}, {x1, y1}, {x0, y1}};
		const D2 s = all[i];
		for (size_t j = 0; j < all.size() && !poly.empty(); ++j) {
			if (j == i) continue;
			const double a = all[j].x - s.x, b = all[j].y - s.y;
			if (a == 0.0 && b == 0.0) continue; // a coincident site has no bisector
			const double c = ((all[j].x * all[j].x + all[j].y * all[j].y) - (s.x * s.x + s.y * s.y)) / 2.0;
			clip_half_plane(poly, a, b, c);
		}
		for (const D2 &p : poly) cells[i].push_back({(float)p.x, (float)p.y}); // Centroid::from_xy
	}
	return cells;
}

struct SplitMix64 {
	uint64_t s;
	uint64_t next() {
		uint64_t z = (s += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
};

} // namespace

extern "C" {

int32_t treee_segment_slices(const float *xyz, uint64_t n, const float bbox_min_xz[2], const float bbox_max_xz[2], float max_distance, uint64_t seed,
                             uint32_t *segment_out, uint32_t *n_segments_out) {
	if ((n && (!xyz || !segment_out)) || !bbox_min_xz || !bbox_max_xz) return set_err(TREEE_ERR_INVALID, "treee_segment_slices: null argument");
	if (!(max_distance > 0.0f)) return set_err(TREEE_ERR_INVALID, "max_distance must be positive (the reference's slider: 0.1 ..= 2.0)");
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large");
	const P2 bmin = {bbox_min_xz[0], bbox_min_xz[1]}, bmax = {bbox_max_xz[0], bbox_max_xz[1]};
	// loading.rs:97-101 + segmenting.rs:121-133: 1 m layers of height, keyed floor(y), points in arrival order
	std::map<int64_t, std::vector<uint32_t>> slices;
	for (uint64_t i = 0; i < n; ++i) {
		const float f = std::floor(xyz[3 * i + 1]);
		const int64_t key = !(f == f) ? 0 : (f >= 9.2233720368547758e18f ? INT64_MAX : (f <= -9.2233720368547758e18f ? INT64_MIN : (int64_t)f)); // `as isize`
		slices[key].push_back((uint32_t)i);
	}
	// The reference gives every slice to a rayon worker (:164-234): TreeSet::new needs nothing from the other slices, the
	// centroids travel down a chain of channels (:135-146, :173-179), and the cells and the assignment of a slice need only
	// its own centroid map.  Same three stages here -- the outer two over a thread pool, the chain sequential -- and the
	// same result whatever the number of threads.
	uint32_t next_id = 1;
	if (!slices.empty()) {
		const int64_t lo = slices.begin()->first, hi = slices.rbegin()->first;
		const size_t layers = (size_t)(hi - lo + 1); // a layer without points still hands the centroids on (:141)
		struct Layer {
			std::vector<uint32_t> order;
			TreeSet set;
			std::map<uint32_t, P2> centroids;
		};
		std::vector<Layer> layer(layers); // layer[0] = top slice
		auto parallel_for = [&](const std::function<void(size_t)> &fn) {
			unsigned workers = treee::host_threads();
			if (workers == 0) workers = 1;
			if (workers > layers) workers = (unsigned)layers;
			std::atomic<size_t> next{0};
			std::vector<std::thread> pool;
			for (unsigned w = 0; w + 1 < workers; ++w)
				pool.emplace_back([&] {
					for (size_t i; (i = next.fetch_add(1)) < layers;) fn(i);
				});
			for (size_t i; (i = next.fetch_add(1)) < layers;) fn(i);
			for (auto &t : pool) t.join();
		};
		parallel_for([&](size_t li) { // shuffle (D1) + TreeSet::new (:423-435)
			const int64_t key = hi - (int64_t)li;
			auto it = slices.find(key);
			if (it != slices.end()) layer[li].order = it->second;
			std::vector<uint32_t> &order = layer[li].order;
			SplitMix64 rng{seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(key - lo + 1))};
			for (size_t i = order.size(); i > 1; --i) std::swap(order[i - 1], order[(size_t)(rng.next() % i)]);
			for (const uint32_t p : order) layer[li].set.add_point({xyz[3ull * p + 0], xyz[3ull * p + 2]}, max_distance);
			layer[li].set.filter_trees(max_distance);
		});
		std::map<uint32_t, P2> centroids; // handed from a slice to the one below it
		for (size_t li = 0; li < layers; ++li) {
			centroids = layer[li].set.tree_positions(centroids, max_distance, &next_id);
			layer[li].centroids = centroids;
		}
		parallel_for([&](size_t li) { // cells (D4) + assignment (:181-203)
			std::vector<uint32_t> ids;
			std::vector<P2> sites;
			for (const auto &kv : layer[li].centroids) { // D2
				ids.push_back(kv.first);
				sites.push_back(kv.second);
			}
			if (layer[li].order.empty()) return; // (no point to assign: the cells would not be looked at)
			const std::vector<std::vector<P2>> cells = voronoi_cells(bmin, bmax, sites);
			std::deque<std::pair<Tree, uint32_t>> trees;
			for (size_t c = 0; c < cells.size(); ++c) trees.emplace_back(Tree::from_points(cells[c]), ids[c]);
			for (const uint32_t p : layer[li].order) { // first cell that contains the point (0.1 slack), moved to the front
				const P2 q = {xyz[3ull * p + 0], xyz[3ull * p + 2]};
				size_t at = trees.size();
				for (size_t c = 0; c < trees.size(); ++c)
					if (trees[c].first.contains(q, 0.1f)) {
						at = c;
						break;
					}
				if (at == trees.size()) {
					segment_out[p] = 0u; // DELETED_INDEX
					continue;
				}
				std::pair<Tree, uint32_t> elem = std::move(trees[at]);
				trees.erase(trees.begin() + at);
				segment_out[p] = elem.second;
				trees.push_front(std::move(elem));
			}
		});
	}
	if (n_segments_out) *n_segments_out = next_id;
	return TREEE_OK;
}

} // extern "C"
