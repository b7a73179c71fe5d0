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
#include <chrono>
#include <cstdlib>
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

// The outward unit normal of an edge depends on the edge only: it is computed when the hull changes and kept beside the
// vertices (edge_dist's sqrt and two divisions once per edge and change instead of once per edge and tested point).  The
// value is the one edge_dist computes -- same operations, same order -- so every distance keeps its bits.
inline P2 edge_normal(P2 a, P2 b) {
	const float dx = b.x - a.x, dy = b.y - a.y;
	const float ox = dy, oy = -dx;
	const float norm = std::sqrt(ox * ox + oy * oy);
	return {ox / norm, oy / norm};
}
inline float edge_dist_n(P2 a, P2 nrm, P2 point) {
	const float fx = point.x - a.x, fy = point.y - a.y;
	return nrm.x * fx + nrm.y * fy;
}

struct Tree { // segmenting.rs:242-247
	std::vector<P2> points;
	std::vector<P2> normals; // normals[i]: edge points[i] -> points[i + 1]
	P2 min, max;
	// a disc that lies inside the area with room to spare: a point in it is inside every edge whatever the rounding of the
	// edge tests, so `distance` would answer <= 0 -- known without walking the hull (most points of a slice fall into an
	// area that already covers them; the exact tests decide everything outside the disc)
	P2 disc_c = {0.0f, 0.0f};
	float disc_r2 = 0.0f;

	void update_normals() {
		const size_t len = points.size();
		normals.resize(len);
		float sx = 0.0f, sy = 0.0f, big = 0.0f;
		for (size_t i = 0; i < len; ++i) {
			normals[i] = edge_normal(points[i], points[i + 1 < len ? i + 1 : 0]);
			sx += points[i].x;
			sy += points[i].y;
			big = std::fmax(big, std::fmax(std::fabs(points[i].x), std::fabs(points[i].y)));
		}
		disc_c = {sx / (float)len, sy / (float)len};
		float r = 3.402823466e38f;
		for (size_t i = 0; i < len; ++i) {
			const float d = -edge_dist_n(points[i], normals[i], disc_c); // > 0 when the centre is inside edge i
			r = d < r ? d : (d == d ? r : -1.0f);                         // (a NaN normal -- a zero-length edge -- switches the disc off)
		}
		r -= 1e-3f + big * 4e-6f; // the edge tests' own rounding is below 8 ulp of the coordinates
		disc_r2 = r > 0.0f && r < 3.0e38f ? r * r : 0.0f;
	}
	bool surely_inside(P2 p) const {
		const float dx = p.x - disc_c.x, dy = p.y - disc_c.y;
		return dx * dx + dy * dy < disc_r2;
	}
	static Tree make(P2 p) { // Tree::new :258-269
		Tree t;
		t.points = {p, {p.x + 0.1f, p.y}, {p.x, p.y + 0.1f}};
		t.min = p;
		t.max = {p.x + 0.1f, p.y + 0.1f};
		t.update_normals();
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
		t.update_normals();
		return t;
	}
	bool outside_bounds(P2 p, float md) const { // :318-323
		return p.x + md < min.x || max.x + md <= p.x || p.y + md < min.y || max.y + md <= p.y;
	}
	float distance(P2 p, float md) const { // :298-316
		if (outside_bounds(p, md)) return 3.402823466e38f;
		float best = -3.402823466e38f;
		for (size_t i = 0; i < points.size(); ++i) {
			const float d = edge_dist_n(points[i], normals[i], p);
			if (d > md) return 3.402823466e38f;
			best = std::fmax(best, d);
		}
		return best;
	}
	bool contains(P2 p, float md) const { // :330-345
		if (outside_bounds(p, md)) return false;
		for (size_t i = 0; i < points.size(); ++i)
			if (edge_dist_n(points[i], normals[i], p) > md) return false;
		return true;
	}
	bool insert(P2 point) { // :346-396; true when the hull changed
		const size_t len = points.size();
		bool last = edge_dist_n(points[len - 1], normals[len - 1], point) > 0.0f;
		bool has_s = false, has_e = false;
		size_t start = 0, end = 0;
		for (size_t i = 0; i < len; ++i) {
			const bool out = edge_dist_n(points[i], normals[i], point) > 0.0f;
			if (!last && out) {
				start = i;
				has_s = true;
			} else if (last && !out) {
				end = i;
				has_e = true;
			}
			last = out;
		}
		if (!has_s || !has_e) return false;
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
		update_normals();
		return true;
	}
};

// A uniform grid over boxes in the plane: which boxes (dilated by `reach`) cover a position?  Only ever a SUPERSET filter in
// front of the reference's own exact tests (outside_bounds, distance, contains), so it changes no result: add_point's
// "which areas are within max_distance" (segmenting.rs:437-467) and the assignment's "which cell contains the point"
// (:187-203) visit a handful of candidates instead of every area of the slice.
struct BoxGrid {
	float x0 = 0, y0 = 0, cell = 1, reach = 0;
	int nx = 1, ny = 1;
	std::vector<std::vector<uint32_t>> cells;
	struct Rect {
		int cx0, cy0, cx1, cy1; // registered cells, inclusive; cx1 < cx0: none yet
	};
	std::vector<Rect> rects; // per box id
	void init(P2 lo, P2 hi, float cell_size, float reach_) {
		cell = cell_size;
		reach = reach_ * 1.01f + 1e-3f; // (the exact tests add in f32: a little slack keeps the filter a superset)
		x0 = lo.x - reach;
		y0 = lo.y - reach;
		const float w = (hi.x - lo.x) + 2.0f * reach, h = (hi.y - lo.y) + 2.0f * reach;
		nx = (int)std::fmin(std::fmax(std::ceil(w / cell), 1.0f), 1024.0f);
		ny = (int)std::fmin(std::fmax(std::ceil(h / cell), 1.0f), 1024.0f);
		cells.assign((size_t)nx * ny, {});
		rects.clear();
	}
	int cx(float x) const {
		const float f = std::floor((x - x0) / cell);
		return !(f >= 0.0f) ? 0 : (f >= (float)nx ? nx - 1 : (int)f);
	}
	int cy(float y) const {
		const float f = std::floor((y - y0) / cell);
		return !(f >= 0.0f) ? 0 : (f >= (float)ny ? ny - 1 : (int)f);
	}
	std::vector<uint32_t> &at(P2 p) { return cells[(size_t)cy(p.y) * nx + cx(p.x)]; }
	// box `id` now spans [mn, mx]: register it in the cells its dilated box covers and was not in yet (boxes only grow)
	void cover(uint32_t id, P2 mn, P2 mx) {
		if (id >= rects.size()) rects.resize((size_t)id + 1, Rect{0, 0, -1, -1});
		Rect &r = rects[id];
		const Rect want = {cx(mn.x - reach), cy(mn.y - reach), cx(mx.x + reach), cy(mx.y + reach)};
		if (r.cx1 >= r.cx0 && want.cx0 >= r.cx0 && want.cx1 <= r.cx1 && want.cy0 >= r.cy0 && want.cy1 <= r.cy1) return;
		const Rect grown = r.cx1 < r.cx0 ? want : Rect{std::min(r.cx0, want.cx0), std::min(r.cy0, want.cy0), std::max(r.cx1, want.cx1), std::max(r.cy1, want.cy1)};
		for (int y = grown.cy0; y <= grown.cy1; ++y)
			for (int x = grown.cx0; x <= grown.cx1; ++x)
				if (r.cx1 < r.cx0 || x < r.cx0 || x > r.cx1 || y < r.cy0 || y > r.cy1) cells[(size_t)y * nx + x].push_back(id);
		r = grown;
	}
};

struct TreeSet { // segmenting.rs:414-524
	// `trees` in creation order; Vec::remove becomes a tombstone (alive = 0): removal keeps the relative order of the others,
	// which is all the reference's index order is used for (near[0] = the first near area; reverse removal; j < i)
	std::vector<Tree> trees;
	std::vector<char> alive;
	BoxGrid grid;
	std::vector<uint32_t> near; // scratch

	void add_point(P2 p, float md) { // :437-467
		near.clear();
		std::vector<uint32_t> &cand = grid.at(p); // every area NOT listed here answers f32::MAX (outside_bounds)
		for (size_t k = 0; k < cand.size();) {
			const uint32_t i = cand[k];
			if (!alive[i]) { // merged away or never to come back: drop it from this cell's list
				cand[k] = cand.back();
				cand.pop_back();
				continue;
			}
			// inside an area: nothing happens, whichever areas were seen before -- so the order of the scan is free, and the
			// area that swallowed the last point of this cell is tried first
			const float dist = trees[i].surely_inside(p) ? 0.0f : trees[i].distance(p, md);
			if (dist <= 0.0f) {
				std::swap(cand[0], cand[k]);
				return;
			}
			if (dist <= md) near.push_back(i);
			++k;
		}
		if (near.empty()) {
			trees.push_back(Tree::make(p));
			alive.push_back(1);
			grid.cover((uint32_t)trees.size() - 1u, trees.back().min, trees.back().max);
			return;
		}
		std::sort(near.begin(), near.end()); // index order, as the reference's scan over `trees` finds them
		const uint32_t target = near[0];
		for (size_t k = near.size(); k-- > 1;) { // near[1..].iter().rev()
			alive[near[k]] = 0;
			const std::vector<P2> pts = std::move(trees[near[k]].points);
			for (const P2 q : pts) trees[target].insert(q);
		}
		if (trees[target].insert(p) || near.size() > 1) grid.cover(target, trees[target].min, trees[target].max);
	}
	void filter_trees(float md) { // :469-486
		for (size_t i = trees.size(); i-- > 0;) {
			if (!alive[i]) continue;
			P2 center;
			float area;
			centroid(trees[i].points, &center, &area);
			if (area < (md * md) / 4.0f) {
				alive[i] = 0;
				continue;
			}
			for (size_t j = 0; j < i; ++j)
				if (alive[j] && trees[j].contains(center, 0.1f)) {
					alive[i] = 0;
					break;
				}
		}
	}
	// :488-524; D2: `prev` in ascending key order, D3: new ids from *next_id
	std::map<uint32_t, P2> tree_positions(const std::map<uint32_t, P2> &prev, float md, uint32_t *next_id) const {
		std::map<uint32_t, P2> res;
		std::vector<P2> centroids;
		for (size_t i = 0; i < trees.size(); ++i) {
			if (!alive[i]) continue;
			P2 c;
			float area;
			centroid(trees[i].points, &c, &area);
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

namespace {

constexpr size_t MAX_LAYERS = size_t(1) << 20; // 1 m layers: a cloud taller than 1 000 km is a unit error, not a forest

int32_t segment_slices(const float *xyz, uint64_t n, const float *bbox_min_xz, const float *bbox_max_xz, float max_distance, uint64_t seed, uint32_t *segment_out,
                       uint32_t *n_segments_out) {
	// loading.rs:97-101 + segmenting.rs:121-133: 1 m layers of height, keyed floor(y), points in arrival order (a stable
	// counting sort by key: the reference's HashMap<isize, Vec<Point>> densified top-down).  One pass over the cloud, in
	// pieces over the host's threads: every point's key, the key range, and the x / z box when the caller did not give it.
	auto key_of = [&](uint64_t i) -> int64_t {
		const float f = std::floor(xyz[3 * i + 1]);
		return !(f == f) ? 0 : (f >= 9.2233720368547758e18f ? INT64_MAX : (f <= -9.2233720368547758e18f ? INT64_MIN : (int64_t)f)); // `as isize`
	};
	const bool own_box = !(bbox_min_xz && bbox_max_xz);
	P2 bmin = {0.0f, 0.0f}, bmax = {0.0f, 0.0f};
	if (!own_box) {
		bmin = {bbox_min_xz[0], bbox_min_xz[1]};
		bmax = {bbox_max_xz[0], bbox_max_xz[1]};
	}
	uint32_t next_id = 1;
	const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	if (n) {
		std::vector<int64_t> key(n);
		int64_t lo = INT64_MAX, hi = INT64_MIN;
		{
			const unsigned parts = std::max(1u, std::min(treee::host_threads(), (unsigned)(n / 65536 + 1)));
			struct Part {
				int64_t lo = INT64_MAX, hi = INT64_MIN;
				P2 mn = {INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY};
			};
			std::vector<Part> part(parts);
			auto scan = [&](unsigned t) {
				Part &pt = part[t];
				for (uint64_t i = n * t / parts, e = n * (t + 1) / parts; i < e; ++i) {
					const int64_t k = key[i] = key_of(i);
					pt.lo = std::min(pt.lo, k);
					pt.hi = std::max(pt.hi, k);
					if (own_box) { // the box `Loading` keeps (loading.rs:16-17), taken from the points themselves
						pt.mn = {std::fmin(pt.mn.x, xyz[3 * i]), std::fmin(pt.mn.y, xyz[3 * i + 2])};
						pt.mx = {std::fmax(pt.mx.x, xyz[3 * i]), std::fmax(pt.mx.y, xyz[3 * i + 2])};
					}
				}
			};
			std::vector<std::thread> pool;
			for (unsigned t = 1; t < parts; ++t) pool.emplace_back(scan, t);
			scan(0);
			for (auto &t : pool) t.join();
			if (own_box) bmin = {INFINITY, INFINITY}, bmax = {-INFINITY, -INFINITY};
			for (const Part &pt : part) {
				lo = std::min(lo, pt.lo);
				hi = std::max(hi, pt.hi);
				if (own_box) {
					bmin = {std::fmin(bmin.x, pt.mn.x), std::fmin(bmin.y, pt.mn.y)};
					bmax = {std::fmax(bmax.x, pt.mx.x), std::fmax(bmax.y, pt.mx.y)};
				}
			}
		}
		if ((uint64_t)hi - (uint64_t)lo >= MAX_LAYERS)
			return set_err(TREEE_ERR_UNSUPPORTED, "treee_segment_slices: the cloud spans %llu one-metre layers (y from %lld to %lld); more than %zu is refused",
			               (unsigned long long)((uint64_t)hi - (uint64_t)lo + 1u), (long long)lo, (long long)hi, MAX_LAYERS);
		const size_t layers = (size_t)(hi - lo + 1); // a layer without points still hands the centroids on (:141)
		struct Layer {
			std::vector<uint32_t> order;
			TreeSet set;
			std::map<uint32_t, P2> centroids;
		};
		std::vector<Layer> layer(layers); // layer[0] = top slice
		{
			std::vector<uint32_t> count(layers, 0);
			for (uint64_t i = 0; i < n; ++i) ++count[(size_t)(hi - key[i])];
			for (size_t li = 0; li < layers; ++li) layer[li].order.reserve(count[li]);
			for (uint64_t i = 0; i < n; ++i) layer[(size_t)(hi - key[i])].order.push_back((uint32_t)i);
			std::vector<int64_t>().swap(key);
		}
		// The reference gives every slice to a rayon worker (:164-234): TreeSet::new needs nothing from the other slices, the
		// centroids travel down a chain of channels (:135-146, :173-179), and the cells and the assignment of a slice need only
		// its own centroid map.  Same three stages here -- the outer two over a thread pool (largest slices first: the ground
		// slices of a plot hold a fifth of its points each), the chain sequential -- and the same result whatever the number
		// of threads.  An exception in a worker (allocation) is carried to the caller: nothing unwinds through a thread.
		const double t_layers = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
		std::vector<size_t> by_size(layers);
		for (size_t li = 0; li < layers; ++li) by_size[li] = li;
		std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return layer[a].order.size() > layer[b].order.size(); });
		std::atomic<bool> failed{false};
		auto parallel_for = [&](const std::function<void(size_t)> &fn) {
			unsigned workers = treee::host_threads();
			if (workers == 0) workers = 1;
			if (workers > layers) workers = (unsigned)layers;
			std::atomic<size_t> next{0};
			auto run = [&] {
				try {
					for (size_t i; (i = next.fetch_add(1)) < layers;) fn(by_size[i]);
				} catch (...) {
					failed = true;
				}
			};
			std::vector<std::thread> pool;
			try {
				for (unsigned w = 0; w + 1 < workers; ++w) pool.emplace_back(run);
			} catch (...) { // (thread creation refused: the threads that exist and this one do the work)
			}
			run();
			for (auto &t : pool) t.join();
		};
		parallel_for([&](size_t li) { // shuffle (D1) + TreeSet::new (:423-435)
			std::vector<uint32_t> &order = layer[li].order;
			if (order.empty()) return;
			const int64_t key = hi - (int64_t)li;
			SplitMix64 rng{seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(key - lo + 1))};
			for (size_t i = order.size(); i > 1; --i) std::swap(order[i - 1], order[(size_t)(rng.next() % i)]);
			P2 mn = {INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY};
			for (const uint32_t p : order) {
				const float x = xyz[3ull * p + 0], z = xyz[3ull * p + 2];
				if (x == x) mn.x = std::fmin(mn.x, x), mx.x = std::fmax(mx.x, x);
				if (z == z) mn.y = std::fmin(mn.y, z), mx.y = std::fmax(mx.y, z);
			}
			if (!(mn.x <= mx.x)) mn.x = mx.x = 0.0f;
			if (!(mn.y <= mx.y)) mn.y = mx.y = 0.0f;
			layer[li].set.grid.init(mn, mx, std::fmax(2.0f * max_distance, 1.0f), max_distance);
			for (const uint32_t p : order) layer[li].set.add_point({xyz[3ull * p + 0], xyz[3ull * p + 2]}, max_distance);
			layer[li].set.filter_trees(max_distance);
		});
		if (failed) return set_err(TREEE_ERR_NOMEM, "treee_segment_slices: a worker ran out of memory");
		const bool times = getenv("TREEE_SEG_TIMES") != nullptr;
		auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		double t1 = now();
		if (times) fprintf(stderr, "keys+layers %.3f s, stage 1 (hulls) %.3f s\n", t_layers - t_begin, t1 - t_layers);
		std::map<uint32_t, P2> centroids; // handed from a slice to the one below it
		for (size_t li = 0; li < layers; ++li) {
			centroids = layer[li].set.tree_positions(centroids, max_distance, &next_id);
			layer[li].centroids = centroids;
			layer[li].set = TreeSet(); // (its hulls are not needed again)
		}
		const double t2 = now();
		std::atomic<long long> vor_us{0};
		struct Assign { // a slice's cells, ready for the assignment
			std::vector<uint32_t> ids;
			std::vector<Tree> trees;
			BoxGrid grid;
			std::vector<uint32_t> first; // per point of the slice: a cell that contains it (NONE32: none)
		};
		std::vector<Assign> assign(layers);
		auto parallel_items = [&](size_t count, const std::function<void(size_t)> &fn) {
			unsigned workers = treee::host_threads();
			if (workers == 0) workers = 1;
			if (workers > count) workers = (unsigned)(count ? count : 1);
			std::atomic<size_t> next{0};
			auto run = [&] {
				try {
					for (size_t i; (i = next.fetch_add(1)) < count;) fn(i);
				} catch (...) {
					failed = true;
				}
			};
			std::vector<std::thread> pool;
			try {
				for (unsigned w = 0; w + 1 < workers; ++w) pool.emplace_back(run);
			} catch (...) {
			}
			run();
			for (auto &t : pool) t.join();
		};
		parallel_for([&](size_t li) { // cells (D4) + assignment (:181-203)
			if (layer[li].order.empty()) return; // (no point to assign: the cells would not be looked at)
			std::vector<uint32_t> ids;
			std::vector<P2> sites;
			for (const auto &kv : layer[li].centroids) { // D2
				ids.push_back(kv.first);
				sites.push_back(kv.second);
			}
			const double tv = now();
			const std::vector<std::vector<P2>> cells = voronoi_cells(bmin, bmax, sites);
			vor_us += (long long)((now() - tv) * 1e6);
			// The reference keeps the cells in a VecDeque, takes the FIRST one that contains the point (0.1 slack: cells overlap
			// along their edges) and moves it to the front (:187-203).  The deque's order is therefore: cells by their last
			// hit, most recent first, then the never-hit ones in their original order -- so "first containing cell" = of the
			// containing cells the most recently hit one, else the one with the smallest original position.  A grid finds the
			// few cells whose box covers the point; a time stamp per cell replaces the deque.
			std::vector<Tree> trees;
			trees.reserve(cells.size());
			BoxGrid grid;
			grid.init(bmin, bmax, std::fmax(std::fmax(bmax.x - bmin.x, bmax.y - bmin.y) / 96.0f, 0.5f), 0.1f);
			for (size_t c = 0; c < cells.size(); ++c) {
				trees.push_back(Tree::from_points(cells[c]));
				if (!trees.back().points.empty()) grid.cover((uint32_t)c, trees.back().min, trees.back().max);
			}
			// which cells contain each point does not depend on the deque's state: found for all points first (in chunks, over
			// the pool), then one cheap pass in slice order settles the points that lie in more than one cell
			Assign &as = assign[li];
			as.ids = std::move(ids);
			as.trees = std::move(trees);
			as.grid = std::move(grid);
			as.first.assign(layer[li].order.size(), NONE32);
		});
		if (failed) return set_err(TREEE_ERR_NOMEM, "treee_segment_slices: a worker ran out of memory");
		constexpr size_t CHUNK = 8192;
		std::vector<std::pair<uint32_t, uint32_t>> chunks; // (layer, first point)
		std::vector<size_t> chunk_begin(layers + 1, 0);
		for (size_t li = 0; li < layers; ++li) {
			chunk_begin[li] = chunks.size();
			if (assign[li].first.empty()) continue; // (a slice without points, or one whose cells were never built)
			for (size_t b = 0; b < layer[li].order.size(); b += CHUNK) chunks.push_back({(uint32_t)li, (uint32_t)b});
		}
		chunk_begin[layers] = chunks.size();
		std::vector<std::vector<std::pair<uint32_t, uint32_t>>> extra(chunks.size()); // (point position in the slice, another containing cell)
		parallel_items(chunks.size(), [&](size_t ci) {
			const size_t li = chunks[ci].first, b = chunks[ci].second;
			Assign &as = assign[li];
			const std::vector<uint32_t> &order = layer[li].order;
			const size_t e = std::min(order.size(), b + CHUNK);
			for (size_t k = b; k < e; ++k) {
				const uint32_t p = order[k];
				const P2 q = {xyz[3ull * p + 0], xyz[3ull * p + 2]};
				if (q.x == q.x && q.y == q.y) {
					for (const uint32_t c : as.grid.at(q)) {
						if (!as.trees[c].contains(q, 0.1f)) continue;
						if (as.first[k] == NONE32)
							as.first[k] = c;
						else
							extra[ci].push_back({(uint32_t)k, c});
					}
				} else { // NaN compares false everywhere: `contains` holds for EVERY cell -- a cell clipped to nothing included, as
					// in the reference (from_points of no vertices, segmenting.rs:272-278: no bound and no edge can fail, and
					// position() at :195-197 takes the first of the move-to-front order); tests/test_segmenting.py pins it
					for (size_t c = 0; c < as.trees.size(); ++c) {
						if (!as.trees[c].contains(q, 0.1f)) continue;
						if (as.first[k] == NONE32)
							as.first[k] = (uint32_t)c;
						else
							extra[ci].push_back({(uint32_t)k, (uint32_t)c});
					}
				}
			}
		});
		if (failed) return set_err(TREEE_ERR_NOMEM, "treee_segment_slices: a worker ran out of memory");
		parallel_for([&](size_t li) {
			Assign &as = assign[li];
			const std::vector<uint32_t> &order = layer[li].order;
			std::vector<uint64_t> last_hit(as.trees.size(), 0);
			uint64_t clock = 0;
			for (size_t ci = chunk_begin[li]; ci < chunk_begin[li + 1]; ++ci) {
				const size_t b = chunks[ci].second, e = std::min(order.size(), b + CHUNK);
				size_t xi = 0; // (the other cells that contain a point follow each other in its chunk's list, points ascending)
				for (size_t k = b; k < e; ++k) {
				uint32_t at = as.first[k];
				while (xi < extra[ci].size() && extra[ci][xi].first == (uint32_t)k) {
					const uint32_t c = extra[ci][xi++].second;
					if (last_hit[c] > last_hit[at] || (last_hit[c] == last_hit[at] && c < at)) at = c;
				}
				if (at == NONE32) {
					segment_out[order[k]] = 0u; // DELETED_INDEX
					continue;
				}
				segment_out[order[k]] = as.ids[at];
				last_hit[at] = ++clock;
				}
			}
		});
		if (failed) return set_err(TREEE_ERR_NOMEM, "treee_segment_slices: a worker ran out of memory");
		if (times) fprintf(stderr, "chain %.3f s, stage 3 (cells + assignment) %.3f s of which voronoi (summed over threads) %.3f s\n", t2 - t1, now() - t2, vor_us / 1e6);
	}
	if (n_segments_out) *n_segments_out = next_id;
	return TREEE_OK;
}

} // namespace

extern "C" {

int32_t treee_segment_slices(const float *xyz, uint64_t n, const float bbox_min_xz[2], const float bbox_max_xz[2], float max_distance, uint64_t seed,
                             uint32_t *segment_out, uint32_t *n_segments_out) {
	if ((n && (!xyz || !segment_out)) || (!bbox_min_xz != !bbox_max_xz)) return set_err(TREEE_ERR_INVALID, "treee_segment_slices: null argument");
	if (!(max_distance > 0.0f)) return set_err(TREEE_ERR_INVALID, "max_distance must be positive (the reference's slider: 0.1 ..= 2.0)");
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large");
	try { // nothing unwinds across the C ABI
		return segment_slices(xyz, n, bbox_min_xz, bbox_max_xz, max_distance, seed, segment_out, n_segments_out);
	} catch (const std::exception &e) {
		return set_err(TREEE_ERR_NOMEM, "treee_segment_slices: %s", e.what());
	}
}

} // extern "C"
