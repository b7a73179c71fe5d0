// Parity of the C++ host-side mirror (include/treee_hip.hpp -> C ABI -> HIP kernels) against the CPU oracle
// (oracle/treee_oracle.h), written the way the reference's call sites use the interfaces:
//   k-nearest/readme.md usage + kd_tree.rs:82-102, treee/src/calculations.rs:328-373,392-419,
//   render/src/point_cloud.rs:148-173, treee/src/laz.rs:193-213, loading.rs:99.
// Test infrastructure: links the oracle only as the checker.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "treee_hip.hpp"
#include "treee_oracle.h"

using treee::Entry;
using treee::Point3;

static int failures = 0;
#define CHECK(cond)                                                          \
	do {                                                                     \
		if (!(cond)) {                                                       \
			std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
			++failures;                                                      \
		}                                                                    \
	} while (0)

static uint64_t splitmix(uint64_t &s) {
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static float u01(uint64_t &s) { return (float)(splitmix(s) >> 40) * (1.0f / 16777216.0f); }
static uint32_t bits(float f) {
	uint32_t u;
	std::memcpy(&u, &f, 4);
	return u;
}
static float dist_sq(const Point3 &p, const Point3 &q) { // metric.rs:21-26
	float d = 0.0f;
	for (int a = 0; a < 3; ++a) {
		const float t = p[a] - q[a];
		d = d + t * t;
	}
	return d;
}

int main() {
	uint64_t seed = 0x7EEE0001ull;
	const size_t n = 20000;
	std::vector<Point3> points(n);
	for (auto &p : points) { // 1 mm grid: distance ties on purpose
		p = {std::round(u01(seed) * 30.0f * 1000.0f) / 1000.0f, std::round(u01(seed) * 10.0f * 1000.0f) / 1000.0f,
		     std::round(u01(seed) * 30.0f * 1000.0f) / 1000.0f};
	}
	orc_kdtree *ref = orc_kdtree_new(points[0].data(), n);

	// ---- KDTree::new / k_nearest / nearest / empty ------------------------------------------------------
	treee::KDTree tree(points);
	CHECK(tree.len() == n);
	for (float max_distance : {treee::NEIGHBORS_MAX_DISTANCE, 0.25f}) {
		for (size_t qi = 0; qi < 600; ++qi) {
			const Point3 q = qi % 3 ? points[(qi * 31) % n] : Point3{u01(seed) * 32.0f - 1.0f, u01(seed) * 12.0f - 1.0f, u01(seed) * 32.0f - 1.0f};
			std::vector<Entry> location(31);
			const size_t l = tree.k_nearest(q, location, max_distance);
			std::vector<orc_entry> want(31);
			const uint64_t wl = orc_kdtree_k_nearest(ref, q.data(), want.data(), 31, max_distance);
			CHECK(l == wl);
			for (size_t i = 0; i < l && i < wl; ++i) {
				CHECK(bits(location[i].distance) == bits(want[i].distance)); // bit-identical distances
				if (location[i].index != want[i].index)                       // only inside a distance tie group
					CHECK(bits(dist_sq(points[location[i].index], q)) == bits(want[i].distance));
				CHECK(location[i].distance < max_distance);
				if (i) CHECK(location[i - 1].distance <= location[i].distance);
			}
		}
	}
	{
		const Point3 q = points[77];
		const auto got = tree.nearest(q, 0.5f);
		std::vector<orc_entry> want(got.size() + 8);
		const uint64_t found = orc_kdtree_nearest(ref, q.data(), 0.5f, want.data(), want.size());
		CHECK(found == got.size());
		for (size_t i = 0; i < got.size() && i < found; ++i) CHECK(bits(got[i].distance) == bits(want[i].distance));
		CHECK(tree.empty(q, 1e-12f) == (orc_kdtree_empty(ref, q.data(), 1e-12f) != 0));
		CHECK(tree.empty(Point3{500.0f, 0.0f, 0.0f}, 1.0f));
		CHECK(!tree.empty(q, 0.01f));
	}
	// the reference panics: KDTree::new(&[]) (kd_tree.rs:53,60) and k == 0 (best_set.rs:74)
	{
		bool threw = false;
		try {
			treee::KDTree bad(std::vector<Point3>{});
		} catch (const treee::Error &e) {
			threw = e.code == TREEE_ERR_EMPTY;
		}
		CHECK(threw);
		threw = false;
		try {
			std::vector<Entry> none;
			tree.k_nearest(points[0], none, 1.0f);
		} catch (const treee::Error &e) {
			threw = e.code == TREEE_ERR_INVALID;
		}
		CHECK(threw);
	}
	// ---- NeighborsTree + the curve loop (calculations.rs:328-373) -----------------------------------------
	{
		treee::NeighborsTree nt(points);
		std::vector<Entry> location(treee::NEIGHBORS_COUNT);
		const auto nb = nt.get(5, points, location, treee::NEIGHBORS_MAX_DISTANCE);
		CHECK(nb.second == 31 && nb.first[0].index == 5 && nb.first[0].distance == 0.0f);
		const std::vector<float> curve = treee::curvature(points);
		std::vector<float> want(n);
		orc_curvature(points[0].data(), n, 31, treee::NEIGHBORS_MAX_DISTANCE, want.data(), 8);
		size_t same = 0;
		for (size_t i = 0; i < n; ++i) {
			CHECK(std::isnan(curve[i]) == std::isnan(want[i]));
			if (!std::isnan(want[i])) {
				CHECK(std::fabs(curve[i] - want[i]) < 1e-4f); // rows with a distance tie may hold another member or sum in another order
				same += std::memcmp(&curve[i], &want[i], 4) == 0; // everywhere else: bit for bit (acos / cos: include/treee_f32_trig.h on both sides)
			}
		}
		CHECK(same > n * 0.995);
		// the tie flags say WHICH rows may differ: every unflagged row is the reference's value bit for bit
		const auto tied = treee::curvature_ties(points);
		size_t flagged = 0;
		for (size_t i = 0; i < n; ++i) {
			CHECK(std::memcmp(&tied.first[i], &curve[i], 4) == 0);
			flagged += tied.second[i] != 0;
			if (!tied.second[i]) CHECK(std::memcmp(&curve[i], &want[i], 4) == 0 || (std::isnan(curve[i]) && std::isnan(want[i])));
		}
		CHECK(flagged < n * 0.01);
		for (float v : {-1.0f, 0.0f, 1e-10f, 0.25f, 0.9999999f, 1.0f, 7.0f, NAN}) CHECK(treee::map_to_u32(v) == orc_map_to_u32(v));
		// every segment at once == segment by segment
		std::vector<uint32_t> segment(n);
		for (size_t i = 0; i < n; ++i) segment[i] = points[i][0] < 15.0f ? 0u : 1u;
		const std::vector<float> both = treee::curvature_segments(points, segment, 2);
		for (uint32_t sg = 0; sg < 2; ++sg) {
			std::vector<Point3> part;
			std::vector<size_t> where;
			for (size_t i = 0; i < n; ++i)
				if (segment[i] == sg) {
					part.push_back(points[i]);
					where.push_back(i);
				}
			std::vector<float> w(part.size());
			orc_curvature(part[0].data(), part.size(), 31, treee::NEIGHBORS_MAX_DISTANCE, w.data(), 8);
			for (size_t i = 0; i < part.size(); ++i)
				if (!std::isnan(w[i])) CHECK(std::fabs(both[where[i]] - w[i]) < 1e-4f);
		}
	}
	// ---- whole segments over a device list (calculations.rs:97-118): the single-device result, bit for bit -------------
	{
		std::vector<uint32_t> segment(n);
		for (size_t i = 0; i < n; ++i) segment[i] = (uint32_t)(points[i][0] / 4.0f) % 7u;
		const std::vector<float> one = treee::curvature_segments(points, segment, 7);
		for (const std::vector<int32_t> &devices : {std::vector<int32_t>{0}, std::vector<int32_t>{0, 0}}) {
			const std::vector<float> many = treee::curvature_segments(devices, points, segment, 7);
			CHECK(many.size() == one.size() && std::memcmp(many.data(), one.data(), one.size() * 4) == 0);
		}
	}
	// ---- the sharded octree's host arithmetic and the loopback communicator (treee_hip.h, "many GPUs") ------------------
	{
		const uint32_t L = 2, ncell = 64;
		std::vector<uint32_t> counts(2 * ncell, 0);
		for (uint32_t c = 0; c < ncell; ++c) {
			counts[c] = c < 40 ? 500 + c : 0;          // rank 0's input: the low cells
			counts[ncell + c] = c >= 24 ? 300 + c : 0; // rank 1's: the high ones, overlapping in 24 .. 39
		}
		treee::ShardPlan plan(counts, 2, L, 64, 16);
		const treee_shard_plan_rank r0 = plan.rank(0), r1 = plan.rank(1);
		uint64_t total = 0;
		for (uint32_t v : counts) total += v;
		CHECK(plan.info.n_total == total && r0.count + r1.count == total && r0.first == 0 && r1.first == r0.count);
		CHECK(r0.cell_begin == 0 && r0.cell_end == r1.cell_begin && r1.cell_end == ncell && r1.index_base > 0);
		const std::vector<treee_octree_node> top = plan.top_nodes();
		CHECK(!top.empty() && top[0].count == total && top[0].level == 0 && top[0].first_child == 1);
		size_t level_nodes = 0;
		for (const treee_octree_node &nd : top) CHECK(nd.level < L);
		for (int r = 0; r < 2; ++r)
			for (const treee_octree_node &nd : plan.level_nodes(r)) {
				CHECK(nd.level == L && nd.count > 0 && nd.key_prefix >= plan.rank(r).cell_begin && nd.key_prefix < plan.rank(r).cell_end);
				++level_nodes;
			}
		CHECK(level_nodes == plan.info.n_level_nodes);
		treee::Comm comm = treee::Comm::loopback();
		int32_t rank = -1, world = -1, rccl = -1;
		CHECK(treee_comm_info(comm.handle(), &rank, &world, &rccl) == TREEE_OK && rank == 0 && world == 1 && rccl == 0);
	}
	// ---- SegmentData::new + SegmentInformation::new / update for all segments (calculations.rs:133-327) ---------
	{
		std::vector<uint32_t> segment(n);
		for (size_t i = 0; i < n; ++i) segment[i] = points[i][0] < 10.0f ? 0u : (points[i][0] < 20.0f ? 1u : 3u); // segment 2 stays empty
		const treee::SegmentsData got = treee::segments_calculate(points, segment, 4);
		for (uint32_t sg = 0; sg < 4; ++sg) {
			std::vector<Point3> part;
			std::vector<size_t> where;
			for (size_t i = 0; i < n; ++i)
				if (segment[i] == sg) {
					part.push_back(points[i]);
					where.push_back(i);
				}
			float mn[3], mx[3], info[4];
			std::vector<uint8_t> cls(part.size() + 1);
			std::vector<float> ex(part.size() + 1), hg(part.size() + 1);
			orc_segment_data(part.empty() ? nullptr : part[0].data(), part.size(), mn, mx, info, cls.data());
			orc_segment_update(part.empty() ? nullptr : part[0].data(), cls.data(), part.size(), mn[1], mx[1], info, ex.data(), hg.data());
			CHECK(std::memcmp(got.min[sg].data(), mn, 12) == 0 && std::memcmp(got.max[sg].data(), mx, 12) == 0);
			CHECK(std::memcmp(&got.info[sg], info, 16) == 0);
			size_t same = 0;
			for (size_t i = 0; i < part.size(); ++i)
				same += (uint32_t)got.classifications[where[i]] == cls[i] && std::memcmp(&got.properties.expansion[where[i]], &ex[i], 4) == 0 &&
				        std::memcmp(&got.properties.height[where[i]], &hg[i], 4) == 0;
			CHECK(same == part.size());
		}
		bool threw = false;
		try {
			segment.pop_back();
			treee::segments_calculate(points, segment, 4);
		} catch (const treee::Error &) {
			threw = true;
		}
		CHECK(threw);
	}
	// ---- segmentation (segmenting.rs:110-235): separate columns keep one id each through all slices -------------
	{
		std::vector<Point3> cols;
		std::vector<uint32_t> truth;
		const float cx[3] = {-9.0f, 0.0f, 9.0f}, cz[3] = {-7.0f, 8.0f, -6.0f};
		for (uint32_t t = 0; t < 3; ++t)
			for (int i = 0; i < 700; ++i) {
				const float a = u01(seed) * 6.2831853f, r = 0.7f * std::sqrt(u01(seed));
				cols.push_back({cx[t] + r * std::cos(a), u01(seed) * 5.5f, cz[t] + r * std::sin(a)});
				truth.push_back(t);
			}
		const treee::Segmentation sg = treee::segmentation(cols, {-10.0f, 0.0f, -8.0f}, {10.0f, 5.5f, 9.0f});
		CHECK(sg.n_segments == 4);
		uint32_t id_of[3] = {sg.segment[0], sg.segment[700], sg.segment[1400]};
		CHECK(id_of[0] != id_of[1] && id_of[1] != id_of[2] && id_of[0] != id_of[2] && id_of[0] >= 1 && id_of[0] <= 3);
		size_t same = 0;
		for (size_t i = 0; i < cols.size(); ++i) same += sg.segment[i] == id_of[truth[i]];
		CHECK(same == cols.size());
	}
	// ---- PointCloud::render into depth + id (point_cloud.rs:148-173, point_cloud.wgsl) --------------------
	{
		const uint32_t w = 160, h = 90;
		const auto proj = treee::perspective((float)w / h, 45.0f * 2.0f * 3.14159265358979323846f / 360.0f, 0.1f, 10000.0f);
		float want_proj[16];
		orc_perspective((float)w / h, 45.0f * 2.0f * 3.14159265358979323846f / 360.0f, 0.1f, 10000.0f, want_proj);
		CHECK(std::memcmp(proj.data(), want_proj, 64) == 0);
		// view = inverse of translate(15, 5, 60): camera in front of the cloud (camera.rs:20-28 with another offset)
		const float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -15, -5, -60, 1};
		std::vector<uint32_t> property(n);
		for (size_t i = 0; i < n; ++i) property[i] = (uint32_t)(splitmix(seed) >> 32);
		treee::PointCloudEnvironment env;
		env.scale = 0.3f;
		const treee::Frame f = treee::render(points, property, view, proj.data(), env, w, h);
		std::vector<float> depth(w * h);
		std::vector<uint32_t> id(w * h);
		orc_splat(points[0].data(), property.data(), n, view, proj.data(), env.scale, env.min, env.max, w, h, depth.data(), id.data(), nullptr);
		CHECK(std::memcmp(f.depth.data(), depth.data(), depth.size() * 4) == 0);
		CHECK(std::memcmp(f.id.data(), id.data(), id.size() * 4) == 0);
		size_t drawn = 0;
		for (float d : f.depth) drawn += d < 1.0f;
		CHECK(drawn > 500);
		bool threw = false;
		try {
			property.pop_back();
			treee::render(points, property, view, proj.data(), env, w, h);
		} catch (const treee::Error &) {
			threw = true;
		}
		CHECK(threw);
	}
	// ---- laz::Chunk::read + slice keys (laz.rs:193-213, loading.rs:99) --------------------------------------
	{
		const size_t m = 50000, stride = 28;
		std::vector<uint8_t> rec(m * stride);
		for (size_t i = 0; i < m; ++i) {
			int32_t v[3] = {(int32_t)(splitmix(seed) >> 33) - (1 << 30), (int32_t)(splitmix(seed) >> 40), -(int32_t)(splitmix(seed) >> 44)};
			std::memcpy(&rec[i * stride], v, 12);
		}
		treee::laz::Chunk chunk;
		chunk.slice = rec.data();
		chunk.slice_len = rec.size();
		chunk.point_length = stride;
		const double scale[3] = {0.001, 0.002, 0.0005}, offset[3] = {500000.25, 4.5e6, 120.125}, center[3] = {500100.0, 4.5e6 + 3.0, 130.0};
		std::memcpy(chunk.scale, scale, 24);
		std::memcpy(chunk.offset, offset, 24);
		std::memcpy(chunk.center, center, 24);
		const std::vector<Point3> got = chunk.read();
		std::vector<Point3> want(m);
		orc_decode_las_records(rec.data(), m, stride, scale, offset, center, want[0].data());
		CHECK(got.size() == m && std::memcmp(got[0].data(), want[0].data(), m * 12) == 0);
		const std::vector<int64_t> keys = treee::slice_keys(got);
		std::vector<int64_t> wkeys(m);
		orc_slice_keys(got[0].data(), m, wkeys.data());
		CHECK(keys == wkeys);
	}
	orc_kdtree_free(ref);
	if (failures) {
		std::fprintf(stderr, "%d check(s) failed\n", failures);
		return 1;
	}
	std::puts("host mirror: all checks passed");
	return 0;
}
