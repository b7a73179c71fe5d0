// multi.cpp -- the per-segment phases of the reference over several GPUs of ONE process, the way the reference spreads them
// over CPU workers: whole segments are independent units (Calculations::new, treee/src/calculations.rs:97-118, hands every
// segment to a rayon task; :329 builds one KD-tree per segment), so nothing is exchanged.  Host code: deal the segments,
// one host thread + one context per device, scatter the results back.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <string>
#include <system_error>
#include <thread>

using namespace treee;

namespace {

// longest segment first to the least loaded device (ties: lower segment id, lower list position) -- the rule of
// treee_amd/distributed.py::partition_segments
std::vector<uint32_t> deal_segments(const std::vector<uint64_t> &counts, uint32_t n_dev) {
	std::vector<uint32_t> order(counts.size()), owner(counts.size(), 0);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return counts[a] > counts[b]; });
	std::vector<uint64_t> load(n_dev, 0);
	for (uint32_t sg : order) {
		const uint32_t d = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
		owner[sg] = d;
		load[d] += counts[sg];
	}
	return owner;
}

struct Share { // one device's part of the cloud
	std::vector<uint32_t> seg_ids;  // local segment -> global segment
	std::vector<uint32_t> point_of; // local point -> input position
	std::vector<float> xyz;
	std::vector<uint32_t> seg;      // dense local ids
};

int32_t make_shares(const float *xyz, const uint32_t *segment, uint64_t n, uint32_t n_segments, uint32_t n_dev, std::vector<Share> *out) {
	std::vector<uint64_t> counts(n_segments, 0);
	for (uint64_t i = 0; i < n; ++i) {
		if (segment[i] >= n_segments) return set_err(TREEE_ERR_INVALID, "segment id >= n_segments (%u)", n_segments);
		++counts[segment[i]];
	}
	const std::vector<uint32_t> owner = deal_segments(counts, n_dev);
	std::vector<uint32_t> local_id(n_segments, 0);
	out->assign(n_dev, Share{});
	for (uint32_t sg = 0; sg < n_segments; ++sg) {
		Share &sh = (*out)[owner[sg]];
		local_id[sg] = (uint32_t)sh.seg_ids.size();
		sh.seg_ids.push_back(sg);
	}
	std::vector<uint64_t> points(n_dev, 0);
	for (uint32_t sg = 0; sg < n_segments; ++sg) points[owner[sg]] += counts[sg];
	for (uint32_t d = 0; d < n_dev; ++d) {
		(*out)[d].point_of.reserve(points[d]);
		(*out)[d].xyz.reserve(points[d] * 3);
		(*out)[d].seg.reserve(points[d]);
	}
	for (uint64_t i = 0; i < n; ++i) {
		Share &sh = (*out)[owner[segment[i]]];
		sh.point_of.push_back((uint32_t)i);
		sh.xyz.insert(sh.xyz.end(), xyz + 3 * i, xyz + 3 * i + 3);
		sh.seg.push_back(local_id[segment[i]]);
	}
	return TREEE_OK;
}

// the cube treee_knn_build_segments_dev would compute for the whole cloud (bbox_dev, radix_sort.hip)
void whole_cube(const float *xyz, uint64_t n, float cube[4]) {
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (uint64_t i = 0; i < n; ++i)
		for (int a = 0; a < 3; ++a) {
			mn[a] = fminf(mn[a], xyz[3 * i + a]);
			mx[a] = fmaxf(mx[a], xyz[3 * i + a]);
		}
	float size = 0.0f;
	for (int a = 0; a < 3; ++a) {
		const float e = mx[a] - mn[a];
		if (e > size) size = e;
		cube[a] = (mn[a] > -INFINITY && mn[a] < INFINITY) ? mn[a] : 0.0f;
	}
	if (!(size > 0.0f) || !(size < INFINITY)) size = 1.0f;
	cube[3] = size;
}

struct DeviceRun { // a context on a private stream, destroyed with the frame
	treee_ctx *ctx = nullptr;
	hipStream_t stream = nullptr;
	~DeviceRun() {
		if (ctx) treee_ctx_destroy(ctx);
		if (stream) (void)hipStreamDestroy(stream);
	}
	int32_t open(int32_t device) {
		TREEE_HIP(hipSetDevice(device));
		TREEE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
		return treee_ctx_create(device, stream, &ctx);
	}
};

template <class F> int32_t run_on_devices(const int32_t *devices, uint32_t n_dev, F &&body) {
	std::vector<int32_t> rc(n_dev, TREEE_OK);
	std::vector<std::string> msg(n_dev);
	auto run = [&](uint32_t d) {
		try {
			rc[d] = body(d, devices[d]);
		} catch (const std::exception &e) {
			rc[d] = set_err(TREEE_ERR_NOMEM, "device %d: %s", devices[d], e.what());
		}
		if (rc[d] != TREEE_OK) msg[d] = treee_last_error(); // (the message is thread-local: carry it to the caller's thread)
	};
	std::vector<std::thread> pool;
	pool.reserve(n_dev); // (emplace_back below never reallocates: a failure can only come from the thread itself)
	for (uint32_t d = 0; d < n_dev; ++d) {
		try {
			pool.emplace_back(run, d);
		} catch (const std::system_error &) {
			// no more threads: this device's share runs on the calling thread -- nothing may unwind through joinable
			// threads (std::terminate) or across the C ABI
			run(d);
		}
	}
	for (auto &t : pool) t.join();
	for (uint32_t d = 0; d < n_dev; ++d)
		if (rc[d] != TREEE_OK) return set_err(rc[d], "device %d (list position %u): %s", devices[d], d, msg[d].c_str());
	return TREEE_OK;
}

int32_t check_devices(const int32_t *devices, uint32_t n_devices) {
	if (!devices || n_devices == 0) return set_err(TREEE_ERR_INVALID, "empty device list");
	int count = 0;
	TREEE_HIP(hipGetDeviceCount(&count));
	for (uint32_t d = 0; d < n_devices; ++d)
		if (devices[d] < 0 || devices[d] >= count) return set_err(TREEE_ERR_INVALID, "device %d is not one of the %d visible devices", devices[d], count);
	return TREEE_OK;
}

} // namespace

extern "C" {

int32_t treee_curvature_segments_multi(const int32_t *devices, uint32_t n_devices, const float *xyz, const uint32_t *segment, uint64_t n,
                                       uint32_t n_segments, uint32_t k, float max_dist_sq, float *curve_out) {
	if (n == 0) return TREEE_OK;
	if (!xyz || !segment || !curve_out) return set_err(TREEE_ERR_INVALID, "treee_curvature_segments_multi: null argument");
	if (k == 0) return set_err(TREEE_ERR_INVALID, "k == 0");
	if (n_segments == 0 || n_segments > (1u << 24)) return set_err(TREEE_ERR_INVALID, "n_segments %u outside [1, 2^24]", n_segments);
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large");
	TREEE_TRY(check_devices(devices, n_devices));
	try {
		std::vector<Share> shares;
		TREEE_TRY(make_shares(xyz, segment, n, n_segments, n_devices, &shares));
		float cube[4];
		whole_cube(xyz, n, cube);
		return run_on_devices(devices, n_devices, [&](uint32_t d, int32_t device) -> int32_t {
			const Share &sh = shares[d];
			const uint64_t m = sh.point_of.size();
			if (m == 0) return TREEE_OK;
			DeviceRun run;
			TREEE_TRY(run.open(device));
			DevBuf<float> d_xyz, d_curve;
			DevBuf<uint32_t> d_seg;
			TREEE_TRY(d_xyz.alloc(m * 3));
			TREEE_TRY(d_seg.alloc(m));
			TREEE_TRY(d_curve.alloc(m));
			TREEE_HIP(hipMemcpyAsync(d_xyz.p, sh.xyz.data(), m * 12, hipMemcpyHostToDevice, run.stream));
			TREEE_HIP(hipMemcpyAsync(d_seg.p, sh.seg.data(), m * 4, hipMemcpyHostToDevice, run.stream));
			std::vector<float> h(m); // (before the handle exists: an allocation failure here has nothing to leak)
			treee_knn *t = nullptr;
			TREEE_TRY(knn_build_segments_cube(run.ctx, d_xyz.p, d_seg.p, m, (uint32_t)sh.seg_ids.size(), cube, &t));
			int32_t rc = treee_curvature_dev(run.ctx, t, k, max_dist_sq, d_curve.p);
			if (rc == TREEE_OK) {
				hipError_t e = hipMemcpyAsync(h.data(), d_curve.p, m * 4, hipMemcpyDeviceToHost, run.stream);
				if (e == hipSuccess) e = hipStreamSynchronize(run.stream);
				if (e != hipSuccess) rc = set_err(TREEE_ERR_HIP, "curvature copy back: %s", hipGetErrorString(e));
			}
			treee_knn_free(t);
			if (rc != TREEE_OK) return rc;
			for (uint64_t i = 0; i < m; ++i) curve_out[sh.point_of[i]] = h[i]; // (the devices' point sets are disjoint)
			return TREEE_OK;
		});
	} catch (const std::exception &e) {
		return set_err(TREEE_ERR_NOMEM, "treee_curvature_segments_multi: %s", e.what());
	}
}

int32_t treee_segments_calculate_multi(const int32_t *devices, uint32_t n_devices, const float *xyz, const uint32_t *segment, uint64_t n,
                                       uint32_t n_segments, float *seg_min_out, float *seg_max_out, float *seg_info_out,
                                       uint32_t *classification_out, float *expansion_out, float *height_out) {
	if (n && (!xyz || !segment)) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate_multi: null argument");
	if (!seg_min_out || !seg_max_out || !seg_info_out) return set_err(TREEE_ERR_INVALID, "treee_segments_calculate_multi: null per-segment output");
	if (n_segments == 0 || n_segments > (1u << 24)) return set_err(TREEE_ERR_INVALID, "n_segments %u outside [1, 2^24]", n_segments);
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large");
	TREEE_TRY(check_devices(devices, n_devices));
	try {
		std::vector<Share> shares;
		TREEE_TRY(make_shares(xyz, segment, n, n_segments, n_devices, &shares));
		return run_on_devices(devices, n_devices, [&](uint32_t d, int32_t device) -> int32_t {
			const Share &sh = shares[d];
			const uint64_t m = sh.point_of.size();
			const uint32_t ns = (uint32_t)sh.seg_ids.size();
			if (ns == 0) return TREEE_OK;
			DeviceRun run;
			TREEE_TRY(run.open(device));
			DevBuf<float> d_xyz, d_min, d_max, d_info, d_exp, d_hgt;
			DevBuf<uint32_t> d_seg, d_cls;
			TREEE_TRY(d_xyz.alloc(m * 3));
			TREEE_TRY(d_seg.alloc(m));
			TREEE_TRY(d_cls.alloc(m));
			TREEE_TRY(d_min.alloc((size_t)ns * 3));
			TREEE_TRY(d_max.alloc((size_t)ns * 3));
			TREEE_TRY(d_info.alloc((size_t)ns * 4));
			if (expansion_out) TREEE_TRY(d_exp.alloc(m));
			if (height_out) TREEE_TRY(d_hgt.alloc(m));
			hipStream_t st = run.stream;
			if (m) {
				TREEE_HIP(hipMemcpyAsync(d_xyz.p, sh.xyz.data(), m * 12, hipMemcpyHostToDevice, st));
				TREEE_HIP(hipMemcpyAsync(d_seg.p, sh.seg.data(), m * 4, hipMemcpyHostToDevice, st));
			}
			TREEE_TRY(treee_segments_calculate_dev(run.ctx, d_xyz.p, d_seg.p, m, ns, d_min.p, d_max.p, d_info.p, d_cls.p, expansion_out ? d_exp.p : nullptr,
			                                       height_out ? d_hgt.p : nullptr));
			std::vector<float> h_min((size_t)ns * 3), h_max((size_t)ns * 3), h_info((size_t)ns * 4), h_exp(expansion_out ? m : 0), h_hgt(height_out ? m : 0);
			std::vector<uint32_t> h_cls(classification_out ? m : 0);
			TREEE_HIP(hipMemcpyAsync(h_min.data(), d_min.p, h_min.size() * 4, hipMemcpyDeviceToHost, st));
			TREEE_HIP(hipMemcpyAsync(h_max.data(), d_max.p, h_max.size() * 4, hipMemcpyDeviceToHost, st));
			TREEE_HIP(hipMemcpyAsync(h_info.data(), d_info.p, h_info.size() * 4, hipMemcpyDeviceToHost, st));
			if (m && classification_out) TREEE_HIP(hipMemcpyAsync(h_cls.data(), d_cls.p, m * 4, hipMemcpyDeviceToHost, st));
			if (m && expansion_out) TREEE_HIP(hipMemcpyAsync(h_exp.data(), d_exp.p, m * 4, hipMemcpyDeviceToHost, st));
			if (m && height_out) TREEE_HIP(hipMemcpyAsync(h_hgt.data(), d_hgt.p, m * 4, hipMemcpyDeviceToHost, st));
			TREEE_HIP(hipStreamSynchronize(st));
			for (uint32_t s = 0; s < ns; ++s) {
				const uint32_t g = sh.seg_ids[s];
				std::memcpy(seg_min_out + 3 * (size_t)g, h_min.data() + 3 * (size_t)s, 12);
				std::memcpy(seg_max_out + 3 * (size_t)g, h_max.data() + 3 * (size_t)s, 12);
				std::memcpy(seg_info_out + 4 * (size_t)g, h_info.data() + 4 * (size_t)s, 16);
			}
			for (uint64_t i = 0; i < m; ++i) {
				const uint32_t p = sh.point_of[i];
				if (classification_out) classification_out[p] = h_cls[i];
				if (expansion_out) expansion_out[p] = h_exp[i];
				if (height_out) height_out[p] = h_hgt[i];
			}
			return TREEE_OK;
		});
	} catch (const std::exception &e) {
		return set_err(TREEE_ERR_NOMEM, "treee_segments_calculate_multi: %s", e.what());
	}
}

} // extern "C"
