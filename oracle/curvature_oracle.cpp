// curvature_oracle.cpp -- CPU ORACLE (test infrastructure): the only consumer of
// k-nearest in treee: SegmentInformation::update curve branch
// (treee/src/calculations.rs:328-373) + fast_eigenvalues (:467-498) + map_to_u32 (:455-457).
// Third-party arithmetic restated from published source, parity unpinned:
//   nalgebra 0.33.0 (Cargo.lock:1562-1563): Point3 += Vector3, Point3 / f32 (component-wise
//   true division), Matrix3::determinant (cofactor expansion along the first row);
//   f32::powi(-3) = compiler-rt __powisf2 / LLVM ExpandPowI = 1 / (p * (p * p)).
// acos / cos: Rust forwards them to the platform libm, HIP to OCML; both sides of the parity tests use the one
// shared, correctly rounding definition of include/treee_f32_trig.h instead (checked against this host's libm in
// tests/test_oracle.py: never more than 1 ulp apart, identical for > 99.7 % of the arguments), so that `curve` is
// compared bit for bit wherever the neighbour lists agree.
#include "treee_oracle.h"

#include "../include/treee_f32_trig.h"

#include <cmath>
#include <thread>
#include <vector>

namespace {

inline float square(float x) { return x * x; }

// nalgebra Matrix3::determinant, m[r][c]
inline float det3(const float m[3][3]) {
	float minor_m12_m23 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
	float minor_m11_m23 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
	float minor_m11_m22 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
	return m[0][0] * minor_m12_m23 - m[0][1] * minor_m11_m23 + m[0][2] * minor_m11_m22;
}

// calculations.rs:467-498.  LIBM = false: acos / cos of include/treee_f32_trig.h (shared with the kernels: bit parity);
// LIBM = true: this host's libm acosf / cosf, i.e. what Rust's f32::acos / f32::cos forward to on a linux-gnu target -- the
// reference's own last bits on this machine, kept so that the shared definition is measured against the reference and not
// only against itself (tests: explicit absolute bound).
template <bool LIBM> void fast_eigenvalues_t(const float mat[3][3], float out[3]) {
	float p1 = square(mat[0][1]) + square(mat[0][2]) + square(mat[1][2]);
	if (p1 == 0.0f) { // :473-476 diagonal returned unsorted
		out[0] = mat[0][0];
		out[1] = mat[1][1];
		out[2] = mat[2][2];
		return;
	}
	float q = (mat[0][0] + mat[1][1] + mat[2][2]) / 3.0f;
	float p2 = square(mat[0][0] - q) + square(mat[1][1] - q) + square(mat[2][2] - q) + 2.0f * p1;
	float p = std::sqrt(p2 / 6.0f);
	float mat_b[3][3];
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) mat_b[r][c] = mat[r][c];
	for (int i = 0; i < 3; ++i) mat_b[i][i] -= q;
	float powi_m3 = 1.0f / (p * (p * p)); // p.powi(-3)
	float r = det3(mat_b) / 2.0f * powi_m3;
	const float PI = 3.14159265358979323846f; // std::f32::consts::PI
	float phi;
	if (r <= -1.0f)
		phi = PI / 3.0f;
	else if (r >= 1.0f)
		phi = 0.0f;
	else
		phi = (LIBM ? std::acos(r) : treee_acos_f32(r)) / 3.0f;
	float eig_1 = q + 2.0f * p * (LIBM ? std::cos(phi) : treee_cos_f32(phi));
	float eig_3 = q + 2.0f * p * (LIBM ? std::cos(phi + (2.0f * PI / 3.0f)) : treee_cos_f32(phi + (2.0f * PI / 3.0f)));
	float eig_2 = 3.0f * q - eig_1 - eig_3;
	out[0] = eig_1;
	out[1] = eig_2;
	out[2] = eig_3;
}
void fast_eigenvalues(const float mat[3][3], float out[3]) { fast_eigenvalues_t<false>(mat, out); }

// calculations.rs:342-368 for one point, neighbours in returned (ascending) order
template <bool LIBM = false> float curve_from(const float *xyz, const orc_entry *nb, uint64_t count) {
	float len = (float)count; // neighbors.len() as f32
	float mean[3] = {0.0f, 0.0f, 0.0f};
	for (uint64_t j = 0; j < count; ++j) { // :344-346
		const float *p = xyz + 3 * nb[j].index;
		mean[0] += p[0];
		mean[1] += p[1];
		mean[2] += p[2];
	}
	mean[0] = mean[0] / len; // :347
	mean[1] = mean[1] / len;
	mean[2] = mean[2] / len;
	float var[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; // Matrix3::default() = zeros
	for (uint64_t j = 0; j < count; ++j) { // :351-358
		const float *p = xyz + 3 * nb[j].index;
		float diff[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
		for (int x = 0; x < 3; ++x)
			for (int y = 0; y < 3; ++y) var[x][y] += diff[x] * diff[y];
	}
	for (int x = 0; x < 3; ++x) // :359-363
		for (int y = 0; y < 3; ++y) var[x][y] /= len;
	float eig[3];
	fast_eigenvalues_t<LIBM>(var, eig);
	return (3.0f * eig[2]) / (eig[0] + eig[1] + eig[2]); // :368
}

} // namespace

extern "C" {

void orc_fast_eigenvalues(const float m[9], float out[3]) {
	float mat[3][3];
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) mat[r][c] = m[3 * r + c];
	fast_eigenvalues(mat, out);
}

float orc_curve_from_neighbors(const float *xyz, const orc_entry *nb, uint64_t count) {
	return curve_from(xyz, nb, count);
}

// the loop body of calculations.rs:342-368 over GIVEN neighbour lists (u32 index[nq][k], the first counts[i] of a row
// are valid, summed in the order given): separates the arithmetic from the neighbour search when an implementation's
// lists differ from the KD-tree's inside distance ties
// libm != 0: acos / cos of this host's libm instead of the shared header (see fast_eigenvalues_t)
void orc_curve_from_lists(const float *xyz, const uint32_t *index, const uint32_t *counts, uint64_t nq, uint64_t k,
                          float *curve_out, int threads) {
	orc_curve_from_lists_trig(xyz, index, counts, nq, k, curve_out, threads, 0);
}
void orc_curve_from_lists_trig(const float *xyz, const uint32_t *index, const uint32_t *counts, uint64_t nq, uint64_t k,
                               float *curve_out, int threads, int libm) {
	auto work = [&](uint64_t lo, uint64_t hi) {
		std::vector<orc_entry> loc(k);
		for (uint64_t i = lo; i < hi; ++i) {
			const uint64_t c = counts ? counts[i] : k;
			for (uint64_t j = 0; j < c; ++j) loc[j].index = index[i * k + j];
			curve_out[i] = libm ? curve_from<true>(xyz, loc.data(), c) : curve_from<false>(xyz, loc.data(), c);
		}
	};
	if (threads <= 1) return work(0, nq);
	std::vector<std::thread> pool;
	const uint64_t per = (nq + threads - 1) / threads;
	for (int th = 0; th < threads; ++th) {
		const uint64_t lo = per * th < nq ? per * th : nq, hi = lo + per < nq ? lo + per : nq;
		if (lo < hi) pool.emplace_back(work, lo, hi);
	}
	for (auto &th : pool) th.join();
}

// the two shared functions themselves, and this host's libm beside them (tests)
void orc_trig_f32(const float *x, uint64_t n, float *acos_out, float *cos_out, float *libm_acos_out, float *libm_cos_out) {
	for (uint64_t i = 0; i < n; ++i) {
		acos_out[i] = treee_acos_f32(x[i]);
		cos_out[i] = treee_cos_f32(x[i]);
		libm_acos_out[i] = std::acos(x[i]);
		libm_cos_out[i] = std::cos(x[i]);
	}
}

void orc_curvature(const float *xyz, uint64_t n, uint64_t k, float max_distance, float *curve_out,
                   int threads) {
	if (n == 0) return;
	orc_kdtree *tree = orc_kdtree_new(xyz, n); // NeighborsTree::new (calculations.rs:329)
	auto work = [&](uint64_t lo, uint64_t hi) {
		std::vector<orc_entry> loc(k); // neighbors_location, reused (calculations.rs:331)
		for (uint64_t i = lo; i < hi; ++i) {
			uint64_t c = orc_kdtree_k_nearest(tree, xyz + 3 * i, loc.data(), k, max_distance);
			curve_out[i] = curve_from(xyz, loc.data(), c);
		}
	};
	if (threads <= 1) {
		work(0, n); // as-reference: sequential iterator (calculations.rs:332)
	} else {
		std::vector<std::thread> pool;
		uint64_t per = (n + threads - 1) / threads;
		for (int th = 0; th < threads; ++th) {
			uint64_t lo = per * th < n ? per * th : n, hi = lo + per < n ? lo + per : n;
			if (lo < hi) pool.emplace_back(work, lo, hi);
		}
		for (auto &th : pool) th.join();
	}
	orc_kdtree_free(tree);
}

uint32_t orc_map_to_u32(float v) {
	// (value * u32::MAX as f32) as u32 : Rust `as` saturates, NaN -> 0
	float x = v * 4294967296.0f; // u32::MAX as f32 rounds to 2^32
	if (!(x > 0.0f)) return 0;   // negative, zero, NaN
	if (x >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)x;
}

} // extern "C"
