// knn_sim.cpp -- CPU model of what a wave of the kNN kernel DOES (buckets, candidate steps, buffer flushes, box tests), for the
// shipped structure ("cur": wave-shared broadcast scans + transposed scans, csrc/knn.hip) and for per-lane candidate iteration
// over LDS-resident sub-buckets ("pl").  Not product code and not a checker: a design tool for profiles/README.md (round 4).
// Input: profiles/sim/make_cloud.py (a forest patch in the structure's Hilbert order).  Thresholds evolve exactly as in the
// kernel: per-lane 32-entry list, 16-slot buffer, coarse filter d_bits <= thr, thr only moves at a flush, a flush when any
// lane holds more than 12 pending entries (checked every 4 candidates).
//   g++ -O2 -std=c++17 -ffp-contract=off -fopenmp knn_sim.cpp -o knn_sim && ./knn_sim /tmp/knn_sim_cloud.f32
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct P3 {
	float x, y, z;
};
struct Box {
	float lo[3], hi[3];
};
static inline uint32_t fbits(float f) {
	uint32_t u;
	memcpy(&u, &f, 4);
	return u;
}
static inline float dist2(const P3 &c, const P3 &q) {
	const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z;
	return (dx * dx + dy * dy) + dz * dz;
}
static inline float box_point(const Box &b, const P3 &q) {
	const float qq[3] = {q.x, q.y, q.z};
	float d[3];
	for (int a = 0; a < 3; ++a) d[a] = std::fmax(std::fmax(b.lo[a] - qq[a], qq[a] - b.hi[a]), 0.0f);
	return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
}
static inline float box_box(const Box &n, const Box &g) {
	float d[3];
	for (int a = 0; a < 3; ++a) d[a] = std::fmax(std::fmax(n.lo[a] - g.hi[a], g.lo[a] - n.hi[a]), 0.0f);
	return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
}
static Box box_of(const P3 *p, int n) {
	Box b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
	for (int i = 0; i < n; ++i) {
		const float c[3] = {p[i].x, p[i].y, p[i].z};
		for (int a = 0; a < 3; ++a) {
			b.lo[a] = std::fmin(b.lo[a], c[a]);
			b.hi[a] = std::fmax(b.hi[a], c[a]);
		}
	}
	return b;
}

constexpr int SK = 32, SBUF = 16, BUCKET = 64;
constexpr uint32_t MAXBITS = 0x7F7FFFFFu;

struct Cloud {
	std::vector<P3> pts;
	uint32_t n = 0, nb = 0, P = 1;
	std::vector<Box> bvh;                 // node i, leaves at P + b
	std::vector<Box> sub16, sub32;        // [nb * 4], [nb * 2]
	void build() {
		n = (uint32_t)pts.size();
		nb = (n + 63) / 64;
		while (P < nb) P <<= 1;
		bvh.assign(2 * (size_t)P, Box{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}});
		sub16.resize((size_t)nb * 4);
		sub32.resize((size_t)nb * 2);
		for (uint32_t b = 0; b < nb; ++b) {
			const int cnt = (int)std::min<uint32_t>(64, n - b * 64);
			bvh[P + b] = box_of(&pts[b * 64], cnt);
			for (int s = 0; s < 4; ++s) sub16[b * 4 + s] = box_of(&pts[b * 64 + s * 16], std::max(0, std::min(16, cnt - s * 16)));
			for (int s = 0; s < 2; ++s) sub32[b * 2 + s] = box_of(&pts[b * 64 + s * 32], std::max(0, std::min(32, cnt - s * 32)));
		}
		for (uint32_t i = P - 1; i >= 1; --i) {
			Box &o = bvh[i];
			const Box &l = bvh[2 * i], &r = bvh[2 * i + 1];
			for (int a = 0; a < 3; ++a) {
				o.lo[a] = std::fmin(l.lo[a], r.lo[a]);
				o.hi[a] = std::fmax(l.hi[a], r.hi[a]);
			}
		}
	}
};

struct Counters {
	double waves = 0, buckets = 0, steps = 0 /* candidate steps of lane-parallel or per-lane scans (one candidate per lane) */, flushes = 0, appended = 0,
	       tq = 0 /* transposed (query, bucket) pairs */, tq_pass = 0, boxtests = 0 /* per-lane box tests, wave instructions groups */, bfs_batches = 0,
	       listed = 0, rounds = 0, evals_useful = 0, subtests = 0, lanes_need_sum = 0, full_scans = 0, small_scans = 0, idle_lane_steps = 0, retests = 0,
	       max_lane_appended = 0, fill_hist[17] = {}, flush_phase[4] = {}, lanes_over8_at_flush = 0, sum_final_thr_ratio = 0,
	       pd_full = 0, pd_paired = 0, pd_inter_small = 0,
	       gq_any = 0, gq_max4 = 0, gq_mean4 = 0, gq_max2 = 0, gq_max8 = 0, gq_buckets = 0,
	       q_total = 0, q_zero = 0, q_low = 0, q_need_sum = 0, h_total = 0, h_zero = 0, flush_ce = 0, flush_wide = 0, flush_lo_untouched = 0, flush_lo8_untouched = 0, lanes_lo_touched = 0, nbr_need_hist[5] = {}, nbr_total = 0;
	void add(const Counters &o) {
		double *a = &waves;
		const double *b = &o.waves;
		for (size_t i = 0; i < sizeof(Counters) / sizeof(double); ++i) a[i] += b[i];
	}
};

struct Wave {
	const Cloud &c;
	Counters &k;
	P3 q[64];
	uint64_t L[64][SK];
	uint32_t thr[64];
	std::vector<uint64_t> buf[64];
	uint32_t start;
	int phase = 0;
	uint32_t lane_app[64] = {};
	Wave(const Cloud &c_, Counters &k_, uint32_t b) : c(c_), k(k_), start(b) {
		for (int l = 0; l < 64; ++l) {
			q[l] = c.pts[b * 64 + l];
			for (int i = 0; i < SK; ++i) L[l][i] = (uint64_t)MAXBITS << 32;
			thr[l] = MAXBITS;
			buf[l].reserve(SBUF);
		}
	}
	int pair_disjoint = 0; // 1: a wave-shared scan takes a second listed bucket along when the two need masks are disjoint (statistics only)
	int list_order = 0;   // 1: listed buckets nearest first (box distance to the wave's query box), 2: by needing lanes, most first
	int tighten_mode = 0; // 1: after every bucket thr = min(thr, max(L[31 - m], largest pending)); 2: the exact 32nd of list + pending
	void tighten() {
		if (!tighten_mode) return;
		for (int l = 0; l < 64; ++l) {
			const size_t m = buf[l].size();
			if (!m) continue;
			if (tighten_mode == 1) {
				uint32_t pmax = 0;
				for (uint64_t e : buf[l]) pmax = std::max(pmax, (uint32_t)(e >> 32));
				const uint32_t lm = (uint32_t)(L[l][SK - 1 - std::min<size_t>(m, SK - 1)] >> 32);
				thr[l] = std::min(thr[l], std::max(lm, pmax));
			} else {
				uint64_t tmp[SK + 4 * SBUF];
				memcpy(tmp, L[l], sizeof(uint64_t) * SK);
				memcpy(tmp + SK, buf[l].data(), sizeof(uint64_t) * m);
				std::nth_element(tmp, tmp + SK - 1, tmp + SK + m);
				thr[l] = std::min(thr[l], (uint32_t)(tmp[SK - 1] >> 32));
			}
		}
	}
	int pair_xor = 0; // > 0: lanes l and l ^ pair_xor share 2 x SBUF slots
	int sbuf = SBUF;
	bool any_pending(size_t more_than) const {
		if (more_than == 0 || pair_xor == 0) {
			const size_t lim = more_than == 0 ? 0 : (size_t)sbuf - 4;
			for (int l = 0; l < 64; ++l)
				if (buf[l].size() > lim) return true;
			return false;
		}
		for (int l = 0; l < 64; ++l)
			if (buf[l].size() + buf[l ^ pair_xor].size() > (size_t)(2 * sbuf - 8)) return true;
		return false;
	}
	void flush() {
		k.flush_phase[phase] += 1;
		{
			size_t mx = 0;
			for (int l = 0; l < 64; ++l) mx = std::max(mx, buf[l].size());
			// compare-exchanges of the merge: 16 entries per lane 60 + 16 + 80; up to 32: 191 + 32 + 80; beyond: one more 16-pass each
			k.flush_ce += mx <= 16 ? 156 : (mx <= 32 ? 303 : 303 + 156 * (double)((mx - 32 + 15) / 16));
			k.flush_wide += mx > 16;
		}
		{
			bool lo_touched = false, lo8_touched = false;
			for (int l = 0; l < 64; ++l)
				for (uint64_t e : buf[l]) {
					if (e < L[l][15]) {
						lo_touched = true;
						k.lanes_lo_touched += 1;
						break;
					}
				}
			for (int l = 0; l < 64 && !lo8_touched; ++l)
				for (uint64_t e : buf[l])
					if (e < L[l][7]) {
						lo8_touched = true;
						break;
					}
			k.flush_lo_untouched += !lo_touched;
			k.flush_lo8_untouched += !lo8_touched;
		}
		for (int l = 0; l < 64; ++l) {
			k.fill_hist[std::min<size_t>(16, buf[l].size())] += 1;
			if (buf[l].size() > 8) k.lanes_over8_at_flush += 1;
			if (!buf[l].empty()) {
				uint64_t tmp[SK + 4 * SBUF];
				memcpy(tmp, L[l], sizeof(uint64_t) * SK);
				memcpy(tmp + SK, buf[l].data(), sizeof(uint64_t) * buf[l].size());
				const size_t m = SK + buf[l].size();
				std::partial_sort(tmp, tmp + SK, tmp + m);
				memcpy(L[l], tmp, sizeof(uint64_t) * SK);
				buf[l].clear();
			}
			const uint32_t nt = (uint32_t)(L[l][SK - 1] >> 32);
			thr[l] = std::min(thr[l], nt);
		}
		k.flushes += 1;
	}
	void offer(int l, uint32_t pos) {
		const uint32_t db = fbits(dist2(c.pts[pos], q[l]));
		if (db <= thr[l]) { // NaN never (no NaN here)
			buf[l].push_back(((uint64_t)db << 32) | pos);
			k.appended += 1;
			lane_app[l] += 1;
		}
	}
	// wave-shared broadcast scan of bucket b, candidates [first, 64), lanes in `mask` append
	void scan_tile(uint32_t b, uint64_t mask, int first) {
		const uint32_t base = b * 64;
		for (int c0 = first; c0 < 64; c0 += 4) {
			if (any_pending(SBUF - 4)) flush();
			for (int j = 0; j < 4; ++j) {
				const uint32_t pos = base + c0 + j;
				if (pos >= c.n) continue;
				for (int l = 0; l < 64; ++l)
					if ((mask >> l) & 1) offer(l, pos);
			}
			k.steps += 4;
		}
	}
	void own_bucket() {
		const uint32_t base = start * 64;
		for (int l = 0; l < 64; ++l) {
			for (int j = 0; j < SK; ++j) {
				const uint32_t db = fbits(dist2(c.pts[base + j], q[l]));
				L[l][j] = db < MAXBITS ? (((uint64_t)db << 32) | (base + j)) : ((uint64_t)MAXBITS << 32);
			}
			std::sort(L[l], L[l] + SK);
			thr[l] = (uint32_t)(L[l][SK - 1] >> 32);
		}
		scan_tile(start, ~0ull, SK);
		k.buckets += 1;
	}
	bool prescanned(uint32_t b, int nbr) const {
		const uint32_t below = start - b, above = b - start;
		return (below >= 1u && below <= (uint32_t)((nbr + 1) / 2)) || (above >= 1u && above <= (uint32_t)(nbr / 2));
	}
	std::vector<uint32_t> nbr_list(int nbr) const {
		std::vector<uint32_t> v;
		for (int t = 1; t <= nbr; ++t) {
			const uint32_t b = (t & 1) ? start - (uint32_t)((t + 1) >> 1) : start + (uint32_t)(t >> 1);
			if (b < c.nb) v.push_back(b);
		}
		return v;
	}
	// the kernel's BFS: 4 query sub-boxes of 16 lanes with their largest threshold, 4-ary steps; list in frontier order
	std::vector<uint32_t> bfs(int nbr) {
		constexpr int NG = 4, GL = 16;
		Box g[NG];
		uint32_t gt[NG];
		for (int i = 0; i < NG; ++i) {
			g[i] = box_of(&q[i * GL], GL);
			gt[i] = 0;
			for (int l = 0; l < GL; ++l) gt[i] = std::max(gt[i], thr[i * GL + l]);
		}
		std::vector<uint32_t> front, next, list;
		for (uint32_t nd = c.P + start; nd > 1u; nd >>= 1) front.push_back(nd ^ 1u);
		while (!front.empty()) {
			next.clear();
			for (size_t base = 0; base < front.size(); base += 64) {
				k.bfs_batches += 1;
				const size_t e = std::min(front.size(), base + 64);
				for (size_t i = base; i < e; ++i) {
					const uint32_t node = front[i];
					bool hit = false;
					for (int gi = 0; gi < NG && !hit; ++gi) hit = fbits(box_box(c.bvh[node], g[gi])) <= gt[gi];
					if (!hit) continue;
					if (node >= c.P) {
						const uint32_t b = node - c.P;
						if (b < c.nb && !prescanned(b, nbr)) list.push_back(b);
					} else if (2u * node < c.P) {
						for (uint32_t ch = 0; ch < 4; ++ch) next.push_back(4u * node + ch);
					} else {
						next.push_back(2u * node);
						next.push_back(2u * node + 1u);
					}
				}
			}
			front.swap(next);
		}
		k.listed += (double)list.size();
		return list;
	}
	uint64_t need_mask(const Box &b) {
		uint64_t m = 0;
		for (int l = 0; l < 64; ++l)
			if (fbits(box_point(b, q[l])) <= thr[l]) m |= 1ull << l;
		k.boxtests += 1;
		return m;
	}
	// quarters a wave-shared scan walks today (needed by ANY lane) against what groups of 16 / 32 / 8 Hilbert-adjacent lanes would
	// walk if every group followed its own list of quarters (the scan then takes as many trips as its busiest group)
	void group_stats(uint32_t b) {
		int any = 0, per[8][4] = {};
		for (int sgm = 0; sgm < 4; ++sgm) {
			if (b * 64u + (uint32_t)(sgm * 16) >= c.n) continue;
			bool a = false;
			for (int l = 0; l < 64; ++l)
				if (fbits(box_point(c.sub16[(size_t)b * 4 + sgm], q[l])) <= thr[l]) {
					a = true;
					per[l / 8][sgm] = 1;
				}
			any += a;
		}
		auto busiest = [&](int lanes) {
			int mx = 0;
			for (int g0 = 0; g0 < 8; g0 += lanes / 8) {
				int cnt = 0;
				for (int sgm = 0; sgm < 4; ++sgm) {
					int need = 0;
					for (int g = g0; g < g0 + lanes / 8; ++g) need |= per[g][sgm];
					cnt += need;
				}
				mx = std::max(mx, cnt);
			}
			return mx;
		};
		int sum4 = 0;
		for (int g0 = 0; g0 < 8; g0 += 2)
			for (int sgm = 0; sgm < 4; ++sgm) sum4 += per[g0][sgm] | per[g0 + 1][sgm];
		k.gq_any += any;
		k.gq_max4 += busiest(16);
		k.gq_max2 += busiest(32);
		k.gq_max8 += busiest(8);
		k.gq_mean4 += sum4 / 4.0;
		k.gq_buckets += 1;
	}
	void quarter_stats(uint32_t b) {
		group_stats(b);
		for (int sgm = 0; sgm < 4; ++sgm) {
			if (b * 64u + (uint32_t)(sgm * 16) >= c.n) continue;
			int need = 0;
			for (int l = 0; l < 64; ++l) need += fbits(box_point(c.sub16[(size_t)b * 4 + sgm], q[l])) <= thr[l];
			k.q_total += 1;
			k.q_zero += need == 0;
			k.q_low += need > 0 && need < 8;
			k.q_need_sum += need;
		}
		for (int sgm = 0; sgm < 2; ++sgm) {
			int need = 0;
			for (int l = 0; l < 64; ++l) need += fbits(box_point(c.sub32[(size_t)b * 2 + sgm], q[l])) <= thr[l];
			k.h_total += 1;
			k.h_zero += need == 0;
		}
	}
	// ---- the shipped structure ---------------------------------------------------------------------------------------
	void run_current(int nbr, int transpose_below) {
		phase = 0;
		own_bucket();
		phase = 1;
		for (uint32_t b : nbr_list(nbr)) {
			quarter_stats(b);
			{
				int need = 0;
				for (int l = 0; l < 64; ++l) need += fbits(box_point(c.bvh[c.P + b], q[l])) <= thr[l];
				k.nbr_total += 1;
				k.nbr_need_hist[need == 0 ? 0 : (need < 12 ? 1 : (need < 24 ? 2 : (need < 48 ? 3 : 4)))] += 1;
			}
			scan_tile(b, ~0ull, 0);
			k.buckets += 1;
			tighten();
		}
		if (any_pending(0)) flush();
		phase = 2;
		std::vector<uint32_t> list = bfs(nbr);
		if (list_order == 1) {
			const Box qb = box_of(q, 64);
			std::stable_sort(list.begin(), list.end(), [&](uint32_t x, uint32_t y) { return box_box(c.bvh[c.P + x], qb) < box_box(c.bvh[c.P + y], qb); });
		} else if (list_order == 2) {
			std::vector<std::pair<int, uint32_t>> v;
			for (uint32_t leaf : list) {
				int need = 0;
				for (int l = 0; l < 64; ++l) need += fbits(box_point(c.bvh[c.P + leaf], q[l])) <= thr[l];
				v.push_back({-need, leaf});
			}
			std::stable_sort(v.begin(), v.end(), [](const std::pair<int, uint32_t> &x, const std::pair<int, uint32_t> &y) { return x.first < y.first; });
			for (size_t i = 0; i < v.size(); ++i) list[i] = v[i].second;
		}
		size_t list_pos = 0;
		for (uint32_t leaf : list) {
			const size_t my_pos = list_pos++;
			const uint64_t need = need_mask(c.bvh[c.P + leaf]);
			if (!need) continue;
			k.buckets += 1;
			k.lanes_need_sum += __builtin_popcountll(need);
			if (__builtin_popcountll(need) < transpose_below) {
				k.small_scans += 1;
				uint64_t full = 0;
				for (int l = 0; l < 64; ++l) {
					if (!((need >> l) & 1)) continue;
					k.tq += 1;
					uint32_t pass[64];
					int np = 0;
					for (int j = 0; j < 64; ++j) {
						const uint32_t pos = leaf * 64 + j;
						if (pos >= c.n) continue;
						if (fbits(dist2(c.pts[pos], q[l])) <= thr[l]) pass[np++] = pos;
					}
					if (!np) continue;
					k.tq_pass += 1;
					if ((size_t)np > (pair_xor ? 2 * sbuf - buf[l].size() - buf[l ^ pair_xor].size() : sbuf - buf[l].size())) {
						full |= 1ull << l;
						continue;
					}
					for (int i = 0; i < np; ++i) offer(l, pass[i]);
				}
				if (full) scan_tile(leaf, full, 0);
			} else {
				k.full_scans += 1;
				quarter_stats(leaf);
				scan_tile(leaf, ~0ull, 0);
				// statistics: would the NEXT listed bucket that gets a wave-shared scan have had a disjoint (or nearly disjoint) mask?
				k.pd_full += 1;
				for (size_t jj = my_pos + 1; jj < list.size(); ++jj) {
					uint64_t m2 = 0;
					for (int l = 0; l < 64; ++l)
						if (fbits(box_point(c.bvh[c.P + list[jj]], q[l])) <= thr[l]) m2 |= 1ull << l;
					if (__builtin_popcountll(m2) < transpose_below) continue;
					if ((m2 & need) == 0) k.pd_paired += 1;
					else if (__builtin_popcountll(m2 & need) <= 4) k.pd_inter_small += 1;
					break;
				}
			}
			tighten();
		}
		if (any_pending(0)) flush();
		k.waves += 1;
		k.max_lane_appended += *std::max_element(lane_app, lane_app + 64);
	}
	// ---- per-lane iteration over sub-buckets resident in LDS --------------------------------------------------------------
	// SUB: points per sub-bucket (16 / 32 / 64); C: listed buckets staged at a time; nbr: neighbours scanned wave-shared before
	// the BFS (the rest of them go to the head of the list); lazy: re-test a sub-box with the threshold of the moment before
	// scanning it
	void run_per_lane(int SUB, int C, int nbr_pre, int nbr_total, bool lazy, bool own_per_lane) {
		own_bucket();
		std::vector<uint32_t> head;
		{
			const std::vector<uint32_t> nl = nbr_list(nbr_total);
			for (size_t i = 0; i < nl.size(); ++i) {
				if ((int)i < nbr_pre) {
					scan_tile(nl[i], ~0ull, 0);
					k.buckets += 1;
				} else
					head.push_back(nl[i]);
			}
		}
		(void)own_per_lane;
		if (any_pending(0)) flush();
		std::vector<uint32_t> list = bfs(nbr_total);
		list.insert(list.begin(), head.begin(), head.end());
		const int S = 64 / SUB;
		const std::vector<Box> &sb = SUB == 16 ? c.sub16 : c.sub32;
		for (size_t g0 = 0; g0 < list.size(); g0 += (size_t)C) {
			const size_t g1 = std::min(list.size(), g0 + (size_t)C);
			// per-lane queues of (bucket, sub) in list order
			std::vector<uint32_t> todo[64];
			for (size_t j = g0; j < g1; ++j) {
				const uint32_t leaf = list[j];
				const uint64_t need = need_mask(c.bvh[c.P + leaf]);
				if (!need) continue;
				k.buckets += 1;
				k.lanes_need_sum += __builtin_popcountll(need);
				for (int s = 0; s < S; ++s) {
					if (S == 1) {
						for (int l = 0; l < 64; ++l)
							if ((need >> l) & 1) todo[l].push_back(leaf * 4u);
						break;
					}
					if (leaf * 64u + (uint32_t)(s * SUB) >= c.n) continue;
					const Box &bx = sb[(size_t)leaf * S + s];
					k.subtests += 1;
					for (int l = 0; l < 64; ++l)
						if (((need >> l) & 1) && fbits(box_point(bx, q[l])) <= thr[l]) todo[l].push_back(leaf * 4u + (uint32_t)(s * (SUB / 16)));
				}
			}
			size_t cursor[64] = {};
			for (;;) {
				// every lane takes its next sub-bucket (lazy: the next one that still passes its box test; each skipped one costs the
				// wave a re-test step in which the other lanes wait)
				uint32_t cur[64];
				bool active[64];
				bool any = false;
				int retest_steps = 0;
				for (int l = 0; l < 64; ++l) {
					active[l] = false;
					int tries = 0;
					while (cursor[l] < todo[l].size()) {
						const uint32_t code = todo[l][cursor[l]++];
						if (lazy && S > 1) {
							++tries;
							const uint32_t leaf = code >> 2, s16 = code & 3u;
							const Box &bx = sb[(size_t)leaf * S + s16 / (uint32_t)(SUB / 16)];
							if (!(fbits(box_point(bx, q[l])) <= thr[l])) continue;
						}
						cur[l] = code;
						active[l] = true;
						break;
					}
					retest_steps = std::max(retest_steps, tries);
					any = any || active[l];
				}
				k.retests += retest_steps;
				if (!any) break;
				k.rounds += 1;
				for (int i0 = 0; i0 < SUB; i0 += 4) {
					if (any_pending(SBUF - 4)) flush();
					for (int i = i0; i < i0 + 4; ++i)
						for (int l = 0; l < 64; ++l) {
							if (!active[l]) {
								k.idle_lane_steps += 1;
								continue;
							}
							const uint32_t pos = (cur[l] >> 2) * 64u + (cur[l] & 3u) * 16u + (uint32_t)i;
							if (pos < c.n) {
								offer(l, pos);
								k.evals_useful += 1;
							}
						}
					k.steps += 4;
				}
			}
		}
		if (any_pending(0)) flush();
		k.waves += 1;
	}
};

int main(int argc, char **argv) {
	if (argc < 2) {
		fprintf(stderr, "usage: knn_sim cloud.f32 [stride]\n");
		return 1;
	}
	Cloud c;
	{
		FILE *f = fopen(argv[1], "rb");
		if (!f) return 1;
		fseek(f, 0, SEEK_END);
		const long bytes = ftell(f);
		fseek(f, 0, SEEK_SET);
		c.pts.resize((size_t)bytes / 12);
		if (fread(c.pts.data(), 12, c.pts.size(), f) != c.pts.size()) return 1;
		fclose(f);
	}
	const uint32_t kd_block = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
	if (kd_block >= 128) {
		// buckets as the leaves of a small k-d tree over every block of kd_block curve-consecutive points instead of runs of the curve
		struct R { static void split(P3 *p, size_t n) {
			if (n <= 64) return;
			Box b = box_of(p, (int)n);
			int ax = 0; float ext = -1;
			for (int a = 0; a < 3; ++a) if (b.hi[a] - b.lo[a] > ext) { ext = b.hi[a] - b.lo[a]; ax = a; }
			const size_t half = (n / 2 + 63) / 64 * 64 == n ? n / 2 : (n / 128) * 64; // whole buckets on the left
			const size_t h = half ? half : 64;
			std::nth_element(p, p + h, p + n, [ax](const P3 &x, const P3 &y) { return (&x.x)[ax] < (&y.x)[ax]; });
			split(p, h); split(p + h, n - h);
		} };
		for (size_t i = 0; i + 64 <= c.pts.size(); i += kd_block) R::split(&c.pts[i], std::min<size_t>(kd_block, c.pts.size() - i));
	}
	c.build();
	const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 4;
	struct Variant {
		std::string name;
		int kind, SUB, C, nbr_pre, nbr_total, lazy, tb;
		int pair_xor = 0, sbuf = SBUF, tighten = 0, order = 0;
	};
	std::vector<Variant> vs = {
	    {"cur nbr4 tb20", 0, 0, 0, 4, 4, 0, 20},
	    {"cur tb12", 0, 0, 0, 4, 4, 0, 12},
	    {"cur tb12 listed nearest first", 0, 0, 0, 4, 4, 0, 12, 0, 16, 0, 1},
	    {"cur tb12 listed most-needed first", 0, 0, 0, 4, 4, 0, 12, 0, 16, 0, 2},
	    {"near nbr2 tb12", 0, 0, 0, 2, 2, 0, 12, 0, 16, 0, 1},
	    {"near nbr0 tb12", 0, 0, 0, 0, 0, 0, 12, 0, 16, 0, 1},
	    {"near nbr6 tb12", 0, 0, 0, 6, 6, 0, 12, 0, 16, 0, 1},
	    {"near nbr4 tb8", 0, 0, 0, 4, 4, 0, 8, 0, 16, 0, 1},
	    {"near nbr4 tb16", 0, 0, 0, 4, 4, 0, 16, 0, 16, 0, 1},
	    {"near nbr2 tb16", 0, 0, 0, 2, 2, 0, 16, 0, 16, 0, 1},
	    {"cur tighten max(L[31-m],pmax)", 0, 0, 0, 4, 4, 0, 20, 0, 16, 1},
	    {"cur tighten exact per bucket", 0, 0, 0, 4, 4, 0, 20, 0, 16, 2},
	    {"cur pair l^32", 0, 0, 0, 4, 4, 0, 20, 32, 16},
	    {"cur pair l^1", 0, 0, 0, 4, 4, 0, 20, 1, 16},
	    {"cur pair l^16", 0, 0, 0, 4, 4, 0, 20, 16, 16},
	    {"cur sbuf 24", 0, 0, 0, 4, 4, 0, 20, 0, 24},
	    {"cur sbuf 32", 0, 0, 0, 4, 4, 0, 20, 0, 32},
	    {"cur sbuf 64", 0, 0, 0, 4, 4, 0, 20, 0, 64},
	    {"pl sub64 C=all nbr4", 1, 64, 100000, 4, 4, 0, 0},
	    {"pl sub32 C=all nbr4", 1, 32, 100000, 4, 4, 0, 0},
	    {"pl sub16 C=all nbr4", 1, 16, 100000, 4, 4, 0, 0},
	    {"pl sub16 C=all nbr4 lazy", 1, 16, 100000, 4, 4, 1, 0},
	    {"pl sub16 C=all nbr0/4 lazy", 1, 16, 100000, 0, 4, 1, 0},
	    {"pl sub16 C=all nbr2/4 lazy", 1, 16, 100000, 2, 4, 1, 0},
	    {"pl sub16 C=16 nbr0/4 lazy", 1, 16, 16, 0, 4, 1, 0},
	    {"pl sub16 C=8 nbr0/4 lazy", 1, 16, 8, 0, 4, 1, 0},
	    {"pl sub16 C=4 nbr0/4 lazy", 1, 16, 4, 0, 4, 1, 0},
	    {"pl sub32 C=8 nbr0/4 lazy", 1, 32, 8, 0, 4, 1, 0},
	    {"pl sub32 C=all nbr0/4 lazy", 1, 32, 100000, 0, 4, 1, 0},
	};
	printf("%-30s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s %7s | %8s\n", "variant", "bkts", "listed", "steps", "flush", "append", "tq", "boxt", "subt", "rounds", "retest", "idle%", "VALU~");
	const bool only_first = getenv("SIM_ONLY_CUR") != nullptr;
	for (const Variant &v : vs) {
		if (only_first && (&v - &vs[0]) > 9) break;
		Counters total;
#pragma omp parallel
		{
			Counters k;
#pragma omp for schedule(dynamic, 16)
			for (uint32_t b = 2; b < c.nb - 3; b += stride) {
				Wave w(c, k, b);
				w.pair_xor = v.pair_xor;
				w.sbuf = v.sbuf;
				w.tighten_mode = v.tighten;
				w.list_order = v.order;
				if (v.kind == 0)
					w.run_current(v.nbr_total, v.tb);
				else
					w.run_per_lane(v.SUB, v.C, v.nbr_pre, v.nbr_total, v.lazy != 0, false);
			}
#pragma omp critical
			total.add(k);
		}
		const double W = total.waves;
		// instruction model per wave (VALU): a candidate step 8.75 (+2 for a per-lane address), a flush 300, a box test 16, a transposed
		// (query, bucket) pair 22 (+14 when something passes), a BFS batch 80, a per-lane round 30, a re-test step 20; fixed: initial sort
		// 382, own first-32 128, epilogue 1000
		const double step_cost = v.kind == 0 ? 8.75 : 10.75;
		const double own_steps = 32 + (v.kind == 0 ? 0 : 0);
		(void)own_steps;
		double valu = 382 + 128 + 1000 + total.flushes / W * 300 + total.boxtests / W * 16 + total.subtests / W * 16 + total.tq / W * 22 + total.tq_pass / W * 14 +
		              total.bfs_batches / W * 80 + total.rounds / W * 30 + total.retests / W * 20;
		// steps: broadcast scans cost 8.75, per-lane steps 10.75: the own bucket and prescanned neighbours are broadcast in both
		const double bsteps = v.kind == 0 ? total.steps / W : (32.0 + 64.0 * v.nbr_pre);
		const double psteps = total.steps / W - bsteps;
		valu += bsteps * 8.75 + (v.kind == 0 ? 0.0 : psteps * step_cost);
		printf("%-30s %7.2f %7.2f %7.1f %7.2f %7.1f %7.1f %7.1f %7.1f %7.2f %7.2f %7.1f | %8.0f\n", v.name.c_str(), total.buckets / W, total.listed / W, total.steps / W,
		       total.flushes / W, total.appended / W / 64.0, total.tq / W, total.boxtests / W, total.subtests / W, total.rounds / W, total.retests / W,
		       total.steps > 0 ? 100.0 * total.idle_lane_steps / (total.steps * 64.0) : 0.0, valu);
		if (v.kind == 0) printf("   flush compare-exchanges per wave %.0f (wide flushes %.2f) | flushes in which no lane gets an entry below its 16th: %.2f, below its 8th: %.2f; lanes per flush with one below the 16th %.1f\n", total.flush_ce / W, total.flush_wide / W, total.flush_lo_untouched / W, total.flush_lo8_untouched / W, total.lanes_lo_touched / total.flushes);
		if (v.kind == 0 && v.pair_xor == 0 && v.sbuf == SBUF && v.tighten == 0)
			printf("   neighbour buckets by needing lanes: none %.1f%%, 1-11 %.1f%%, 12-23 %.1f%%, 24-47 %.1f%%, 48-64 %.1f%%\n", 100 * total.nbr_need_hist[0] / total.nbr_total,
			       100 * total.nbr_need_hist[1] / total.nbr_total, 100 * total.nbr_need_hist[2] / total.nbr_total, 100 * total.nbr_need_hist[3] / total.nbr_total, 100 * total.nbr_need_hist[4] / total.nbr_total);
		if (v.kind == 0 && v.pair_xor == 0 && v.sbuf == SBUF && v.tighten == 0)
			printf("   quarters walked per wave-shared bucket: any lane %.2f | busiest of 2 groups of 32 lanes %.2f | of 4 groups of 16 %.2f (mean group %.2f) | of 8 groups of 8 %.2f | buckets per wave %.2f\n",
			       total.gq_any / total.gq_buckets, total.gq_max2 / total.gq_buckets, total.gq_max4 / total.gq_buckets, total.gq_mean4 / total.gq_buckets, total.gq_max8 / total.gq_buckets, total.gq_buckets / W);
		if (v.kind == 0 && v.pair_xor == 0 && v.sbuf == SBUF && v.tighten == 0)
			printf("   listed wave-shared scans per wave %.2f; the next one has a disjoint mask in %.1f%%, an overlap of <= 4 lanes in another %.1f%%\n", total.pd_full / W, 100.0 * total.pd_paired / total.pd_full, 100.0 * total.pd_inter_small / total.pd_full);
		if (v.kind == 0 && v.pair_xor == 0 && v.sbuf == SBUF) {
			printf("   max-lane appended per wave %.1f | flushes by phase own %.2f nbr %.2f list %.2f | lanes with > 8 pending at a flush %.2f | fill histogram:", total.max_lane_appended / W,
			       total.flush_phase[0] / W, total.flush_phase[1] / W, total.flush_phase[2] / W, total.lanes_over8_at_flush / total.flushes);
			for (int i = 0; i <= 16; ++i) printf(" %.1f%%", 100.0 * total.fill_hist[i] / (total.flushes * 64));
			printf("\n   lane-parallel scanned buckets (nbrs + listed): quarters %.2f per wave, needed by no lane %.1f%%, by 1-7 lanes %.1f%%, mean need %.1f | halves needed by no lane %.1f%%\n",
			       total.q_total / W, 100.0 * total.q_zero / total.q_total, 100.0 * total.q_low / total.q_total, total.q_need_sum / total.q_total, 100.0 * total.h_zero / total.h_total);
		}
	}
	return 0;
}
