// instr_rate.hip -- issue cost of the instructions the kNN kernel is made of, measured on the GPU it runs on.
// Each test runs a loop of 64 independent copies of one instruction (8 rotating destination registers) and reports
// cycles per wave-instruction per SIMD at 1..4 waves per SIMD (s_memtime around the loop, median over waves).
// Build: hipcc --offload-arch=gfx950 -O3 -o instr_rate instr_rate.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
			exit(1);                                                                  \
		}                                                                             \
	} while (0)

constexpr int ITER = 256; // loop trips; 64 instructions per trip

// 8 independent chains, 8 instructions each per trip
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define TRIP(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int T> __global__ __launch_bounds__(1024) void k_rate(unsigned long long *out, float seed) {
	extern __shared__ char dyn[];
	__shared__ float lds[2048];
	const int lane = threadIdx.x & 63;
	float f[8], g[8];
	double d[8], e[8];
	unsigned u[8], w[8];
	typedef float v2f __attribute__((ext_vector_type(2)));
	typedef float v16f __attribute__((ext_vector_type(16)));
	typedef float v4f __attribute__((ext_vector_type(4)));
	typedef short v8s __attribute__((ext_vector_type(8)));
	v2f p[8], q[8];
	v16f acc = {0};
	v4f a4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
	v8s fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 2, 2, 2, 2};
	for (int i = 0; i < 8; ++i) {
		f[i] = seed + i + lane;
		g[i] = seed * 0.5f + i;
		d[i] = (double)f[i];
		e[i] = (double)g[i];
		u[i] = (unsigned)(lane * 7 + i);
		w[i] = (unsigned)(lane + i * 3);
		p[i] = v2f{f[i], g[i]};
		q[i] = v2f{g[i], f[i]};
	}
	for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (float)i;
	__syncthreads();
	unsigned ofs = lane * 4 + (threadIdx.x >> 6) * 512;
	unsigned ofs8 = (ofs * 2u) & 0x1FF8u, tmpv = 0u, c256 = 256u;
	asm volatile("" : "+v"(ofs8), "+v"(tmpv), "+v"(c256));
	unsigned long long cond = __ballot(f[0] > 20.0f);
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int it = 0; it < ITER; ++it) {
		if constexpr (T == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 1) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 3) {
#define X(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 4) {
#define X(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 5) { // 64-bit unsigned compare + 2 selects (half a compare-exchange)
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : "+v"(d[i]) : "v"(e[i]), "v"(u[i]), "v"(w[i]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 6) {
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 7) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(w[i]), "s"(cond));
			TRIP(X)
#undef X
		} else if constexpr (T == 8) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 9) {
#define X(i) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(u[i]), "v"(w[i]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 10) { // branch-free append as the kernel does it: 2 stores, compare, select, add
#define X(i) asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:16" : : "v"(ofs), "v"(u[i]), "v"(w[i]) : "memory");
			TRIP(X)
#undef X
		} else if constexpr (T == 11) {
#define X(i) asm volatile("ds_write_b32 %0, %1" : : "v"(ofs), "v"(u[i]) : "memory");
			TRIP(X)
#undef X
		} else if constexpr (T == 70) { // one 8-byte store per lane, lane stride 8 (an interleaved (position, distance) entry)
#define X(i) asm volatile("ds_write_b64 %0, %1" : : "v"(ofs8), "v"(d[i]) : "memory");
			TRIP(X)
#undef X
		} else if constexpr (T == 71) { // per-lane 8-byte read, lane stride 8
#define X(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(ofs8) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if constexpr (T == 72) { // per-lane two-address read (the flush's read of one buffer slot today)
#define X(i) asm volatile("ds_read2st64_b32 %0, %1 offset1:16" : "=v"(d[i]) : "v"(ofs) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if constexpr (T == 73) { // per-lane 4-byte read, lane stride 4
#define X(i) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(ofs) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if constexpr (T == 74) { // v_cmpx + add under the new exec + exec restore: the conditional advance of the append
#define X(i) asm volatile("v_cmpx_le_u32 %1, %2\n\tv_add_u32 %0, 0x100, %0\n\ts_mov_b64 exec, -1" : "+v"(u[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
			TRIP(X)
#undef X
		} else if constexpr (T == 75) { // the same advance as compare + select + add
#define X(i) asm volatile("v_cmp_le_u32 vcc, %1, %2\n\tv_cndmask_b32 %3, 0, %4, vcc\n\tv_add_u32 %0, %3, %0" : "+v"(u[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]), "v"(tmpv), "v"(c256) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 12) { // broadcast read, 8 bytes
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(0u), "n"(i * 8) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if constexpr (T == 13) {
#define X(i) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(u[i]) : "s20");
			TRIP(X)
#undef X
		} else if constexpr (T == 14) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 15) { // 64 MFMAs per trip
#define X(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
			TRIP(X)
#undef X
		} else if constexpr (T == 16) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 17) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 18) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 19) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 20) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g[i]), "v"(g[(i + 1) & 7]));
			TRIP(X)
#undef X
		} else if constexpr (T == 21) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g[i]), "v"(g[(i + 1) & 7]));
			TRIP(X)
#undef X
		} else if constexpr (T == 22) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 23) {
#define X(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 24) { // dual 32-bit compare-exchange on unsigned keys
#define X(i) asm volatile("v_min_u32 %0, %1, %2\n\tv_max_u32 %1, %1, %2" : "=&v"(u[(i + 1) & 7]), "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 25) {
#define X(i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(f[i]), "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 26) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a4[i & 1]) : "v"(0u), "n"(i * 16) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if constexpr (T == 27) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 28) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 29) {
#define X(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(f[i]), "v"(u[i]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 30) { // mbcnt pair
#define X(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 31) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(f[i]), "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 32) {
#define X(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[i & 1]) : "v"(fa), "v"(fb));
			TRIP(X)
#undef X
		} else if constexpr (T >= 60 && T <= 65) { // ds_write2st64_b32 under an exec mask
			constexpr unsigned long long M = T == 60 ? ~0ull : T == 61 ? 1ull : T == 62 ? 0xFFFFull : T == 63 ? 0x1111111111111111ull : T == 64 ? 0x0001000100010001ull : 0ull;
			const unsigned long long save = __builtin_amdgcn_read_exec();
			asm volatile("s_mov_b64 exec, %0" : : "s"(M));
#define X(i) asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:16" : : "v"(ofs), "v"(u[i]), "v"(w[i]) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_mov_b64 exec, %0" : : "s"(save));
		} else if constexpr (T == 33) { // full 64-bit compare-exchange without f64: compare + 4 selects
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc" : "+v"(d[i]) : "v"(e[i]), "v"(u[i]), "v"(w[i]), "v"(f[i]), "v"(g[i]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 34) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[i]) : "v"(ofs), "v"(w[i]) : "memory");
			TRIP(X)
#undef X
			asm volatile("s_waitcnt lgkmcnt(0)");
		} else if constexpr (T == 35) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 36) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 37) {
#define X(i) asm volatile("v_cmp_le_u32 %0, %1, %2" : "=s"(cond) : "v"(u[i]), "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 38) { // compare + add-with-carry (count passers)
#define X(i) asm volatile("v_cmp_le_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]) : "vcc");
			TRIP(X)
#undef X
		} else if constexpr (T == 39) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 40) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 41) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %0" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 42) {
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 43) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 44) {
#define X(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 45) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 46) {
#define X(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 47) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 48) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 49) {
#define X(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 50) {
#define X(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
			TRIP(X)
#undef X
		} else if constexpr (T == 51) {
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(w[i]), "v"(w[(i+1)&7]));
			TRIP(X)
#undef X
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
	double sd = 0;
	unsigned su = 0;
	for (int i = 0; i < 8; ++i) {
		s += f[i] + p[i].x + p[i].y;
		sd += d[i];
		su += u[i] + w[i];
	}
	for (int i = 0; i < 16; ++i) s += acc[i];
	s += a4[0].x + a4[1].y + a4[0].z + a4[1].w;
	if (s == 12345.678f && sd == 1.0 && su == 77u) out[1] = 1; // keep everything alive
	if (lane == 0) {
		unsigned hw;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		out[2 + blockIdx.x * 16 + (threadIdx.x >> 6)] = ((t1 - t0) << 4) | ((hw >> 4) & 3u);
	}
}

template <int T> void run(const char *name, unsigned long long *d_out, int per_instr_mult = 1) {
	printf("%-44s", name);
	std::vector<double> byn[17];
	// one workgroup of 16 waves per CU (> 80 KiB of LDS): the hardware does not spread them evenly over the 4 SIMDs,
	// so every wave records its SIMD (HW_REG_HW_ID) and the result is grouped by the number n of waves that shared
	// a SIMD: cycles per instruction per SIMD = (cycles a wave saw per instruction) / n
	for (int waves : {4, 8, 12, 16}) {
		const size_t lds = 96 * 1024;
		const int grid = 256;
		CHECK(hipFuncSetAttribute((const void *)k_rate<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL(k_rate<T>, dim3(grid), dim3(64 * waves), lds, 0, d_out, 1.0f);
		hipLaunchKernelGGL(k_rate<T>, dim3(grid), dim3(64 * waves), lds, 0, d_out, 1.0f);
		CHECK(hipDeviceSynchronize());
		std::vector<unsigned long long> h(2 + grid * 16);
		CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
		for (int b = 0; b < grid; ++b) {
			int cnt[4] = {0, 0, 0, 0};
			for (int w = 0; w < waves; ++w) cnt[h[2 + b * 16 + w] & 3]++;
			for (int w = 0; w < waves; ++w) {
				const int n = cnt[h[2 + b * 16 + w] & 3];
				byn[n].push_back((double)(h[2 + b * 16 + w] >> 4) / (ITER * 64.0 * per_instr_mult));
			}
		}
	}
	for (int n = 1; n <= 6; ++n) {
		if (byn[n].size() < 16) continue;
		std::sort(byn[n].begin(), byn[n].end());
		const double per = byn[n][byn[n].size() / 2];
		printf("  n%d: %6.2f (simd %5.2f)", n, per, per / n);
	}
	printf("\n");
}

int main() {
	unsigned long long *d_out;
	CHECK(hipMalloc(&d_out, (2 + 256 * 16) * 8));
	CHECK(hipMemset(d_out, 0, (2 + 256 * 16) * 8));
	printf("cycles per wave-instruction seen by one wave (and per SIMD = /waves) grouped by n = waves sharing the SIMD\n");
	run<0>("v_add_f32", d_out);
	run<16>("v_mul_f32", d_out);
	run<28>("v_sub_f32", d_out);
	run<17>("v_fma_f32", d_out);
	run<1>("v_pk_add_f32", d_out);
	run<2>("v_pk_mul_f32", d_out);
	run<27>("v_pk_fma_f32", d_out);
	run<3>("v_min_f64", d_out);
	run<4>("v_max_f64", d_out);
	run<23>("v_cmp_lt_f64", d_out);
	run<6>("v_cmp_lt_u64", d_out);
	run<5>("v_cmp_lt_u64 + v_cndmask (2 instr)", d_out, 2);
	run<7>("v_cndmask_b32", d_out);
	run<8>("v_min_u32", d_out);
	run<24>("v_min_u32 + v_max_u32 (2 instr)", d_out, 2);
	run<9>("v_cmp_le_u32", d_out);
	run<18>("v_add_u32", d_out);
	run<19>("v_lshl_add_u32", d_out);
	run<20>("v_max3_f32", d_out);
	run<21>("v_med3_f32", d_out);
	run<29>("v_cmp_class_f32", d_out);
	run<22>("v_mov_b32_dpp row_shr:1", d_out);
	run<13>("v_readlane_b32", d_out);
	run<14>("v_permlane32_swap_b32", d_out);
	run<30>("v_mbcnt_lo + v_mbcnt_hi (2 instr)", d_out, 2);
	run<31>("v_cvt_pk_bf16_f32", d_out);
	run<33>("v_cmp_lt_u64 + 4 v_cndmask (5 instr)", d_out, 5);
	run<38>("v_cmp_le_u32 + v_addc_co_u32 (2 instr)", d_out, 2);
	run<37>("v_cmp_le_u32 -> sgpr pair", d_out);
	run<35>("v_and_b32", d_out);
	run<42>("v_or_b32", d_out);
	run<48>("v_xor_b32", d_out);
	run<36>("v_lshlrev_b32", d_out);
	run<43>("v_bfe_u32", d_out);
	run<39>("v_max_f32", d_out);
	run<45>("v_min_f32", d_out);
	run<44>("v_max_u32", d_out);
	run<46>("v_min_i32", d_out);
	run<51>("v_min3_u32", d_out);
	run<50>("v_pk_min_u16", d_out);
	run<49>("v_sub_u32", d_out);
	run<40>("v_mov_b32", d_out);
	run<47>("v_mul_f64", d_out);
	run<41>("v_sub_f32 + v_mul_f32 dependent (2 instr)", d_out, 2);
	run<34>("ds_bpermute_b32", d_out);
	run<60>("ds_write2st64_b32 exec=all", d_out);
	run<61>("ds_write2st64_b32 exec=1 lane", d_out);
	run<62>("ds_write2st64_b32 exec=lanes 0-15", d_out);
	run<63>("ds_write2st64_b32 exec=every 4th lane", d_out);
	run<64>("ds_write2st64_b32 exec=every 16th lane", d_out);
	run<65>("ds_write2st64_b32 exec=0", d_out);
	run<70>("ds_write_b64 (lane stride 8)", d_out);
	run<71>("ds_read_b64 (lane stride 8)", d_out);
	run<72>("ds_read2st64_b32 (per lane)", d_out);
	run<73>("ds_read_b32 (per lane)", d_out);
	run<74>("v_cmpx + v_add + s_mov exec (3 instr)", d_out, 3);
	run<75>("v_cmp + v_cndmask + v_add (3 instr)", d_out, 3);
	run<10>("ds_write2st64_b32", d_out);
	run<11>("ds_write_b32", d_out);
	run<12>("ds_read_b64 (broadcast)", d_out);
	run<26>("ds_read_b128 (broadcast)", d_out);
	run<15>("v_mfma_f32_32x32x16_bf16", d_out);
	run<32>("v_mfma_f32_16x16x32_bf16", d_out);
	run<25>("v_mfma_f32_32x32x2_f32", d_out);
	return 0;
}
