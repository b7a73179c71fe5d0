// gather_rate.hip -- what does one wave-level 16-byte gather cost when every lane reads a different cache line?
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; run on the MI355X box.
// Each lane walks its own 256-byte block (a 16-point bucket of float4 records) of a region that fits the vector L1 / the L2;
// 12 waves per CU as the kNN kernel has.  Prints cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(64) void k(const float4 *__restrict__ p, unsigned nblocks, float *out, int iters) {
	const unsigned lane = threadIdx.x;
	unsigned b = (blockIdx.x * 64u + lane) * 2654435761u;
	float acc = 0.0f;
	for (int it = 0; it < iters; ++it) {
		b = b * 1664525u + 1013904223u;
		// MODE 0: every lane its own block; MODE 1: all lanes the same block (coalesced: the wave reads 1 KiB = 4 blocks);
		// MODE 2: groups of 16 lanes share a block (4 blocks per instruction)
		unsigned blk = MODE == 0 ? (b >> 8) % nblocks : (MODE == 1 ? (__builtin_amdgcn_readfirstlane(b) >> 8) % nblocks : (__shfl(b, lane & 48u, 64) >> 8) % nblocks);
		const float4 *q = p + (size_t)blk * 16u;
#pragma unroll
		for (int c = 0; c < 16; ++c) {
			const float4 v = MODE == 1 ? q[(c * 64 + lane) & 15u] : q[c];
			acc += v.x * v.y + v.z;
		}
	}
	out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE> void run(const char *name, const float4 *d, unsigned nblocks, float *o, int iters) {
	const int grid = 256 * 12 * 8;
	hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, nblocks, o, 2);
	hipDeviceSynchronize();
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipEventRecord(a);
	hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, nblocks, o, iters);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double instr_per_cu = (double)grid / 256.0 * iters * 16.0;
	printf("%-44s region %7.1f KiB: %8.3f ms, %7.1f ns per wave-load per CU (~%.0f cycles at 2.4 GHz)\n", name, nblocks * 256.0 / 1024.0, ms, ms * 1e6 / instr_per_cu,
	       ms * 1e6 / instr_per_cu * 2.4);
}

int main() {
	const unsigned maxblocks = 1u << 20; // 256 MiB
	float4 *d;
	float *o;
	hipMalloc(&d, (size_t)maxblocks * 256);
	hipMalloc(&o, 256 * 12 * 8 * 64 * 4);
	hipMemset(d, 0, (size_t)maxblocks * 256);
	for (unsigned nb : {64u, 4096u, 65536u, 1u << 20}) {
		run<0>("every lane its own 256-B block", d, nb, o, 200);
		run<2>("16 lanes per block (4 blocks / instruction)", d, nb, o, 200);
		run<1>("all lanes one block (coalesced)", d, nb, o, 200);
	}
	return 0;
}
