// Host instantiation of the library's own Hilbert index (treee_amd/csrc/common.h: the same source the device code
// compiles) printed for a deterministic set of cells; tests/test_oracle.py compares the lines with the oracle's
// orc_hilbert_keys.  Test infrastructure: built by the CPU test suite with hipcc, needs no GPU.
#include "../../treee_amd/csrc/common.h"

#include <cstdio>

static uint32_t morton_of(uint32_t x, uint32_t y, uint32_t z) { // bit b of x -> bit 3 b, y -> 3 b + 1, z -> 3 b + 2 (radix_sort.hip)
	uint32_t m = 0;
	for (int b = 0; b < 10; ++b) m |= (((x >> b) & 1u) << (3 * b)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b + 2));
	return m;
}

int main() {
	// all cells of the 16^3 grid of every 64th coordinate, then a SplitMix64 stream of arbitrary cells
	for (uint32_t z = 0; z < 1024; z += 64)
		for (uint32_t y = 0; y < 1024; y += 64)
			for (uint32_t x = 0; x < 1024; x += 64) {
				const uint32_t m = morton_of(x, y, z);
				std::printf("%u %u %u %u %u\n", x, y, z, treee::hilbert30(x, y, z), treee::hilbert_of_morton(m));
			}
	unsigned long long s = 0x7EEE0001ull;
	for (int i = 0; i < 20000; ++i) {
		s += 0x9E3779B97F4A7C15ull;
		unsigned long long v = s;
		v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
		v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
		v ^= v >> 31;
		const uint32_t x = (uint32_t)(v & 1023u), y = (uint32_t)((v >> 10) & 1023u), z = (uint32_t)((v >> 20) & 1023u);
		const uint32_t m = morton_of(x, y, z);
		std::printf("%u %u %u %u %u\n", x, y, z, treee::hilbert30(x, y, z), treee::hilbert_of_morton(m));
	}
	return 0;
}
