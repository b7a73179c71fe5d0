// ingest.hip -- LAS record decode and y-slice keys / binning.
//   decode   Chunk::read / next_point           treee/src/laz.rs:193-213
//   keys     floor(p.y) as isize                treee/src/loading.rs:99
//   binning  HashMap<isize, Vec<P3>> densified  treee/src/loading.rs:97-101,
//                                               treee/src/segmenting.rs:121-133
// The reference serialises binning behind one mutex; here it is a key + stable radix sort
// (radix_sort.hip).  "Arrival order" inside a slice = point index order (the deterministic
// sequential order the oracle uses).  i32 -> f64 -> f32 arithmetic uses only *, +, -, casts
// and is compiled with -ffp-contract=off: bit-exact against oracle/ingest_oracle.cpp.
#include "common.h"

using namespace treee;

namespace {

struct DecodeParams {
	double scale[3], offset[3], center[3];
};

__global__ __launch_bounds__(256) void k_decode_las(const uint8_t *__restrict__ rec, uint32_t n, uint32_t stride, int aligned4,
                                                     DecodeParams P, float *__restrict__ xyz) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint8_t *s = rec + (size_t)i * stride;
	int32_t v[3];
	if (aligned4) {
		const uint32_t *w = (const uint32_t *)s;
		v[0] = (int32_t)w[0];
		v[1] = (int32_t)w[1];
		v[2] = (int32_t)w[2];
	} else {
#pragma unroll
		for (int a = 0; a < 3; ++a)
			v[a] = (int32_t)((uint32_t)s[4 * a] | ((uint32_t)s[4 * a + 1] << 8) | ((uint32_t)s[4 * a + 2] << 16) | ((uint32_t)s[4 * a + 3] << 24));
	}
	const double vx = P.offset[0] + (double)v[0] * P.scale[0];
	const double vy = P.offset[1] + (double)v[1] * P.scale[1];
	const double vz = P.offset[2] + (double)v[2] * P.scale[2];
	xyz[3ull * i + 0] = (float)(vx - P.center[0]); // (v.x, v.z, -v.y) - center, `as f32`
	xyz[3ull * i + 1] = (float)(vz - P.center[1]);
	xyz[3ull * i + 2] = (float)(-vy - P.center[2]);
}

__device__ __forceinline__ long long floor_as_isize(float y) {
	const float f = floorf(y);
	if (f != f) return 0; // Rust `as`: NaN -> 0, saturating
	if (f >= 9223372036854775808.0f) return 0x7FFFFFFFFFFFFFFFll;
	if (f <= -9223372036854775808.0f) return (long long)0x8000000000000000ull;
	return (long long)f;
}

__global__ __launch_bounds__(256) void k_slice_keys(const float *__restrict__ xyz, uint32_t n, long long *__restrict__ keys) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) keys[i] = floor_as_isize(xyz[3ull * i + 1]);
}

__global__ __launch_bounds__(256) void k_minmax_i64(const long long *__restrict__ keys, uint32_t n, long long *__restrict__ mm) {
	long long mn = 0x7FFFFFFFFFFFFFFFll, mx = (long long)0x8000000000000000ull;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const long long k = keys[i];
		mn = k < mn ? k : mn;
		mx = k > mx ? k : mx;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const long long a = __shfl_down(mn, o, 64), b = __shfl_down(mx, o, 64);
		mn = a < mn ? a : mn;
		mx = b > mx ? b : mx;
	}
	if ((threadIdx.x & 63) == 0) {
		atomicMin(&mm[0], mn);
		atomicMax(&mm[1], mx);
	}
}

__global__ __launch_bounds__(256) void k_offset_keys(const long long *__restrict__ keys, uint32_t n, long long mn, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) out[i] = (uint32_t)(keys[i] - mn);
}

__global__ __launch_bounds__(256) void k_slice_offsets(const uint32_t *__restrict__ sorted_keys, uint32_t n, uint32_t layers,
                                                        unsigned long long *__restrict__ offsets) {
	const uint32_t l = blockIdx.x * 256u + threadIdx.x;
	if (l > layers) return;
	uint32_t lo = 0, hi = n;
	while (lo < hi) { // first position with key >= l
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (sorted_keys[mid] < l)
			lo = mid + 1;
		else
			hi = mid;
	}
	offsets[l] = lo;
}

} // namespace

extern "C" {

int32_t treee_decode_las_records_dev(treee_ctx *ctx, const uint8_t *d_rec, uint64_t n, uint32_t stride, const double scale[3],
                                     const double offset[3], const double center[3], float *d_xyz) {
	if (!ctx || !scale || !offset || !center || (n && (!d_rec || !d_xyz))) return set_err(TREEE_ERR_INVALID, "treee_decode_las_records_dev: null argument");
	if (stride < 12) return set_err(TREEE_ERR_INVALID, "point_data_record_length %u < 12", stride);
	if (n == 0) return TREEE_OK;
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large for one call");
	TREEE_HIP(hipSetDevice(ctx->device));
	DecodeParams P;
	for (int a = 0; a < 3; ++a) {
		P.scale[a] = scale[a];
		P.offset[a] = offset[a];
		P.center[a] = center[a];
	}
	const int aligned4 = (stride % 4 == 0) && (((uintptr_t)d_rec) % 4 == 0);
	hipLaunchKernelGGL(k_decode_las, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_rec, (uint32_t)n, stride, aligned4, P, d_xyz);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_decode_las_records(const uint8_t *rec, uint64_t n, uint32_t stride, const double scale[3], const double offset[3],
                                 const double center[3], float *xyz_out) {
	if (n == 0) return TREEE_OK;
	if (!rec || !xyz_out) return set_err(TREEE_ERR_INVALID, "treee_decode_las_records: null argument");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<uint8_t> d_rec;
	DevBuf<float> d_xyz;
	TREEE_TRY(d_rec.alloc(n * stride));
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_HIP(hipMemcpyAsync(d_rec.p, rec, n * stride, hipMemcpyHostToDevice, ctx->stream));
	TREEE_TRY(treee_decode_las_records_dev(ctx, d_rec.p, n, stride, scale, offset, center, d_xyz.p));
	TREEE_HIP(hipMemcpyAsync(xyz_out, d_xyz.p, n * 12, hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_slice_keys_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, int64_t *d_keys) {
	if (!ctx || (n && (!d_xyz || !d_keys))) return set_err(TREEE_ERR_INVALID, "treee_slice_keys_dev: null argument");
	if (n == 0) return TREEE_OK;
	if (n >= (1ull << 32)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large for one call");
	TREEE_HIP(hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_slice_keys, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, (long long *)d_keys);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_slice_keys(const float *xyz, uint64_t n, int64_t *key_out) {
	if (n == 0) return TREEE_OK;
	if (!xyz || !key_out) return set_err(TREEE_ERR_INVALID, "treee_slice_keys: null argument");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_xyz;
	DevBuf<int64_t> d_keys;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_TRY(d_keys.alloc(n));
	TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
	TREEE_TRY(treee_slice_keys_dev(ctx, d_xyz.p, n, d_keys.p));
	TREEE_HIP(hipMemcpyAsync(key_out, d_keys.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_slice_bin_dev(treee_ctx *ctx, const float *d_xyz, uint64_t n, int64_t *min_key_out, uint64_t *layers_out, uint32_t *d_order_out,
                            uint64_t *d_offsets_out, uint64_t offsets_cap) {
	if (!ctx || !layers_out) return set_err(TREEE_ERR_INVALID, "treee_slice_bin_dev: null argument");
	*layers_out = 0;
	if (n == 0) return TREEE_OK;
	if (!d_xyz) return set_err(TREEE_ERR_INVALID, "treee_slice_bin_dev: xyz is null");
	if (n >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large for one call");
	TREEE_HIP(hipSetDevice(ctx->device));
	TREEE_TRY(ctx->reset());
	hipStream_t st = ctx->stream;
	long long *d_keys = nullptr, *d_mm = nullptr;
	TREEE_TRY(ctx->alloc_t(n, &d_keys));
	TREEE_TRY(ctx->alloc_t(2, &d_mm));
	hipLaunchKernelGGL(k_slice_keys, dim3(div_up(n, 256)), dim3(256), 0, st, d_xyz, (uint32_t)n, d_keys);
	const long long init[2] = {0x7FFFFFFFFFFFFFFFll, (long long)0x8000000000000000ull};
	TREEE_HIP(hipMemcpyAsync(d_mm, init, 16, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_minmax_i64, dim3(n < 262144 ? div_up(n, 256) : 1024), dim3(256), 0, st, d_keys, (uint32_t)n, d_mm);
	TREEE_LAUNCH_CHECK();
	long long mm[2];
	TREEE_HIP(hipMemcpyAsync(mm, d_mm, 16, hipMemcpyDeviceToHost, st));
	TREEE_HIP(hipStreamSynchronize(st)); // the layer count sizes the caller's offsets array
	const unsigned long long span = (unsigned long long)mm[1] - (unsigned long long)mm[0];
	if (span >= (1ull << 24))
		return set_err(TREEE_ERR_UNSUPPORTED, "slice key range %llu too large (the reference would allocate one Vec per layer)", span + 1);
	const uint32_t layers = (uint32_t)span + 1;
	*layers_out = layers;
	if (min_key_out) *min_key_out = mm[0];
	if (!d_offsets_out && !d_order_out) return TREEE_OK;
	if (d_offsets_out && offsets_cap < (uint64_t)layers + 1) return set_err(TREEE_ERR_INVALID, "offsets_cap %llu < layers+1 = %u", (unsigned long long)offsets_cap, layers + 1);
	uint32_t *d_k32 = nullptr, *d_ks = nullptr, *d_order = d_order_out;
	unsigned long long *d_off = (unsigned long long *)d_offsets_out;
	TREEE_TRY(ctx->alloc_t(n, &d_k32));
	TREEE_TRY(ctx->alloc_t(n, &d_ks));
	if (!d_order) TREEE_TRY(ctx->alloc_t(n, &d_order));
	hipLaunchKernelGGL(k_offset_keys, dim3(div_up(n, 256)), dim3(256), 0, st, d_keys, (uint32_t)n, mm[0], d_k32);
	int bits = 0;
	while ((1ull << bits) < layers) ++bits;
	TREEE_TRY(sort_pairs_dev(ctx, d_k32, nullptr, n, bits, d_ks, d_order));
	if (d_off) hipLaunchKernelGGL(k_slice_offsets, dim3(div_up((uint64_t)layers + 1, 256)), dim3(256), 0, st, d_ks, (uint32_t)n, layers, d_off);
	TREEE_LAUNCH_CHECK();
	return TREEE_OK;
}

int32_t treee_slice_bin(const float *xyz, uint64_t n, int64_t *min_key_out, uint64_t *layers_out, uint32_t *order_out,
                        uint64_t *offsets_out, uint64_t offsets_cap) {
	if (!layers_out) return set_err(TREEE_ERR_INVALID, "treee_slice_bin: layers_out is null");
	*layers_out = 0;
	if (n == 0) return TREEE_OK;
	if (!xyz) return set_err(TREEE_ERR_INVALID, "treee_slice_bin: xyz is null");
	if (n >= (1ull << 31)) return set_err(TREEE_ERR_UNSUPPORTED, "n too large for one call");
	treee_ctx *ctx = nullptr;
	TREEE_DEFAULT_CTX(ctx);
	TREEE_HIP(hipSetDevice(ctx->device));
	DevBuf<float> d_xyz;
	DevBuf<uint32_t> d_order;
	DevBuf<uint64_t> d_off;
	TREEE_TRY(d_xyz.alloc(n * 3));
	TREEE_HIP(hipMemcpyAsync(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice, ctx->stream));
	if (order_out) TREEE_TRY(d_order.alloc(n));
	if (offsets_out) TREEE_TRY(d_off.alloc(offsets_cap ? offsets_cap : 1));
	TREEE_TRY(treee_slice_bin_dev(ctx, d_xyz.p, n, min_key_out, layers_out, d_order.p, d_off.p, offsets_cap));
	if (order_out) TREEE_HIP(hipMemcpyAsync(order_out, d_order.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
	if (offsets_out) TREEE_HIP(hipMemcpyAsync(offsets_out, d_off.p, ((size_t)*layers_out + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

} // extern "C"
