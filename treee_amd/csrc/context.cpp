// context.cpp -- error state, treee_ctx (stream + workspace arena), default context.
#include "common.h"
#include <cstdlib>

#include <mutex>
#include <set>

#include <cstdio>
#include <thread>

namespace treee {

unsigned host_threads() {
	unsigned n = std::thread::hardware_concurrency();
	if (n == 0) n = 1;
	if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
		long long quota = 0, period = 0;
		if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
			const unsigned q = (unsigned)((quota + period - 1) / period);
			if (q >= 1 && q < n) n = q;
		}
		std::fclose(f);
	}
	return n;
}


static thread_local char g_err[512] = "";

int32_t set_err(int32_t code, const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

static std::mutex g_live_mu; // contexts that are still alive (handles may outlive their context)
static std::set<treee_ctx *> g_live;
constexpr size_t POOL_GRANULE = size_t(1) << 20;

int32_t pool_alloc(treee_ctx *ctx, size_t bytes, void **out) {
	bytes = ((bytes ? bytes : 1) + POOL_GRANULE - 1) & ~(POOL_GRANULE - 1);
	{
		std::lock_guard<std::mutex> lk(ctx->pool_mu);
		size_t best = SIZE_MAX, best_cap = SIZE_MAX;
		for (size_t i = 0; i < ctx->pool_idle.size(); ++i) {
			const size_t cap = ctx->pool_idle[i].cap;
			if (cap >= bytes && cap <= 2 * bytes && cap < best_cap) {
				best = i;
				best_cap = cap;
			}
		}
		if (best != SIZE_MAX) {
			*out = ctx->pool_idle[best].p;
			ctx->pool_live[*out] = best_cap;
			ctx->pool_idle.erase(ctx->pool_idle.begin() + best);
			return TREEE_OK;
		}
	}
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) { // give idle blocks back to the driver and retry once
		{
			std::lock_guard<std::mutex> lk(ctx->pool_mu);
			(void)hipStreamSynchronize(ctx->stream);
			for (auto &b : ctx->pool_idle) (void)hipFree(b.p);
			ctx->pool_idle.clear();
		}
		e = hipMalloc(&p, bytes);
		if (e != hipSuccess) { // the other contexts of this device keep idle blocks too (up to pool_idle_limit each)
			(void)hipDeviceSynchronize();
			std::lock_guard<std::mutex> live(g_live_mu);
			for (treee_ctx *other : g_live) {
				if (other == ctx || other->device != ctx->device) continue;
				std::lock_guard<std::mutex> lk(other->pool_mu);
				for (auto &b : other->pool_idle) (void)hipFree(b.p);
				other->pool_idle.clear();
			}
			e = hipMalloc(&p, bytes);
		}
		if (e != hipSuccess) return set_err(TREEE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
	}
	std::lock_guard<std::mutex> lk(ctx->pool_mu);
	ctx->pool_live[p] = bytes;
	*out = p;
	return TREEE_OK;
}

void pool_release(treee_ctx *ctx, void *p) {
	if (!p) return;
	std::lock_guard<std::mutex> live(g_live_mu);
	if (ctx && g_live.count(ctx)) {
		std::lock_guard<std::mutex> lk(ctx->pool_mu);
		auto it = ctx->pool_live.find(p);
		if (it != ctx->pool_live.end()) {
			ctx->pool_idle.push_back({(char *)p, it->second});
			ctx->pool_live.erase(it);
			size_t idle = 0;
			for (auto &b : ctx->pool_idle) idle += b.cap;
			while (idle > ctx->pool_idle_limit && !ctx->pool_idle.empty()) { // trim oldest first
				(void)hipStreamSynchronize(ctx->stream);
				idle -= ctx->pool_idle.front().cap;
				(void)hipFree(ctx->pool_idle.front().p);
				ctx->pool_idle.erase(ctx->pool_idle.begin());
			}
			return;
		}
	}
	(void)hipDeviceSynchronize(); // owner gone: the block is an ordinary allocation
	(void)hipFree(p);
}

bool ctx_is_live(treee_ctx *ctx) {
	std::lock_guard<std::mutex> live(g_live_mu);
	return ctx && g_live.count(ctx);
}

static bool foreign(treee_ctx *ctx, treee_ctx *owner) { return ctx && owner && ctx != owner && ctx->stream != owner->stream; }

int32_t mark_ready(treee_ctx *owner, HandleSync *hs) {
	std::lock_guard<std::mutex> lk(hs->mu);
	if (!hs->ready) TREEE_HIP(hipEventCreateWithFlags(&hs->ready, hipEventDisableTiming));
	TREEE_HIP(hipEventRecord(hs->ready, owner->stream));
	return TREEE_OK;
}

int32_t use_begin(treee_ctx *ctx, treee_ctx *owner, HandleSync *hs) {
	if (!foreign(ctx, owner) || !ctx_is_live(owner)) return TREEE_OK; // (a destroyed owner synchronised its stream: the build is done)
	{
		std::lock_guard<std::mutex> lk(hs->mu);
		if (hs->ready) { // the build's own event: later work of the owner's stream (a kNN kernel, say) is not waited for
			TREEE_HIP(hipStreamWaitEvent(ctx->stream, hs->ready, 0));
			return TREEE_OK;
		}
	}
	hipEvent_t ev = nullptr;
	TREEE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	hipError_t e = hipEventRecord(ev, owner->stream);
	if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
	(void)hipEventDestroy(ev); // the wait holds its own reference to the recorded state
	if (e != hipSuccess) return set_err(TREEE_ERR_HIP, "cross-stream wait for a handle's owner: %s", hipGetErrorString(e));
	return TREEE_OK;
}

int32_t use_end(treee_ctx *ctx, treee_ctx *owner, HandleSync *hs) {
	if (!foreign(ctx, owner)) return TREEE_OK;
	std::lock_guard<std::mutex> lk(hs->mu);
	hipEvent_t ev = nullptr;
	for (auto &u : hs->users)
		if (u.first == ctx->stream) ev = u.second;
	if (!ev) {
		TREEE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		hs->users.push_back({ctx->stream, ev});
	}
	TREEE_HIP(hipEventRecord(ev, ctx->stream));
	return TREEE_OK;
}

// the owner's stream waits for the uses recorded so far; the events and `ready` stay (the handle lives on)
void wait_users(treee_ctx *owner, HandleSync *hs) {
	std::lock_guard<std::mutex> lk(hs->mu);
	if (!ctx_is_live(owner)) return;
	for (auto &u : hs->users)
		if (hipStreamWaitEvent(owner->stream, u.second, 0) != hipSuccess) (void)hipEventSynchronize(u.second);
}

void retire(treee_ctx *owner, HandleSync *hs) {
	std::lock_guard<std::mutex> lk(hs->mu);
	const bool live = ctx_is_live(owner);
	for (auto &u : hs->users) {
		// a live owner's stream waits (no host stall); otherwise the block becomes an ordinary allocation and the
		// host waits here
		if (!live || hipStreamWaitEvent(owner->stream, u.second, 0) != hipSuccess) (void)hipEventSynchronize(u.second);
		(void)hipEventDestroy(u.second);
	}
	hs->users.clear();
	if (hs->ready) {
		(void)hipEventDestroy(hs->ready); // (destroying a recorded event does not disturb waits already enqueued on it)
		hs->ready = nullptr;
	}
}

HandleSync::~HandleSync() {
	if (ready) (void)hipEventDestroy(ready);
	for (auto &u : users) {
		(void)hipEventSynchronize(u.second);
		(void)hipEventDestroy(u.second);
	}
}

static std::mutex g_default_mu;
std::recursive_mutex &default_ctx_mutex() {
	static std::recursive_mutex m;
	return m;
}
static treee_ctx *g_default = nullptr;

int32_t default_ctx(treee_ctx **out) {
	std::lock_guard<std::mutex> lk(g_default_mu);
	if (!g_default) {
		int dev = 0;
		TREEE_HIP(hipGetDevice(&dev));
		TREEE_TRY(treee_ctx_create(dev, nullptr, &g_default));
		// the host-pointer API gets a private non-blocking stream: it synchronises itself
		hipStream_t s = nullptr;
		TREEE_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		g_default->stream = s;
		g_default->own_stream = true;
	}
	*out = g_default;
	return TREEE_OK;
}

} // namespace treee

using namespace treee;

int32_t treee_ctx::fork() {
	if (!aux) { // all three or none: a half-made set would leave later fork() / join() calls recording on a null event
		hipStream_t s = nullptr;
		hipEvent_t ef = nullptr, ej = nullptr, em = nullptr;
		// the auxiliary stream carries chains of small dependent launches (the octree's levels) beside a kernel that fills the
		// card (the record gather): at the default priority each of them queues behind the big kernel's waiting workgroups
		// (flags + block tables 400 us instead of 100, measured with TREEE_TRACE_BUILD); the highest priority lets them through
		int lo = 0, hi = 0;
		hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
		static const bool flat = getenv("TREEE_AUX_DEFAULT_PRIORITY") != nullptr; // A/B hook
		if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, flat ? 0 : hi);
		// (ONE auxiliary stream: a second one for the kNN order's tables made the side chains shorter and every dispatch of the process
		// ~2 us slower -- the splat's five launches 0.161 instead of 0.149 ms, the structure stage 1.23 instead of 1.21 ms)
		if (e == hipSuccess) e = hipEventCreateWithFlags(&ef, hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&ej, hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&em, hipEventDisableTiming);
		if (e != hipSuccess) {
			if (em) (void)hipEventDestroy(em);
			if (ej) (void)hipEventDestroy(ej);
			if (ef) (void)hipEventDestroy(ef);
			if (s) (void)hipStreamDestroy(s);
			return set_err(TREEE_ERR_HIP, "auxiliary stream of the context: %s", hipGetErrorString(e));
		}
		aux = s;
		ev_fork = ef;
		ev_join = ej;
		ev_mid = em;
	}
	TREEE_HIP(hipEventRecord(ev_fork, stream));
	TREEE_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
	return TREEE_OK;
}
void treee_ctx::on_aux(bool yes) {
	if (yes && aux && stream != aux) {
		main_stream = stream;
		stream = aux;
	} else if (!yes && aux && stream == aux) {
		stream = main_stream;
	}
}
int32_t treee_ctx::join() {
	on_aux(false);
	if (!aux) return TREEE_OK;
	TREEE_HIP(hipEventRecord(ev_join, aux));
	TREEE_HIP(hipStreamWaitEvent(stream, ev_join, 0));
	return TREEE_OK;
}

int32_t treee_ctx::reset() {
	// coalesce into one block if the previous call had to chain several
	if (blocks.size() > 1) {
		size_t total = 0;
		for (auto &b : blocks) total += b.cap;
		TREEE_HIP(hipStreamSynchronize(stream));
		for (auto &b : blocks) (void)hipFree(b.p);
		blocks.clear();
		char *p = nullptr;
		hipError_t e = hipMalloc((void **)&p, total);
		if (e != hipSuccess) return set_err(TREEE_ERR_NOMEM, "workspace hipMalloc(%zu): %s", total, hipGetErrorString(e));
		blocks.push_back({p, total});
	}
	cur = 0;
	off = 0;
	return TREEE_OK;
}

int32_t treee_ctx::alloc(size_t bytes, void **out) {
	bytes = (bytes + 255) & ~size_t(255);
	if (bytes == 0) bytes = 256;
	while (cur < blocks.size()) {
		if (off + bytes <= blocks[cur].cap) {
			*out = blocks[cur].p + off;
			off += bytes;
			return TREEE_OK;
		}
		cur++;
		off = 0;
	}
	size_t cap = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes;
	char *p = nullptr;
	hipError_t e = hipMalloc((void **)&p, cap);
	if (e != hipSuccess) return set_err(TREEE_ERR_NOMEM, "workspace hipMalloc(%zu): %s", cap, hipGetErrorString(e));
	blocks.push_back({p, cap});
	cur = blocks.size() - 1;
	*out = p;
	off = bytes;
	return TREEE_OK;
}

extern "C" {

int32_t treee_abi_version(void) { return TREEE_ABI_VERSION; }

const char *treee_last_error(void) { return g_err; }

int32_t treee_device_count(int32_t *count_out) {
	if (!count_out) return set_err(TREEE_ERR_INVALID, "count_out is null");
	int c = 0;
	hipError_t e = hipGetDeviceCount(&c);
	if (e != hipSuccess) {
		*count_out = 0;
		return set_err(TREEE_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
	}
	*count_out = c;
	return TREEE_OK;
}

int32_t treee_ctx_create(int32_t device, void *hip_stream, treee_ctx **out) {
	if (!out) return set_err(TREEE_ERR_INVALID, "out is null");
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count == 0)
		return set_err(TREEE_ERR_HIP, "no HIP device available (%s); libtreee_hip has no CPU fallback",
		               e != hipSuccess ? hipGetErrorString(e) : "0 devices");
	if (device < 0 || device >= count) return set_err(TREEE_ERR_INVALID, "device %d out of range [0,%d)", device, count);
	TREEE_HIP(hipSetDevice(device));
	treee_ctx *c = new treee_ctx;
	c->device = device;
	// the handle is used verbatim; NULL is HIP's legacy default stream (what torch's default
	// stream is), NOT a request for a private stream
	c->stream = (hipStream_t)hip_stream;
	c->own_stream = false;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
		c->num_cus = prop.multiProcessorCount;
		if (prop.totalGlobalMem / 4 > c->pool_idle_limit) c->pool_idle_limit = prop.totalGlobalMem / 4;
	}
	e = hipHostMalloc((void **)&c->h_scalars, 64 * sizeof(uint32_t), hipHostMallocDefault);
	if (e != hipSuccess) {
		delete c;
		return set_err(TREEE_ERR_NOMEM, "pinned scalar block: %s", hipGetErrorString(e));
	}
	memset(c->h_scalars, 0, 64 * sizeof(uint32_t));
	{
		std::lock_guard<std::mutex> lk(g_live_mu);
		g_live.insert(c);
	}
	*out = c;
	return TREEE_OK;
}

int32_t treee_ctx_destroy(treee_ctx *ctx) {
	if (!ctx) return TREEE_OK;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	{
		std::lock_guard<std::mutex> lk(g_live_mu);
		g_live.erase(ctx); // blocks still held by handles stay valid; they are hipFree'd on release
	}
	for (auto &b : ctx->pool_idle) (void)hipFree(b.p);
	for (auto &b : ctx->blocks) (void)hipFree(b.p);
	if (ctx->aux) {
		(void)hipStreamSynchronize(ctx->aux);
		(void)hipStreamDestroy(ctx->aux);
		(void)hipEventDestroy(ctx->ev_fork);
		(void)hipEventDestroy(ctx->ev_join);
		(void)hipEventDestroy(ctx->ev_mid);
	}
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
	if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
	delete ctx;
	return TREEE_OK;
}

int32_t treee_ctx_sync(treee_ctx *ctx) {
	if (!ctx) return set_err(TREEE_ERR_INVALID, "ctx is null");
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	return TREEE_OK;
}

int32_t treee_ctx_reserve(treee_ctx *ctx, uint64_t bytes) {
	if (!ctx) return set_err(TREEE_ERR_INVALID, "ctx is null");
	TREEE_HIP(hipSetDevice(ctx->device));
	size_t have = 0;
	for (auto &b : ctx->blocks) have += b.cap;
	if (have >= bytes && ctx->blocks.size() == 1) return TREEE_OK;
	TREEE_HIP(hipStreamSynchronize(ctx->stream));
	for (auto &b : ctx->blocks) (void)hipFree(b.p);
	ctx->blocks.clear();
	char *p = nullptr;
	size_t cap = bytes > have ? bytes : have;
	hipError_t e = hipMalloc((void **)&p, cap);
	if (e != hipSuccess) return set_err(TREEE_ERR_NOMEM, "workspace hipMalloc(%zu): %s", cap, hipGetErrorString(e));
	ctx->blocks.push_back({p, cap});
	ctx->cur = 0;
	ctx->off = 0;
	return TREEE_OK;
}

} // extern "C"
