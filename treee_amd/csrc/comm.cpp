// comm.cpp -- the two transports of treee_comm (treee_hip.h, "many GPUs -- communicator"): RCCL over xGMI, opened at run
// time, and caller-supplied host callbacks.  Host code; no reference counterpart (the reference is one process).
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums only: the functions are looked up in whichever librccl the process holds

#include <cstdlib>

using namespace treee;

namespace {

struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mu;
RcclApi g_rccl;

// The copy the process already holds first (under PyTorch that is torch's own librccl: two RCCL copies in one process
// would each bring their own transport state), then the system's.  TREEE_RCCL_LIB names ONE library instead: it alone is
// tried, and a path that cannot be opened means "no RCCL here" (how a host without xGMI peers -- or a test -- says so).
int32_t rccl_api(const RcclApi **out) {
	std::lock_guard<std::mutex> lk(g_rccl_mu);
	if (!g_rccl.lib) {
		void *h = nullptr;
		const char *names[] = {"librccl.so.1", "librccl.so"};
		if (const char *env = std::getenv("TREEE_RCCL_LIB")) {
			h = dlopen(env, RTLD_NOW | RTLD_NOLOAD);
			if (!h) {
				// the override is taken -- but if the process already holds a librccl (under PyTorch: torch's own), this loads a
				// SECOND copy with its own transport state, which the paragraph above warns against.  Say so, unless the caller
				// states that the named file is the tests' stand-in (TREEE_RCCL_LIB_IS_TEST_DOUBLE=1), which brings no RCCL state.
				void *held = nullptr;
				for (const char *nm : names)
					if (!held) held = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
				if (held && !std::getenv("TREEE_RCCL_LIB_IS_TEST_DOUBLE"))
					std::fprintf(stderr, "treee: TREEE_RCCL_LIB=%s is loaded beside the librccl this process already holds -- two RCCL copies in one process "
					                     "(unset it to use the held one)\n", env);
				if (held) (void)dlclose(held);
				h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
			}
			if (!h) return set_err(TREEE_ERR_UNSUPPORTED, "TREEE_RCCL_LIB=%s cannot be opened: %s", env, dlerror());
		}
		for (const char *nm : names)
			if (!h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
		for (const char *nm : names)
			if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!h) return set_err(TREEE_ERR_UNSUPPORTED, "librccl.so not found: %s", dlerror());
		RcclApi a;
		a.lib = h;
#define TREEE_RCCL_SYM(field, name)                                                                  \
	a.field = (decltype(a.field))dlsym(h, name);                                                     \
	if (!a.field) return set_err(TREEE_ERR_UNSUPPORTED, "librccl.so lacks %s", name)
		TREEE_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
		TREEE_RCCL_SYM(CommInitRank, "ncclCommInitRank");
		TREEE_RCCL_SYM(CommDestroy, "ncclCommDestroy");
		TREEE_RCCL_SYM(AllGather, "ncclAllGather");
		TREEE_RCCL_SYM(AllReduce, "ncclAllReduce");
		TREEE_RCCL_SYM(Send, "ncclSend");
		TREEE_RCCL_SYM(Recv, "ncclRecv");
		TREEE_RCCL_SYM(GroupStart, "ncclGroupStart");
		TREEE_RCCL_SYM(GroupEnd, "ncclGroupEnd");
		TREEE_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef TREEE_RCCL_SYM
		g_rccl = a;
	}
	*out = &g_rccl;
	return TREEE_OK;
}

#define TREEE_RCCL(api, expr)                                                                                       \
	do {                                                                                                            \
		ncclResult_t r__ = (expr);                                                                                  \
		if (r__ != ncclSuccess)                                                                                     \
			return set_err(TREEE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r__), __FILE__, __LINE__); \
	} while (0)

struct RcclComm final : treee_comm {
	const RcclApi *api = nullptr;
	ncclComm_t comm = nullptr;
	bool owned = false;
	~RcclComm() override {
		if (owned && comm) (void)api->CommDestroy(comm);
	}
	int32_t all_gather_dev(treee_ctx *ctx, const void *d_send, void *d_recv, size_t bytes) override {
		if (bytes == 0) return TREEE_OK;
		TREEE_RCCL(api, api->AllGather(d_send, d_recv, bytes, ncclUint8, comm, ctx->stream));
		return TREEE_OK;
	}
	// xGMI is point to point: one grouped send / receive per peer is the all-to-all (each pair uses its own link)
	int32_t all_to_all_v_dev(treee_ctx *ctx, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
	                         const uint64_t *recv_off, const uint64_t *recv_bytes) override {
		TREEE_RCCL(api, api->GroupStart());
		ncclResult_t bad = ncclSuccess;
		for (int r = 0; r < world && bad == ncclSuccess; ++r) {
			if (send_bytes[r]) bad = api->Send((const char *)d_send + send_off[r], send_bytes[r], ncclUint8, r, comm, ctx->stream);
			if (bad == ncclSuccess && recv_bytes[r]) bad = api->Recv((char *)d_recv + recv_off[r], recv_bytes[r], ncclUint8, r, comm, ctx->stream);
		}
		const ncclResult_t end = api->GroupEnd();
		if (bad != ncclSuccess) return set_err(TREEE_ERR_HIP, "ncclSend / ncclRecv failed: %s", api->GetErrorString(bad));
		TREEE_RCCL(api, end);
		return TREEE_OK;
	}
	int32_t all_reduce_min_u64_dev(treee_ctx *ctx, uint64_t *d_buf, size_t count) override {
		if (count == 0) return TREEE_OK;
		TREEE_RCCL(api, api->AllReduce(d_buf, d_buf, count, ncclUint64, ncclMin, comm, ctx->stream));
		return TREEE_OK;
	}
	int32_t all_gather_host(treee_ctx *ctx, const void *send, void *recv, size_t bytes) override {
		if (bytes == 0) return TREEE_OK;
		char *d = nullptr;
		TREEE_TRY(ctx->alloc_t(bytes * (size_t)(world + 1), &d));
		TREEE_HIP(hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, ctx->stream));
		TREEE_RCCL(api, api->AllGather(d, d + bytes, bytes, ncclUint8, comm, ctx->stream));
		TREEE_HIP(hipMemcpyAsync(recv, d + bytes, bytes * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		return TREEE_OK;
	}
};

struct HostComm final : treee_comm {
	treee_comm_host_ops ops{};
	std::vector<char> a, b; // staging
	int32_t call(int32_t rc, const char *what) { return rc == TREEE_OK ? TREEE_OK : set_err(TREEE_ERR_HIP, "host transport: %s failed with %d", what, rc); }
	int32_t all_gather_dev(treee_ctx *ctx, const void *d_send, void *d_recv, size_t bytes) override {
		if (bytes == 0) return TREEE_OK;
		try {
			a.resize(bytes);
			b.resize(bytes * (size_t)world);
		} catch (const std::exception &) {
			return set_err(TREEE_ERR_NOMEM, "host transport: staging buffer");
		}
		TREEE_HIP(hipMemcpyAsync(a.data(), d_send, bytes, hipMemcpyDeviceToHost, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		TREEE_TRY(call(ops.all_gather(ops.user, a.data(), b.data(), bytes), "all_gather"));
		TREEE_HIP(hipMemcpyAsync(d_recv, b.data(), b.size(), hipMemcpyHostToDevice, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream)); // (the staging buffer is reused by the next call)
		return TREEE_OK;
	}
	int32_t all_to_all_v_dev(treee_ctx *ctx, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
	                         const uint64_t *recv_off, const uint64_t *recv_bytes) override {
		uint64_t s_end = 0, r_end = 0;
		for (int r = 0; r < world; ++r) {
			if (send_off[r] + send_bytes[r] > s_end) s_end = send_off[r] + send_bytes[r];
			if (recv_off[r] + recv_bytes[r] > r_end) r_end = recv_off[r] + recv_bytes[r];
		}
		try {
			a.resize(s_end ? s_end : 1);
			b.resize(r_end ? r_end : 1);
		} catch (const std::exception &) {
			return set_err(TREEE_ERR_NOMEM, "host transport: staging buffer");
		}
		for (int r = 0; r < world; ++r) // (only what is sent: the send buffer is the rank's whole sorted cloud)
			if (send_bytes[r]) TREEE_HIP(hipMemcpyAsync(a.data() + send_off[r], (const char *)d_send + send_off[r], send_bytes[r], hipMemcpyDeviceToHost, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		TREEE_TRY(call(ops.all_to_all_v(ops.user, a.data(), send_off, send_bytes, b.data(), recv_off, recv_bytes), "all_to_all_v"));
		for (int r = 0; r < world; ++r)
			if (recv_bytes[r]) TREEE_HIP(hipMemcpyAsync((char *)d_recv + recv_off[r], b.data() + recv_off[r], recv_bytes[r], hipMemcpyHostToDevice, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		return TREEE_OK;
	}
	int32_t all_reduce_min_u64_dev(treee_ctx *ctx, uint64_t *d_buf, size_t count) override {
		if (count == 0) return TREEE_OK;
		try {
			a.resize(count * 8);
		} catch (const std::exception &) {
			return set_err(TREEE_ERR_NOMEM, "host transport: staging buffer");
		}
		TREEE_HIP(hipMemcpyAsync(a.data(), d_buf, count * 8, hipMemcpyDeviceToHost, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		TREEE_TRY(call(ops.all_reduce_min_u64(ops.user, (uint64_t *)a.data(), count), "all_reduce_min_u64"));
		TREEE_HIP(hipMemcpyAsync(d_buf, a.data(), count * 8, hipMemcpyHostToDevice, ctx->stream));
		TREEE_HIP(hipStreamSynchronize(ctx->stream));
		return TREEE_OK;
	}
	int32_t all_gather_host(treee_ctx *, const void *send, void *recv, size_t bytes) override {
		if (bytes == 0) return TREEE_OK;
		return call(ops.all_gather(ops.user, send, recv, bytes), "all_gather");
	}
};

} // namespace

extern "C" {

int32_t treee_comm_create_host(const treee_comm_host_ops *ops, int32_t rank, int32_t world, treee_comm **out) {
	if (!ops || !out || !ops->all_gather || !ops->all_to_all_v || !ops->all_reduce_min_u64 || world < 1 || rank < 0 || rank >= world)
		return set_err(TREEE_ERR_INVALID, "treee_comm_create_host: bad argument");
	HostComm *c = new HostComm;
	c->ops = *ops;
	c->rank = rank;
	c->world = world;
	*out = c;
	return TREEE_OK;
}

int32_t treee_rccl_available(void) {
	// only opens the library and looks the entry points up: no communicator, no collective, nothing a peer has to answer --
	// so that the ranks can agree on the transport BEFORE any of them enters ncclCommInitRank
	const RcclApi *api = nullptr;
	return rccl_api(&api);
}

int32_t treee_rccl_unique_id(uint8_t id_out[128]) {
	if (!id_out) return set_err(TREEE_ERR_INVALID, "treee_rccl_unique_id: null argument");
	const RcclApi *api = nullptr;
	TREEE_TRY(rccl_api(&api));
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
	ncclUniqueId id;
	TREEE_RCCL(api, api->GetUniqueId(&id));
	std::memcpy(id_out, &id, 128);
	return TREEE_OK;
}

int32_t treee_comm_create_rccl(const uint8_t unique_id[128], int32_t rank, int32_t world, int32_t device, treee_comm **out) {
	if (!unique_id || !out || world < 1 || rank < 0 || rank >= world) return set_err(TREEE_ERR_INVALID, "treee_comm_create_rccl: bad argument");
	*out = nullptr;
	const RcclApi *api = nullptr;
	TREEE_TRY(rccl_api(&api));
	TREEE_HIP(hipSetDevice(device));
	ncclUniqueId id;
	std::memcpy(&id, unique_id, 128);
	ncclComm_t comm = nullptr;
	TREEE_RCCL(api, api->CommInitRank(&comm, world, id, rank));
	RcclComm *c = new RcclComm;
	c->api = api;
	c->comm = comm;
	c->owned = true;
	c->rccl = true;
	c->rank = rank;
	c->world = world;
	*out = c;
	return TREEE_OK;
}

int32_t treee_comm_wrap_rccl(void *nccl_comm, int32_t rank, int32_t world, treee_comm **out) {
	if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return set_err(TREEE_ERR_INVALID, "treee_comm_wrap_rccl: bad argument");
	const RcclApi *api = nullptr;
	TREEE_TRY(rccl_api(&api));
	RcclComm *c = new RcclComm;
	c->api = api;
	c->comm = (ncclComm_t)nccl_comm;
	c->rccl = true;
	c->rank = rank;
	c->world = world;
	*out = c;
	return TREEE_OK;
}

int32_t treee_comm_destroy(treee_comm *c) {
	delete c;
	return TREEE_OK;
}

int32_t treee_comm_info(const treee_comm *c, int32_t *rank_out, int32_t *world_out, int32_t *is_rccl_out) {
	if (!c) return set_err(TREEE_ERR_INVALID, "treee_comm_info: null argument");
	if (rank_out) *rank_out = c->rank;
	if (world_out) *world_out = c->world;
	if (is_rccl_out) *is_rccl_out = c->rccl ? 1 : 0;
	return TREEE_OK;
}

} // extern "C"
