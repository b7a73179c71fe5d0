// rccl_peer_double.cpp -- TEST INFRASTRUCTURE, not RCCL and not part of the product.
//
// A stand-in for the ten librccl entry points csrc/comm.cpp looks up (TREEE_RCCL_LIB=<this .so>), for boxes with ONE GPU:
// real RCCL refuses two ranks on one device, so the library's RCCL transport (RcclComm in csrc/comm.cpp: offsets and byte
// counts of the grouped send / receive, the in-place MIN all-reduce, the all-gather, the unique-id hand-shake of
// treee_amd/distributed.py Comm.rccl) never meets a peer there.  Here the ranks are processes of one host sharing the GPU;
// the bytes travel device -> POSIX shared memory -> device, and every call CHECKS what a real peer would only corrupt
// silently: a receive whose size differs from the matching send, a second send to the same peer inside one group, a call
// outside its communicator's size.  Collectives are synchronous on the stream they are given (the stream is drained first,
// the result is in place when the call returns): what the library may assume of RCCL is weaker, so stream-ORDER mistakes
// are not found here -- only what is sent where.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // the types of the interface it stands in for

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr double TIMEOUT_S = 120.0;

struct Control { // one per communicator, in shared memory, zero-filled by ftruncate
	std::atomic<int> arrived;          // ranks that have mapped the segment
	std::atomic<int> barrier_count;    // sense-reversing barrier
	std::atomic<int> barrier_sense;
	std::atomic<int> failed;           // a rank found an inconsistency: everyone returns an error
	std::atomic<uint64_t> box_bytes[MAX_RANKS]; // size of each rank's outbox file
	struct Entry {
		uint64_t off, bytes;
	} dir[MAX_RANKS][MAX_RANKS]; // dir[src][dst]: where src put what it sends to dst (grouped send / receive)
};

struct Box {
	int fd = -1;
	char *p = nullptr;
	size_t mapped = 0;
};

struct Comm {
	std::string token;
	int rank = 0, world = 1;
	Control *ctl = nullptr;
	Box box[MAX_RANKS]; // [rank] is mine (read-write), the others are the peers' (read)
	int sense = 0;
	std::vector<char> host;
};

struct Pending {
	bool send;
	void *buf;
	size_t bytes;
	int peer;
	Comm *comm;
	hipStream_t stream;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_pending;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::string box_name(const std::string &token, int r) { return "/treee_rccl_double_" + token + "_" + std::to_string(r); }

bool map_box(Comm *c, int r, size_t need, bool mine) {
	Box &b = c->box[r];
	if (b.mapped >= need && b.p) return true;
	if (b.p) munmap(b.p, b.mapped);
	b.p = nullptr;
	if (b.fd < 0) {
		b.fd = shm_open(box_name(c->token, r).c_str(), mine ? (O_CREAT | O_RDWR) : O_RDONLY, 0600);
		if (b.fd < 0) return false;
	}
	if (mine && ftruncate(b.fd, (off_t)need) != 0) return false;
	void *p = mmap(nullptr, need, mine ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, b.fd, 0);
	if (p == MAP_FAILED) return false;
	b.p = (char *)p;
	b.mapped = need;
	return true;
}

// every rank calls it the same number of times; false = a peer reported a failure or did not arrive in time.  After a
// failure the communicator is dead: nobody waits for anybody any more (the ranks leave a failed call at different points)
bool barrier(Comm *c) {
	if (c->world == 1 || c->ctl->failed.load()) return c->ctl->failed.load() == 0;
	c->sense ^= 1;
	if (c->ctl->barrier_count.fetch_add(1) + 1 == c->world) {
		c->ctl->barrier_count.store(0);
		c->ctl->barrier_sense.store(c->sense);
	} else {
		const double t0 = now();
		while (c->ctl->barrier_sense.load() != c->sense) {
			if (c->ctl->failed.load()) return false;
			if (now() - t0 > TIMEOUT_S) {
				c->ctl->failed.store(1);
				std::fprintf(stderr, "rccl_peer_double: rank %d waited %.0f s in a barrier\n", c->rank, TIMEOUT_S);
				return false;
			}
			std::this_thread::sleep_for(std::chrono::microseconds(50));
		}
	}
	return c->ctl->failed.load() == 0;
}

ncclResult_t fail(Comm *c, const char *what) {
	std::fprintf(stderr, "rccl_peer_double: rank %d: %s\n", c->rank, what);
	c->ctl->failed.store(1);
	return ncclInvalidUsage;
}

size_t type_bytes(ncclDataType_t t) {
	switch (t) {
	case ncclUint8: case ncclInt8: return 1;
	case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
	case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
	default: return 0;
	}
}

// my outbox <- device bytes (after everything queued on the stream before the call)
bool put(Comm *c, size_t off, const void *d_src, size_t bytes, hipStream_t s) {
	if (!map_box(c, c->rank, off + bytes, true)) return false;
	return hipMemcpyAsync(c->box[c->rank].p + off, d_src, bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
}

ncclResult_t run_group() {
	std::vector<Pending> ops;
	ops.swap(g_pending);
	if (ops.empty()) return ncclSuccess;
	Comm *c = ops[0].comm;
	hipStream_t s = ops[0].stream;
	for (auto &o : ops)
		if (o.comm != c || o.stream != s) return fail(c, "one group, two communicators or streams: not what csrc/comm.cpp does");
	// sends: packed into my outbox, one directory entry per destination
	for (int r = 0; r < c->world; ++r) c->ctl->dir[c->rank][r] = {0, 0};
	size_t total = 0;
	bool seen[MAX_RANKS] = {};
	for (auto &o : ops)
		if (o.send) {
			if (seen[o.peer]) return fail(c, "two sends to one peer in one group");
			seen[o.peer] = true;
			total += o.bytes;
		}
	if (!map_box(c, c->rank, total ? total : 1, true)) return fail(c, "outbox");
	size_t off = 0;
	for (auto &o : ops)
		if (o.send) {
			if (!put(c, off, o.buf, o.bytes, s)) return fail(c, "device -> outbox copy");
			c->ctl->dir[c->rank][o.peer] = {off, o.bytes};
			off += o.bytes;
		}
	if (hipStreamSynchronize(s) != hipSuccess) return fail(c, "stream");
	c->ctl->box_bytes[c->rank].store(total ? total : 1);
	if (!barrier(c)) return ncclSystemError;
	// receives: the sender's directory must name exactly the bytes this rank expects
	bool got[MAX_RANKS] = {};
	ncclResult_t rc = ncclSuccess;
	for (auto &o : ops)
		if (!o.send && rc == ncclSuccess) {
			if (got[o.peer]) rc = fail(c, "two receives from one peer in one group");
			got[o.peer] = true;
			const Control::Entry e = c->ctl->dir[o.peer][c->rank];
			if (e.bytes != o.bytes) {
				char msg[160];
				std::snprintf(msg, sizeof msg, "receive of %zu bytes from rank %d, which sends %llu", o.bytes, o.peer, (unsigned long long)e.bytes);
				rc = fail(c, msg);
			} else if (!map_box(c, o.peer, c->ctl->box_bytes[o.peer].load(), o.peer == c->rank)) {
				rc = fail(c, "peer outbox");
			} else if (hipMemcpyAsync(o.buf, c->box[o.peer].p + e.off, o.bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
				rc = fail(c, "outbox -> device copy");
			}
		}
	// a send nobody receives is a hang with real RCCL
	if (rc == ncclSuccess)
		for (int r = 0; r < c->world; ++r)
			if (c->ctl->dir[r][c->rank].bytes && !got[r]) rc = fail(c, "a peer sends to this rank, which posts no receive");
	if (hipStreamSynchronize(s) != hipSuccess && rc == ncclSuccess) rc = fail(c, "stream");
	if (!barrier(c) && rc == ncclSuccess) rc = ncclSystemError; // (outboxes are rewritten by the next call)
	return rc;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
	static std::atomic<int> n{0};
	std::memset(id, 0, sizeof *id);
	std::snprintf(id->internal, sizeof id->internal, "%d_%d_%llx", (int)getpid(), n.fetch_add(1), (unsigned long long)(now() * 1e6));
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
	if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
	Comm *c = new Comm;
	c->token.assign(id.internal, strnlen(id.internal, sizeof id.internal));
	c->rank = rank;
	c->world = nranks;
	const std::string name = "/treee_rccl_double_" + c->token + "_ctl";
	const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) return ncclSystemError; // (new bytes are zero: a valid initial state)
	void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) return ncclSystemError;
	c->ctl = (Control *)p;
	c->ctl->arrived.fetch_add(1);
	const double t0 = now();
	while (c->ctl->arrived.load() < nranks) {
		if (now() - t0 > TIMEOUT_S) return ncclSystemError;
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
	if (!map_box(c, rank, 1, true)) return ncclSystemError;
	c->ctl->box_bytes[rank].store(1);
	if (!barrier(c)) return ncclSystemError;
	*out = (ncclComm_t)c;
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
	Comm *c = (Comm *)comm;
	for (int r = 0; r < MAX_RANKS; ++r) {
		if (c->box[r].p) munmap(c->box[r].p, c->box[r].mapped);
		if (c->box[r].fd >= 0) close(c->box[r].fd);
	}
	shm_unlink(box_name(c->token, c->rank).c_str());
	if (c->rank == 0) shm_unlink(("/treee_rccl_double_" + c->token + "_ctl").c_str());
	munmap(c->ctl, sizeof(Control));
	delete c;
	return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t s) {
	Comm *c = (Comm *)comm;
	const size_t bytes = count * type_bytes(type);
	if (!bytes) return fail(c, "ncclAllGather of nothing (csrc/comm.cpp returns before the call)");
	if (!put(c, 0, send, bytes, s) || hipStreamSynchronize(s) != hipSuccess) return fail(c, "device -> outbox copy");
	c->ctl->box_bytes[c->rank].store(bytes);
	if (!barrier(c)) return ncclSystemError;
	ncclResult_t rc = ncclSuccess;
	for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
		if (c->ctl->box_bytes[r].load() != bytes) rc = fail(c, "ncclAllGather: the ranks disagree on the block size");
		else if (!map_box(c, r, bytes, r == c->rank)) rc = fail(c, "peer outbox");
		else if (hipMemcpyAsync((char *)recv + (size_t)r * bytes, c->box[r].p, bytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = fail(c, "outbox -> device copy");
	}
	if (hipStreamSynchronize(s) != hipSuccess && rc == ncclSuccess) rc = fail(c, "stream");
	if (!barrier(c) && rc == ncclSuccess) rc = ncclSystemError;
	return rc;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
	Comm *c = (Comm *)comm;
	if (type != ncclUint64 || op != ncclMin) return fail(c, "ncclAllReduce: only MIN over u64 is what csrc/comm.cpp asks for");
	const size_t bytes = count * 8;
	if (!bytes) return fail(c, "ncclAllReduce of nothing");
	if (!put(c, 0, send, bytes, s) || hipStreamSynchronize(s) != hipSuccess) return fail(c, "device -> outbox copy");
	c->ctl->box_bytes[c->rank].store(bytes);
	if (!barrier(c)) return ncclSystemError;
	ncclResult_t rc = ncclSuccess;
	c->host.resize(bytes);
	uint64_t *acc = (uint64_t *)c->host.data();
	std::memcpy(acc, c->box[c->rank].p, bytes);
	for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
		if (r == c->rank) continue;
		if (c->ctl->box_bytes[r].load() != bytes) rc = fail(c, "ncclAllReduce: the ranks disagree on the count");
		else if (!map_box(c, r, bytes, false)) rc = fail(c, "peer outbox");
		else {
			const uint64_t *q = (const uint64_t *)c->box[r].p;
			for (size_t i = 0; i < count; ++i) acc[i] = q[i] < acc[i] ? q[i] : acc[i];
		}
	}
	if (rc == ncclSuccess && hipMemcpyAsync(recv, acc, bytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = fail(c, "result -> device copy");
	if (hipStreamSynchronize(s) != hipSuccess && rc == ncclSuccess) rc = fail(c, "stream");
	if (!barrier(c) && rc == ncclSuccess) rc = ncclSystemError;
	return rc;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
	Comm *c = (Comm *)comm;
	if (peer < 0 || peer >= c->world) return fail(c, "ncclSend: peer outside the communicator");
	if (!count) return fail(c, "ncclSend of nothing (csrc/comm.cpp skips empty blocks)");
	g_pending.push_back({true, (void *)buf, count * type_bytes(type), peer, c, s});
	return g_group_depth ? ncclSuccess : run_group();
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
	Comm *c = (Comm *)comm;
	if (peer < 0 || peer >= c->world) return fail(c, "ncclRecv: peer outside the communicator");
	if (!count) return fail(c, "ncclRecv of nothing (csrc/comm.cpp skips empty blocks)");
	g_pending.push_back({false, buf, count * type_bytes(type), peer, c, s});
	return g_group_depth ? ncclSuccess : run_group();
}

ncclResult_t ncclGroupStart() {
	++g_group_depth;
	return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
	if (g_group_depth <= 0) return ncclInvalidUsage;
	if (--g_group_depth) return ncclSuccess;
	return run_group();
}

const char *ncclGetErrorString(ncclResult_t r) {
	switch (r) {
	case ncclSuccess: return "no error";
	case ncclSystemError: return "peer double: a rank failed or did not arrive";
	case ncclInvalidArgument: return "peer double: invalid argument";
	case ncclInvalidUsage: return "peer double: a call the real library would corrupt data or hang on (see stderr)";
	default: return "peer double: error";
	}
}

} // extern "C"
