// rccl_peer_double.cpp -- TEST INFRASTRUCTURE, not RCCL and not part of the product.
//
// A stand-in for the ten librccl entry points csrc/comm.cpp looks up (TREEE_RCCL_LIB=<this .so>), for boxes with ONE GPU:
// real RCCL refuses two ranks on one device, so the library's RCCL transport (RcclComm in csrc/comm.cpp: offsets and byte
// counts of the grouped send / receive, the in-place MIN all-reduce, the all-gather, the unique-id hand-shake of
// treee_amd/distributed.py Comm.rccl) never meets a peer there.  Here the ranks are processes of one host sharing the GPU;
// the bytes travel device -> pinned staging -> POSIX shared memory -> pinned staging -> device, and every call CHECKS what a
// real peer would only corrupt silently: a receive whose size differs from the matching send, a second send to the same peer
// inside one group, a call outside its communicator's size.
//
// STREAM SEMANTICS ARE RCCL'S (round 5; rounds 3-4 drained the caller's stream before and after every call, which hid every
// stream-order mistake): a call only exchanges SIZES with its peers on the host -- that is where a mismatch is refused --
// and ENQUEUES the data movement: the send buffers are read after what the caller's stream held at the call (an event), on an
// internal stream whose host functions carry the bytes between the processes, and the caller's stream waits for that
// stream's completion event.  Nothing is in place when the call returns; a consumer on ANOTHER stream that does not wait
// for the caller's stream reads stale bytes, a producer on another stream that the caller's stream did not wait for is read
// too early -- as with the real library (test_peer_double_exposes_a_missing_stream_wait).  TREEE_DOUBLE_DELAY_MS adds a
// pause before a rank publishes its bytes, so that such a race loses every time instead of most times.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // the types of the interface it stands in for

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr double TIMEOUT_S = 120.0;

struct Control { // one per communicator, in shared memory, zero-filled by ftruncate
	std::atomic<int> arrived;          // ranks that have mapped the segment
	std::atomic<int> barrier_count;    // sense-reversing barrier (host side of a call: sizes only)
	std::atomic<int> barrier_sense;
	std::atomic<int> failed;           // a rank found an inconsistency: everyone returns an error
	std::atomic<uint64_t> box_bytes[MAX_RANKS]; // size of each rank's outbox file
	std::atomic<uint64_t> ready[MAX_RANKS];     // sequence number of the call whose bytes lie in the rank's outbox
	std::atomic<uint64_t> consumed[MAX_RANKS];  // sequence number of the last call whose peers' outboxes the rank has read
	struct Meta {
		uint64_t seq, kind, bytes; // kind: 1 grouped send / receive, 2 all-gather, 3 all-reduce
	} meta[MAX_RANKS];
	struct Entry {
		uint64_t off, bytes;
	} dir[MAX_RANKS][MAX_RANKS]; // dir[src][dst]: where src puts what it sends to dst (grouped send / receive)
};

struct Box {
	int fd = -1;
	char *p = nullptr;
	size_t mapped = 0;
};

struct Piece { // one contiguous transfer between a device buffer and a staging buffer
	void *dev;
	size_t stage_off, bytes;
	int peer;        // receive: whose outbox
	size_t peer_off; // receive: where in it
};

struct Comm;
struct Job { // what the host functions of one call do (owned by the communicator until the call has run)
	Comm *comm;
	uint64_t seq;
	int kind;
	size_t out_bytes;          // bytes of staging_out that go to my outbox
	std::vector<Piece> recv;   // outbox pieces -> staging_in (kind 1, 2)
	size_t reduce_count = 0;   // kind 3: u64 words
};

struct Comm {
	std::string token;
	int rank = 0, world = 1;
	Control *ctl = nullptr;
	Box box[MAX_RANKS]; // [rank] is mine (read-write), the others are the peers' (read)
	int sense = 0;
	uint64_t seq = 0;
	hipStream_t is = nullptr; // the internal stream the transfers are ordered on
	char *stage_out = nullptr, *stage_in = nullptr; // pinned
	size_t stage_out_cap = 0, stage_in_cap = 0;
	std::vector<Job *> jobs;
	std::mutex mu;
	int delay_ms = 0;
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
	// (a host function of an earlier call may still be reading the old mapping)
	if (c->is && hipStreamSynchronize(c->is) != hipSuccess) return false;
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

bool grow_stage(Comm *c, char **p, size_t *cap, size_t need) {
	if (*cap >= need) return true;
	if (hipStreamSynchronize(c->is) != hipSuccess) return false; // the old buffer may be in flight
	if (*p) (void)hipHostFree(*p);
	*p = nullptr;
	*cap = 0;
	size_t want = need + need / 2 + 4096;
	if (hipHostMalloc((void **)p, want, hipHostMallocDefault) != hipSuccess) return false;
	*cap = want;
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

// waits (inside a host function: the internal stream stands still meanwhile) until every rank's counter has reached seq
bool wait_all(Comm *c, std::atomic<uint64_t> *counter, uint64_t seq, const char *what) {
	const double t0 = now();
	for (int r = 0; r < c->world; ++r)
		while (counter[r].load(std::memory_order_acquire) < seq) {
			if (c->ctl->failed.load()) return false;
			if (now() - t0 > TIMEOUT_S) {
				c->ctl->failed.store(1);
				std::fprintf(stderr, "rccl_peer_double: rank %d waited %.0f s for rank %d (%s, call %llu)\n", c->rank, TIMEOUT_S, r, what, (unsigned long long)seq);
				return false;
			}
			std::this_thread::sleep_for(std::chrono::microseconds(20));
		}
	return true;
}

// The host function of a call, on the internal stream between the device -> staging copies and the staging -> device ones:
// my bytes into my outbox (once every rank has read the previous call's), then the peers' bytes out of theirs.
void carry(void *user) {
	Job *j = (Job *)user;
	Comm *c = j->comm;
	if (c->ctl->failed.load()) return;
	if (!wait_all(c, c->ctl->consumed, j->seq - 1, "outbox still being read")) return;
	if (c->delay_ms) std::this_thread::sleep_for(std::chrono::milliseconds(c->delay_ms));
	if (j->out_bytes) std::memcpy(c->box[c->rank].p, c->stage_out, j->out_bytes);
	c->ctl->ready[c->rank].store(j->seq, std::memory_order_release);
	if (!wait_all(c, c->ctl->ready, j->seq, "peer bytes")) return;
	if (j->kind == 3) {
		uint64_t *acc = (uint64_t *)c->stage_in;
		std::memcpy(acc, c->box[c->rank].p, j->reduce_count * 8);
		for (int r = 0; r < c->world; ++r) {
			if (r == c->rank) continue;
			const uint64_t *q = (const uint64_t *)c->box[r].p;
			for (size_t i = 0; i < j->reduce_count; ++i) acc[i] = q[i] < acc[i] ? q[i] : acc[i];
		}
	} else {
		for (const Piece &p : j->recv) std::memcpy(c->stage_in + p.stage_off, c->box[p.peer].p + p.peer_off, p.bytes);
	}
	c->ctl->consumed[c->rank].store(j->seq, std::memory_order_release);
}

// the stream-ordered half of a call: sends (device -> staging_out) after what `s` holds now, the carrying host function,
// receives (staging_in -> device), and `s` waits for all of it.  Returns at once.
ncclResult_t enqueue(Comm *c, hipStream_t s, Job *j, const std::vector<Piece> &sends, const std::vector<Piece> &recv_to_device) {
	hipEvent_t e1 = nullptr, e2 = nullptr;
	bool ok = hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
	ok = ok && hipEventRecord(e1, s) == hipSuccess && hipStreamWaitEvent(c->is, e1, 0) == hipSuccess;
	for (const Piece &p : sends)
		ok = ok && hipMemcpyAsync(c->stage_out + p.stage_off, p.dev, p.bytes, hipMemcpyDeviceToHost, c->is) == hipSuccess;
	c->jobs.push_back(j);
	ok = ok && hipLaunchHostFunc(c->is, carry, j) == hipSuccess;
	for (const Piece &p : recv_to_device)
		ok = ok && hipMemcpyAsync(p.dev, c->stage_in + p.stage_off, p.bytes, hipMemcpyHostToDevice, c->is) == hipSuccess;
	ok = ok && hipEventRecord(e2, c->is) == hipSuccess && hipStreamWaitEvent(s, e2, 0) == hipSuccess;
	if (e1) (void)hipEventDestroy(e1); // (the waits hold their own references to the recorded states)
	if (e2) (void)hipEventDestroy(e2);
	return ok ? ncclSuccess : fail(c, "enqueueing the transfers");
}

// the host half of every call: publish what this rank is about to do, meet the peers, check that they do the same
ncclResult_t meet(Comm *c, uint64_t kind, uint64_t bytes, size_t out_bytes) {
	if (c->ctl->failed.load()) return ncclSystemError;
	++c->seq;
	c->ctl->meta[c->rank] = {c->seq, kind, bytes};
	if (!map_box(c, c->rank, out_bytes ? out_bytes : 1, true)) return fail(c, "outbox");
	const uint64_t have = c->ctl->box_bytes[c->rank].load();
	if ((out_bytes ? out_bytes : 1) > have) c->ctl->box_bytes[c->rank].store(c->box[c->rank].mapped);
	if (!barrier(c)) return ncclSystemError;
	for (int r = 0; r < c->world; ++r) {
		const Control::Meta m = c->ctl->meta[r];
		if (m.seq != c->seq || m.kind != kind) return fail(c, "the ranks are not in the same call (another collective, or another number of calls so far)");
		if (kind != 1 && m.bytes != bytes) return fail(c, kind == 2 ? "ncclAllGather: the ranks disagree on the block size" : "ncclAllReduce: the ranks disagree on the count");
	}
	return ncclSuccess;
}

ncclResult_t run_group() {
	std::vector<Pending> ops;
	ops.swap(g_pending);
	if (ops.empty()) return ncclSuccess;
	Comm *c = ops[0].comm;
	hipStream_t s = ops[0].stream;
	std::lock_guard<std::mutex> lk(c->mu);
	for (auto &o : ops)
		if (o.comm != c || o.stream != s) return fail(c, "one group, two communicators or streams: not what csrc/comm.cpp does");
	// sends: packed into my outbox, one directory entry per destination
	for (int r = 0; r < c->world; ++r) c->ctl->dir[c->rank][r] = {0, 0};
	size_t total = 0, recv_total = 0;
	bool seen[MAX_RANKS] = {}, got[MAX_RANKS] = {};
	std::vector<Piece> sends, to_device;
	for (auto &o : ops) {
		if (o.send) {
			if (seen[o.peer]) return fail(c, "two sends to one peer in one group");
			seen[o.peer] = true;
			c->ctl->dir[c->rank][o.peer] = {total, o.bytes};
			sends.push_back({o.buf, total, o.bytes, o.peer, 0});
			total += o.bytes;
		} else {
			if (got[o.peer]) return fail(c, "two receives from one peer in one group");
			got[o.peer] = true;
			recv_total += o.bytes;
		}
	}
	const ncclResult_t m = meet(c, 1, 0, total);
	if (m != ncclSuccess) return m;
	// receives: the sender's directory must name exactly the bytes this rank expects
	Job *j = new Job{c, c->seq, 1, total, {}, 0};
	ncclResult_t rc = ncclSuccess;
	size_t off = 0;
	for (auto &o : ops)
		if (!o.send && rc == ncclSuccess) {
			const Control::Entry e = c->ctl->dir[o.peer][c->rank];
			if (e.bytes != o.bytes) {
				char msg[160];
				std::snprintf(msg, sizeof msg, "receive of %zu bytes from rank %d, which sends %llu", o.bytes, o.peer, (unsigned long long)e.bytes);
				rc = fail(c, msg);
			} else if (!map_box(c, o.peer, c->ctl->box_bytes[o.peer].load(), o.peer == c->rank)) {
				rc = fail(c, "peer outbox");
			} else {
				j->recv.push_back({nullptr, off, o.bytes, o.peer, e.off});
				to_device.push_back({o.buf, off, o.bytes, o.peer, e.off});
				off += o.bytes;
			}
		}
	// a send nobody receives is a hang with real RCCL
	if (rc == ncclSuccess)
		for (int r = 0; r < c->world; ++r)
			if (c->ctl->dir[r][c->rank].bytes && !got[r]) rc = fail(c, "a peer sends to this rank, which posts no receive");
	if (rc == ncclSuccess && (!grow_stage(c, &c->stage_out, &c->stage_out_cap, total) || !grow_stage(c, &c->stage_in, &c->stage_in_cap, recv_total)))
		rc = fail(c, "staging");
	if (!barrier(c) && rc == ncclSuccess) rc = ncclSystemError; // (the directory is rewritten by the next call)
	if (rc != ncclSuccess) {
		delete j;
		return rc;
	}
	return enqueue(c, s, j, sends, to_device);
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
	if (const char *d = std::getenv("TREEE_DOUBLE_DELAY_MS")) c->delay_ms = std::atoi(d);
	const std::string name = "/treee_rccl_double_" + c->token + "_ctl";
	const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) return ncclSystemError; // (new bytes are zero: a valid initial state)
	void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) return ncclSystemError;
	c->ctl = (Control *)p;
	if (hipStreamCreateWithFlags(&c->is, hipStreamNonBlocking) != hipSuccess) return ncclSystemError;
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
	if (c->is) {
		(void)hipStreamSynchronize(c->is); // every enqueued call has run (or given up after its timeout)
		(void)hipStreamDestroy(c->is);
	}
	for (Job *j : c->jobs) delete j;
	if (c->stage_out) (void)hipHostFree(c->stage_out);
	if (c->stage_in) (void)hipHostFree(c->stage_in);
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
	std::lock_guard<std::mutex> lk(c->mu);
	const size_t bytes = count * type_bytes(type);
	if (!bytes) return fail(c, "ncclAllGather of nothing (csrc/comm.cpp returns before the call)");
	const ncclResult_t m = meet(c, 2, bytes, bytes);
	if (m != ncclSuccess) return m;
	ncclResult_t rc = ncclSuccess;
	Job *j = new Job{c, c->seq, 2, bytes, {}, 0};
	std::vector<Piece> sends{{(void *)send, 0, bytes, c->rank, 0}}, to_device;
	for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
		if (!map_box(c, r, c->ctl->box_bytes[r].load(), r == c->rank)) rc = fail(c, "peer outbox");
		j->recv.push_back({nullptr, (size_t)r * bytes, bytes, r, 0});
	}
	to_device.push_back({recv, 0, bytes * (size_t)c->world, 0, 0});
	if (rc == ncclSuccess && (!grow_stage(c, &c->stage_out, &c->stage_out_cap, bytes) || !grow_stage(c, &c->stage_in, &c->stage_in_cap, bytes * (size_t)c->world)))
		rc = fail(c, "staging");
	if (!barrier(c) && rc == ncclSuccess) rc = ncclSystemError;
	if (rc != ncclSuccess) {
		delete j;
		return rc;
	}
	return enqueue(c, s, j, sends, to_device);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
	Comm *c = (Comm *)comm;
	std::lock_guard<std::mutex> lk(c->mu);
	if (type != ncclUint64 || op != ncclMin) return fail(c, "ncclAllReduce: only MIN over u64 is what csrc/comm.cpp asks for");
	const size_t bytes = count * 8;
	if (!bytes) return fail(c, "ncclAllReduce of nothing");
	const ncclResult_t m = meet(c, 3, bytes, bytes);
	if (m != ncclSuccess) return m;
	ncclResult_t rc = ncclSuccess;
	Job *j = new Job{c, c->seq, 3, bytes, {}, count};
	for (int r = 0; r < c->world && rc == ncclSuccess; ++r)
		if (!map_box(c, r, c->ctl->box_bytes[r].load(), r == c->rank)) rc = fail(c, "peer outbox");
	std::vector<Piece> sends{{(void *)send, 0, bytes, c->rank, 0}}, to_device{{recv, 0, bytes, 0, 0}};
	if (rc == ncclSuccess && (!grow_stage(c, &c->stage_out, &c->stage_out_cap, bytes) || !grow_stage(c, &c->stage_in, &c->stage_in_cap, bytes))) rc = fail(c, "staging");
	if (!barrier(c) && rc == ncclSuccess) rc = ncclSystemError;
	if (rc != ncclSuccess) {
		delete j;
		return rc;
	}
	return enqueue(c, s, j, sends, to_device);
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
