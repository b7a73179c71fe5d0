// comm.h -- the communicator behind treee_comm_* (comm.cpp): what the sharded entry points need from a transport
#pragma once
#include "common.h"

struct treee_comm {
	int rank = 0, world = 1;
	bool rccl = false;
	virtual ~treee_comm() {}
	// device payloads.  RCCL: enqueued on ctx->stream, nothing synchronises.  Host transport: stages through host memory and
	// synchronises the stream.  Offsets and sizes in bytes.
	virtual int32_t all_gather_dev(treee_ctx *ctx, const void *d_send, void *d_recv, size_t bytes) = 0;
	virtual int32_t all_to_all_v_dev(treee_ctx *ctx, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
	                                 const uint64_t *recv_off, const uint64_t *recv_bytes) = 0;
	virtual int32_t all_reduce_min_u64_dev(treee_ctx *ctx, uint64_t *d_buf, size_t count) = 0;
	// small control data in host memory (counts, cubes); synchronises
	virtual int32_t all_gather_host(treee_ctx *ctx, const void *send, void *recv, size_t bytes) = 0;
};
