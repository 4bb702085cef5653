/*
 * mdb_dist.hip - the multi-GPU exchange behind the C-ABI (include/mdb_dist.h): RCCL communicators, the
 * per-table sequence  partition by destination -> counts -> uneven all-to-all,  and the sharded north-star
 * operator built on the split form of the fused join + GROUP BY (mdb_dev_join.hip).  One process per GPU.
 *
 * xGMI is what bounds this step, not HBM: at 8 ranks 7/8 of every table leaves each GPU, one
 * eighth per point-to-point link.  So the sequence keeps the links busy - table L's all-to-all runs while
 * table R is partitioned by destination, table R's while the received L is hashed and radix-partitioned
 * locally - and ships 4-byte keys whenever the column statistics allow it.  Transfers run on a stream of
 * their own; the count exchanges on a communicator of their own.  The only host synchronisations are the ones
 * the sizes force: the send counts come from the destination partition (exact layout: it ends with a
 * read-back), the receive counts from the count exchange (128 bytes).
 */
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <rccl/rccl.h>
#include "mdb_dev_internal.h"
#include "mdb_dist.h"

#define SH_MAX_PENDING 1024

struct mdb_dist {
	mdb_dev_ctx *ctx;
	int world, rank;
	mdb_dist_transport t;
	bool own_transport;		/* t.self is an rccl_transport created here */
	hipStream_t comm_stream;	/* key transfers */
	hipEvent_t ev_ready, ev_a, ev_b;
	int wire_mode, last_wire32, last_pruned, last_fused;
	bool have_ranges;		/* mdb_dist_set_key_ranges(): promised global key ranges of the two tables of the next calls */
	int64_t promised_lo[2], promised_hi[2];
	/* exchange buffers (grow-only): send = this rank's keys grouped by destination, recv = what arrived */
	void *send[2], *recv[2];
	uint64_t send_cap[2], recv_cap[2];	/* in 8-byte words */
	uint64_t last_recv_left;
	/* row shuffles (mdb_dist_shuffle_rows): buffers the posted transfers still read - released by mdb_dist_wait_transfers() */
	hipEvent_t ev_sh;
	hipEvent_t ev_tab[MDB_SHARD_MAX_TABS];	/* the sharded operator: table x has arrived */
	/* what the last regions-on-the-wire call planned (mdb_dist_last_plan) and, on request, where its time went (mdb_dist_last_phases) */
	mdb_shard_plan last_plan;
	bool have_plan;
	bool time_phases;
	hipEvent_t ev_t0, ev_t1, ev_part[MDB_SHARD_MAX_TABS], ev_arr[MDB_SHARD_MAX_TABS];	/* (created on first use, with timing) */
	double phase_ms[MDB_DIST_PHASES];
	uint64_t seq;			/* collective calls made through this handle: every rank's must agree (checked with the counts) */
	bool ko_had_dups;		/* the last keys-only join through this handle met a key twice on some rank (the same on every rank) */
	bool sh_posted;
	void *pend[SH_MAX_PENDING];
	int npend;
	char err[512];
};

static const char *transport_err(mdb_dist *d);

static int dist_err(mdb_dist *d, int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(d->err, sizeof(d->err), fmt, ap);
	va_end(ap);
	return code;
}

#define DIST_HIP(d, call)                                                                              \
	do {                                                                                           \
		hipError_t e__ = (call);                                                               \
		if (e__ != hipSuccess)                                                                 \
			return dist_err((d), -MIDORIDB_INTERNAL, "%s failed: %s (%s:%d)", #call,          \
					hipGetErrorString(e__), __FILE__, __LINE__);                   \
	} while (0)

/* ------------------------------------------------------------------ RCCL transport */

struct rccl_transport {
	ncclComm_t data, small;		/* key transfers / counts + reductions (always on small_stream, host-synchronous) */
	ncclComm_t status;		/* the sharded operator's status all-reduce, always on the context's stream: a communicator of its own,
					 * so that no communicator is ever used on two streams (operations of one communicator must be issued in
					 * the same order on every rank; with one stream per communicator program order is that order) */
	int world, rank, device;
	hipStream_t small_stream;
	uint64_t *d_buf, *h_buf;	/* 2 x world x 8 counters: device staging + pinned mirror */
	char err[256];
};

#define RT_MAX_COUNTERS 12

static int rt_fail(rccl_transport *rt, const char *what, ncclResult_t r)
{
	snprintf(rt->err, sizeof(rt->err), "%s: %s", what, ncclGetErrorString(r));
	return -MIDORIDB_INTERNAL;
}

static int rt_counts(void *self, const uint64_t *send, uint64_t *recv, int n)
{
	rccl_transport *rt = (rccl_transport *)self;
	if (n < 1 || n > RT_MAX_COUNTERS)
		return -MIDORIDB_ERROR;
	const size_t words = (size_t)rt->world * (size_t)n;
	uint64_t *d_send = rt->d_buf, *d_recv = rt->d_buf + (size_t)rt->world * RT_MAX_COUNTERS;
	uint64_t *h_send = rt->h_buf, *h_recv = rt->h_buf + (size_t)rt->world * RT_MAX_COUNTERS;
	memcpy(h_send, send, words * 8);
	if (hipMemcpyAsync(d_send, h_send, words * 8, hipMemcpyHostToDevice, rt->small_stream) != hipSuccess)
		return -MIDORIDB_INTERNAL;
	ncclResult_t r = ncclAllToAll(d_send, d_recv, (size_t)n, ncclUint64, rt->small, rt->small_stream);
	if (r != ncclSuccess)
		return rt_fail(rt, "ncclAllToAll(counts)", r);
	if (hipMemcpyAsync(h_recv, d_recv, words * 8, hipMemcpyDeviceToHost, rt->small_stream) != hipSuccess ||
	    hipStreamSynchronize(rt->small_stream) != hipSuccess)
		return -MIDORIDB_INTERNAL;
	memcpy(recv, h_recv, words * 8);
	return MIDORIDB_OK;
}

static int rt_alltoallv(void *self, const void *d_send, const size_t *sendcounts, const size_t *sdispls, void *d_recv,
			const size_t *recvcounts, const size_t *rdispls, size_t elem_bytes, void *stream)
{
	rccl_transport *rt = (rccl_transport *)self;
	const ncclDataType_t ty = elem_bytes == 8 ? ncclUint64 : (elem_bytes == 4 ? ncclUint32 : ncclUint8);
	if (elem_bytes != 8 && elem_bytes != 4 && elem_bytes != 1)
		return -MIDORIDB_ERROR;
	/* the block a rank keeps for itself does not go through the collective: RCCL moves a self-send with a copy kernel of a few
	 * workgroups (~1 TB/s: 0.2 ms of the forced-shuffle step at world 1 for 0.2 GB), a device-to-device copy on the same stream runs
	 * at the HBM rate - and at world 1 nothing is left for the collective at all */
	size_t sc[1 << MDB_MAX_RADIX_BITS], rc[1 << MDB_MAX_RADIX_BITS];
	for (int p = 0; p < rt->world; p++) {
		sc[p] = sendcounts[p];
		rc[p] = recvcounts[p];
	}
	const size_t self_n = sc[rt->rank] < rc[rt->rank] ? sc[rt->rank] : rc[rt->rank];
	if (sc[rt->rank] == rc[rt->rank]) {
		if (self_n && hipMemcpyAsync((char *)d_recv + rdispls[rt->rank] * elem_bytes, (const char *)d_send + sdispls[rt->rank] * elem_bytes,
					     self_n * elem_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
			snprintf(rt->err, sizeof(rt->err), "copying this rank's own block failed");
			return -MIDORIDB_INTERNAL;
		}
		sc[rt->rank] = rc[rt->rank] = 0;
	}
	if (rt->world == 1 && sc[0] == 0)
		return MIDORIDB_OK;
	ncclResult_t r = ncclAllToAllv(d_send, sc, sdispls, d_recv, rc, rdispls, ty, rt->data, (hipStream_t)stream);
	if (r != ncclSuccess)
		return rt_fail(rt, "ncclAllToAllv", r);
	return MIDORIDB_OK;
}

static int rt_allreduce(void *self, uint64_t *vals, int n)
{
	rccl_transport *rt = (rccl_transport *)self;
	if (n < 1 || n > rt->world * RT_MAX_COUNTERS)
		return -MIDORIDB_ERROR;
	memcpy(rt->h_buf, vals, (size_t)n * 8);
	if (hipMemcpyAsync(rt->d_buf, rt->h_buf, (size_t)n * 8, hipMemcpyHostToDevice, rt->small_stream) != hipSuccess)
		return -MIDORIDB_INTERNAL;
	ncclResult_t r = ncclAllReduce(rt->d_buf, rt->d_buf, (size_t)n, ncclUint64, ncclSum, rt->small, rt->small_stream);
	if (r != ncclSuccess)
		return rt_fail(rt, "ncclAllReduce", r);
	if (hipMemcpyAsync(rt->h_buf, rt->d_buf, (size_t)n * 8, hipMemcpyDeviceToHost, rt->small_stream) != hipSuccess ||
	    hipStreamSynchronize(rt->small_stream) != hipSuccess)
		return -MIDORIDB_INTERNAL;
	memcpy(vals, rt->h_buf, (size_t)n * 8);
	return MIDORIDB_OK;
}

static void rt_destroy(void *self)
{
	rccl_transport *rt = (rccl_transport *)self;
	if (!rt)
		return;
	if (rt->data)
		ncclCommDestroy(rt->data);
	if (rt->small)
		ncclCommDestroy(rt->small);
	if (rt->status)
		ncclCommDestroy(rt->status);
	if (rt->small_stream)
		(void)hipStreamDestroy(rt->small_stream);
	if (rt->d_buf)
		(void)hipFree(rt->d_buf);
	if (rt->h_buf)
		(void)hipHostFree(rt->h_buf);
	free(rt);
}

extern "C" int mdb_dist_unique_id(void *id_out)
{
	ncclUniqueId id;
	static_assert(sizeof(id) == MDB_DIST_ID_BYTES, "RCCL unique id size");
	if (!id_out || ncclGetUniqueId(&id) != ncclSuccess)
		return -MIDORIDB_INTERNAL;
	memcpy(id_out, &id, sizeof(id));
	return MIDORIDB_OK;
}

/* The id through a file, safe against what an earlier run left behind.  A file that merely holds an id cannot be told from a
 * stale one, and a rank that initialises RCCL with a stale id never comes back.  So every rank r > 0 announces itself with a
 * fresh random nonce (path.hello.<r>), rank 0 - which removes any old id file first - writes the id together with the nonces
 * it has seen, and rank r only accepts a file that carries ITS nonce, then withdraws its announcement; rank 0 returns when
 * every announcement is gone (and rewrites the file when it finds one that changed: a leftover of a crashed run, overwritten
 * by the live rank a moment later). */
#define IDF_MAGIC 0x3230304449424D4Dull	/* "MMBIDI002"-ish: layout version of the id file */

static double idf_now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + (double)t.tv_nsec * 1e-9;
}

static uint64_t idf_nonce(void)
{
	uint64_t v = 0;
	FILE *f = fopen("/dev/urandom", "rb");
	if (f) {
		if (fread(&v, 1, sizeof(v), f) != sizeof(v))
			v = 0;
		fclose(f);
	}
	struct timespec t;
	clock_gettime(CLOCK_REALTIME, &t);
	v ^= mdb_fmix64((uint64_t)t.tv_nsec ^ ((uint64_t)t.tv_sec << 30) ^ ((uint64_t)getpid() << 48));
	return v ? v : 1;
}

static int idf_write_atomic(const char *path, const void *buf, size_t len)
{
	char tmp[4096];
	if (snprintf(tmp, sizeof(tmp), "%s.tmp.%ld", path, (long)getpid()) >= (int)sizeof(tmp))
		return -MIDORIDB_ERROR;
	FILE *f = fopen(tmp, "wb");
	if (!f)
		return -MIDORIDB_ERROR;
	const bool ok = fwrite(buf, 1, len, f) == len;
	if (fclose(f) != 0 || !ok || rename(tmp, path) != 0) {
		(void)remove(tmp);
		return -MIDORIDB_ERROR;
	}
	return MIDORIDB_OK;
}

static bool idf_read(const char *path, void *buf, size_t len)
{
	FILE *f = fopen(path, "rb");
	if (!f)
		return false;
	const size_t got = fread(buf, 1, len, f);
	fclose(f);
	return got == len;
}

extern "C" int mdb_dist_id_via_file(const char *path, int world, int rank, double timeout_s, void *id_out)
{
	if (!path || !id_out || world < 1 || rank < 0 || rank >= world || world > (1 << MDB_MAX_RADIX_BITS))
		return -MIDORIDB_ERROR;
	char hello[4096];
	const size_t flen = 16 + 8 * (size_t)world + MDB_DIST_ID_BYTES;	/* magic, world, nonce of every rank (slot 0 unused), id */
	uint64_t file[2 + (1 << MDB_MAX_RADIX_BITS) + MDB_DIST_ID_BYTES / 8];
	const double t0 = idf_now();
	if (rank == 0) {
		uint64_t known[1 << MDB_MAX_RADIX_BITS];
		bool have[1 << MDB_MAX_RADIX_BITS], acked[1 << MDB_MAX_RADIX_BITS];
		(void)remove(path);		/* whatever an earlier run left */
		int rc = mdb_dist_unique_id(id_out);
		if (rc)
			return rc;
		for (int r = 0; r < world; r++) {
			known[r] = 0;
			have[r] = acked[r] = r == 0;
		}
		bool written = false;
		for (;;) {
			bool changed = false, all_have = true, all_acked = true;
			for (int r = 1; r < world; r++) {
				if (acked[r])
					continue;
				uint64_t n = 0;
				if (snprintf(hello, sizeof(hello), "%s.hello.%d", path, r) >= (int)sizeof(hello))
					return -MIDORIDB_ERROR;
				if (idf_read(hello, &n, sizeof(n)) && n) {
					if (n != known[r]) {
						known[r] = n;
						changed = true;
					}
					have[r] = true;
				} else if (have[r] && written && access(hello, F_OK) != 0) {
					acked[r] = true;	/* the rank took the id and withdrew its announcement */
				}
				all_have = all_have && have[r];
				all_acked = all_acked && acked[r];
			}
			if (all_have && (changed || !written)) {
				file[0] = IDF_MAGIC;
				file[1] = (uint64_t)world;
				for (int r = 0; r < world; r++)
					file[2 + r] = known[r];
				memcpy(&file[2 + world], id_out, MDB_DIST_ID_BYTES);
				rc = idf_write_atomic(path, file, flen);
				if (rc)
					return rc;
				written = true;
			}
			if (all_acked && (written || world == 1))
				return MIDORIDB_OK;
			if (idf_now() - t0 > timeout_s)
				return -MIDORIDB_ERROR;
			usleep(1000);
		}
	}
	const uint64_t mine = idf_nonce();
	if (snprintf(hello, sizeof(hello), "%s.hello.%d", path, rank) >= (int)sizeof(hello))
		return -MIDORIDB_ERROR;
	int rc = idf_write_atomic(hello, &mine, sizeof(mine));
	if (rc)
		return rc;
	for (;;) {
		if (idf_read(path, file, flen) && file[0] == IDF_MAGIC && file[1] == (uint64_t)world && file[2 + rank] == mine) {
			memcpy(id_out, &file[2 + world], MDB_DIST_ID_BYTES);
			(void)remove(hello);
			return MIDORIDB_OK;
		}
		if (idf_now() - t0 > timeout_s) {
			(void)remove(hello);
			return -MIDORIDB_ERROR;
		}
		usleep(1000);
	}
}

/* ------------------------------------------------------------------ handle */

static int dist_new(mdb_dev_ctx *ctx, int world, int rank, mdb_dist **out)
{
	if (!ctx || !out || world < 1 || rank < 0 || rank >= world || world > (1 << MDB_MAX_RADIX_BITS))
		return -MIDORIDB_ERROR;
	mdb_dist *d = (mdb_dist *)calloc(1, sizeof(*d));
	if (!d)
		return -MIDORIDB_NOMEM;
	d->ctx = ctx;
	d->world = world;
	d->rank = rank;
	d->wire_mode = MDB_WIRE_AUTO;
	if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_a, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_b, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_sh, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_tab[0], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_tab[1], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_tab[2], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->ev_tab[3], hipEventDisableTiming) != hipSuccess) {
		mdb_dist_destroy(d);
		return -MIDORIDB_INTERNAL;
	}
	*out = d;
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_init(mdb_dev_ctx *ctx, int world, int rank, const void *id, mdb_dist **out)
{
	if (!id || !out)
		return -MIDORIDB_ERROR;
	*out = NULL;
	mdb_dist *d = NULL;
	int rc = dist_new(ctx, world, rank, &d);
	if (rc)
		return rc;
	rccl_transport *rt = (rccl_transport *)calloc(1, sizeof(*rt));
	if (!rt) {
		mdb_dist_destroy(d);
		return -MIDORIDB_NOMEM;
	}
	rt->world = world;
	rt->rank = rank;
	rt->device = ctx->device;
	d->t.self = rt;
	d->t.counts = rt_counts;
	d->t.alltoallv = rt_alltoallv;
	d->t.allreduce_sum_u64 = rt_allreduce;
	d->t.destroy = rt_destroy;
	d->own_transport = true;
	ncclUniqueId uid;
	memcpy(&uid, id, sizeof(uid));
	ncclResult_t r = ncclCommInitRank(&rt->data, world, uid, rank);
	if (r != ncclSuccess) {
		snprintf(ctx->err, sizeof(ctx->err), "ncclCommInitRank: %s", ncclGetErrorString(r));
		mdb_dist_destroy(d);
		return -MIDORIDB_INTERNAL;
	}
	const size_t words = (size_t)world * RT_MAX_COUNTERS * 2;
	if (hipStreamCreateWithFlags(&rt->small_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc((void **)&rt->d_buf, (words > 64 ? words : 64) * 8) != hipSuccess ||
	    hipHostMalloc((void **)&rt->h_buf, (words > 64 ? words : 64) * 8) != hipSuccess) {
		mdb_dist_destroy(d);
		return -MIDORIDB_INTERNAL;
	}
	/* the further communicators each get an id of their own, agreed on through the first: rank 0 creates it and every rank
	 * learns it from an all-reduce (the other ranks contribute zeros) */
	ncclComm_t *further[2] = { &rt->small, &rt->status };
	for (int c = 0; c < 2; c++) {
		ncclUniqueId uid2;
		memset(&uid2, 0, sizeof(uid2));
		if (rank == 0 && ncclGetUniqueId(&uid2) != ncclSuccess) {
			mdb_dist_destroy(d);
			return -MIDORIDB_INTERNAL;
		}
		uint64_t *h = rt->h_buf;
		memcpy(h, &uid2, sizeof(uid2));
		r = ncclSuccess;
		if (hipMemcpyAsync(rt->d_buf, h, sizeof(uid2), hipMemcpyHostToDevice, rt->small_stream) != hipSuccess)
			r = ncclSystemError;
		if (r == ncclSuccess)
			r = ncclAllReduce(rt->d_buf, rt->d_buf, sizeof(uid2) / 8, ncclUint64, ncclSum, rt->data, rt->small_stream);
		if (r == ncclSuccess &&
		    (hipMemcpyAsync(h, rt->d_buf, sizeof(uid2), hipMemcpyDeviceToHost, rt->small_stream) != hipSuccess ||
		     hipStreamSynchronize(rt->small_stream) != hipSuccess))
			r = ncclSystemError;
		if (r == ncclSuccess) {
			memcpy(&uid2, h, sizeof(uid2));
			r = ncclCommInitRank(further[c], world, uid2, rank);
		}
		if (r != ncclSuccess) {
			snprintf(ctx->err, sizeof(ctx->err), "communicator %d: %s", c + 2, ncclGetErrorString(r));
			mdb_dist_destroy(d);
			return -MIDORIDB_INTERNAL;
		}
	}
	*out = d;
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_init_transport(mdb_dev_ctx *ctx, int world, int rank, const struct mdb_dist_transport *t, mdb_dist **out)
{
	if (!t || !t->counts || !t->alltoallv || !t->allreduce_sum_u64 || !out)
		return -MIDORIDB_ERROR;
	*out = NULL;
	mdb_dist *d = NULL;
	int rc = dist_new(ctx, world, rank, &d);
	if (rc)
		return rc;
	d->t = *t;
	d->own_transport = false;
	*out = d;
	return MIDORIDB_OK;
}

extern "C" void mdb_dist_destroy(mdb_dist *d)
{
	if (!d)
		return;
	if (d->ctx)
		(void)hipSetDevice(d->ctx->device);
	if (d->comm_stream)
		(void)hipStreamSynchronize(d->comm_stream);
	if (d->t.destroy && d->t.self)
		d->t.destroy(d->t.self);
	for (int i = 0; i < 2; i++) {
		if (d->send[i])
			(void)hipFree(d->send[i]);
		if (d->recv[i])
			(void)hipFree(d->recv[i]);
	}
	if (d->ev_ready)
		(void)hipEventDestroy(d->ev_ready);
	if (d->ev_a)
		(void)hipEventDestroy(d->ev_a);
	if (d->ev_b)
		(void)hipEventDestroy(d->ev_b);
	if (d->ev_sh)
		(void)hipEventDestroy(d->ev_sh);
	for (int x = 0; x < MDB_SHARD_MAX_TABS; x++) {
		if (d->ev_tab[x])
			(void)hipEventDestroy(d->ev_tab[x]);
		if (d->ev_part[x])
			(void)hipEventDestroy(d->ev_part[x]);
		if (d->ev_arr[x])
			(void)hipEventDestroy(d->ev_arr[x]);
	}
	if (d->ev_t0)
		(void)hipEventDestroy(d->ev_t0);
	if (d->ev_t1)
		(void)hipEventDestroy(d->ev_t1);
	for (int i = 0; i < d->npend; i++)
		(void)mdb_dev_free(d->ctx, d->pend[i]);
	if (d->comm_stream)
		(void)hipStreamDestroy(d->comm_stream);
	free(d);
}

extern "C" int mdb_dist_world(const mdb_dist *d) { return d ? d->world : 0; }
extern "C" int mdb_dist_rank(const mdb_dist *d) { return d ? d->rank : -1; }
extern "C" const char *mdb_dist_last_error(const mdb_dist *d) { return d ? d->err : "no distributed handle"; }
extern "C" int mdb_dist_last_wire32(const mdb_dist *d) { return d ? d->last_wire32 : 0; }
extern "C" int mdb_dist_last_pruned(const mdb_dist *d) { return d ? d->last_pruned : 0; }
extern "C" int mdb_dist_last_fused(const mdb_dist *d) { return d ? d->last_fused : 0; }

static void plan_info_fill(const mdb_shard_plan *p, bool completed, struct mdb_dist_plan_info *out);

extern "C" int mdb_dist_last_plan(const mdb_dist *d, struct mdb_dist_plan_info *out)
{
	if (!d || !out)
		return -MIDORIDB_ERROR;
	memset(out, 0, sizeof(*out));
	if (!d->have_plan)
		return 1;
	plan_info_fill(&d->last_plan, d->last_fused != 0, out);
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_plan_preview(int world, int tables, const uint64_t *rows_per_rank, const int64_t left[2], const int64_t right[2],
				     struct mdb_dist_plan_info *out)
{
	if (!out || !rows_per_rank || !left || !right || tables < 2 || tables > MDB_SHARD_MAX_TABS)
		return -MIDORIDB_ERROR;
	memset(out, 0, sizeof(*out));
	mdb_shard_plan plan;
	if (mdb_shard_plan_make((uint32_t)world, 0u, (uint32_t)tables, rows_per_rank, left[0], left[1], right[0], right[1], &plan))
		return 1;
	plan_info_fill(&plan, false, out);
	return MIDORIDB_OK;
}

static void plan_info_fill(const mdb_shard_plan *p, bool completed, struct mdb_dist_plan_info *out)
{
	out->world = p->world;
	out->tables = p->ntab;
	out->digit_bits = p->dbits;
	out->digits_per_rank = p->Dp;
	out->key_bits = p->kbits;
	out->receiver_bits = (uint32_t)p->b2;
	out->leaf_bits = p->rem;
	out->word_bytes = p->wbytes;
	out->completed = completed ? 1u : 0u;
	for (uint32_t x = 0; x < p->ntab; x++) {
		if (p->right_only && x == 0)
			continue;	/* (GROUP BY of one table: there is no left table, nothing of it travels) */
		out->region_words[x] = p->cap[x];
		out->block_bytes[x] = p->block_words[x] * p->wbytes;
		out->bytes_per_peer += p->block_words[x] * p->wbytes + (uint64_t)p->D * p->nsub * 4;
	}
}

extern "C" int mdb_dist_set_phase_timing(mdb_dist *d, int on)
{
	if (!d)
		return -MIDORIDB_ERROR;
	if (on && !d->ev_t0) {
		DIST_HIP(d, hipSetDevice(d->ctx->device));
		DIST_HIP(d, hipEventCreate(&d->ev_t0));
		DIST_HIP(d, hipEventCreate(&d->ev_t1));
		for (int x = 0; x < MDB_SHARD_MAX_TABS; x++) {
			DIST_HIP(d, hipEventCreate(&d->ev_part[x]));
			DIST_HIP(d, hipEventCreate(&d->ev_arr[x]));
		}
	}
	d->time_phases = on != 0;
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_last_phases(const mdb_dist *d, double *ms)
{
	if (!d || !ms)
		return -MIDORIDB_ERROR;
	memcpy(ms, d->phase_ms, sizeof(d->phase_ms));
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_set_key_ranges(mdb_dist *d, const int64_t left[2], const int64_t right[2])
{
	if (!d)
		return -MIDORIDB_ERROR;
	d->have_ranges = left && right;
	if (d->have_ranges) {
		d->promised_lo[0] = left[0];
		d->promised_hi[0] = left[1];
		d->promised_lo[1] = right[0];
		d->promised_hi[1] = right[1];
	}
	return MIDORIDB_OK;
}
extern "C" uint64_t mdb_dist_last_received_left(const mdb_dist *d) { return d ? d->last_recv_left : 0; }

extern "C" int mdb_dist_set_wire(mdb_dist *d, int mode)
{
	if (!d || mode < MDB_WIRE_AUTO || mode > MDB_WIRE_32)
		return -MIDORIDB_ERROR;
	d->wire_mode = mode;
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_allreduce_sum_u64(mdb_dist *d, uint64_t *vals, int n)
{
	if (!d || !vals)
		return -MIDORIDB_ERROR;
	const int rc = d->t.allreduce_sum_u64(d->t.self, vals, n);
	return rc ? dist_err(d, rc, "all-reduce failed%s%s", d->own_transport ? ": " : "", d->own_transport ? ((rccl_transport *)d->t.self)->err : "") : rc;
}

extern "C" int mdb_dist_allgather_u64(mdb_dist *d, const uint64_t *mine, int n, uint64_t *all)
{
	if (!d || !mine || !all || n < 1 || n > 8)
		return -MIDORIDB_ERROR;
	/* (values travel as 32-bit halves: a transport's counters need not survive values beyond 2^63) */
	uint64_t sendv[16 << MDB_MAX_RADIX_BITS], recvv[16 << MDB_MAX_RADIX_BITS];
	for (int p = 0; p < d->world; p++)
		for (int i = 0; i < n; i++) {
			sendv[2 * n * p + 2 * i] = mine[i] >> 32;
			sendv[2 * n * p + 2 * i + 1] = mine[i] & 0xFFFFFFFFull;
		}
	int rc = 0;
	for (int half = 0; half < 2 * n && !rc; half += RT_MAX_COUNTERS) {	/* at most RT_MAX_COUNTERS counters per exchange */
		const int cnt = 2 * n - half < RT_MAX_COUNTERS ? 2 * n - half : RT_MAX_COUNTERS;
		uint64_t s2[RT_MAX_COUNTERS << MDB_MAX_RADIX_BITS], r2[RT_MAX_COUNTERS << MDB_MAX_RADIX_BITS];
		for (int p = 0; p < d->world; p++)
			for (int i = 0; i < cnt; i++)
				s2[cnt * p + i] = sendv[2 * n * p + half + i];
		rc = d->t.counts(d->t.self, s2, r2, cnt);
		for (int p = 0; p < d->world && !rc; p++)
			for (int i = 0; i < cnt; i++)
				recvv[2 * n * p + half + i] = r2[cnt * p + i];
	}
	if (rc)
		return dist_err(d, rc, "all-gather failed%s%s", d->own_transport ? ": " : "", transport_err(d));
	for (int p = 0; p < d->world; p++)
		for (int i = 0; i < n; i++)
			all[n * p + i] = (recvv[2 * n * p + 2 * i] << 32) | recvv[2 * n * p + 2 * i + 1];
	return MIDORIDB_OK;
}

/* every rank's `n` bytes to every rank (HOST buffers; *all = malloc'd concatenation in rank order, counts[p] = rank p's bytes):
 * small variable-length payloads - the new entries of the ranks' string dictionaries.  Collective, blocking. */
extern "C" int mdb_dist_allgather_bytes(mdb_dist *d, const void *mine, uint64_t n, void **all, uint64_t *counts)
{
	if (!d || !all || !counts || (n && !mine))
		return d ? dist_err(d, -MIDORIDB_ERROR, "allgather_bytes: bad arguments") : -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = d->ctx;
	const int W = d->world;
	*all = NULL;
	DIST_HIP(d, hipSetDevice(ctx->device));
	uint64_t sendv[1 << MDB_MAX_RADIX_BITS], recvv[1 << MDB_MAX_RADIX_BITS];
	size_t sc[1 << MDB_MAX_RADIX_BITS], sd[1 << MDB_MAX_RADIX_BITS], rcn[1 << MDB_MAX_RADIX_BITS], rd[1 << MDB_MAX_RADIX_BITS];
	for (int p = 0; p < W; p++)
		sendv[p] = n;
	int rc = d->t.counts(d->t.self, sendv, recvv, 1);
	if (rc)
		return dist_err(d, rc, "count exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d));
	uint64_t total = 0;
	for (int p = 0; p < W; p++) {
		counts[p] = recvv[p];
		sc[p] = (size_t)n;
		sd[p] = 0;
		rcn[p] = (size_t)recvv[p];
		rd[p] = (size_t)total;
		total += recvv[p];
	}
	char *host = (char *)malloc(total ? total : 1);
	void *dsend = NULL, *drecv = NULL;
	int arc = host ? MIDORIDB_OK : -MIDORIDB_NOMEM;
	if (!arc)
		arc = mdb_dev_alloc(ctx, n ? n : 1, &dsend);
	if (!arc)
		arc = mdb_dev_alloc(ctx, total ? total : 1, &drecv);
	/* (the ranks agree on their buffers before anything is posted) */
	for (int p = 0; p < W; p++)
		sendv[p] = arc ? 1 : 0;
	rc = d->t.counts(d->t.self, sendv, recvv, 1);
	for (int p = 0; p < W && !rc; p++)
		if (recvv[p] && !arc)
			arc = -MIDORIDB_NOMEM;
	if (!rc && !arc && total) {
		if (n && hipMemcpyAsync(dsend, mine, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
			arc = -MIDORIDB_INTERNAL;
		if (!arc && (hipEventRecord(d->ev_ready, ctx->stream) != hipSuccess || hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0) != hipSuccess))
			arc = -MIDORIDB_INTERNAL;
		if (!arc)
			rc = d->t.alltoallv(d->t.self, dsend, sc, sd, drecv, rcn, rd, 1, d->comm_stream);
		if (!rc && !arc && (hipStreamSynchronize(d->comm_stream) != hipSuccess ||
				    hipMemcpy(host, drecv, total, hipMemcpyDeviceToHost) != hipSuccess))
			arc = -MIDORIDB_INTERNAL;
	}
	(void)mdb_dev_free(ctx, dsend);
	(void)mdb_dev_free(ctx, drecv);
	if (rc || arc) {
		free(host);
		return rc ? dist_err(d, rc, "all-gather failed%s%s", d->own_transport ? ": " : "", transport_err(d))
			  : dist_err(d, arc, "all-gather of %llu bytes: buffers or copies failed on some rank", (unsigned long long)total);
	}
	*all = host;
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_barrier(mdb_dist *d)
{
	uint64_t one = 1;
	return mdb_dist_allreduce_sum_u64(d, &one, 1);
}

static const char *transport_err(mdb_dist *d)
{
	return d->own_transport ? ((rccl_transport *)d->t.self)->err : "";
}

/* ------------------------------------------------------------------ the exchange */

static int dist_reserve(mdb_dist *d, void **buf, uint64_t *cap, uint64_t words)
{
	if (words <= *cap)
		return MIDORIDB_OK;
	DIST_HIP(d, hipStreamSynchronize(d->comm_stream));	/* nothing may still read the old buffer */
	DIST_HIP(d, hipStreamSynchronize(d->ctx->stream));
	if (*buf)
		DIST_HIP(d, hipFree(*buf));
	*buf = NULL;
	*cap = 0;
	const uint64_t want = words + words / 8 + 4096;
	DIST_HIP(d, hipMalloc(buf, want * 8));
	*cap = want;
	return MIDORIDB_OK;
}

/* one table: partition by destination into send[i], exchange counts, post the all-to-all into recv[i] on the transfer
 * stream.  *n_recv = rows this rank receives (they are in recv[i] once `done` has happened) */
static int dist_send_table(mdb_dist *d, int i, const int64_t *keys, const uint64_t *nulls, uint64_t n, bool wire32, uint64_t *n_recv,
			   hipEvent_t done, int64_t keep_lo = INT64_MIN, int64_t keep_hi = INT64_MAX, int64_t own_lo = INT64_MIN,
			   int64_t own_hi = INT64_MAX)
{
	const int W = d->world;
	uint64_t scnt[1 << MDB_MAX_RADIX_BITS], rcnt[1 << MDB_MAX_RADIX_BITS];
	size_t sc[1 << MDB_MAX_RADIX_BITS], sd[1 << MDB_MAX_RADIX_BITS], rc_[1 << MDB_MAX_RADIX_BITS], rd[1 << MDB_MAX_RADIX_BITS];
	int rc = dist_reserve(d, &d->send[i], &d->send_cap[i], n + 2);
	/* a failure on this rank (buffer, a key outside its promised range or the 4-byte wire format) travels WITH the counts: every
	 * rank reaches the count exchange, sees it, and all of them leave together - none is left waiting in the all-to-all */
	int prc = rc;
	if (!prc)
		prc = mdb_dev_partition_by_dest_pruned(d->ctx, keys, nulls, n, (uint32_t)W, wire32 ? 1 : 0, keep_lo, keep_hi, own_lo, own_hi, d->send[i],
						       NULL, scnt);	/* (synchronises) */
	uint64_t cs[2 << MDB_MAX_RADIX_BITS], cr[2 << MDB_MAX_RADIX_BITS];
	for (int p = 0; p < W; p++) {
		cs[2 * p] = prc ? 0 : scnt[p];
		cs[2 * p + 1] = prc ? 1 : 0;
	}
	rc = d->t.counts(d->t.self, cs, cr, 2);
	if (rc)
		return dist_err(d, rc, "count exchange failed%s%s", d->own_transport ? ": " : "", d->own_transport ? ((rccl_transport *)d->t.self)->err : "");
	if (prc)
		return dist_err(d, prc, "partition by destination: %s", mdb_dev_last_error(d->ctx));
	uint64_t total = 0, sent = 0;
	for (int p = 0; p < W; p++) {
		if (cr[2 * p + 1])
			return dist_err(d, -MIDORIDB_ERROR, "rank %d failed while it partitioned its rows (its own message says why); nothing was exchanged", p);
		rcnt[p] = cr[2 * p];
		sc[p] = (size_t)scnt[p];
		sd[p] = (size_t)sent;
		rc_[p] = (size_t)rcnt[p];
		rd[p] = (size_t)total;
		sent += scnt[p];
		total += rcnt[p];
	}
	if (total >= 0xFFFFFFFFull)
		return dist_err(d, -MIDORIDB_ERROR, "%llu rows for one GPU shard exceed the 32-bit row-id limit", (unsigned long long)total);
	rc = dist_reserve(d, &d->recv[i], &d->recv_cap[i], total + 2);
	if (rc)
		return rc;
	rc = d->t.alltoallv(d->t.self, d->send[i], sc, sd, d->recv[i], rc_, rd, wire32 ? 4 : 8, d->comm_stream);
	if (rc)
		return dist_err(d, rc, "all-to-all failed%s%s", d->own_transport ? ": " : "", d->own_transport ? ((rccl_transport *)d->t.self)->err : "");
	DIST_HIP(d, hipEventRecord(done, d->comm_stream));
	*n_recv = total;
	return MIDORIDB_OK;
}

/* the operator's flag word as three counters the ranks can sum: a region overflowed (the exact path answers), a right key outside
 * the window, anything else */
__global__ void k_status_words(const uint32_t *status, uint64_t *out)
{
	if (threadIdx.x == 0) {
		const uint32_t f = status[0];
		out[0] = (f & (2u | 2048u)) ? 1u : 0u;
		out[1] = (f & 128u) ? 1u : 0u;
		out[2] = (f & ~(2u | 128u | 2048u)) ? 1u : 0u;
	}
}

static int fused_fail(mdb_dev_ctx *ctx, bool alloc_out, int64_t *out_key, int64_t *out_count, int rc)
{
	if (alloc_out) {
		(void)mdb_dev_free(ctx, out_key);
		(void)mdb_dev_free(ctx, out_count);
	}
	return rc;
}

/* ------------------------------------------------------------------ the sharded operator with first-level regions on the wire
 *
 * mdb_dev_shard.hip: each table is partitioned ONCE, by the join's own first level, whose digit's top bits are the
 * destination; the regions of a destination are one block of a size every rank knows, so both all-to-alls are posted without
 * a count reaching a host, and the receiver joins what arrived without hashing or partitioning it again.  Host round trips
 * per call: one tiny exchange up front (the ranks' row counts: region capacities must be agreed on) and one at the end (the
 * ranks' status words: a region that overflowed anywhere sends EVERY rank to the exact path below) - two, where the
 * key-by-destination path needs four plus two read-backs.  Needs the two tables' global key ranges (promised, or measured
 * by MDB_WIRE_AUTO) and a right-table range of at most 2^30 values; MDB_DIST_FUSED=0 switches it off.
 * Returns 0 = done, 1 = not served / fell back (agreed by all ranks), < 0 = error. */
/* MDB_DIST_FAULT="<step>:<rank>" (the fault-injection mode of the test suite): this rank behaves as if `step` had failed */
static bool dist_fault_injected(const mdb_dist *d, const char *step)
{
	const char *e = mdb_knob("MDB_DIST_FAULT");
	const size_t l = strlen(step);
	return e && strncmp(e, step, l) == 0 && e[l] == ':' && atoi(e + l + 1) == d->rank;
}

#define DIST_PEER_FAILED 16384u	/* status bit: a peer's region counters say that its first level failed (k_shard_regions) */

static int dist_join_fused(mdb_dist *d, int ntab, const int64_t *const *keys, const uint64_t *const *nulls, const uint64_t *ns, const int64_t glo[2],
			   const int64_t ghi[2], bool promised, bool alloc_out, int64_t **out_key_p, int64_t **out_count_p, uint64_t cap,
			   uint64_t *out_groups, uint64_t *out_joined, bool right_only = false /* GROUP BY of table [1] alone: table [0] has no rows and does not travel */,
			   bool no_counts = false /* the caller wants the group keys alone (and J, G): the COUNT column is not written */)
{
	/* tables: [0] the left one, [1] the right one, [2 ...] further right tables joined on the same key */
	mdb_dev_ctx *ctx = d->ctx;
	const int W = d->world;
	if (mdb_knob("MDB_DIST_FUSED") && mdb_knob("MDB_DIST_FUSED")[0] == '0')
		return 1;
	if (ntab < 2 || ntab > MDB_SHARD_MAX_TABS)
		return 1;
	/* ---- every rank's row counts (region capacities are sized by the largest), whether its output buffers can take the
	 *      groups - a decision every rank takes from the same numbers - and WHICH call this is: the exchange rests on every rank
	 *      issuing the same collectives in the same order, so the ranks compare (kind of call, tables, calls made so far) before
	 *      anything is posted; ranks that disagree all return an error instead of pairing transfers of different calls */
	const int NC = MDB_SHARD_MAX_TABS + 2;
	const uint64_t guard = (1ull << 56) | ((uint64_t)ntab << 48) | ((uint64_t)(right_only ? 1 : 0) << 47) | (++d->seq & 0xFFFFFFFFFFull);
	uint64_t sendv[(MDB_SHARD_MAX_TABS + 2) << MDB_MAX_RADIX_BITS], recvv[(MDB_SHARD_MAX_TABS + 2) << MDB_MAX_RADIX_BITS];
	for (int p = 0; p < W; p++) {
		for (int x = 0; x < MDB_SHARD_MAX_TABS; x++)
			sendv[NC * p + x] = x < ntab ? ns[x] : 0;
		sendv[NC * p + MDB_SHARD_MAX_TABS] = alloc_out ? (1ull << 62) : cap;	/* (a transport's counters need not survive values beyond 2^63) */
		sendv[NC * p + MDB_SHARD_MAX_TABS + 1] = guard;
	}
	int rc = d->t.counts(d->t.self, sendv, recvv, NC);
	if (rc)
		return dist_err(d, rc, "count exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d));
	uint64_t n_max[MDB_SHARD_MAX_TABS] = { 0, 0, 0, 0 }, cap_min = ~0ull, nl_sum = 0;
	for (int p = 0; p < W; p++) {
		if (recvv[NC * p + MDB_SHARD_MAX_TABS + 1] != guard) {
			/* every rank has the same numbers in front of it: all continue from the largest call number seen, so the handle
			 * keeps working once the callers are in step again (a mismatch used to fail every later call) */
			uint64_t seq_max = 0;
			for (int q = 0; q < W; q++) {
				const uint64_t sq = recvv[NC * q + MDB_SHARD_MAX_TABS + 1] & 0xFFFFFFFFFFull;
				seq_max = sq > seq_max ? sq : seq_max;
			}
			d->seq = seq_max;
			return dist_err(d, -MIDORIDB_ERROR, "the ranks are not in the same collective call: rank %d is in call %llu (kind %llu), rank %d in call %llu "
					"(kind %llu); nothing was exchanged", p, (unsigned long long)(recvv[NC * p + MDB_SHARD_MAX_TABS + 1] & 0xFFFFFFFFFFull),
					(unsigned long long)(recvv[NC * p + MDB_SHARD_MAX_TABS + 1] >> 47), d->rank,
					(unsigned long long)(guard & 0xFFFFFFFFFFull), (unsigned long long)(guard >> 47));
		}
		nl_sum += recvv[NC * p + (right_only ? 1 : 0)];
		for (int x = 0; x < ntab; x++)
			n_max[x] = recvv[NC * p + x] > n_max[x] ? recvv[NC * p + x] : n_max[x];
		cap_min = recvv[NC * p + MDB_SHARD_MAX_TABS] < cap_min ? recvv[NC * p + MDB_SHARD_MAX_TABS] : cap_min;
	}
	mdb_shard_plan plan;
	if (mdb_shard_plan_make((uint32_t)W, (uint32_t)d->rank, (uint32_t)ntab, n_max, glo[0], ghi[0], glo[1], ghi[1], &plan))
		return 1;
	plan.right_only = right_only;
	d->last_plan = plan;
	d->have_plan = true;
	memset(d->phase_ms, 0, sizeof(d->phase_ms));
	/* the groups of a rank: at most the left rows it can receive, at most the key values that hash to it */
	const uint64_t recv_bound_l = plan.block_words[right_only ? 1 : 0] * (uint64_t)W;
	const uint64_t values = ((uint64_t)1 << plan.kbits) / (uint64_t)W;
	uint64_t group_bound = recv_bound_l < values ? recv_bound_l : values;
	group_bound = nl_sum < group_bound ? nl_sum : group_bound;	/* ... and never more than there are left rows at all */
	if (cap_min < group_bound)
		return 1;	/* (a caller's buffer that may not hold this path's groups: the exact path knows its sizes) */
	int64_t *out_key = alloc_out ? NULL : *out_key_p, *out_count = alloc_out ? NULL : *out_count_p;
	if (alloc_out) {
		/* (an allocation that fails on one rank only would leave the others in the all-to-all: sized by the agreed bound, it
		 * fails everywhere or nowhere on equal GPUs; what fails later - a rank's first level - is told to the peers, below) */
		if (mdb_dev_alloc(ctx, (group_bound ? group_bound : 1) * 8, (void **)&out_key) ||
		    mdb_dev_alloc(ctx, (group_bound ? group_bound : 1) * 8, (void **)&out_count)) {
			(void)mdb_dev_free(ctx, out_key);
			return dist_err(d, -MIDORIDB_NOMEM, "allocating group outputs: %s", mdb_dev_last_error(ctx));
		}
		cap = group_bound ? group_bound : 1;
	}
	const size_t wb = plan.wbytes, ncur = (size_t)plan.D * plan.nsub;
	size_t need = mdb_shard_arena_bytes(&plan);
	for (int x = 0; x < ntab; x++)
		need += mdb_align_up(plan.block_words[x] * (size_t)W * wb) + mdb_align_up(ncur * (size_t)W * 4) + 512;
	rc = mdb_arena_begin(ctx, need);
	if (rc)
		return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, rc, "%s", mdb_dev_last_error(ctx)));
	DIST_HIP(d, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	const bool timed = d->time_phases && d->ev_t0;
	if (timed)
		DIST_HIP(d, hipEventRecord(d->ev_t0, ctx->stream));
	void *recv[MDB_SHARD_MAX_TABS];
	uint32_t *rcnt[MDB_SHARD_MAX_TABS];
	for (int x = 0; x < ntab; x++) {
		recv[x] = mdb_arena_take(ctx, plan.block_words[x] * (size_t)W * wb);
		rcnt[x] = (uint32_t *)mdb_arena_take(ctx, ncur * (size_t)W * 4);
		if (!recv[x] || !rcnt[x])
			return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, -MIDORIDB_INTERNAL, "%s", mdb_dev_last_error(ctx)));
	}
	/* ---- each table: ONE partition pass, then its blocks and its region counters travel (fixed sizes: nothing to wait for).
	 *      A rank whose first level fails (a launch or sizing error, not a flag on the device) still posts what its peers are
	 *      waiting for - blocks nobody will read and region counters of 0xFFFFFFFF, which every receiver's descriptor kernel reads
	 *      as "this peer failed" - so that every rank reaches the status exchange and returns an error; none is left in the
	 *      all-to-all */
	size_t bc[1 << MDB_MAX_RADIX_BITS], bd[1 << MDB_MAX_RADIX_BITS], cc[1 << MDB_MAX_RADIX_BITS], cd0[1 << MDB_MAX_RADIX_BITS],
		cdr[1 << MDB_MAX_RADIX_BITS];
	int prc = MIDORIDB_OK;
	char perr[256] = "";
	void *fail_blocks = NULL;
	uint32_t *fail_counters = NULL;
	for (int x = 0; x < ntab; x++) {
		const void *regions = NULL;
		const uint32_t *cursors = NULL;
		if (right_only && x == 0) {	/* no left table: nothing travels, every region counter reads zero */
			DIST_HIP(d, hipMemsetAsync(rcnt[0], 0, ncur * (size_t)W * 4, ctx->stream));
			DIST_HIP(d, hipEventRecord(d->ev_tab[0], ctx->stream));
			continue;
		}
		if (!prc) {
			if (dist_fault_injected(d, "first_level"))
				prc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "fault injected into the first level of table %d (MDB_DIST_FAULT)", x);
			else
				prc = mdb_shard_partition(ctx, &plan, x, keys[x], nulls[x], ns[x], &regions, &cursors);
			if (prc)
				snprintf(perr, sizeof(perr), "%s", mdb_dev_last_error(ctx));
		}
		if (prc) {
			if (!fail_blocks) {
				size_t words = 0;
				for (int y = 0; y < ntab; y++)
					words = plan.block_words[y] > words ? plan.block_words[y] : words;
				if (mdb_dev_alloc(ctx, words * (size_t)W * wb + 64, &fail_blocks) || mdb_dev_alloc(ctx, ncur * 4, (void **)&fail_counters) ||
				    hipMemsetAsync(fail_counters, 0xFF, ncur * 4, ctx->stream) != hipSuccess) {
					(void)mdb_dev_free(ctx, fail_blocks);
					(void)mdb_dev_free(ctx, fail_counters);
					/* (nothing left to tell the peers with: fatal for the communicator) */
					return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, prc, "sharded first level: %s (and the peers could not be told)", perr));
				}
			}
			regions = fail_blocks;
			cursors = fail_counters;
		}
		for (int p = 0; p < W; p++) {
			bc[p] = (size_t)plan.block_words[x] * wb;
			bd[p] = (size_t)p * plan.block_words[x] * wb;
			cc[p] = ncur;
			cd0[p] = 0;		/* every rank gets the whole counter array */
			cdr[p] = (size_t)p * ncur;
		}
		DIST_HIP(d, hipEventRecord(d->ev_ready, ctx->stream));
		if (timed)
			DIST_HIP(d, hipEventRecord(d->ev_part[x], ctx->stream));
		DIST_HIP(d, hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0));
		rc = d->t.alltoallv(d->t.self, regions, bc, bd, recv[x], bc, bd, 1, d->comm_stream);
		if (!rc)
			rc = d->t.alltoallv(d->t.self, cursors, cc, cd0, rcnt[x], cc, cdr, 4, d->comm_stream);
		if (rc)
			return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, rc, "all-to-all failed%s%s", d->own_transport ? ": " : "", transport_err(d)));
		DIST_HIP(d, hipEventRecord(d->ev_tab[x], d->comm_stream));
		if (timed)
			DIST_HIP(d, hipEventRecord(d->ev_arr[x], d->comm_stream));
	}
	/* (the receiver waits for a table right before the first kernel that reads it: its own level over the left table runs while
	 * the right table is still on the wire) */
	void *arrived[MDB_SHARD_MAX_TABS] = { d->ev_tab[0], d->ev_tab[1], d->ev_tab[2], d->ev_tab[3] };
	rc = mdb_shard_join(ctx, &plan, recv, rcnt, out_key, no_counts ? NULL : out_count, cap, arrived);
	if (rc && !prc) {
		/* (the same holds for the receiver's launches: the peers are told through the status exchange) */
		prc = rc;
		snprintf(perr, sizeof(perr), "%s", mdb_dev_last_error(ctx));
		uint32_t *h1 = reinterpret_cast<uint32_t *>(ctx->h_pinned) + 520;
		h1[0] = DIST_PEER_FAILED;
		DIST_HIP(d, hipMemcpyAsync(ctx->d_status, h1, 4, hipMemcpyHostToDevice, ctx->stream));
	}
	/* ---- G, J and the flags come back with ONE host synchronisation; what went wrong anywhere sends every rank the same way:
	 *      over RCCL the ranks' flags are summed on the device, on the operator's own stream, before that read-back (a host
	 *      all-reduce is a second round trip); a host's own transport is asked the usual way */
	uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
	uint64_t st[3] = { 0, 0, 0 };
	bool reduced_on_device = false;
	if (d->own_transport) {
		rccl_transport *rt = (rccl_transport *)d->t.self;
		uint64_t *dflags = reinterpret_cast<uint64_t *>(ctx->d_status + 32);	/* (three 8-byte words beyond what the operators use) */
		hipLaunchKernelGGL(k_status_words, dim3(1), dim3(64), 0, ctx->stream, ctx->d_status, dflags);
		ncclResult_t nr = ncclAllReduce(dflags, dflags, 3, ncclUint64, ncclSum, rt->status, ctx->stream);
		if (nr != ncclSuccess)
			return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, -MIDORIDB_INTERNAL, "status all-reduce: %s", ncclGetErrorString(nr)));
		reduced_on_device = true;
	}
	if (timed)
		DIST_HIP(d, hipEventRecord(d->ev_t1, ctx->stream));
	DIST_HIP(d, hipMemcpyAsync(h, ctx->d_status, 40 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	DIST_HIP(d, hipStreamSynchronize(ctx->stream));
	if (fail_blocks) {
		DIST_HIP(d, hipStreamSynchronize(d->comm_stream));
		(void)mdb_dev_free(ctx, fail_blocks);
		(void)mdb_dev_free(ctx, fail_counters);
	}
	const uint32_t flags = h[0];
	const uint64_t G = h[1], J = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
	if (reduced_on_device) {
		memcpy(st, h + 32, sizeof(st));
	} else {
		st[0] = (flags & (2u | 2048u)) ? 1u : 0u;
		st[1] = (flags & 128u) ? 1u : 0u;
		st[2] = ((flags & ~(2u | 128u | 2048u)) || prc) ? 1u : 0u;
		rc = d->t.allreduce_sum_u64(d->t.self, st, 3);
	}
	if (timed) {
		/* (events of two streams share one clock) */
		float first = 0.f, arr_last = 0.f, total = 0.f, v = 0.f;
		const int x_first = right_only ? 1 : 0;
		for (int x = x_first; x < ntab; x++) {
			if (hipEventElapsedTime(&v, d->ev_t0, d->ev_part[x]) == hipSuccess && v > first)
				first = v;
			if (hipStreamSynchronize(d->comm_stream) == hipSuccess && hipEventElapsedTime(&v, d->ev_t0, d->ev_arr[x]) == hipSuccess && v > arr_last)
				arr_last = v;
		}
		(void)hipEventElapsedTime(&total, d->ev_t0, d->ev_t1);
		d->phase_ms[0] = first;
		d->phase_ms[1] = arr_last > first ? arr_last - first : 0.0;
		d->phase_ms[2] = total - (arr_last > first ? arr_last : first);
		d->phase_ms[3] = total;
	}
	if (prc)
		return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, prc, "sharded operator failed on this rank (the peers have been told): %s", perr));
	if (rc)
		return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, rc, "status exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d)));
	if (st[2]) {
		if (flags & DIST_PEER_FAILED)
			return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, -MIDORIDB_ERROR, "sharded join: a peer failed in its first partition level (its own message says why); no result"));
		return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, -MIDORIDB_INTERNAL, "sharded join: failed on %llu rank(s) (status %u on this rank)", (unsigned long long)st[2], flags));
	}
	if (st[1]) {
		if (promised)
			return fused_fail(ctx, alloc_out, out_key, out_count, dist_err(d, -MIDORIDB_ERROR, "partition_by_dest: a key lies outside the range [%lld, %lld] promised for its column",
					(long long)glo[1], (long long)ghi[1]));
		return fused_fail(ctx, alloc_out, out_key, out_count, 1);
	}
	if (st[0])
		return fused_fail(ctx, alloc_out, out_key, out_count, 1);	/* skewed keys outgrew a fixed-capacity region somewhere (or a product of counts 32 bits): the exact path */
	d->last_recv_left = G;	/* (rows are not counted on this path: the groups are a lower bound) */
	if (alloc_out) {
		*out_key_p = out_key;
		*out_count_p = out_count;
	}
	*out_groups = G;
	if (out_joined)
		*out_joined = J;
	d->last_fused = 1;
	return 0;
}

/* GROUP BY key + COUNT(*) of ONE sharded column as (key, COUNT) pairs, every key on the rank its hash belongs to, in unspecified
 * order (include/mdb_dist.h): each rank's ONE partition pass writes the 2- or 4-byte words of its keys into first-level regions,
 * the regions travel, the receiver counts per leaf slot - no rows are exchanged, no row ids exist, nothing is ordered.
 * 0 = done (outputs allocated by the call), 1 = not served (the global key range is unknown or spans more than 2^30 values, NULL
 * keys, skew: every rank gets the same answer - the caller exchanges rows and groups locally), < 0 = error. */
static int dist_group_keys_impl(mdb_dist *d, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t **out_key, int64_t **out_count,
				uint64_t *out_groups);
/* the outputs a call allocated do not outlive its failure */
static int dist_join_fail(mdb_dev_ctx *ctx, bool alloc_out, int64_t **out_key, int64_t **out_count, uint32_t **out_first, int rc)
{
	if (alloc_out) {
		(void)mdb_dev_free(ctx, *out_key);
		(void)mdb_dev_free(ctx, *out_count);
		*out_key = *out_count = NULL;
		if (out_first) {
			(void)mdb_dev_free(ctx, *out_first);
			*out_first = NULL;
		}
	}
	return rc;
}

/* MDB_WIRE_AUTO: the GLOBAL [smallest, largest] key of the left and of the right table - one statistics pass per table on
 * every rank, one tiny exchange - and whether all of them fit 32 bits (the 4-byte wire format) */
static int dist_measure_ranges(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			       const uint64_t *null_r, uint64_t n_r, int64_t glo[2], int64_t ghi[2], bool *fits32)
{
	mdb_dev_ctx *ctx = d->ctx;
	{
		const int W = d->world;
		const int64_t *cols[2] = { keys_l, keys_r };
		const uint64_t *nb[2] = { null_l, null_r };
		const uint64_t ns[2] = { n_l, n_r };
		/* (eight 32-bit halves per rank: the counters of a transport need not survive values beyond 2^63) */
		uint64_t mine[4], all[9 << MDB_MAX_RADIX_BITS], sendv[9 << MDB_MAX_RADIX_BITS];
		int range_rc = MIDORIDB_OK;
		for (int i = 0; i < 2; i++) {
			int64_t lo = 0, hi = -1;
			if (ns[i] && !range_rc) {
				range_rc = mdb_dev_key_range(ctx, cols[i], nb[i], ns[i], &lo, &hi);
				if (range_rc)
					dist_err(d, range_rc, "key range: %s", mdb_dev_last_error(ctx));
			}
			if (lo > hi) {		/* no key at all on this rank */
				lo = INT64_MAX;
				hi = INT64_MIN;
			}
			mine[2 * i] = (uint64_t)lo;
			mine[2 * i + 1] = (uint64_t)hi;
		}
		for (int p = 0; p < W; p++) {
			for (int q = 0; q < 4; q++) {
				sendv[9 * p + 2 * q] = mine[q] >> 32;
				sendv[9 * p + 2 * q + 1] = mine[q] & 0xFFFFFFFFull;
			}
			sendv[9 * p + 8] = range_rc ? 1 : 0;	/* (status: a rank whose statistics pass failed takes every rank out with it) */
		}
		int rc = d->t.counts(d->t.self, sendv, all, 9);		/* every rank's four numbers (+ status) to every rank */
		if (rc)
			return dist_err(d, rc, "key range exchange failed");
		if (range_rc)
			return range_rc;
		for (int p = 0; p < W; p++)
			if (all[9 * p + 8])
				return dist_err(d, -MIDORIDB_ERROR, "rank %d failed while it measured its key ranges; nothing was exchanged", p);
		for (int i = 0; i < 2; i++) {
			int64_t lo = INT64_MAX, hi = INT64_MIN;
			for (int p = 0; p < W; p++) {
				const int64_t plo = (int64_t)((all[9 * p + 4 * i] << 32) | all[9 * p + 4 * i + 1]);
				const int64_t phi = (int64_t)((all[9 * p + 4 * i + 2] << 32) | all[9 * p + 4 * i + 3]);
				lo = plo < lo ? plo : lo;
				hi = phi > hi ? phi : hi;
			}
			glo[i] = lo;
			ghi[i] = hi;
		}
		bool wide = false;
		for (int i = 0; i < 2; i++)
			if (glo[i] <= ghi[i] && (glo[i] < -(1ll << 31) || ghi[i] >= (1ll << 31)))
				wide = true;
		*fits32 = !wide;
	}
	return MIDORIDB_OK;
}

/* common part: exchange (the left table only unless it is already in place), local join into buffers that are either the
 * caller's (capacity cap) or allocated here once the number of received left rows is known */
static int dist_join_impl(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			  const uint64_t *null_r, uint64_t n_r, bool left_in_place, bool alloc_out, int64_t **out_key, int64_t **out_count,
			  uint32_t **out_first, uint64_t cap, uint64_t *out_groups, uint64_t *out_joined, bool by_key_hash = false)
{
	mdb_dev_ctx *ctx = d->ctx;
	*out_groups = 0;
	if (out_joined)
		*out_joined = 0;
	DIST_HIP(d, hipSetDevice(ctx->device));
	/* ---- wire format: every rank must take the same decision */
	bool wire32 = d->wire_mode == MDB_WIRE_32 && !left_in_place;
	/* the two tables' GLOBAL key ranges (MDB_WIRE_AUTO: the column statistics are computed anyway): what lies outside the
	 * other table's range joins nothing on any GPU and stays home - min-max pruning before the shuffle; a fact table whose
	 * dimension covers a sixteenth of its key range sends a sixteenth of its rows */
	int64_t glo[2] = { INT64_MIN, INT64_MIN }, ghi[2] = { INT64_MAX, INT64_MAX };
	if (d->wire_mode == MDB_WIRE_AUTO && !left_in_place) {
		const int mrc = dist_measure_ranges(d, keys_l, null_l, n_l, keys_r, null_r, n_r, glo, ghi, &wire32);
		if (mrc)
			return mrc;
	}
	d->last_wire32 = wire32 ? 1 : 0;
	/* (a table without any key makes the other one's range empty: lo > hi drops every row, the join is empty) */
	const bool measured = d->wire_mode == MDB_WIRE_AUTO && !left_in_place;
	if (!measured && d->have_ranges && !left_in_place)
		for (int i = 0; i < 2; i++) {
			glo[i] = d->promised_lo[i];
			ghi[i] = d->promised_hi[i];
		}
	const bool prune = (measured || (d->have_ranges && !left_in_place)) && !(mdb_knob("MDB_MINMAX_PRUNE") && mdb_knob("MDB_MINMAX_PRUNE")[0] == '0');
	const bool verify = prune && !measured;		/* promised ranges are checked while each table is partitioned */
	d->last_pruned = prune ? 1 : 0;

	d->last_fused = 0;
	if (prune && !left_in_place && !out_first && !by_key_hash) {
		/* both global key ranges are known: the first partition level IS the exchange (mdb_dev_shard.hip) */
		const int64_t *fk[2] = { keys_l, keys_r };
		const uint64_t *fn[2] = { null_l, null_r };
		const uint64_t fs[2] = { n_l, n_r };
		const int frc = dist_join_fused(d, 2, fk, fn, fs, glo, ghi, verify, alloc_out, out_key, out_count, cap, out_groups, out_joined);
		if (frc <= 0)
			return frc;
	}
	/* the transfer stream starts behind whatever the caller queued on the context's stream (its key columns) */
	DIST_HIP(d, hipEventRecord(d->ev_ready, ctx->stream));
	DIST_HIP(d, hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0));

	/* ---- table L travels while table R is partitioned; table R travels while the received L is prepared */
	uint64_t got_l = n_l, got_r = 0;
	int rc;
	if (!left_in_place) {
		rc = dist_send_table(d, 0, keys_l, null_l, n_l, wire32, &got_l, d->ev_a, prune ? glo[1] : INT64_MIN, prune ? ghi[1] : INT64_MAX,
				     verify ? glo[0] : INT64_MIN, verify ? ghi[0] : INT64_MAX);
		if (rc)
			return rc;
	}
	rc = dist_send_table(d, 1, keys_r, null_r, n_r, wire32, &got_r, d->ev_b, prune ? glo[0] : INT64_MIN, prune ? ghi[0] : INT64_MAX,
			     verify ? glo[1] : INT64_MIN, verify ? ghi[1] : INT64_MAX);
	if (rc)
		return rc;
	d->last_recv_left = got_l;
	if (alloc_out) {
		const uint64_t c = got_l ? got_l : 1;
		*out_key = NULL;
		*out_count = NULL;
		if (out_first)
			*out_first = NULL;
		if (mdb_dev_alloc(ctx, c * 8, (void **)out_key) || mdb_dev_alloc(ctx, c * 8, (void **)out_count) ||
		    (out_first && mdb_dev_alloc(ctx, c * 4, (void **)out_first))) {
			if (*out_key)
				mdb_dev_free(ctx, *out_key);
			if (*out_count)
				mdb_dev_free(ctx, *out_count);
			*out_key = *out_count = NULL;
			return dist_err(d, -MIDORIDB_NOMEM, "allocating group outputs: %s", mdb_dev_last_error(ctx));
		}
		cap = c;
	} else if (got_l > cap) {
		return dist_err(d, -MIDORIDB_ERROR, "group output capacity %llu is below the %llu left rows this rank received",
				(unsigned long long)cap, (unsigned long long)got_l);
	}
	if (left_in_place) {
		rc = mdb_dev_join_group_count_begin(ctx, keys_l, null_l, n_l, got_r);
	} else {
		DIST_HIP(d, hipStreamWaitEvent(ctx->stream, d->ev_a, 0));
		if (wire32)
			rc = mdb_dev_join_group_count_begin_i32(ctx, (const int32_t *)d->recv[0], got_l, got_r);
		else
			rc = mdb_dev_join_group_count_begin(ctx, (const int64_t *)d->recv[0], NULL, got_l, got_r);
	}
	if (rc) {
		dist_err(d, rc, "local join (begin): %s", mdb_dev_last_error(ctx));
		return dist_join_fail(ctx, alloc_out, out_key, out_count, out_first, rc);
	}
	DIST_HIP(d, hipStreamWaitEvent(ctx->stream, d->ev_b, 0));
	uint64_t G = 0, J = 0;
	uint32_t *first = out_first ? *out_first : NULL;
	if (wire32 && !left_in_place)
		rc = mdb_dev_join_group_count_finish_i32(ctx, (const int32_t *)d->recv[1], got_r, MDB_ORDER_FIRST, *out_key, *out_count, first, cap, &G,
							 &J);
	else	/* (a left table in place is int64: its right table always travels as 8-byte keys) */
		rc = mdb_dev_join_group_count_finish(ctx, (const int64_t *)d->recv[1], NULL, got_r, MDB_ORDER_FIRST, *out_key, *out_count, first, cap,
						     &G, &J);
	if (rc) {
		dist_err(d, rc, "local join (finish): %s", mdb_dev_last_error(ctx));
		return dist_join_fail(ctx, alloc_out, out_key, out_count, out_first, rc);
	}
	*out_groups = G;
	if (out_joined)
		*out_joined = J;
	return MIDORIDB_OK;
}

extern "C" int mdb_dist_join_group_count(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
					 const uint64_t *null_r, uint64_t n_r, int64_t *out_key, int64_t *out_count, uint64_t cap,
					 uint64_t *out_groups, uint64_t *out_joined)
{
	if (!d || !out_groups || !out_key || !out_count)
		return -MIDORIDB_ERROR;
	return dist_join_impl(d, keys_l, null_l, n_l, keys_r, null_r, n_r, false, false, &out_key, &out_count, NULL, cap, out_groups, out_joined);
}

extern "C" int mdb_dist_join_group_count_alloc(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
					       const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, uint32_t flags, int64_t **out_key,
					       int64_t **out_count, uint32_t **out_first, uint64_t *out_groups, uint64_t *out_joined)
{
	if (!d || !out_groups || !out_key || !out_count)
		return -MIDORIDB_ERROR;
	return dist_join_impl(d, keys_l, null_l, n_l, keys_r, null_r, n_r, (flags & MDB_DIST_LEFT_IN_PLACE) != 0, true, out_key, out_count, out_first,
			      0, out_groups, out_joined, (flags & MDB_DIST_PLACE_BY_KEY_HASH) != 0);
}

/* ------------------------------------------------------------------ row shuffles: keys + payload columns
 *
 * The materialising joins of the path (reference executor_select.c:1076-1149, 1151-1280) over sharded tables: every row
 * travels to the rank its key hashes to, with the columns the statement reads.  Per shuffle: ONE destination partition of
 * the key column (mdb_dev_partition_by_dest, which also returns every outgoing key's source position), ONE count
 * exchange (rows per peer + which columns carry NULL bits + this rank's status: a rank that failed says so there and all
 * ranks leave together), the columns gathered into send order in one launch per 16 columns (mdb_dev_gather_cols through
 * the source positions), one uneven all-to-all per column on the transfer stream.  NULL bits of all columns travel as ONE
 * extra 8-byte word per row (bit c = column c is NULL), only when some rank has a NULL bitmap at all. */

#define SH_THREADS 256

struct sh_null_src {
	const uint64_t *nb[MDB_DIST_SHUFFLE_MAX_COLS];
	const uint32_t *rid[MDB_DIST_SHUFFLE_MAX_COLS];
};
struct sh_null_dst {
	uint64_t *bits[MDB_DIST_SHUFFLE_MAX_COLS];
};

static inline uint32_t sh_grid(uint64_t n)
{
	const uint64_t g = (n + SH_THREADS - 1) / SH_THREADS;
	return (uint32_t)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

/* mask[i] = bit c set when column c of the row at send position i is NULL */
__global__ __launch_bounds__(SH_THREADS) void k_nullmask_pack(sh_null_src s, uint64_t colmask, const uint32_t *__restrict__ pos, uint64_t m,
							       uint64_t *__restrict__ mask)
{
	for (uint64_t i = (uint64_t)blockIdx.x * SH_THREADS + threadIdx.x; i < m; i += (uint64_t)gridDim.x * SH_THREADS) {
		const uint32_t p = pos ? pos[i] : (uint32_t)i;	/* (pos == NULL: the rows as they stand - a broadcast) */
		uint64_t w = 0;
		for (uint64_t cm = colmask; cm; cm &= cm - 1) {
			const int c = __builtin_ctzll(cm);
			const uint32_t row = s.rid[c] ? s.rid[c][p] : p;
			if (mdb_bit_is_set(s.nb[c], row))
				w |= 1ull << c;
		}
		mask[i] = w;
	}
}

/* the received mask words back into one NULL bitmap per column (a wave per 64 rows: ballot) */
__global__ __launch_bounds__(SH_THREADS) void k_nullmask_unpack(const uint64_t *__restrict__ mask, uint64_t n, uint64_t colmask, sh_null_dst d)
{
	const uint64_t nwords = (n + 63) / 64;
	for (uint64_t w = ((uint64_t)blockIdx.x * SH_THREADS + threadIdx.x) >> 6; w < nwords; w += ((uint64_t)gridDim.x * SH_THREADS) >> 6) {
		const uint64_t i = w * 64 + mdb_lane();
		const uint64_t mw = i < n ? mask[i] : 0;
		for (uint64_t cm = colmask; cm; cm &= cm - 1) {
			const int c = __builtin_ctzll(cm);
			const uint64_t word = __ballot((mw >> c) & 1);
			if (mdb_lane() == 0)
				d.bits[c][w] = word;
		}
	}
}

/* partitioning key of a shuffle that keeps its NULL-key rows: one fixed value under every NULL cell, so that they all
 * hash to the same rank (whatever bytes an UPDATE ... SET col = NULL left there) */
__global__ __launch_bounds__(SH_THREADS) void k_zero_null_keys(const int64_t *__restrict__ keys, const uint64_t *__restrict__ nulls, uint64_t n,
							        int64_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * SH_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SH_THREADS)
		out[i] = mdb_bit_is_set(nulls, i) ? 0 : keys[i];
}

static void sh_pend(mdb_dist *d, void *p)
{
	if (!p)
		return;
	if (d->npend < SH_MAX_PENDING)
		d->pend[d->npend++] = p;
	/* (cannot happen: a shuffle keeps at most 2 * MDB_DIST_SHUFFLE_MAX_COLS + 16 buffers and waits when the list is half full) */
}

extern "C" int mdb_dist_wait_transfers(mdb_dist *d)
{
	if (!d)
		return -MIDORIDB_ERROR;
	DIST_HIP(d, hipSetDevice(d->ctx->device));
	if (d->sh_posted)
		DIST_HIP(d, hipStreamWaitEvent(d->ctx->stream, d->ev_sh, 0));
	d->sh_posted = false;
	/* the send buffers go back to the context's cache: its reuse is ordered on the context's stream, behind the wait */
	for (int i = 0; i < d->npend; i++)
		(void)mdb_dev_free(d->ctx, d->pend[i]);
	d->npend = 0;
	return MIDORIDB_OK;
}

#define SH_COUNTERS 3	/* per peer: rows, columns that carry NULL bits, status */


extern "C" int mdb_dist_shuffle_rows(mdb_dist *d, const int64_t *keys, const uint64_t *key_nulls, uint64_t n, uint32_t flags,
				     const struct mdb_dist_col *cols, int ncols, void **out_values, uint64_t **out_nullbits, uint64_t *out_n)
{
	if (!d || !out_n || ncols < 0 || ncols > MDB_DIST_SHUFFLE_MAX_COLS || (ncols && (!cols || !out_values || !out_nullbits)) || (n && !keys))
		return d ? dist_err(d, -MIDORIDB_ERROR, "shuffle_rows: bad arguments") : -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = d->ctx;
	const int W = d->world;
	*out_n = 0;
	for (int c = 0; c < ncols; c++) {
		out_values[c] = NULL;
		out_nullbits[c] = NULL;
	}
	DIST_HIP(d, hipSetDevice(ctx->device));
	if (d->npend > SH_MAX_PENDING / 2) {
		int rc = mdb_dist_wait_transfers(d);
		if (rc)
			return rc;
	}
	uint64_t scnt[1 << MDB_MAX_RADIX_BITS], sendv[SH_COUNTERS << MDB_MAX_RADIX_BITS], recvv[SH_COUNTERS << MDB_MAX_RADIX_BITS];
	size_t sc[1 << MDB_MAX_RADIX_BITS], sd[1 << MDB_MAX_RADIX_BITS], rcn[1 << MDB_MAX_RADIX_BITS], rd[1 << MDB_MAX_RADIX_BITS];
	char local_err[384];
	local_err[0] = 0;
	int status = MIDORIDB_OK;

	/* ---- 1. destination partition of the key column: send order + the source position of every outgoing row */
	void *kbuf = NULL, *canon = NULL;
	uint32_t *pos = NULL;
	for (int p = 0; p < W; p++)
		scnt[p] = 0;
	if (n >= 0xFFFFFFFFull) {
		status = -MIDORIDB_ERROR;
		snprintf(local_err, sizeof(local_err), "%llu rows exceed the 32-bit row-id limit of one shard", (unsigned long long)n);
	}
	if (!status && (mdb_dev_alloc(ctx, (n ? n : 1) * 8, &kbuf) || mdb_dev_alloc(ctx, (n ? n : 1) * 4, (void **)&pos))) {
		status = -MIDORIDB_NOMEM;
		snprintf(local_err, sizeof(local_err), "%s", mdb_dev_last_error(ctx));
	}
	const int64_t *pkeys = keys;
	const uint64_t *pnulls = key_nulls;
	if (!status && n && (flags & MDB_DIST_KEEP_NULL_KEYS) && key_nulls) {
		if (mdb_dev_alloc(ctx, n * 8, &canon)) {
			status = -MIDORIDB_NOMEM;
			snprintf(local_err, sizeof(local_err), "%s", mdb_dev_last_error(ctx));
		} else {
			hipLaunchKernelGGL(k_zero_null_keys, dim3(sh_grid(n)), dim3(SH_THREADS), 0, ctx->stream, keys, key_nulls, n, (int64_t *)canon);
			pkeys = (const int64_t *)canon;
			pnulls = NULL;
		}
	}
	if (!status && n) {
		status = mdb_dev_partition_by_dest(ctx, pkeys, pnulls, n, (uint32_t)W, 0, kbuf, pos, scnt);	/* (synchronises) */
		if (status) {
			snprintf(local_err, sizeof(local_err), "partition by destination: %s", mdb_dev_last_error(ctx));
			for (int p = 0; p < W; p++)
				scnt[p] = 0;
		}
	}

	/* ---- 2. counts + NULL-carrying columns + status: every rank gets here, whatever happened to it so far */
	uint64_t colmask = 0;
	for (int c = 0; c < ncols; c++)
		if (cols[c].nullbits)
			colmask |= 1ull << c;
	for (int p = 0; p < W; p++) {
		sendv[SH_COUNTERS * p] = scnt[p];
		sendv[SH_COUNTERS * p + 1] = colmask;
		sendv[SH_COUNTERS * p + 2] = status ? 1 : 0;
	}
	int rc = d->t.counts(d->t.self, sendv, recvv, SH_COUNTERS);
	if (rc) {
		(void)mdb_dev_free(ctx, kbuf);
		(void)mdb_dev_free(ctx, pos);
		(void)mdb_dev_free(ctx, canon);
		return dist_err(d, rc, "count exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d));
	}
	uint64_t total = 0, m = 0, gmask = 0;
	int failed_rank = -1;
	for (int p = 0; p < W; p++) {
		sc[p] = (size_t)scnt[p];
		sd[p] = (size_t)m;
		rcn[p] = (size_t)recvv[SH_COUNTERS * p];
		rd[p] = (size_t)total;
		m += scnt[p];
		total += recvv[SH_COUNTERS * p];
		gmask |= recvv[SH_COUNTERS * p + 1];
		if (recvv[SH_COUNTERS * p + 2] && failed_rank < 0)
			failed_rank = p;
	}
	if (failed_rank < 0 && total >= 0xFFFFFFFFull) {	/* (every rank sees its own total: agree on it the hard way - it is an error everywhere or nowhere) */
		status = -MIDORIDB_ERROR;
		snprintf(local_err, sizeof(local_err), "%llu rows for one GPU shard exceed the 32-bit row-id limit", (unsigned long long)total);
	}
	if (failed_rank >= 0 || status) {
		(void)mdb_dev_free(ctx, kbuf);
		(void)mdb_dev_free(ctx, pos);
		(void)mdb_dev_free(ctx, canon);
		if (status)
			return dist_err(d, status, "shuffle: %s", local_err);
		return dist_err(d, -MIDORIDB_ERROR, "shuffle: rank %d failed (its own message says why); nothing was exchanged", failed_rank);
	}

	/* ---- 3. the columns in send order, receive buffers */
	void *send[MDB_DIST_SHUFFLE_MAX_COLS + 1], *comp[MDB_GATHER_MAX_RIDS];
	const uint32_t *comp_of[MDB_GATHER_MAX_RIDS];
	int ncomp = 0;
	void *send_mask = NULL, *recv_mask = NULL;
	bool key_buf_used = false;
	rc = MIDORIDB_OK;
	for (int c = 0; c <= MDB_DIST_SHUFFLE_MAX_COLS; c++)
		send[c] = NULL;
	struct mdb_gather_col gl[MDB_GATHER_MAX_COLS];
	int ngl = 0, nrid_in_batch = 0;
	const uint32_t *batch_rids[MDB_GATHER_MAX_RIDS];
	for (int c = 0; c < ncols && !rc; c++) {
		rc = mdb_dev_alloc(ctx, (total ? total : 1) * 8, &out_values[c]);
		if (rc)
			break;
		if (gmask >> c & 1) {
			rc = mdb_dev_alloc(ctx, ((total + 63) / 64 + 1) * 8, (void **)&out_nullbits[c]);
			if (rc)
				break;
		}
		if (!m)
			continue;
		if (cols[c].values == (const void *)keys && !cols[c].rid && !key_buf_used) {
			send[c] = kbuf;		/* the key column itself: the partition wrote it in send order already */
			key_buf_used = true;
			continue;
		}
		const uint32_t *idx = pos;
		if (cols[c].rid) {		/* stream position -> row of the base column: composed once per row-id vector */
			int k = 0;
			while (k < ncomp && comp_of[k] != cols[c].rid)
				k++;
			if (k == ncomp) {
				if (ncomp == MDB_GATHER_MAX_RIDS) {
					rc = -MIDORIDB_ERROR;
					mdb_set_err(ctx, rc, "more than %d row-id vectors in one shuffle", MDB_GATHER_MAX_RIDS);
					break;
				}
				rc = mdb_dev_alloc(ctx, m * 4, &comp[ncomp]);
				if (!rc)
					rc = mdb_dev_gather32(ctx, cols[c].rid, pos, m, (uint32_t *)comp[ncomp]);
				if (rc)
					break;
				comp_of[ncomp++] = cols[c].rid;
			}
			idx = (const uint32_t *)comp[k];
		}
		rc = mdb_dev_alloc(ctx, m * 8, &send[c]);
		if (rc)
			break;
		int t = 0;
		while (t < nrid_in_batch && batch_rids[t] != idx)
			t++;
		if (ngl == MDB_GATHER_MAX_COLS || (t == nrid_in_batch && nrid_in_batch == MDB_GATHER_MAX_RIDS)) {
			rc = mdb_dev_gather_cols(ctx, gl, ngl, m);
			ngl = nrid_in_batch = 0;
			t = 0;
			if (rc)
				break;
		}
		if (t == nrid_in_batch)
			batch_rids[nrid_in_batch++] = idx;
		gl[ngl].src = cols[c].values;
		gl[ngl].src_nullbits = NULL;
		gl[ngl].rid = idx;
		gl[ngl].dst = send[c];
		gl[ngl].dst_nullbits = NULL;
		ngl++;
	}
	if (!rc && ngl)
		rc = mdb_dev_gather_cols(ctx, gl, ngl, m);
	if (!rc && gmask) {
		rc = mdb_dev_alloc(ctx, (m ? m : 1) * 8, &send_mask);
		if (!rc)
			rc = mdb_dev_alloc(ctx, (total ? total : 1) * 8, &recv_mask);
		if (!rc && m) {
			if (colmask) {
				sh_null_src s;
				memset(&s, 0, sizeof(s));
				for (int c = 0; c < ncols; c++) {
					s.nb[c] = cols[c].nullbits;
					s.rid[c] = cols[c].rid;
				}
				hipLaunchKernelGGL(k_nullmask_pack, dim3(sh_grid(m)), dim3(SH_THREADS), 0, ctx->stream, s, colmask, pos, m, (uint64_t *)send_mask);
			} else {
				rc = mdb_dev_memset(ctx, send_mask, 0, m * 8);
			}
		}
	}
	if (rc) {
		/* a failure past the count exchange (allocation): the peers are about to post their transfers.  Nothing sane is
		 * left but to post ours from whatever buffers exist - so the buffers are checked BEFORE anything is posted and the
		 * failure is reported through the collective that follows */
		snprintf(local_err, sizeof(local_err), "%s", mdb_dev_last_error(ctx));
	}
	{
		/* second agreement (one word per peer): did every rank get its buffers?  Costs one tiny exchange; without it an
		 * allocation failure on one rank would leave the others inside the all-to-all */
		for (int p = 0; p < W; p++)
			sendv[p] = rc ? 1 : 0;
		int rc2 = d->t.counts(d->t.self, sendv, recvv, 1);
		bool any = rc2 != 0;
		for (int p = 0; p < W && !rc2; p++)
			any = any || recvv[p] != 0;
		if (any) {
			for (int c = 0; c < ncols; c++) {
				(void)mdb_dev_free(ctx, out_values[c]);
				(void)mdb_dev_free(ctx, out_nullbits[c]);
				out_values[c] = NULL;
				out_nullbits[c] = NULL;
				if (send[c] && send[c] != kbuf)
					(void)mdb_dev_free(ctx, send[c]);
			}
			for (int k = 0; k < ncomp; k++)
				(void)mdb_dev_free(ctx, comp[k]);
			(void)mdb_dev_free(ctx, send_mask);
			(void)mdb_dev_free(ctx, recv_mask);
			(void)mdb_dev_free(ctx, kbuf);
			(void)mdb_dev_free(ctx, pos);
			(void)mdb_dev_free(ctx, canon);
			if (rc2)
				return dist_err(d, rc2, "count exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d));
			if (rc)
				return dist_err(d, rc, "shuffle: %s", local_err);
			return dist_err(d, -MIDORIDB_ERROR, "shuffle: another rank could not allocate its buffers; nothing was exchanged");
		}
	}

	/* ---- 4. transfers, on the transfer stream behind everything queued so far on the context's stream */
	DIST_HIP(d, hipEventRecord(d->ev_ready, ctx->stream));
	DIST_HIP(d, hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0));
	for (int c = 0; c < ncols; c++) {
		rc = d->t.alltoallv(d->t.self, send[c] ? send[c] : kbuf, sc, sd, out_values[c], rcn, rd, 8, d->comm_stream);
		if (rc)
			return dist_err(d, rc, "all-to-all failed%s%s", d->own_transport ? ": " : "", transport_err(d));
	}
	if (gmask) {
		rc = d->t.alltoallv(d->t.self, send_mask, sc, sd, recv_mask, rcn, rd, 8, d->comm_stream);
		if (rc)
			return dist_err(d, rc, "all-to-all failed%s%s", d->own_transport ? ": " : "", transport_err(d));
		sh_null_dst dst;
		memset(&dst, 0, sizeof(dst));
		for (int c = 0; c < ncols; c++)
			dst.bits[c] = out_nullbits[c];
		/* (on the transfer stream, right behind the mask's arrival) */
		hipLaunchKernelGGL(k_nullmask_unpack, dim3(sh_grid(total ? total : 1)), dim3(SH_THREADS), 0, d->comm_stream, (const uint64_t *)recv_mask, total,
				   gmask & (ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull)), dst);
	}
	DIST_HIP(d, hipEventRecord(d->ev_sh, d->comm_stream));
	d->sh_posted = true;
	for (int c = 0; c < ncols; c++)
		if (send[c] && send[c] != kbuf)
			sh_pend(d, send[c]);
	for (int k = 0; k < ncomp; k++)
		sh_pend(d, comp[k]);
	sh_pend(d, send_mask);
	sh_pend(d, recv_mask);
	sh_pend(d, kbuf);
	sh_pend(d, pos);
	sh_pend(d, canon);
	*out_n = total;
	if (!(flags & MDB_DIST_NO_WAIT))
		return mdb_dist_wait_transfers(d);
	return MIDORIDB_OK;
}

/* ---- every rank's rows to EVERY rank: the small side of a join that has no equi-join key (FROM A, B; a general ON expression -
 * reference _join_nested_loop_tbl2tbl with any predicate, executor_select.c:1096-1141, optimiser_select.c:395-464): nothing says which
 * rank a row's partners live on, so the table is replicated and every rank pairs ITS rows of the other side with all of it - each
 * (l, r) pair is produced exactly once, on the rank that holds l.  One count exchange (rows per rank + which columns carry NULL bits +
 * status), one all-to-all per column in which every peer is sent the SAME buffer, NULL bits as one word per row. */
extern "C" int mdb_dist_broadcast_rows(mdb_dist *d, uint64_t n, const struct mdb_dist_col *cols, int ncols, void **out_values,
				       uint64_t **out_nullbits, uint64_t *out_n)
{
	if (!d || !out_n || ncols < 1 || ncols > MDB_DIST_SHUFFLE_MAX_COLS || !cols || !out_values || !out_nullbits)
		return d ? dist_err(d, -MIDORIDB_ERROR, "broadcast_rows: bad arguments") : -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = d->ctx;
	const int W = d->world;
	*out_n = 0;
	for (int c = 0; c < ncols; c++) {
		out_values[c] = NULL;
		out_nullbits[c] = NULL;
	}
	DIST_HIP(d, hipSetDevice(ctx->device));
	if (d->npend > SH_MAX_PENDING / 2) {
		int wrc = mdb_dist_wait_transfers(d);
		if (wrc)
			return wrc;
	}
	uint64_t sendv[SH_COUNTERS << MDB_MAX_RADIX_BITS], recvv[SH_COUNTERS << MDB_MAX_RADIX_BITS];
	size_t sc[1 << MDB_MAX_RADIX_BITS], sd[1 << MDB_MAX_RADIX_BITS], rcn[1 << MDB_MAX_RADIX_BITS], rd[1 << MDB_MAX_RADIX_BITS];
	/* ---- the rows in one contiguous buffer per column (a stream read through row ids is gathered first), and the mask words */
	uint64_t colmask = 0;
	for (int c = 0; c < ncols; c++)
		if (cols[c].nullbits)
			colmask |= 1ull << c;
	void *send[MDB_DIST_SHUFFLE_MAX_COLS], *own[MDB_DIST_SHUFFLE_MAX_COLS + 2];
	int nown = 0;
	int status = n >= 0xFFFFFFFFull ? -MIDORIDB_ERROR : MIDORIDB_OK;
	for (int c = 0; c < ncols && !status; c++) {
		send[c] = const_cast<void *>(cols[c].values);
		if (cols[c].rid && n) {
			void *buf = NULL;
			status = mdb_dev_alloc(ctx, n * 8, &buf);
			if (!status) {
				own[nown++] = buf;
				status = mdb_dev_gather64(ctx, (const int64_t *)cols[c].values, NULL, cols[c].rid, n, (int64_t *)buf, NULL);
				send[c] = buf;
			}
		}
	}
	void *send_mask = NULL;
	if (!status && colmask && n) {
		status = mdb_dev_alloc(ctx, n * 8, &send_mask);
		if (!status) {
			own[nown++] = send_mask;
			sh_null_src src;
			memset(&src, 0, sizeof(src));
			for (int c = 0; c < ncols; c++) {
				src.nb[c] = cols[c].nullbits;
				src.rid[c] = cols[c].rid;
			}
			hipLaunchKernelGGL(k_nullmask_pack, dim3(sh_grid(n)), dim3(SH_THREADS), 0, ctx->stream, src, colmask, (const uint32_t *)NULL, n, (uint64_t *)send_mask);
		}
	}
	char local_err[256];
	snprintf(local_err, sizeof(local_err), "%s", status ? mdb_dev_last_error(ctx) : "");
	for (int p = 0; p < W; p++) {
		sendv[SH_COUNTERS * p] = status ? 0 : n;
		sendv[SH_COUNTERS * p + 1] = colmask;
		sendv[SH_COUNTERS * p + 2] = status ? 1 : 0;
	}
	int rc = d->t.counts(d->t.self, sendv, recvv, SH_COUNTERS);
	uint64_t total = 0, gmask = 0;
	int failed_rank = -1;
	for (int p = 0; p < W && !rc; p++) {
		sc[p] = (size_t)n;
		sd[p] = 0;		/* every peer is sent the same rows */
		rcn[p] = (size_t)recvv[SH_COUNTERS * p];
		rd[p] = (size_t)total;
		total += recvv[SH_COUNTERS * p];
		gmask |= recvv[SH_COUNTERS * p + 1];
		if (recvv[SH_COUNTERS * p + 2] && failed_rank < 0)
			failed_rank = p;
	}
	if (!rc && failed_rank < 0 && total >= 0xFFFFFFFFull) {
		status = -MIDORIDB_ERROR;
		snprintf(local_err, sizeof(local_err), "%llu broadcast rows exceed the 32-bit row-id limit of one shard", (unsigned long long)total);
	}
	/* receive buffers; whether every rank got them is agreed on before anything is posted (as in mdb_dist_shuffle_rows) */
	void *recv_mask = NULL;
	int arc = MIDORIDB_OK;
	if (!rc && failed_rank < 0 && !status) {
		for (int c = 0; c < ncols && !arc; c++) {
			arc = mdb_dev_alloc(ctx, (total ? total : 1) * 8, &out_values[c]);
			if (!arc && (gmask >> c & 1))
				arc = mdb_dev_alloc(ctx, ((total + 63) / 64 + 1) * 8, (void **)&out_nullbits[c]);
		}
		if (!arc && gmask)
			arc = mdb_dev_alloc(ctx, (total ? total : 1) * 8, &recv_mask);
		if (!arc && gmask && !send_mask && n) {	/* (another rank has NULLs, this one has none: all-zero words) */
			arc = mdb_dev_alloc(ctx, n * 8, &send_mask);
			if (!arc) {
				own[nown++] = send_mask;
				arc = mdb_dev_memset(ctx, send_mask, 0, n * 8);
			}
		}
		for (int p = 0; p < W; p++)
			sendv[p] = arc ? 1 : 0;
		int rc2 = d->t.counts(d->t.self, sendv, recvv, 1);
		for (int p = 0; p < W && !rc2; p++)
			if (recvv[p] && !arc)
				arc = -MIDORIDB_NOMEM;
		if (rc2)
			rc = rc2;
	}
	if (rc || failed_rank >= 0 || status || arc) {
		for (int c = 0; c < ncols; c++) {
			(void)mdb_dev_free(ctx, out_values[c]);
			(void)mdb_dev_free(ctx, out_nullbits[c]);
			out_values[c] = NULL;
			out_nullbits[c] = NULL;
		}
		(void)mdb_dev_free(ctx, recv_mask);
		for (int k = 0; k < nown; k++)
			(void)mdb_dev_free(ctx, own[k]);
		if (rc)
			return dist_err(d, rc, "count exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d));
		if (status)
			return dist_err(d, status, "broadcast: %s", local_err);
		if (failed_rank >= 0)
			return dist_err(d, -MIDORIDB_ERROR, "broadcast: rank %d failed (its own message says why); nothing was exchanged", failed_rank);
		return dist_err(d, arc, "broadcast: a rank could not allocate its buffers; nothing was exchanged");
	}
	DIST_HIP(d, hipEventRecord(d->ev_ready, ctx->stream));
	DIST_HIP(d, hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0));
	/* (a rank without rows still takes part: its sends are empty, and a valid pointer stands in for its buffers) */
	for (int c = 0; c < ncols; c++) {
		rc = d->t.alltoallv(d->t.self, n ? send[c] : out_values[c], sc, sd, out_values[c], rcn, rd, 8, d->comm_stream);
		if (rc)
			return dist_err(d, rc, "all-to-all failed%s%s", d->own_transport ? ": " : "", transport_err(d));
	}
	if (gmask) {
		rc = d->t.alltoallv(d->t.self, n ? send_mask : recv_mask, sc, sd, recv_mask, rcn, rd, 8, d->comm_stream);
		if (rc)
			return dist_err(d, rc, "all-to-all failed%s%s", d->own_transport ? ": " : "", transport_err(d));
		sh_null_dst dst;
		memset(&dst, 0, sizeof(dst));
		for (int c = 0; c < ncols; c++)
			dst.bits[c] = out_nullbits[c];
		hipLaunchKernelGGL(k_nullmask_unpack, dim3(sh_grid(total ? total : 1)), dim3(SH_THREADS), 0, d->comm_stream, (const uint64_t *)recv_mask, total,
				   gmask & (ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull)), dst);
	}
	DIST_HIP(d, hipEventRecord(d->ev_sh, d->comm_stream));
	d->sh_posted = true;
	for (int k = 0; k < nown; k++)
		sh_pend(d, own[k]);
	sh_pend(d, recv_mask);
	*out_n = total;
	return mdb_dist_wait_transfers(d);
}

/* ---- a join whose only output is the key column (BASELINE configs[3]: SELECT * over two key columns): no row has to be
 * identified, so the regions-on-the-wire operator answers it - (key, COUNT) per key that occurs on both sides, every key then
 * written COUNT times (unique keys: the groups ARE the joined rows) */
/* 0 = done, 1 = not served (every rank alike), < 0 = error */
static int dist_join_keys_only(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r, const uint64_t *null_r,
			       uint64_t n_r, int64_t **out_key, uint64_t *out_rows)
{
	mdb_dev_ctx *ctx = d->ctx;
	int64_t glo[2] = { INT64_MIN, INT64_MIN }, ghi[2] = { INT64_MAX, INT64_MAX };
	bool promised = false, fits32 = false;
	if (d->wire_mode == MDB_WIRE_AUTO) {
		const int mrc = dist_measure_ranges(d, keys_l, null_l, n_l, keys_r, null_r, n_r, glo, ghi, &fits32);
		if (mrc)
			return mrc;
	} else if (d->have_ranges) {
		for (int i = 0; i < 2; i++) {
			glo[i] = d->promised_lo[i];
			ghi[i] = d->promised_hi[i];
		}
		promised = true;
	} else {
		return 1;
	}
	const int64_t *fk[2] = { keys_l, keys_r };
	const uint64_t *fn[2] = { null_l, null_r };
	const uint64_t fs[2] = { n_l, n_r };
	int64_t *gk = NULL, *gc = NULL;
	uint64_t G = 0, J = 0;
	d->last_fused = 0;
	/* first WITHOUT the COUNT column (primary-key joins: every COUNT is 1 - 8 of the 16 bytes a group costs the leaf kernel); the ranks
	 * then agree on whether that was right everywhere (one tiny exchange) and, if not, all of them run the call again with counts */
	/* (what the previous keys-only call over this handle found is remembered - every rank saw the same sum, so every rank remembers the
	 * same thing: a foreign-key or N:M join asks for the COUNT column at once instead of paying two whole exchanges per call) */
	const bool with_counts = d->ko_had_dups;
	int rc = dist_join_fused(d, 2, fk, fn, fs, glo, ghi, promised, true, &gk, &gc, 0, &G, &J, false, !with_counts);
	if (rc)
		return rc;
	{
		uint64_t dups = J != G ? 1u : 0u;
		rc = d->t.allreduce_sum_u64(d->t.self, &dups, 1);
		if (rc) {
			(void)mdb_dev_free(ctx, gk);
			(void)mdb_dev_free(ctx, gc);
			return dist_err(d, rc, "status exchange failed%s%s", d->own_transport ? ": " : "", transport_err(d));
		}
		d->ko_had_dups = dups != 0;
		if (dups && !with_counts) {
			(void)mdb_dev_free(ctx, gk);
			(void)mdb_dev_free(ctx, gc);
			gk = gc = NULL;
			d->last_fused = 0;
			rc = dist_join_fused(d, 2, fk, fn, fs, glo, ghi, promised, true, &gk, &gc, 0, &G, &J);
			if (rc)
				return rc;
		}
	}
	if (J == G) {		/* every COUNT is 1: the group keys are the joined rows */
		(void)mdb_dev_free(ctx, gc);
		*out_key = gk;
		*out_rows = G;
		return 0;
	}
	int64_t *rows = NULL;
	rc = mdb_expand_keys_by_count(ctx, gk, gc, G, J, &rows);
	(void)mdb_dev_free(ctx, gk);
	(void)mdb_dev_free(ctx, gc);
	if (rc)
		return dist_err(d, rc, "expanding the joined keys: %s", mdb_dev_last_error(ctx));
	*out_key = rows;
	*out_rows = J;
	return 0;
}

extern "C" int mdb_dist_join_pairs(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const struct mdb_dist_col *cols_l,
				   int ncols_l, const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, const struct mdb_dist_col *cols_r,
				   int ncols_r, int64_t **out_key, void **out_l, uint64_t **out_l_nulls, void **out_r, uint64_t **out_r_nulls,
				   uint64_t *out_rows)
{
	if (!d || !out_rows || ncols_l < 0 || ncols_r < 0 || ncols_l >= MDB_DIST_SHUFFLE_MAX_COLS || ncols_r >= MDB_DIST_SHUFFLE_MAX_COLS ||
	    (ncols_l && (!cols_l || !out_l || !out_l_nulls)) || (ncols_r && (!cols_r || !out_r || !out_r_nulls)))
		return d ? dist_err(d, -MIDORIDB_ERROR, "join_pairs: bad arguments") : -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = d->ctx;
	*out_rows = 0;
	if (out_key)
		*out_key = NULL;
	if (out_key && !ncols_l && !ncols_r) {
		/* only the key column is wanted: nothing has to say WHICH rows met - the regions-on-the-wire operator counts the
		 * keys' partners and every key is written COUNT times (rows then come out in leaf order, not in received-row order) */
		const int krc = dist_join_keys_only(d, keys_l, null_l, n_l, keys_r, null_r, n_r, out_key, out_rows);
		if (krc <= 0)
			return krc;
	}
	/* column 0 of each shuffle = the key itself */
	struct mdb_dist_col cl[MDB_DIST_SHUFFLE_MAX_COLS], cr[MDB_DIST_SHUFFLE_MAX_COLS];
	void *vl[MDB_DIST_SHUFFLE_MAX_COLS], *vr[MDB_DIST_SHUFFLE_MAX_COLS];
	uint64_t *bl[MDB_DIST_SHUFFLE_MAX_COLS], *br[MDB_DIST_SHUFFLE_MAX_COLS];
	cl[0].values = keys_l;
	cl[0].nullbits = NULL;	/* (rows with a NULL key stay home) */
	cl[0].rid = NULL;
	cr[0].values = keys_r;
	cr[0].nullbits = NULL;
	cr[0].rid = NULL;
	for (int c = 0; c < ncols_l; c++)
		cl[c + 1] = cols_l[c];
	for (int c = 0; c < ncols_r; c++)
		cr[c + 1] = cols_r[c];
	uint64_t got_l = 0, got_r = 0, J = 0;
	uint32_t *pl = NULL, *pr = NULL;
	/* table L's transfers run while table R is partitioned and gathered */
	int rc = mdb_dist_shuffle_rows(d, keys_l, null_l, n_l, MDB_DIST_NO_WAIT, cl, ncols_l + 1, vl, bl, &got_l);
	int rc2 = mdb_dist_shuffle_rows(d, keys_r, null_r, n_r, MDB_DIST_NO_WAIT, cr, ncols_r + 1, vr, br, &got_r);	/* (collective: also after a failure) */
	int rc3 = mdb_dist_wait_transfers(d);
	if (!rc)
		rc = rc2 ? rc2 : rc3;
	if (rc) {	/* (a failed shuffle has released its own outputs and left NULLs behind) */
		for (int c = 0; c <= ncols_l; c++) {
			(void)mdb_dev_free(ctx, vl[c]);
			(void)mdb_dev_free(ctx, bl[c]);
		}
		for (int c = 0; c <= ncols_r; c++) {
			(void)mdb_dev_free(ctx, vr[c]);
			(void)mdb_dev_free(ctx, br[c]);
		}
		return rc;
	}
	d->last_recv_left = got_l;
	if (got_l && got_r) {
		rc = mdb_dev_join_pairs(ctx, (const int64_t *)vl[0], NULL, got_l, (const int64_t *)vr[0], NULL, got_r, &pl, &pr, &J);
		if (rc)
			dist_err(d, rc, "local join: %s", mdb_dev_last_error(ctx));
	}
	/* projection: every output column of the joined rows in one launch per 16 */
	struct mdb_gather_col gl[MDB_GATHER_MAX_COLS];
	int ngl = 0;
	const int nout = (out_key ? 1 : 0) + ncols_l + ncols_r;
	for (int o = 0; o < nout && !rc; o++) {
		const bool is_key = out_key && o == 0;
		const int k = o - (out_key ? 1 : 0);
		const bool left = is_key || k < ncols_l;
		const int c = is_key ? 0 : (left ? k + 1 : k - ncols_l + 1);
		void *dst = NULL;
		uint64_t *dnb = NULL;
		const uint64_t *snb = left ? bl[c] : br[c];
		rc = mdb_dev_alloc(ctx, (J ? J : 1) * 8, &dst);
		if (!rc && snb && !is_key)
			rc = mdb_dev_alloc(ctx, ((J + 63) / 64 + 1) * 8, (void **)&dnb);
		if (is_key)
			*out_key = (int64_t *)dst;
		else if (left) {
			out_l[k] = dst;
			out_l_nulls[k] = dnb;
		} else {
			out_r[k - ncols_l] = dst;
			out_r_nulls[k - ncols_l] = dnb;
		}
		if (rc) {
			dist_err(d, rc, "allocating the joined columns: %s", mdb_dev_last_error(ctx));
			break;
		}
		if (!J)
			continue;
		if (ngl == MDB_GATHER_MAX_COLS) {
			rc = mdb_dev_gather_cols(ctx, gl, ngl, J);
			ngl = 0;
			if (rc) {
				dist_err(d, rc, "projection: %s", mdb_dev_last_error(ctx));
				break;
			}
		}
		gl[ngl].src = left ? vl[c] : vr[c];
		gl[ngl].src_nullbits = is_key ? NULL : snb;
		gl[ngl].rid = left ? pl : pr;
		gl[ngl].dst = dst;
		gl[ngl].dst_nullbits = dnb;
		ngl++;
	}
	if (!rc && ngl && (rc = mdb_dev_gather_cols(ctx, gl, ngl, J)))
		dist_err(d, rc, "projection: %s", mdb_dev_last_error(ctx));
	for (int c = 0; c <= ncols_l; c++) {
		(void)mdb_dev_free(ctx, vl[c]);
		(void)mdb_dev_free(ctx, bl[c]);
	}
	for (int c = 0; c <= ncols_r; c++) {
		(void)mdb_dev_free(ctx, vr[c]);
		(void)mdb_dev_free(ctx, br[c]);
	}
	(void)mdb_dev_free(ctx, pl);
	(void)mdb_dev_free(ctx, pr);
	if (rc) {
		if (out_key) {
			(void)mdb_dev_free(ctx, *out_key);
			*out_key = NULL;
		}
		for (int k = 0; k < ncols_l; k++) {
			(void)mdb_dev_free(ctx, out_l[k]);
			(void)mdb_dev_free(ctx, out_l_nulls[k]);
			out_l[k] = NULL;
			out_l_nulls[k] = NULL;
		}
		for (int k = 0; k < ncols_r; k++) {
			(void)mdb_dev_free(ctx, out_r[k]);
			(void)mdb_dev_free(ctx, out_r_nulls[k]);
			out_r[k] = NULL;
			out_r_nulls[k] = NULL;
		}
		return rc;
	}
	*out_rows = J;
	return mdb_dev_sync(ctx) ? dist_err(d, -MIDORIDB_INTERNAL, "%s", mdb_dev_last_error(ctx)) : MIDORIDB_OK;
}

/* One left table and 2 ... 3 right tables, all joined on ONE key (A JOIN B ON a = b JOIN C ON a = c ... GROUP BY a, COUNT(*):
 * BASELINE configs[4]) over sharded tables, as ONE exchange: every table partitioned once with the same window hash, every
 * rank joins the regions of ALL tables it received and multiplies the right tables' counts per key (mdb_dev_shard.hip).
 * Returns 0 = done (outputs allocated by the call, as mdb_dist_join_group_count_alloc), 1 = not served (the key ranges are not
 * known or do not fit, skewed keys ...: every rank gets the same answer and the caller chains two-table calls), < 0 = error. */
static int dist_group_keys_impl(mdb_dist *d, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t **out_key, int64_t **out_count,
				uint64_t *out_groups)
{
	int64_t glo[2] = { 0, 0 }, ghi[2] = { 0, 0 };
	bool promised = false, fits32 = false;
	if (nullbits)
		return 1;	/* (every rank passes NULL or none does: the caller agrees on that first) */
	if (d->wire_mode == MDB_WIRE_AUTO) {
		const int mrc = dist_measure_ranges(d, keys, NULL, n, keys, NULL, n, glo, ghi, &fits32);
		if (mrc)
			return mrc;
	} else if (d->have_ranges) {
		glo[1] = d->promised_lo[1];
		ghi[1] = d->promised_hi[1];
		promised = true;
	} else {
		return 1;
	}
	glo[0] = glo[1];	/* (the absent left table's range: the right table's, nothing is pruned) */
	ghi[0] = ghi[1];
	const int64_t *kk[2] = { NULL, keys };
	const uint64_t *nn[2] = { NULL, NULL };
	const uint64_t ns[2] = { 0, n };
	uint64_t joined = 0;
	d->last_fused = 0;
	return dist_join_fused(d, 2, kk, nn, ns, glo, ghi, promised, true, out_key, out_count, 0, out_groups, &joined, true);
}

extern "C" int mdb_dist_group_count_keys_alloc(mdb_dist *d, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t **out_key,
					       int64_t **out_count, uint64_t *out_groups)
{
	if (!d || !out_key || !out_count || !out_groups)
		return d ? dist_err(d, -MIDORIDB_ERROR, "group_count_keys: bad arguments") : -MIDORIDB_ERROR;
	*out_groups = 0;
	*out_key = NULL;
	*out_count = NULL;
	DIST_HIP(d, hipSetDevice(d->ctx->device));
	return dist_group_keys_impl(d, keys, nullbits, n, out_key, out_count, out_groups);
}

extern "C" int mdb_dist_join_group_count_multi_alloc(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, int n_right,
							     const int64_t *const *keys_r, const uint64_t *const *null_r, const uint64_t *n_r,
							     int64_t **out_key, int64_t **out_count, uint64_t *out_groups, uint64_t *out_joined)
{
	if (!d || !out_key || !out_count || !out_groups || n_right < 1 || n_right + 1 > MDB_SHARD_MAX_TABS || !keys_r || !n_r)
		return d ? dist_err(d, -MIDORIDB_ERROR, "join_group_count_multi: bad arguments") : -MIDORIDB_ERROR;
	*out_groups = 0;
	if (out_joined)
		*out_joined = 0;
	*out_key = NULL;
	*out_count = NULL;
	DIST_HIP(d, hipSetDevice(d->ctx->device));
	int64_t glo[2] = { INT64_MIN, INT64_MIN }, ghi[2] = { INT64_MAX, INT64_MAX };
	bool promised = false, fits32 = false;
	if (d->wire_mode == MDB_WIRE_AUTO) {
		const int mrc = dist_measure_ranges(d, keys_l, null_l, n_l, keys_r[0], null_r ? null_r[0] : NULL, n_r[0], glo, ghi, &fits32);
		if (mrc)
			return mrc;
	} else if (d->have_ranges) {
		for (int i = 0; i < 2; i++) {
			glo[i] = d->promised_lo[i];
			ghi[i] = d->promised_hi[i];
		}
		promised = true;
	} else {
		return 1;
	}
	const int64_t *keys[MDB_SHARD_MAX_TABS];
	const uint64_t *nulls[MDB_SHARD_MAX_TABS];
	uint64_t ns[MDB_SHARD_MAX_TABS];
	keys[0] = keys_l;
	nulls[0] = null_l;
	ns[0] = n_l;
	for (int t = 0; t < n_right; t++) {
		keys[t + 1] = keys_r[t];
		nulls[t + 1] = null_r ? null_r[t] : NULL;
		ns[t + 1] = n_r[t];
	}
	d->last_fused = 0;
	return dist_join_fused(d, n_right + 1, keys, nulls, ns, glo, ghi, promised, true, out_key, out_count, 0, out_groups, out_joined);
}
