/*
 * mdb_dev_core.hip - context, scratch arena, profiling, memory helpers, the exclusive-scan
 * primitive and the small streaming kernels (gather / iota / cross pairs / synthetic keys).
 * Hand-written HIP for gfx950 (wave64); HBM-bound byte work, no MFMA.
 */
#include "mdb_dev_internal.h"
#include <stdarg.h>
#include <stdlib.h>
#include "mdb_gen.h"
#include <cxxabi.h>

int mdb_set_err(mdb_dev_ctx *ctx, int code, const char *fmt, ...)
{
	if (ctx) {
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
		va_end(ap);
	}
	return code;
}

/* ------------------------------------------------------------------ context */

extern "C" int mdb_dev_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return -MIDORIDB_INTERNAL;
	return n;
}


/* ------------------------------------------------------------------ environment knobs: ONE reader
 *
 * Every MDB_* knob of the library (INTEGRATION.md lists them) is read through mdb_knob(): looked up in the environment ONCE per process and
 * kept - the operators ask for a handful per call, on the call path.  A process that changes a knob while it runs (tests, same-process A/B
 * scripts) says so: mdb_dev_reload_knobs() (the Python binding does it whenever os.environ changes an MDB_* variable). */
#include <mutex>
static std::mutex g_knob_mu;
static struct { const char *name; char *value; } g_knobs[192];
static int g_nknobs = 0;

extern "C" const char *mdb_knob(const char *name)
{
	std::lock_guard<std::mutex> g(g_knob_mu);
	for (int i = 0; i < g_nknobs; i++)
		if (g_knobs[i].name == name || !strcmp(g_knobs[i].name, name))
			return g_knobs[i].value;
	const char *v = getenv(name);
	if (g_nknobs == (int)(sizeof(g_knobs) / sizeof(g_knobs[0])))
		return v;	/* (more knobs than slots: read through) */
	g_knobs[g_nknobs].name = name;		/* (string literals of the library) */
	g_knobs[g_nknobs].value = v ? strdup(v) : NULL;
	return g_knobs[g_nknobs++].value;
}

extern "C" void mdb_dev_reload_knobs(void)
{
	std::lock_guard<std::mutex> g(g_knob_mu);
	for (int i = 0; i < g_nknobs; i++)
		free(g_knobs[i].value);
	g_nknobs = 0;
}

static void memo_reset(mdb_col_memo &m)
{
	memset(&m, 0, sizeof(m));	/* (plain data: pointers, counters, flags) */
	m.nh_result = -1;
}

extern "C" int mdb_dev_ctx_create(int device, void *stream, mdb_dev_ctx **out)
{
	if (!out)
		return -MIDORIDB_ERROR;
	*out = NULL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
		return -MIDORIDB_INTERNAL;
	if (hipSetDevice(device) != hipSuccess)
		return -MIDORIDB_INTERNAL;
	mdb_dev_ctx *ctx = new (std::nothrow) mdb_dev_ctx();
	if (!ctx)
		return -MIDORIDB_NOMEM;
	ctx->device = device;
	ctx->num_cus = 256;
	(void)hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device);
	if (ctx->num_cus <= 0)
		ctx->num_cus = 256;
	ctx->err[0] = 0;
	ctx->arena = NULL;
	ctx->arena_cap = ctx->arena_off = 0;
	ctx->prof_on = false;
	ctx->prof_pool_used = 0;
	ctx->d_status = NULL;
	ctx->h_pinned = NULL;
	if (stream == MDB_STREAM_OWN) {
		if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
			delete ctx;
			return -MIDORIDB_INTERNAL;
		}
		ctx->own_stream = true;
	} else {
		ctx->stream = (hipStream_t)stream;	/* NULL = the device's default stream */
		ctx->own_stream = false;
	}
	ctx->aux_stream = NULL;
	ctx->ev_fork = ctx->ev_join = NULL;
	ctx->pending_op = NULL;
	ctx->cache_bytes = 0;
	ctx->narrow_mode = 1;
	ctx->last_narrow = 0;
	ctx->last_semijoin = 0;
	memo_reset(*ctx);
	ctx->memo_key = mdb_memo_key{ NULL, NULL, 0, 0 };
	{
		const char *e = mdb_knob("MDB_NARROW_KEYS");	/* whole-suite soaks: force one form (see mdb_dev_set_narrow_keys) */
		if (e && e[0] >= '0' && e[0] <= '2' && !e[1])
			ctx->narrow_mode = e[0] - '0';
	}
	ctx->overlap = false;	/* measured: no gain on one GPU (each kernel already fills the chip), kept for the multi-GPU exchange */
	if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
		mdb_dev_ctx_destroy(ctx);
		return -MIDORIDB_INTERNAL;
	}
	if (hipMalloc((void **)&ctx->d_status, 64 * sizeof(uint64_t) + (size_t)MDB_ZERO_BLK_WORDS * 4) != hipSuccess ||
	    hipHostMalloc((void **)&ctx->h_pinned, 1024 * sizeof(uint64_t)) != hipSuccess ||
	    hipMemsetAsync(ctx->d_status, 0, 64 * sizeof(uint64_t), ctx->stream) != hipSuccess) {
		mdb_dev_ctx_destroy(ctx);
		return -MIDORIDB_INTERNAL;
	}
	*out = ctx;
	return MIDORIDB_OK;
}

extern "C" void mdb_dev_ctx_destroy(mdb_dev_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	for (auto &p : ctx->prof_pool) {
		(void)hipEventDestroy(p.first);
		(void)hipEventDestroy(p.second);
	}
	if (ctx->aux_stream) {
		(void)hipStreamSynchronize(ctx->aux_stream);
		(void)hipStreamDestroy(ctx->aux_stream);
	}
	if (ctx->ev_fork)
		(void)hipEventDestroy(ctx->ev_fork);
	if (ctx->ev_join)
		(void)hipEventDestroy(ctx->ev_join);
	free(ctx->pending_op);
	for (auto &c : ctx->cache)
		(void)hipFree(c.first);
	for (auto &l : ctx->live)
		(void)hipFree(l.first);
	if (ctx->arena)
		(void)hipFree(ctx->arena);
	if (ctx->d_status)
		(void)hipFree(ctx->d_status);
	if (ctx->h_pinned)
		(void)hipHostFree(ctx->h_pinned);
	if (ctx->own_stream)
		(void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" int mdb_dev_ctx_set_stream(mdb_dev_ctx *ctx, void *stream)
{
	if (!ctx)
		return -MIDORIDB_ERROR;
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->own_stream) {
		MDB_HIP(ctx, hipStreamDestroy(ctx->stream));
		ctx->own_stream = false;
	}
	if (stream == MDB_STREAM_OWN) {
		MDB_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
		ctx->own_stream = true;
	} else {
		ctx->stream = (hipStream_t)stream;
	}
	return MIDORIDB_OK;
}

extern "C" const char *mdb_dev_last_error(mdb_dev_ctx *ctx)
{
	return ctx ? ctx->err : "no context";
}

extern "C" int mdb_dev_sync(mdb_dev_ctx *ctx)
{
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ two-stream fork / join */

extern "C" int mdb_dev_set_overlap(mdb_dev_ctx *ctx, int on)
{
	ctx->overlap = on != 0;
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_last_join_narrow(mdb_dev_ctx *ctx)
{
	return ctx->last_narrow;
}

extern "C" int mdb_dev_last_pairs_identity(mdb_dev_ctx *ctx)
{
	return ctx ? ctx->last_pairs_identity : 0;
}

extern "C" int mdb_dev_last_join_filter(mdb_dev_ctx *ctx)
{
	return ctx->last_semijoin;
}

extern "C" int mdb_dev_call_stats(mdb_dev_ctx *ctx, const void *keys_l, const struct mdb_dev_col_stats *l, const void *keys_r, const struct mdb_dev_col_stats *r)
{
	if (!ctx)
		return -MIDORIDB_ERROR;
	ctx->cs_on = false;
	ctx->cs_has_r = false;
	ctx->cs_kl = ctx->cs_kr = NULL;
	if (!keys_l || !l || (mdb_knob("MDB_CALL_STATS") && mdb_knob("MDB_CALL_STATS")[0] == '0'))	/* (the knob: A/B runs against the sampled decisions) */
		return MIDORIDB_OK;
	if ((keys_r != NULL) != (r != NULL))
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "mdb_dev_call_stats: a right key column and its statistics go together");
	ctx->cs_on = true;
	ctx->cs_kl = keys_l;
	ctx->cs_l = *l;
	if (keys_r) {
		ctx->cs_has_r = true;
		ctx->cs_kr = keys_r;
		ctx->cs_r = *r;
	}
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_last_plan(mdb_dev_ctx *ctx, struct mdb_dev_plan_info *out)
{
	if (!ctx || !out)
		return -MIDORIDB_ERROR;
	memset(out, 0, sizeof(*out));
	const int f = ctx->last_semijoin;
	out->key_form = (uint32_t)ctx->last_narrow;
	out->key_bits = ctx->last_narrow == 2 ? ctx->pl_key_bits : 0u;
	out->levels = (f & 0x1200) ? 1u : 2u;
	out->digits = (f & 0x1000) ? 4096u : 512u;
	out->minmax_pruned = (f & 0x100) ? 1u : 0u;
	out->semijoin = (uint32_t)(f & 0xFF);
	out->any_order = (f & 0x800) ? 1u : 0u;
	out->ranged_order = (f & 0x2000) ? 1u : 0u;
	out->multi_one_pass = (f & 0x400) ? 1u : 0u;
	out->retries = ctx->pl_retries;
	out->samples = ctx->pl_samples;
	out->from_stats = ctx->pl_from_stats;
	out->payload_form = ctx->pl_payload_form;
	out->group_form = ctx->pl_group_form;
	out->groups_as_bits = ctx->pl_bits;
	out->small_form = ctx->pl_small_form;
	out->keys_are_left_column = ctx->pl_keys_left;
	out->counts_all_one = ctx->pl_counts_one;
	out->payload_tables = ctx->pl_payload_tables;
	return MIDORIDB_OK;
}


extern "C" int mdb_dev_counters(mdb_dev_ctx *ctx, struct mdb_dev_counters *out)
{
	if (!ctx || !out)
		return -MIDORIDB_ERROR;
	out->operator_calls = ctx->ct_calls;
	out->retries = ctx->ct_retries;
	out->samples = ctx->ct_samples;
	out->arena_grows = ctx->ct_arena_grows;
	out->alloc_misses = ctx->ct_alloc_misses;
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_set_narrow_keys(mdb_dev_ctx *ctx, int mode)
{
	if (mode < 0 || mode > 2)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "narrow-key mode must be 0, 1 or 2");
	ctx->narrow_mode = mode;
	return MIDORIDB_OK;
}

int mdb_aux_begin(mdb_dev_ctx *ctx, hipStream_t *saved_main)
{
	*saved_main = ctx->stream;
	if (!ctx->overlap)
		return MIDORIDB_OK;
	MDB_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
	MDB_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
	ctx->stream = ctx->aux_stream;
	return MIDORIDB_OK;
}

int mdb_aux_end(mdb_dev_ctx *ctx, hipStream_t saved_main)
{
	if (!ctx->overlap)
		return MIDORIDB_OK;
	hipStream_t aux = ctx->stream;
	ctx->stream = saved_main;
	MDB_HIP(ctx, hipEventRecord(ctx->ev_join, aux));
	return MIDORIDB_OK;
}

int mdb_aux_join(mdb_dev_ctx *ctx)
{
	if (!ctx->overlap)
		return MIDORIDB_OK;
	MDB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ arena */

int mdb_arena_begin(mdb_dev_ctx *ctx, size_t total_bytes)
{
	total_bytes = mdb_align_up(total_bytes) + 4096;
	if (total_bytes > ctx->arena_cap) {
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->arena) {
			MDB_HIP(ctx, hipFree(ctx->arena));
			ctx->arena = NULL;
			ctx->arena_cap = 0;
		}
		size_t want = total_bytes + total_bytes / 16;
		ctx->ct_arena_grows++;
		hipError_t e = hipMalloc((void **)&ctx->arena, want);
		if (e != hipSuccess)
			return mdb_set_err(ctx, -MIDORIDB_NOMEM, "scratch arena of %zu bytes: %s", want, hipGetErrorString(e));
		ctx->arena_cap = want;
	}
	ctx->arena_off = 0;
	return MIDORIDB_OK;
}

void *mdb_arena_take(mdb_dev_ctx *ctx, size_t bytes)
{
	size_t off = ctx->arena_off;
	bytes = mdb_align_up(bytes ? bytes : 1);
	if (off + bytes > ctx->arena_cap) {
		mdb_set_err(ctx, -MIDORIDB_INTERNAL, "scratch arena exhausted (%zu + %zu > %zu): sizing bug", off, bytes,
			    ctx->arena_cap);
		return NULL;
	}
	ctx->arena_off = off + bytes;
	return ctx->arena + off;
}

extern "C" int mdb_dev_reserve(mdb_dev_ctx *ctx, size_t bytes)
{
	return mdb_arena_begin(ctx, bytes);
}

extern "C" size_t mdb_dev_arena_bytes(mdb_dev_ctx *ctx)
{
	return ctx->arena_cap;
}

/* ------------------------------------------------------------------ memory */

#define MDB_CACHE_MAX_BUFFERS 24
#define MDB_CACHE_MAX_BYTES ((size_t)16 << 30)

int mdb_cached_alloc(mdb_dev_ctx *ctx, size_t bytes, void **dptr)
{
	*dptr = NULL;
	if (bytes == 0)
		bytes = 8;
	/* best fit among the released buffers: big enough, at most 2x too big */
	int best = -1;
	for (size_t i = 0; i < ctx->cache.size(); i++) {
		const size_t sz = ctx->cache[i].second;
		if (sz >= bytes && sz <= 2 * bytes + 4096 && (best < 0 || sz < ctx->cache[best].second))
			best = (int)i;
	}
	if (best >= 0) {
		*dptr = ctx->cache[best].first;
		ctx->live[*dptr] = ctx->cache[best].second;
		ctx->cache_bytes -= ctx->cache[best].second;
		ctx->cache.erase(ctx->cache.begin() + best);
		return MIDORIDB_OK;
	}
	ctx->ct_alloc_misses++;
	hipError_t e = hipMalloc(dptr, bytes);
	if (e != hipSuccess) {
		/* memory pressure: drop the cache and retry once */
		(void)hipStreamSynchronize(ctx->stream);
		for (auto &c : ctx->cache)
			(void)hipFree(c.first);
		ctx->cache.clear();
		ctx->cache_bytes = 0;
		e = hipMalloc(dptr, bytes);
	}
	if (e != hipSuccess)
		return mdb_set_err(ctx, -MIDORIDB_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
	ctx->live[*dptr] = bytes;
	return MIDORIDB_OK;
}

/* What the operators remember about a key column - sampled ranges, narrow-form verdicts, duplicate flags ... - is keyed by
 * the column's ADDRESS and length.  A buffer that is released (its address will be handed out again), or written to
 * through the library (UPDATE: mdb_dev_scatter_set64; an upload into an existing buffer), takes what was learned about it
 * along: the next operator over that address samples afresh instead of trusting a verdict about other data (results never
 * depended on it - every verdict is verified on the device - but a wrong one costs a failed attempt, for up to 8 uses). */
static void memo_drop(mdb_col_memo &m, uintptr_t a, uintptr_t b)
{
	auto hit = [a, b](const void *p) { return p && (uintptr_t)p >= a && (uintptr_t)p < b; };
	if (hit(m.nh_kl) || hit(m.nh_kr))
		m.nh_result = -1;
	if (hit(m.sr_kl) || hit(m.sr_kr))
		m.sr_valid = 0;
	if (hit(m.gh_keys))
		m.gh_keys = NULL;
	if (hit(m.ex_keys))
		m.ex_keys = NULL;
	if (hit(m.pw_bad_keys))
		m.pw_bad_keys = NULL;
	if (hit(m.pu_dup_keys))
		m.pu_dup_keys = NULL;
	if (hit(m.pu_dupl_keys))
		m.pu_dupl_keys = NULL;
	if (hit(m.r32_kl) || hit(m.r32_kr))
		m.r32_ok = false;
	if (hit(m.lw_bad_keys))
		m.lw_bad_keys = NULL;
	if (hit(m.l4_bad_keys))
		m.l4_bad_keys = NULL;
	if (hit(m.lg_kl) || hit(m.lg_kr))
		m.lg_valid = false;
	if (hit(m.jk_dup_l) || hit(m.jk_dup_r))
		m.jk_dup_l = m.jk_dup_r = NULL;
	if (hit(m.jp_bad_l) || hit(m.jp_bad_r))
		m.jp_bad_l = m.jp_bad_r = NULL;
}

static void mdb_hints_drop(mdb_dev_ctx *ctx, const void *lo, size_t bytes)
{
	const uintptr_t a = (uintptr_t)lo, b = a + (bytes ? bytes : 1);
	memo_drop(*ctx, a, b);
	/* the sets put aside for other column pairs: a set whose own columns are touched goes altogether */
	for (size_t i = 0; i < ctx->memo_lru.size();) {
		const mdb_memo_key &k = ctx->memo_lru[i].first;
		const bool own = (k.kl && (uintptr_t)k.kl >= a && (uintptr_t)k.kl < b) || (k.kr && (uintptr_t)k.kr >= a && (uintptr_t)k.kr < b);
		if (own) {
			ctx->memo_lru.erase(ctx->memo_lru.begin() + (long)i);
			continue;
		}
		memo_drop(ctx->memo_lru[i].second, a, b);
		i++;
	}
}

void mdb_memo_switch(mdb_dev_ctx *ctx, const void *kl, uint64_t nl, const void *kr, uint64_t nr)
{
	const mdb_memo_key key{ kl, kr, nl, kr ? nr : 0 };
	if (ctx->memo_key == key)
		return;
	/* put the live set aside under the pair it belongs to */
	if (ctx->memo_key.kl) {
		size_t i = 0;
		while (i < ctx->memo_lru.size() && !(ctx->memo_lru[i].first == ctx->memo_key))
			i++;
		if (i < ctx->memo_lru.size())
			ctx->memo_lru.erase(ctx->memo_lru.begin() + (long)i);
		ctx->memo_lru.push_back(std::make_pair(ctx->memo_key, static_cast<const mdb_col_memo &>(*ctx)));
		if (ctx->memo_lru.size() > MDB_MEMO_SLOTS)
			ctx->memo_lru.erase(ctx->memo_lru.begin());
	}
	size_t i = 0;
	while (i < ctx->memo_lru.size() && !(ctx->memo_lru[i].first == key))
		i++;
	if (i < ctx->memo_lru.size()) {
		static_cast<mdb_col_memo &>(*ctx) = ctx->memo_lru[i].second;
		ctx->memo_lru.erase(ctx->memo_lru.begin() + (long)i);
	} else {
		memo_reset(*ctx);
	}
	ctx->memo_key = key;
}

int mdb_cached_free(mdb_dev_ctx *ctx, void *dptr)
{
	if (!dptr)
		return MIDORIDB_OK;
	auto it = ctx->live.find(dptr);
	if (it == ctx->live.end()) {
		/* not ours (or already released): plain free */
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		MDB_HIP(ctx, hipFree(dptr));
		return MIDORIDB_OK;
	}
	{
		auto hs = ctx->holders.find(dptr);
		if (hs != ctx->holders.end()) {		/* (other holders read it: this one is gone, the buffer stays) */
			if (--hs->second == 0)
				ctx->holders.erase(hs);
			return MIDORIDB_OK;
		}
	}
	const size_t sz = it->second;
	mdb_hints_drop(ctx, dptr, sz);
	ctx->live.erase(it);
	/* reuse is ordered by the context's stream: whoever gets the buffer next launches after every kernel
	 * that still reads it */
	ctx->cache.push_back(std::make_pair(dptr, sz));
	ctx->cache_bytes += sz;
	while (ctx->cache.size() > MDB_CACHE_MAX_BUFFERS || ctx->cache_bytes > MDB_CACHE_MAX_BYTES) {
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		ctx->cache_bytes -= ctx->cache.front().second;
		MDB_HIP(ctx, hipFree(ctx->cache.front().first));
		ctx->cache.erase(ctx->cache.begin());
	}
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_alloc(mdb_dev_ctx *ctx, size_t bytes, void **dptr)
{
	return mdb_cached_alloc(ctx, bytes, dptr);
}

extern "C" int mdb_dev_free(mdb_dev_ctx *ctx, void *dptr)
{
	return mdb_cached_free(ctx, dptr);
}

/* out[i] = table[cells[i]] for 0 <= cells[i] < table_n, else 0 (dictionary ids of one rank <-> the ranks' common ids: mdb_exec_shard.c) */
__global__ __launch_bounds__(256) void k_map_ids(const long long *__restrict__ cells, unsigned long long n, const long long *__restrict__ table,
						   unsigned long long table_n, long long *__restrict__ out)
{
	for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
		const long long c = cells[i];
		out[i] = (c >= 0 && (unsigned long long)c < table_n) ? table[c] : 0;
	}
}

extern "C" int mdb_dev_map_ids(mdb_dev_ctx *ctx, const int64_t *cells, uint64_t n, const int64_t *table, uint64_t table_n, int64_t *out)
{
	if (!ctx || (n && (!cells || !out || !table)))
		return -MIDORIDB_ERROR;
	if (!n)
		return MIDORIDB_OK;
	const uint64_t b = (n + 255) / 256;
	MDB_LAUNCH(ctx, "map_ids", k_map_ids, (uint32_t)(b > 16384 ? 16384 : b), 256, reinterpret_cast<const long long *>(cells), (unsigned long long)n,
		   reinterpret_cast<const long long *>(table), (unsigned long long)table_n, reinterpret_cast<long long *>(out));
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_retain(mdb_dev_ctx *ctx, const void *dptr)
{
	if (!ctx || !dptr || ctx->live.find(const_cast<void *>(dptr)) == ctx->live.end())
		return -MIDORIDB_ERROR;
	ctx->holders[const_cast<void *>(dptr)]++;
	return MIDORIDB_OK;
}

extern "C" unsigned mdb_dev_holders(mdb_dev_ctx *ctx, const void *dptr)
{
	if (!ctx || !dptr)
		return 0;
	auto it = ctx->holders.find(const_cast<void *>(dptr));
	return it == ctx->holders.end() ? 0u : it->second;
}

extern "C" size_t mdb_dev_alloc_size(mdb_dev_ctx *ctx, const void *dptr)
{
	if (!ctx || !dptr)
		return 0;
	auto it = ctx->live.find(const_cast<void *>(dptr));
	return it == ctx->live.end() ? 0 : it->second;
}

extern "C" int mdb_dev_memset(mdb_dev_ctx *ctx, void *dptr, int byte, size_t bytes)
{
	if (bytes)
		MDB_HIP(ctx, hipMemsetAsync(dptr, byte, bytes, ctx->stream));
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_h2d(mdb_dev_ctx *ctx, void *dptr, const void *host, size_t bytes)
{
	if (bytes) {
		mdb_hints_drop(ctx, dptr, bytes);
		MDB_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_d2h(mdb_dev_ctx *ctx, void *host, const void *dptr, size_t bytes)
{
	if (bytes) {
		MDB_HIP(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ pinned host pool (result columns) */

#include <mutex>
#include <vector>

#define HOST_POOL_MIN (1u << 20)		/* below 1 MiB pinning costs more than it saves */
#define HOST_POOL_KEEP ((size_t)8 << 30)	/* idle pinned bytes kept for reuse */

struct host_buf {
	void *p;
	size_t bytes;
	bool pinned, in_use;
};
static std::mutex g_host_mu;
static std::vector<host_buf> g_host_pool;

extern "C" void *mdb_dev_host_alloc(size_t bytes)
{
	if (bytes == 0)
		bytes = 8;
	std::lock_guard<std::mutex> lk(g_host_mu);
	if (bytes >= HOST_POOL_MIN) {
		/* best fit among the idle pinned buffers (at most 2x the request) */
		int best = -1;
		for (size_t i = 0; i < g_host_pool.size(); i++) {
			const host_buf &b = g_host_pool[i];
			if (b.pinned && !b.in_use && b.bytes >= bytes && b.bytes <= 2 * bytes && (best < 0 || b.bytes < g_host_pool[best].bytes))
				best = (int)i;
		}
		if (best >= 0) {
			g_host_pool[best].in_use = true;
			return g_host_pool[best].p;
		}
		void *p = NULL;
		if (hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess && p) {
			g_host_pool.push_back({ p, bytes, true, true });
			return p;
		}
		(void)hipGetLastError();
	}
	void *p = malloc(bytes);
	if (p)
		g_host_pool.push_back({ p, bytes, false, true });
	return p;
}

extern "C" void mdb_dev_host_free(void *p)
{
	if (!p)
		return;
	std::lock_guard<std::mutex> lk(g_host_mu);
	size_t idle = 0;
	for (const host_buf &b : g_host_pool)
		if (b.pinned && !b.in_use)
			idle += b.bytes;
	for (size_t i = 0; i < g_host_pool.size(); i++) {
		host_buf &b = g_host_pool[i];
		if (b.p != p)
			continue;
		if (b.pinned && idle + b.bytes <= HOST_POOL_KEEP) {
			b.in_use = false;	/* stays pinned for the next result */
			return;
		}
		if (b.pinned)
			(void)hipHostFree(b.p);
		else
			free(b.p);
		g_host_pool.erase(g_host_pool.begin() + (long)i);
		return;
	}
	free(p);	/* not ours: plain malloc memory */
}

/* ------------------------------------------------------------------ profiling */

void mdb_prof_begin(mdb_dev_ctx *ctx, const char *name, const void *kernel)
{
	if (!ctx->prof_on)
		return;
	int id = -1;
	for (size_t i = 0; i < ctx->prof_names.size(); i++)
		if (ctx->prof_names[i] == name) {
			id = (int)i;
			break;
		}
	if (id < 0) {
		ctx->prof_names.push_back(name);
		ctx->prof_kernels.push_back(std::vector<const void *>());
		id = (int)ctx->prof_names.size() - 1;
	}
	{
		auto &ks = ctx->prof_kernels[(size_t)id];
		bool seen = false;
		for (const void *k : ks)
			seen = seen || k == kernel;
		if (!seen && kernel)
			ks.push_back(kernel);
	}
	if (ctx->prof_pool_used == ctx->prof_pool.size()) {
		hipEvent_t a, b;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
			return;
		ctx->prof_pool.push_back(std::make_pair(a, b));
	}
	auto &ev = ctx->prof_pool[ctx->prof_pool_used++];
	mdb_prof_rec r;
	r.name_id = id;
	r.start = ev.first;
	r.stop = ev.second;
	ctx->prof_recs.push_back(r);
	(void)hipEventRecord(r.start, ctx->stream);
}

void mdb_prof_end(mdb_dev_ctx *ctx)
{
	if (!ctx->prof_on || ctx->prof_recs.empty())
		return;
	(void)hipEventRecord(ctx->prof_recs.back().stop, ctx->stream);
}

extern "C" int mdb_dev_prof_enable(mdb_dev_ctx *ctx, int on)
{
	ctx->prof_on = on != 0;
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_prof_reset(mdb_dev_ctx *ctx)
{
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ctx->prof_recs.clear();
	ctx->prof_pool_used = 0;
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_prof_read(mdb_dev_ctx *ctx, struct mdb_dev_prof_entry *out, int cap, int *n_out)
{
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	int n = (int)ctx->prof_names.size();
	if (n > cap)
		n = cap;
	for (int i = 0; i < n; i++) {
		memset(&out[i], 0, sizeof(out[i]));
		snprintf(out[i].name, sizeof(out[i].name), "%s", ctx->prof_names[i].c_str());
	}
	for (auto &r : ctx->prof_recs) {
		if (r.name_id >= n)
			continue;
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
			out[r.name_id].launches++;
			out[r.name_id].total_ms += ms;
		}
	}
	*n_out = n;
	return MIDORIDB_OK;
}

/* The symbols of the kernels launched under a profiler name since profiling was enabled (the template instances a profiler
 * such as rocprofv3 lists them under), separated by newlines: the one table that ties this library's per-kernel timings to
 * the rows of an external profile - nothing outside has to spell a mangled name. */
extern "C" int mdb_dev_prof_symbols(mdb_dev_ctx *ctx, const char *name, char *out, size_t cap)
{
	if (!out || !cap || !name)
		return -MIDORIDB_ERROR;
	out[0] = 0;
	size_t used = 0;
	for (size_t i = 0; i < ctx->prof_names.size(); i++) {
		if (ctx->prof_names[i] != name)
			continue;
		for (const void *k : ctx->prof_kernels[i]) {
			const char *mangled = hipKernelNameRefByPtr(k, ctx->stream);
			if (!mangled)
				continue;
			/* demangled, without "void " and the argument list: the form rocprofv3's kernel trace shows */
			int st = 0;
			char *dm = abi::__cxa_demangle(mangled, NULL, NULL, &st);
			std::string nm = (st == 0 && dm) ? dm : mangled;
			free(dm);
			if (nm.compare(0, 5, "void ") == 0)
				nm.erase(0, 5);
			{
				int depth = 0;
				for (size_t c = 0; c < nm.size(); c++) {
					if (nm[c] == '<')
						depth++;
					else if (nm[c] == '>')
						depth--;
					else if (nm[c] == '(' && depth == 0) {
						nm.erase(c);
						break;
					}
				}
			}
			const char *sym = nm.c_str();
			const size_t len = nm.size();
			if (used + len + 2 > cap)
				return -MIDORIDB_ERROR;
			if (used)
				out[used++] = '\n';
			memcpy(out + used, sym, len + 1);
			used += len;
		}
	}
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ exclusive scan (uint32)
 *
 * reduce -> spine -> apply.  Each block owns MDB_SCAN_CHUNK consecutive words (16 per thread).
 * HBM traffic: 2 reads + 1 write of the array; it is only ever run over histogram-sized arrays
 * (a few % of the key bytes).
 */
#define SCAN_THREADS 256
#define SCAN_PER_THREAD (MDB_SCAN_CHUNK / SCAN_THREADS)

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t *__restrict__ data, uint64_t len,
							       uint32_t *__restrict__ block_sums)
{
	__shared__ uint32_t tmp[32];
	const uint64_t base = (uint64_t)blockIdx.x * MDB_SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
	uint32_t s = 0;
	if (base + SCAN_PER_THREAD <= len) {
		const uint4 *p = reinterpret_cast<const uint4 *>(data + base);
#pragma unroll
		for (int i = 0; i < SCAN_PER_THREAD / 4; i++) {
			uint4 v = p[i];
			s += v.x + v.y + v.z + v.w;
		}
	} else {
		for (int i = 0; i < SCAN_PER_THREAD; i++)
			if (base + i < len)
				s += data[base + i];
	}
	uint32_t total;
	(void)mdb_block_excl_scan(s, tmp, &total);
	if (threadIdx.x == 0)
		block_sums[blockIdx.x] = total;
}

/* single block: exclusive scan of block_sums[0..nb) in place, carry across 1024-wide sweeps */
__global__ __launch_bounds__(1024) void k_scan_spine(uint32_t *__restrict__ block_sums, uint32_t nb, bool write_total)
{
	__shared__ uint32_t tmp[32];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nb; base += 1024) {
		uint32_t i = base + threadIdx.x;
		uint32_t v = i < nb ? block_sums[i] : 0;
		uint32_t total;
		uint32_t ex = mdb_block_excl_scan(v, tmp, &total);
		if (i < nb)
			block_sums[i] = carry + ex;
		carry += total;
	}
	if (write_total && threadIdx.x == 0)
		block_sums[nb] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t *__restrict__ data, uint64_t len,
							      const uint32_t *__restrict__ block_sums)
{
	__shared__ uint32_t tmp[32];
	const uint64_t base = (uint64_t)blockIdx.x * MDB_SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
	uint32_t v[SCAN_PER_THREAD];
	uint32_t s = 0;
	const bool full = base + SCAN_PER_THREAD <= len;
	if (full) {
		const uint4 *p = reinterpret_cast<const uint4 *>(data + base);
#pragma unroll
		for (int i = 0; i < SCAN_PER_THREAD / 4; i++) {
			uint4 q = p[i];
			v[4 * i + 0] = q.x;
			v[4 * i + 1] = q.y;
			v[4 * i + 2] = q.z;
			v[4 * i + 3] = q.w;
		}
	} else {
#pragma unroll
		for (int i = 0; i < SCAN_PER_THREAD; i++)
			v[i] = base + i < len ? data[base + i] : 0;
	}
#pragma unroll
	for (int i = 0; i < SCAN_PER_THREAD; i++)
		s += v[i];
	uint32_t total;
	uint32_t run = mdb_block_excl_scan(s, tmp, &total) + block_sums[blockIdx.x];
#pragma unroll
	for (int i = 0; i < SCAN_PER_THREAD; i++) {
		uint32_t t = v[i];
		v[i] = run;
		run += t;
	}
	if (full) {
		uint4 *p = reinterpret_cast<uint4 *>(data + base);
#pragma unroll
		for (int i = 0; i < SCAN_PER_THREAD / 4; i++)
			p[i] = make_uint4(v[4 * i + 0], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
	} else {
#pragma unroll
		for (int i = 0; i < SCAN_PER_THREAD; i++)
			if (base + i < len)
				data[base + i] = v[i];
	}
}

/* single block: dst[0..n] = exclusive prefix sums of src[0..n) (dst[n] = the total).  Every wave owns one contiguous part of the n + 1
 * elements and walks it 64 x 8 elements at a time (coalesced, eight loads in flight): summed, the sums scanned across the block, walked
 * again from the cache - up to MDB_SCAN_FROM_MAX elements in less time than the three launches of the general scan take to start */
__global__ __launch_bounds__(1024) void k_scan_excl_from(const uint32_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ dst)
{
	__shared__ uint32_t tmp[16];
	const uint32_t lane = mdb_lane(), wave = threadIdx.x >> 6, N = n + 1u;
	const uint32_t seg = (((N + 15u) / 16u) + 63u) & ~63u, b = wave * seg, e = (b + seg < N) ? b + seg : N;
	uint32_t sum = 0;
	for (uint32_t i0 = b; i0 < e; i0 += 512u) {
		uint32_t v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t i = i0 + 64u * (uint32_t)k + lane;
			v[k] = (i < e && i < n) ? src[i] : 0u;
		}
#pragma unroll
		for (int k = 0; k < 8; k++)
			sum += v[k];
	}
#pragma unroll
	for (int o = 32; o; o >>= 1)
		sum += (uint32_t)__shfl_xor((int)sum, o, MDB_WAVE);
	if (lane == 0)
		tmp[wave] = sum;
	__syncthreads();
	uint32_t carry = (lane < wave) ? tmp[lane] : 0u;
#pragma unroll
	for (int o = 32; o; o >>= 1)
		carry += (uint32_t)__shfl_xor((int)carry, o, MDB_WAVE);
	for (uint32_t i0 = b; i0 < e; i0 += 512u) {
		uint32_t v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t i = i0 + 64u * (uint32_t)k + lane;
			v[k] = (i < e && i < n) ? src[i] : 0u;
		}
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t i = i0 + 64u * (uint32_t)k + lane;
			const uint32_t incl = mdb_wave_incl_scan(v[k]);
			if (i < e)
				dst[i] = carry + incl - v[k];
			carry += (uint32_t)__shfl((int)incl, 63, MDB_WAVE);
		}
	}
}

int mdb_scan_u32_small_from(mdb_dev_ctx *ctx, const uint32_t *src, uint32_t n, uint32_t *dst)
{
	MDB_LAUNCH(ctx, "scan_small", k_scan_excl_from, 1, 1024, src, n, dst);
	return MIDORIDB_OK;
}

int mdb_scan_u32_inplace(mdb_dev_ctx *ctx, uint32_t *data, uint64_t len, uint32_t *block_sums)
{
	if (len == 0)
		return MIDORIDB_OK;
	if (len <= MDB_SCAN_SMALL) {
		/* short arrays (leaf / region tables): one single-workgroup launch instead of three */
		MDB_LAUNCH(ctx, "scan_small", k_scan_spine, 1, 1024, data, (uint32_t)len, false);
		return MIDORIDB_OK;
	}
	uint32_t nb = (uint32_t)((len + MDB_SCAN_CHUNK - 1) / MDB_SCAN_CHUNK);
	MDB_LAUNCH(ctx, "scan_reduce", k_scan_reduce, nb, SCAN_THREADS, data, len, block_sums);
	MDB_LAUNCH(ctx, "scan_spine", k_scan_spine, 1, 1024, block_sums, nb, true);
	MDB_LAUNCH(ctx, "scan_apply", k_scan_apply, nb, SCAN_THREADS, data, len, block_sums);
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ gather / iota / cross pairs */

#define STREAM_THREADS 256
#define STREAM_ROUNDS 8		/* each block handles 256*8 = 2048 consecutive outputs */

__global__ __launch_bounds__(STREAM_THREADS) void k_gather64(const uint64_t *__restrict__ src,
							     const uint64_t *__restrict__ src_null,
							     const uint32_t *__restrict__ idx, uint64_t n,
							     uint64_t *__restrict__ dst, uint64_t *__restrict__ dst_null)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
	if (idx && base + (uint64_t)STREAM_THREADS * STREAM_ROUNDS <= n) {	/* (uniform) a full block: row ids, then gathers, issued together */
		uint32_t row[STREAM_ROUNDS];
		uint64_t v[STREAM_ROUNDS];
#pragma unroll
		for (int r = 0; r < STREAM_ROUNDS; r++)
			row[r] = idx[base + (uint64_t)r * STREAM_THREADS + threadIdx.x];
#pragma unroll
		for (int r = 0; r < STREAM_ROUNDS; r++)
			v[r] = src[row[r]];
#pragma unroll
		for (int r = 0; r < STREAM_ROUNDS; r++)
			dst[base + (uint64_t)r * STREAM_THREADS + threadIdx.x] = v[r];
		if (dst_null) {
#pragma unroll
			for (int r = 0; r < STREAM_ROUNDS; r++) {
				const uint64_t m = __ballot(src_null && mdb_bit_is_set(src_null, row[r]));
				if (mdb_lane() == 0)
					dst_null[(base + (uint64_t)r * STREAM_THREADS + threadIdx.x) >> 6] = m;
			}
		}
		return;
	}
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		/* one wave covers 64 consecutive outputs per round => one NULL word per wave per round */
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		bool in = k < n;
		uint64_t row = 0, v = 0;
		bool isnull = false;
		if (in) {
			row = idx ? (uint64_t)idx[k] : k;
			v = src[row];
			if (src_null)
				isnull = mdb_bit_is_set(src_null, row);
			dst[k] = v;
		}
		if (dst_null) {
			uint64_t m = __ballot(in && isnull);
			if (mdb_lane() == 0 && (k < n))
				dst_null[k >> 6] = m;
		}
	}
}

__global__ __launch_bounds__(STREAM_THREADS) void k_gather32(const uint32_t *__restrict__ src,
							     const uint32_t *__restrict__ idx, uint64_t n,
							     uint32_t *__restrict__ dst)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
	if (base + (uint64_t)STREAM_THREADS * STREAM_ROUNDS <= n) {	/* (uniform) a full block: row ids, then gathers, issued together */
		uint32_t row[STREAM_ROUNDS], v[STREAM_ROUNDS];
#pragma unroll
		for (int r = 0; r < STREAM_ROUNDS; r++)
			row[r] = idx[base + (uint64_t)r * STREAM_THREADS + threadIdx.x];
#pragma unroll
		for (int r = 0; r < STREAM_ROUNDS; r++)
			v[r] = src[row[r]];
#pragma unroll
		for (int r = 0; r < STREAM_ROUNDS; r++)
			dst[base + (uint64_t)r * STREAM_THREADS + threadIdx.x] = v[r];
		return;
	}
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < n)
			dst[k] = src[idx[k]];
	}
}

__global__ __launch_bounds__(STREAM_THREADS) void k_iota32(uint32_t *__restrict__ dst, uint64_t n)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < n)
			dst[k] = (uint32_t)k;
	}
}

__global__ __launch_bounds__(STREAM_THREADS) void k_cross_pairs(uint64_t n_l, uint64_t n_r, uint32_t *__restrict__ out_l,
								uint32_t *__restrict__ out_r)
{
	const uint64_t total = n_l * n_r;
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < total) {
			out_l[k] = (uint32_t)(k / n_r);
			out_r[k] = (uint32_t)(k % n_r);
		}
	}
}

static inline uint32_t stream_grid(uint64_t n)
{
	return (uint32_t)((n + (uint64_t)STREAM_THREADS * STREAM_ROUNDS - 1) / ((uint64_t)STREAM_THREADS * STREAM_ROUNDS));
}

extern "C" int mdb_dev_gather64(mdb_dev_ctx *ctx, const void *src, const uint64_t *src_nullbits, const uint32_t *idx,
				uint64_t n, void *dst, uint64_t *dst_nullbits)
{
	if (n == 0)
		return MIDORIDB_OK;
	if (src_nullbits && !dst_nullbits)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "gather64: source has NULL bits but no destination NULL bits given");
	MDB_LAUNCH(ctx, "gather64", k_gather64, stream_grid(n), STREAM_THREADS, (const uint64_t *)src, src_nullbits, idx, n,
		   (uint64_t *)dst, dst_nullbits);
	return MIDORIDB_OK;
}

/* whole-result projection: see mdb_dev_gather_cols() in mdb_dev.h */
struct gather_cols_args {
	const uint64_t *src[MDB_GATHER_MAX_COLS];
	const uint64_t *src_null[MDB_GATHER_MAX_COLS];
	uint64_t *dst[MDB_GATHER_MAX_COLS];
	uint64_t *dst_null[MDB_GATHER_MAX_COLS];
	const uint32_t *rid[MDB_GATHER_MAX_RIDS];
	uint8_t slot[MDB_GATHER_MAX_COLS];	/* which rid[] a column reads through; 0xFF = identity */
	int ncols, nrids;
	uint64_t n;
};

#define GC_ROUNDS 4	/* 256 x 4 = 1024 consecutive outputs per block: up to 4 x 16 gathers in flight per thread */

/* NR = row-id vectors loaded per output row (the host rounds a.nrids up to 1, 2, 4 or 8 and lets the unused slots alias
 * rid[0]).  A block whose outputs all exist loads its row ids, and then every column's GC_ROUNDS gathers, TOGETHER: behind
 * "k < n" every load sits in a branch of its own, the compiler waits for it before the next one is issued, and a thread has
 * one gather in flight. */
template <int NR>
__global__ __launch_bounds__(STREAM_THREADS) void k_gather_cols(gather_cols_args a)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * GC_ROUNDS);
	if (NR > 0 && base + (uint64_t)STREAM_THREADS * GC_ROUNDS <= a.n) {	/* (uniform) */
		uint32_t row[GC_ROUNDS][NR > 0 ? NR : 1];
#pragma unroll
		for (int r = 0; r < GC_ROUNDS; r++)
#pragma unroll
			for (int t = 0; t < NR; t++)
				row[r][t] = a.rid[t][base + (uint64_t)r * STREAM_THREADS + threadIdx.x];
		for (int c = 0; c < a.ncols; c++) {
			const uint32_t slot = a.slot[c];
			uint64_t v[GC_ROUNDS], src_row[GC_ROUNDS];
#pragma unroll
			for (int r = 0; r < GC_ROUNDS; r++) {
				src_row[r] = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
#pragma unroll
				for (int t = 0; t < NR; t++)
					if (slot == (uint32_t)t)
						src_row[r] = row[r][t];
				v[r] = a.src[c][src_row[r]];
			}
#pragma unroll
			for (int r = 0; r < GC_ROUNDS; r++)
				a.dst[c][base + (uint64_t)r * STREAM_THREADS + threadIdx.x] = v[r];
			if (a.dst_null[c]) {
#pragma unroll
				for (int r = 0; r < GC_ROUNDS; r++) {
					const bool isnull = a.src_null[c] && mdb_bit_is_set(a.src_null[c], src_row[r]);
					const uint64_t m = __ballot(isnull);
					if (mdb_lane() == 0)
						a.dst_null[c][(base + (uint64_t)r * STREAM_THREADS + threadIdx.x) >> 6] = m;
				}
			}
		}
		return;
	}
#pragma unroll
	for (int r = 0; r < GC_ROUNDS; r++) {
		/* one wave covers 64 consecutive outputs per round => one NULL word per wave, round and column */
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		const bool in = k < a.n;
		uint32_t row[MDB_GATHER_MAX_RIDS];
#pragma unroll
		for (int t = 0; t < MDB_GATHER_MAX_RIDS; t++)
			row[t] = (in && t < a.nrids) ? a.rid[t][k] : 0u;
		for (int c = 0; c < a.ncols; c++) {
			uint64_t src_row = k;
			if (a.slot[c] != 0xFF) {
#pragma unroll
				for (int t = 0; t < MDB_GATHER_MAX_RIDS; t++)	/* (a compile-time index keeps row[] in registers) */
					if (a.slot[c] == t)
						src_row = row[t];
			}
			bool isnull = false;
			if (in) {
				a.dst[c][k] = a.src[c][src_row];
				if (a.src_null[c])
					isnull = mdb_bit_is_set(a.src_null[c], src_row);
			}
			if (a.dst_null[c]) {
				const uint64_t m = __ballot(in && isnull);
				if (mdb_lane() == 0 && in)
					a.dst_null[c][k >> 6] = m;
			}
		}
	}
}

extern "C" int mdb_dev_gather_cols(mdb_dev_ctx *ctx, const struct mdb_gather_col *cols, int ncols, uint64_t n)
{
	if (n == 0 || ncols == 0)
		return MIDORIDB_OK;
	if (!cols || ncols < 0 || ncols > MDB_GATHER_MAX_COLS)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "gather_cols: between 1 and %d columns per call", MDB_GATHER_MAX_COLS);
	gather_cols_args a;
	memset(&a, 0, sizeof(a));
	a.ncols = ncols;
	a.n = n;
	for (int c = 0; c < ncols; c++) {
		if (!cols[c].src || !cols[c].dst || (cols[c].src_nullbits && !cols[c].dst_nullbits))
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "gather_cols: column %d lacks a source, a destination or destination NULL bits", c);
		a.src[c] = (const uint64_t *)cols[c].src;
		a.src_null[c] = cols[c].src_nullbits;
		a.dst[c] = (uint64_t *)cols[c].dst;
		a.dst_null[c] = cols[c].dst_nullbits;
		a.slot[c] = 0xFF;
		if (cols[c].rid) {
			int t = 0;
			while (t < a.nrids && a.rid[t] != cols[c].rid)
				t++;
			if (t == a.nrids) {
				if (a.nrids == MDB_GATHER_MAX_RIDS)
					return mdb_set_err(ctx, -MIDORIDB_ERROR, "gather_cols: more than %d row-id vectors", MDB_GATHER_MAX_RIDS);
				a.rid[a.nrids++] = cols[c].rid;
			}
			a.slot[c] = (uint8_t)t;
		}
	}
	const uint32_t grid = (uint32_t)((n + (uint64_t)STREAM_THREADS * GC_ROUNDS - 1) / ((uint64_t)STREAM_THREADS * GC_ROUNDS));
	for (int t = a.nrids; t < MDB_GATHER_MAX_RIDS; t++)
		a.rid[t] = a.rid[0];	/* (the kernel instance may load more slots than are used) */
	if (a.nrids == 0) {
		MDB_LAUNCH(ctx, "gather_cols", k_gather_cols<0>, grid, STREAM_THREADS, a);
	} else if (a.nrids == 1) {
		MDB_LAUNCH(ctx, "gather_cols", k_gather_cols<1>, grid, STREAM_THREADS, a);
	} else if (a.nrids == 2) {
		MDB_LAUNCH(ctx, "gather_cols", k_gather_cols<2>, grid, STREAM_THREADS, a);
	} else if (a.nrids <= 4) {
		MDB_LAUNCH(ctx, "gather_cols", k_gather_cols<4>, grid, STREAM_THREADS, a);
	} else {
		MDB_LAUNCH(ctx, "gather_cols", k_gather_cols<8>, grid, STREAM_THREADS, a);
	}
	return MIDORIDB_OK;
}

/* DOUBLE equi-join keys: the join operators compare 8-byte words, the reference compares IEEE doubles
 * (cmp_double_value_to_value, reference src/engine/executor_select.c:440-460: `==`), which differ in two places -
 * -0.0 == +0.0 (different words) and NaN != NaN (equal words).  This pass makes the word comparison exact: -0.0 is
 * rewritten to +0.0 and a NaN row gets its NULL bit set (a NULL key never matches either, :557-579). */
__global__ __launch_bounds__(STREAM_THREADS) void k_double_join_keys(const uint64_t *__restrict__ src, const uint64_t *__restrict__ src_null,
								     const uint32_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ dst,
								     uint64_t *__restrict__ dst_null)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		const bool in = k < n;
		bool drop = false;
		if (in) {
			const uint64_t row = idx ? (uint64_t)idx[k] : k;
			uint64_t v = src[row];
			if (src_null)
				drop = mdb_bit_is_set(src_null, row);
			if ((v << 1) == 0)
				v = 0;					/* -0.0 -> +0.0 */
			if ((v << 1) > 0xFFE0000000000000ull)
				drop = true;				/* NaN: exponent all ones, mantissa non-zero */
			dst[k] = v;
		}
		const uint64_t m = __ballot(in && drop);
		if (mdb_lane() == 0 && in)
			dst_null[k >> 6] = m;
	}
}

extern "C" int mdb_dev_double_join_keys(mdb_dev_ctx *ctx, const double *src, const uint64_t *src_nullbits, const uint32_t *idx, uint64_t n,
					int64_t *dst, uint64_t *dst_nullbits)
{
	if (n == 0)
		return MIDORIDB_OK;
	if (!dst || !dst_nullbits)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "double_join_keys: destination values and NULL bits are both required");
	MDB_LAUNCH(ctx, "double_join_keys", k_double_join_keys, stream_grid(n), STREAM_THREADS, (const uint64_t *)src, src_nullbits, idx, n,
		   (uint64_t *)dst, dst_nullbits);
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_gather32(mdb_dev_ctx *ctx, const uint32_t *src, const uint32_t *idx, uint64_t n, uint32_t *dst)
{
	if (n == 0)
		return MIDORIDB_OK;
	MDB_LAUNCH(ctx, "gather32", k_gather32, stream_grid(n), STREAM_THREADS, src, idx, n, dst);
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_iota32(mdb_dev_ctx *ctx, uint32_t *dst, uint64_t n)
{
	if (n == 0)
		return MIDORIDB_OK;
	MDB_LAUNCH(ctx, "iota32", k_iota32, stream_grid(n), STREAM_THREADS, dst, n);
	return MIDORIDB_OK;
}

/* UPDATE: one thread per selected row; NULL bits through word atomics (rows of one word may be spread
 * over several threads) */
__global__ __launch_bounds__(STREAM_THREADS) void k_scatter_set64(uint64_t *__restrict__ dst, unsigned long long *__restrict__ nullbits,
								   const uint32_t *__restrict__ idx, uint64_t n, uint64_t value, int set_null)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k >= n)
			continue;
		const uint64_t row = idx ? (uint64_t)idx[k] : k;
		if (!set_null)
			dst[row] = value;
		if (nullbits) {
			const unsigned long long bit = 1ull << (row & 63);
			if (set_null)
				atomicOr(&nullbits[row >> 6], bit);
			else
				atomicAnd(&nullbits[row >> 6], ~bit);
		}
	}
}

extern "C" int mdb_dev_scatter_set64(mdb_dev_ctx *ctx, void *dst, uint64_t *dst_nullbits, const uint32_t *idx, uint64_t n,
				     int64_t value_bits, int set_null)
{
	if (n == 0)
		return MIDORIDB_OK;
	if (set_null && !dst_nullbits)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "scatter_set64: SET NULL needs a NULL bitmap");
	mdb_hints_drop(ctx, dst, 8);	/* (hints are keyed by a column's first byte) */
	MDB_LAUNCH(ctx, "scatter_set64", k_scatter_set64, stream_grid(n), STREAM_THREADS, (uint64_t *)dst,
		   (unsigned long long *)dst_nullbits, idx, n, (uint64_t)value_bits, set_null);
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ chained fused joins: combine two COUNT columns */

__global__ __launch_bounds__(STREAM_THREADS) void k_combine_counts(const int64_t *__restrict__ cnt1, const uint32_t *__restrict__ first1,
								    const uint32_t *__restrict__ idx, const int64_t *__restrict__ cnt2, uint64_t n,
								    int64_t *__restrict__ out_cnt, uint32_t *__restrict__ out_first,
								    unsigned long long *sum)
{
	__shared__ unsigned long long s_sum;
	if (threadIdx.x == 0)
		s_sum = 0ull;
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
	unsigned long long mine = 0;
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < n) {
			const uint32_t g = idx[k];
			const int64_t c = cnt1[g] * cnt2[k];
			out_cnt[k] = c;
			if (out_first)
				out_first[k] = first1 ? first1[g] : g;
			mine += (unsigned long long)c;
		}
	}
	if (mine)
		atomicAdd(&s_sum, mine);
	__syncthreads();
	if (threadIdx.x == 0 && s_sum)
		atomicAdd(sum, s_sum);
}

extern "C" int mdb_dev_combine_counts(mdb_dev_ctx *ctx, const int64_t *cnt1, const uint32_t *first1, const uint32_t *idx, const int64_t *cnt2,
				      uint64_t n, int64_t *out_cnt, uint32_t *out_first, uint64_t *out_sum)
{
	*out_sum = 0;
	if (n == 0)
		return MIDORIDB_OK;
	unsigned long long *d_sum = (unsigned long long *)(ctx->d_status + 12);
	MDB_HIP(ctx, hipMemsetAsync(d_sum, 0, 8, ctx->stream));
	MDB_LAUNCH(ctx, "combine_counts", k_combine_counts, stream_grid(n), STREAM_THREADS, cnt1, first1, idx, cnt2, n, out_cnt, out_first, d_sum);
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h, d_sum, 8, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_sum = h[0];
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ key statistics / 4-byte wire format */

__global__ __launch_bounds__(STREAM_THREADS) void k_key_range(const int64_t *__restrict__ keys, const uint64_t *__restrict__ nullbits, uint64_t n,
							       long long *mm)
{
	__shared__ long long s_lo, s_hi;
	if (threadIdx.x == 0) {
		s_lo = 0x7FFFFFFFFFFFFFFFll;
		s_hi = -0x7FFFFFFFFFFFFFFFll - 1;
	}
	__syncthreads();
	long long lo = 0x7FFFFFFFFFFFFFFFll, hi = -0x7FFFFFFFFFFFFFFFll - 1;
	for (uint64_t k = (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * STREAM_THREADS) {
		if (nullbits && mdb_bit_is_set(nullbits, k))
			continue;
		const long long v = keys[k];
		lo = v < lo ? v : lo;
		hi = v > hi ? v : hi;
	}
	if (lo <= hi) {
		atomicMin(&s_lo, lo);
		atomicMax(&s_hi, hi);
	}
	__syncthreads();
	if (threadIdx.x == 0 && s_lo <= s_hi) {
		atomicMin(&mm[0], s_lo);
		atomicMax(&mm[1], s_hi);
	}
}

extern "C" int mdb_dev_key_range(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t *out_min,
				 int64_t *out_max)
{
	*out_min = 0x7FFFFFFFFFFFFFFFll;
	*out_max = -0x7FFFFFFFFFFFFFFFll - 1;
	if (n == 0)
		return MIDORIDB_OK;
	long long *mm = (long long *)(ctx->d_status + 12);	/* two 8-byte words of the status block */
	long long *h = (long long *)ctx->h_pinned;
	h[0] = *out_min;
	h[1] = *out_max;
	MDB_HIP(ctx, hipMemcpyAsync(mm, h, 16, hipMemcpyHostToDevice, ctx->stream));
	const uint64_t blocks = (n + STREAM_THREADS * STREAM_ROUNDS - 1) / (STREAM_THREADS * STREAM_ROUNDS);
	MDB_LAUNCH(ctx, "key_range", k_key_range, (uint32_t)(blocks < 4096 ? blocks : 4096), STREAM_THREADS, keys, nullbits, n, mm);
	MDB_HIP(ctx, hipMemcpyAsync(h, mm, 16, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_min = h[0];
	*out_max = h[1];
	return MIDORIDB_OK;
}

/* "no key twice" as a measured statistic (include/mdb_dev.h: MDB_COL_DISTINCT): one bit per key value of the column's window */
__global__ __launch_bounds__(STREAM_THREADS) void k_distinct_scan(const int64_t *__restrict__ keys, const uint64_t *__restrict__ nullbits, uint64_t n,
								  int64_t lo, uint64_t bits, uint32_t *seen, uint32_t *flag)
{
	bool twice = false;
	for (uint64_t i = (uint64_t)blockIdx.x * STREAM_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * STREAM_THREADS) {
		if (nullbits && mdb_bit_is_set(nullbits, i))
			continue;
		const uint64_t b = (uint64_t)keys[i] - (uint64_t)lo;
		if (b >= bits) {
			twice = true;	/* (outside the window: nothing can be said) */
			continue;
		}
		const uint32_t m = 1u << (b & 31u);
		twice |= (atomicOr(&seen[b >> 5], m) & m) != 0u;
	}
	if (__ballot(twice) && mdb_lane() == 0)
		mdb_raise(flag, 1u);
}

extern "C" int mdb_dev_distinct_scan(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t window_lo, uint64_t window_bits,
				     uint32_t *seen, int *out_twice)
{
	*out_twice = 0;
	if (n == 0)
		return MIDORIDB_OK;
	uint32_t *flag = ctx->d_status + 12;
	MDB_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
	const uint64_t blocks = (n + STREAM_THREADS * STREAM_ROUNDS - 1) / (STREAM_THREADS * STREAM_ROUNDS);
	MDB_LAUNCH(ctx, "distinct_scan", k_distinct_scan, (uint32_t)(blocks < 8192 ? blocks : 8192), STREAM_THREADS, keys, nullbits, n, window_lo, window_bits,
		   seen, flag);
	uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
	MDB_HIP(ctx, hipMemcpyAsync(h, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_twice = h[0] ? 1 : 0;
	return MIDORIDB_OK;
}

__global__ __launch_bounds__(STREAM_THREADS) void k_widen32(const int32_t *__restrict__ src, uint64_t n, int64_t *__restrict__ dst)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < n)
			dst[k] = (int64_t)src[k];
	}
}

extern "C" int mdb_dev_widen32to64(mdb_dev_ctx *ctx, const int32_t *src, uint64_t n, int64_t *dst)
{
	if (n == 0)
		return MIDORIDB_OK;
	MDB_LAUNCH(ctx, "widen32", k_widen32, stream_grid(n), STREAM_THREADS, src, n, dst);
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_cross_pairs(mdb_dev_ctx *ctx, uint64_t n_l, uint64_t n_r, uint32_t *out_l, uint32_t *out_r)
{
	if (n_l == 0 || n_r == 0)
		return MIDORIDB_OK;
	if (n_l * n_r >= (1ull << 32))
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "cross join of %llu x %llu rows is too large",
				   (unsigned long long)n_l, (unsigned long long)n_r);
	MDB_LAUNCH(ctx, "cross_pairs", k_cross_pairs, stream_grid(n_l * n_r), STREAM_THREADS, n_l, n_r, out_l, out_r);
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ synthetic keys (include/mdb_gen.h) */

__global__ __launch_bounds__(STREAM_THREADS) void k_gen_keys(int64_t *__restrict__ keys, uint64_t n, uint64_t first,
							      mdb_perm perm, uint64_t modulus)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < n) {
			uint64_t v = mdb_perm_apply(&perm, first + k);
			if (modulus)
				v %= modulus;
			keys[k] = (int64_t)v;
		}
	}
}

extern "C" int mdb_dev_gen_keys(mdb_dev_ctx *ctx, int64_t *keys, uint64_t n, uint64_t first_index, uint64_t domain,
				uint64_t seed, uint64_t modulus)
{
	if (n == 0)
		return MIDORIDB_OK;
	if (first_index + n > domain)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "gen_keys: index range exceeds the permutation domain");
	mdb_perm perm = mdb_perm_make(domain, seed);
	MDB_LAUNCH(ctx, "gen_keys", k_gen_keys, stream_grid(n), STREAM_THREADS, keys, n, first_index, perm, modulus);
	return MIDORIDB_OK;
}

__global__ __launch_bounds__(STREAM_THREADS) void k_gen_payload(uint64_t *__restrict__ out, uint64_t n, uint64_t first, uint64_t seed, int kind)
{
	const uint64_t base = (uint64_t)blockIdx.x * (STREAM_THREADS * STREAM_ROUNDS);
#pragma unroll
	for (int r = 0; r < STREAM_ROUNDS; r++) {
		const uint64_t k = base + (uint64_t)r * STREAM_THREADS + threadIdx.x;
		if (k < n) {
			const uint64_t z = mdb_splitmix64_at(seed, first + k);
			if (kind == 1) {
				const double d = (double)(z >> 11) * 0x1.0p-53;
				out[k] = (uint64_t)__double_as_longlong(d);
			} else {
				out[k] = z >> 33;
			}
		}
	}
}

extern "C" int mdb_dev_gen_payload(mdb_dev_ctx *ctx, void *out, uint64_t n, uint64_t first_index, uint64_t seed, int kind)
{
	if (n == 0)
		return MIDORIDB_OK;
	if (kind != 0 && kind != 1)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "gen_payload: kind must be 0 (INT64) or 1 (DOUBLE)");
	MDB_LAUNCH(ctx, "gen_payload", k_gen_payload, stream_grid(n), STREAM_THREADS, (uint64_t *)out, n, first_index, seed, kind);
	return MIDORIDB_OK;
}
