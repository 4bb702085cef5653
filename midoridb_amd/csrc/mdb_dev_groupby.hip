/*
 * mdb_dev_groupby.hip - plain GROUP BY + COUNT(*) fast paths: key columns over a small value range, or with few distinct values (proc_groupby_clause, executor_select.c:1526-1588)
 * (split off mdb_dev_join.hip; what the files share: mdb_dev_join_internal.h).  Hand-written HIP for gfx950, HBM-bound
 * integer work: no MFMA.
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"	/* (struct mdb_bg_comp: a key that is the composite value of several columns) */

/* ------------------------------------------------------------------ GROUP BY over a small value range
 *
 * SELECT fa, COUNT(*) FROM A GROUP BY fa with a thousand distinct values is the other common shape of the operator, and
 * the worst one for the partitioned path: a thousand leaves of 10^5 equal rows each (3.8 ms per 10^8 rows through the
 * hot-key kernels).  When a sample of the column spans at most GD_RANGE / 2 values, every workgroup instead counts its
 * share of the rows directly in an LDS table indexed by (value - base) - COUNT and first row per value, the table
 * replicated per lane group when the range is small so that equal values in a wave do not meet on one LDS address -
 * and flushes it into a global table with one atomic pair per value it saw.  One streaming pass over the column.
 * A value outside the window (the sample missed it) is reported and the partitioned path takes over.
 * Spans up to GD_SPAN_MAX (some 10^4 product / city / customer ids: 10^8 rows took 2.3 ms through two exact partition levels
 * and the hashed leaves) get a window of 1.25 x the span in a table of up to GD_TABLE_MAX entries - 128 KiB of dynamic LDS,
 * one workgroup per CU. */
#define GD_RANGE 8192u		/* entries of the LDS table for spans up to GD_RANGE / 2 (two workgroups per CU), replicated per lane group */
#define GD_TABLE_MAX 16384u	/* entries of a workgroup's LDS table (8 bytes each) */
#define GD_SPAN_MAX 13000u
#define GD_THREADS 1024
#define GD_MIN_ROWS (1u << 18)

struct gd_args {
	const int64_t *keys;
	const uint64_t *nullbits;
	uint64_t n;
	int64_t base;
	uint32_t range;			/* values base .. base + range - 1 have a slot */
	uint32_t copy_shift, copy_mask;	/* slot = (value - base) | ((lane & copy_mask) << copy_shift) */
	uint32_t table;			/* entries of the LDS table: (copy_mask + 1) << copy_shift */
	uint32_t null_group;		/* NULL keys form a group (slot GD_TABLE_MAX of the global table) */
	unsigned long long *g_cnt;	/* [GD_TABLE_MAX + 1] */
	uint32_t *g_first;		/* [GD_TABLE_MAX + 1] */
	uint32_t *status;		/* bit 10: a value outside the window */
	struct mdb_bg_comp comp;	/* k_group_direct<true>: slot = the composite value of these columns (keys, nullbits, base unused) */
};

template <bool COMP = false>
__global__ __launch_bounds__(GD_THREADS) void k_group_direct(gd_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t gd_lds[];
	uint32_t *const s_cnt = gd_lds, *const s_first = gd_lds + a.table;
	__shared__ unsigned long long s_null_cnt;
	__shared__ uint32_t s_null_first;
	for (uint32_t i = threadIdx.x; i < a.table; i += GD_THREADS) {
		s_cnt[i] = 0;
		s_first[i] = 0xFFFFFFFFu;
	}
	if (threadIdx.x == 0) {
		s_null_cnt = 0;
		s_null_first = 0xFFFFFFFFu;
	}
	__syncthreads();
	const uint32_t copy = (mdb_lane() & a.copy_mask) << a.copy_shift;
	bool bad = false;
	for (uint64_t row0 = (uint64_t)blockIdx.x * (2 * GD_THREADS); row0 < a.n; row0 += (uint64_t)gridDim.x * (2 * GD_THREADS)) {
		const uint64_t i0 = row0 + 2 * (uint64_t)threadIdx.x;
		if (COMP) {
			/* GROUP BY over several columns of few values each (group_multi_packed, mdb_dev_sort.hip): the slot is built from the columns,
			 * most significant first - [NULL flag | image - lo] per column; a value outside its field is a value outside the window */
			uint32_t acc[2] = { 0u, 0u };
#pragma unroll
			for (int c = 0; c < MDB_BG_COMP_MAX; c++) {
				if (c >= a.comp.nkeys)
					break;
				const uint64_t *const col = a.comp.values[c], *const nb = a.comp.nullbits[c];
				const uint32_t kb = a.comp.kb[c];
				uint64_t q[2] = { 0ull, 0ull };
				if (i0 + 1 < a.n) {
					const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(col + i0);
					q[0] = t.x;
					q[1] = t.y;
				} else if (i0 < a.n) {
					q[0] = col[i0];
				}
#pragma unroll
				for (int u = 0; u < 2; u++) {
					const uint64_t row = i0 + (uint64_t)u;
					const bool valid = row < a.n, isnull = valid && nb && mdb_bit_is_set(nb, row);
					const uint64_t bits = q[u], im0 = (a.comp.is_double[c] && (bits >> 63)) ? ~bits : (bits ^ 0x8000000000000000ull);
					uint64_t d = (a.comp.desc[c] ? ~im0 : im0) - a.comp.lo[c];
					d = (isnull || !valid) ? 0ull : d;
					bad = bad || d > a.comp.span[c];
					const uint32_t flag = nb ? (uint32_t)(isnull == (a.comp.desc[c] != 0)) : 0u;
					acc[u] = nb ? (acc[u] << (kb + 1u)) | (flag << kb) | (uint32_t)d : (acc[u] << kb) | (uint32_t)d;
				}
			}
#pragma unroll
			for (int u = 0; u < 2; u++)
				if (i0 + (uint64_t)u < a.n && acc[u] < a.range) {
					const uint32_t idx = acc[u] | copy;
					atomicAdd(&s_cnt[idx], 1u);
					atomicMin(&s_first[idx], (uint32_t)(i0 + (uint64_t)u));
				}
			continue;
		}
		int64_t k[2] = { 0, 0 };
		if (i0 + 1 < a.n) {
			const longlong2 q = *reinterpret_cast<const longlong2 *>(a.keys + i0);
			k[0] = q.x;
			k[1] = q.y;
		} else if (i0 < a.n) {
			k[0] = a.keys[i0];
		}
#pragma unroll
		for (int u = 0; u < 2; u++) {
			const uint64_t row = i0 + (uint64_t)u;
			const bool valid = row < a.n;
			const bool isnull = valid && a.nullbits && mdb_bit_is_set(a.nullbits, row);
			if (a.nullbits) {
				const uint64_t nm = __ballot(isnull);
				if (nm && a.null_group && mdb_lane() == (uint32_t)__ffsll((long long)nm) - 1u) {
					atomicAdd(&s_null_cnt, (unsigned long long)__popcll(nm));
					atomicMin(&s_null_first, (uint32_t)row);	/* rows grow with the lane: the first NULL lane holds the smallest */
				}
			}
			if (valid && !isnull) {
				const uint64_t off = (uint64_t)k[u] - (uint64_t)a.base;
				if (off >= a.range) {
					bad = true;
				} else {
					const uint32_t idx = (uint32_t)off | copy;
					atomicAdd(&s_cnt[idx], 1u);
					atomicMin(&s_first[idx], (uint32_t)row);
				}
			}
		}
	}
	if (__ballot(bad) && mdb_lane() == 0)
		mdb_raise(a.status, 1024u);
	__syncthreads();
	const uint32_t copies = a.copy_mask + 1;
	for (uint32_t off = threadIdx.x; off < a.range; off += GD_THREADS) {
		unsigned long long total = 0;
		uint32_t first = 0xFFFFFFFFu;
		for (uint32_t c = 0; c < copies; c++) {
			const uint32_t idx = off | (c << a.copy_shift);
			total += s_cnt[idx];
			const uint32_t f = s_first[idx];
			first = f < first ? f : first;
		}
		if (total) {
			atomicAdd(&a.g_cnt[off], total);
			atomicMin(&a.g_first[off], first);
		}
	}
	if (threadIdx.x == 0 && s_null_cnt) {
		atomicAdd(&a.g_cnt[GD_TABLE_MAX], s_null_cnt);
		atomicMin(&a.g_first[GD_TABLE_MAX], s_null_first);
	}
}

/* g_cnt_r != NULL: join form - a group needs rows on both sides, COUNT(*) = left rows x right rows of the value.  The
 * record carries the SLOT (+ 1) beside the first row, not the count: counts of hot values (10^7 x 10^7 rows of one key)
 * do not fit beside a row id; k_group_direct_counts puts them in once the groups are in order. */
__global__ void k_group_direct_emit(gd_args a, const unsigned long long *g_cnt_r, uint32_t kbits, unsigned long long *rec, uint32_t *rec_n,
				    unsigned long long *joined)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t slot = t < a.range ? t : (t == a.range ? GD_TABLE_MAX : 0xFFFFFFFFu);
	if (slot == 0xFFFFFFFFu)
		return;
	const unsigned long long c = a.g_cnt[slot];
	if (!c)
		return;
	if (g_cnt_r) {
		const unsigned long long cr = g_cnt_r[slot];
		if (!cr || slot == GD_TABLE_MAX)
			return;
		atomicAdd(joined, c * cr);	/* both below 2^32 (row counts of one GPU's tables) */
	}
	rec[atomicAdd(rec_n, 1u)] = ((unsigned long long)a.g_first[slot] << (64 - kbits)) | (unsigned long long)(slot + 1);
}

__global__ void k_group_direct_counts(int64_t *out_count, uint64_t G, const unsigned long long *g_cnt, const unsigned long long *g_cnt_r)
{
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= G)
		return;
	const uint32_t slot = (uint32_t)out_count[g] - 1u;
	out_count[g] = (int64_t)(g_cnt_r ? g_cnt[slot] * g_cnt_r[slot] : g_cnt[slot]);
}

/* GROUP BY over several columns whose composite value has at most 14 bits (mdb_dev_sort.hip, group_multi_packed): the same LDS tables, the slot
 * built from the columns as they are loaded.  0 = done (groups in first-row order), 1 = not served (too many bits, too few rows, a value outside
 * its field - ranges from a sample), < 0 = error.  Synchronises. */
int mdb_group_count_direct_comp(mdb_dev_ctx *ctx, const struct mdb_bg_comp *comp, uint64_t n, uint32_t bits, uint32_t *out_first, int64_t *out_count,
				uint64_t cap, uint64_t *out_groups)
{
	if (bits > 14u || n < GD_MIN_ROWS || n >= 0xFFFFFFFFull || comp->nkeys < 1 || comp->nkeys > MDB_BG_COMP_MAX || ctx->explain)
		return 1;
	for (int c = 0; c < comp->nkeys; c++)
		if ((uintptr_t)comp->values[c] & 15u)
			return 1;
	const uint32_t shift = bits < 6u ? 6u : bits, range = 1u << shift;
	uint32_t copies = GD_RANGE >> shift;
	copies = copies > 64 ? 64 : (copies < 1 ? 1 : copies);
	uint32_t kbits = 0;
	const size_t order_bytes = mdb_order_records_arena_bytes(GD_TABLE_MAX + 1, n, &kbits);
	if (!order_bytes)
		return 1;
	int rc = mdb_arena_begin(ctx, order_bytes + 4 * mdb_align_up((GD_TABLE_MAX + 1) * 8) + 8192);
	if (rc)
		return rc;
	gd_args a;
	memset(&a, 0, sizeof(a));
	a.n = n;
	a.range = range;
	a.copy_shift = shift;
	a.copy_mask = copies - 1;
	a.table = copies << shift;
	a.comp = *comp;
	a.g_cnt = (unsigned long long *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 8);
	a.g_first = (uint32_t *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 4);
	a.status = ctx->d_status;
	unsigned long long *rec = (unsigned long long *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 8);
	if (!a.g_cnt || !a.g_first || !rec)
		return -MIDORIDB_INTERNAL;
	uint32_t *rec_n = ctx->d_status + 1;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_cnt, 0, (GD_TABLE_MAX + 1) * 8, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_first, 0xFF, (GD_TABLE_MAX + 1) * 4, ctx->stream));
	const size_t lds = (size_t)a.table * 8;
	const uint32_t resident = (lds > 80 * 1024 ? 1u : 2u) * (uint32_t)ctx->num_cus;
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_group_direct<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	const uint64_t chunks = (n + 2 * GD_THREADS - 1) / (2 * GD_THREADS);
	MDB_LAUNCH_LDS(ctx, "group_direct_columns", k_group_direct<true>, (uint32_t)(chunks < resident ? chunks : resident), GD_THREADS, lds, a);
	MDB_LAUNCH(ctx, "group_direct_emit", k_group_direct_emit, (range + 1 + 255) / 256, 256, a, (const unsigned long long *)NULL, kbits, rec, rec_n,
		   (unsigned long long *)(ctx->d_status + 2));
	uint32_t *h32 = (uint32_t *)ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h32, ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h32[0] & 1024u)
		return 1;	/* a value outside its field */
	const uint64_t G = h32[1];
	if (G > cap)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups", (unsigned long long)cap,
				   (unsigned long long)G);
	*out_groups = G;
	if (G == 0)
		return 0;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
	rc = mdb_order_records_by_rowid(ctx, rec, G, n, kbits, out_first, out_count);
	if (rc)
		return rc < 0 ? rc : -MIDORIDB_INTERNAL;
	MDB_LAUNCH(ctx, "group_direct_counts", k_group_direct_counts, (uint32_t)((G + 255) / 256), 256, out_count, G, (const unsigned long long *)a.g_cnt,
		   (const unsigned long long *)NULL);
	return mdb_dev_sync(ctx);
}

/* 0 = done, 1 = not applicable (use the partitioned path), < 0 = error.  keys_r != NULL: the join form (both key columns
 * inside one window; NULL keys never join) - also the cheap way through joins on a handful of hot values. */
int group_direct_try(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, const int64_t *keys_r,
			    const uint64_t *null_r, uint64_t n_r, bool null_group, int64_t *out_key, uint32_t *out_first, int64_t *out_count,
			    uint64_t cap, uint64_t *out_groups, uint64_t *out_joined)
{
	if (n + (keys_r ? n_r : 0) < GD_MIN_ROWS || n >= 0xFFFFFFFFull || ((uintptr_t)keys & 15) || ((!out_first || !out_count) && !ctx->explain))
		return 1;
	if (keys_r && (n_r >= 0xFFFFFFFFull || ((uintptr_t)keys_r & 15) || n_r == 0))
		return 1;
	/* range of a sample of the column(s) (shared with the narrow-form decision of the partitioned path) */
	int64_t lo = 0, hi = 0;
	int rc = gc_sample_range(ctx, keys, nullbits, n, keys_r, null_r, n_r, ctx->nh_distrust > 0, &lo, &hi);
	if (rc)
		return rc;
	if (lo > hi)
		return 1;
	const uint64_t span = (uint64_t)hi - (uint64_t)lo + 1;
	if (span > GD_SPAN_MAX)
		return 1;
	/* window: twice the sampled span (at least 64 values), centred on it - 1.25 x beyond GD_RANGE / 2 values, where the 4096
	 * samples lie within a few values of the column's extremes; replicated while copies fit a GD_RANGE-entry table */
	uint32_t range = (uint32_t)(span > GD_RANGE / 2 ? span + span / 4 : (2 * span < 64 ? 64 : 2 * span));
	uint32_t shift = 0;
	while ((1u << shift) < range)
		shift++;
	uint32_t copies = GD_RANGE >> shift;
	copies = copies > 64 ? 64 : (copies < 1 ? 1 : copies);
	uint32_t kbits = 0;
	const size_t order_bytes = mdb_order_records_arena_bytes(GD_TABLE_MAX + 1, n, &kbits);
	if (!order_bytes)
		return 1;
	if (ctx->explain) {	/* (mdb_dev_explain_*: per-workgroup LDS tables over a window of at most 4096 values - nothing is launched) */
		ctx->explain->small_form = 2;
		ctx->explain->from_stats = ctx->pl_from_stats;
		return MIDORIDB_OK;
	}
	rc = mdb_arena_begin(ctx, order_bytes + 6 * mdb_align_up((GD_TABLE_MAX + 1) * 8) + 8192);
	if (rc)
		return rc;
	gd_args a;
	memset(&a, 0, sizeof(a));
	a.keys = keys;
	a.nullbits = nullbits;
	a.n = n;
	a.base = (int64_t)((uint64_t)lo - (uint64_t)((range - span) / 2));
	a.range = range;
	a.copy_shift = shift;
	a.copy_mask = copies - 1;
	a.table = copies << shift;
	a.null_group = (null_group && !keys_r) ? 1u : 0u;
	a.g_cnt = (unsigned long long *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 8);
	a.g_first = (uint32_t *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 4);
	a.status = ctx->d_status;
	unsigned long long *rec = (unsigned long long *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 8);
	unsigned long long *g_cnt_r = keys_r ? (unsigned long long *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 8) : NULL;
	uint32_t *g_first_r = keys_r ? (uint32_t *)mdb_arena_take(ctx, (GD_TABLE_MAX + 1) * 4) : NULL;
	if (!a.g_cnt || !a.g_first || !rec || (keys_r && (!g_cnt_r || !g_first_r)))
		return -MIDORIDB_INTERNAL;
	uint32_t *rec_n = ctx->d_status + 1;
	unsigned long long *d_joined = (unsigned long long *)(ctx->d_status + 2);
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_cnt, 0, (GD_TABLE_MAX + 1) * 8, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_first, 0xFF, (GD_TABLE_MAX + 1) * 4, ctx->stream));
	const size_t lds = (size_t)a.table * 8;
	const uint32_t resident = (lds > 80 * 1024 ? 1u : 2u) * (uint32_t)ctx->num_cus;
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_group_direct<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	{
		const uint64_t chunks = (n + 2 * GD_THREADS - 1) / (2 * GD_THREADS);
		MDB_LAUNCH_LDS(ctx, "group_direct", k_group_direct<false>, (uint32_t)(chunks < resident ? chunks : resident), GD_THREADS, lds, a);
	}
	if (keys_r) {
		gd_args b = a;
		b.keys = keys_r;
		b.nullbits = null_r;
		b.n = n_r;
		b.g_cnt = g_cnt_r;
		b.g_first = g_first_r;
		MDB_HIP(ctx, hipMemsetAsync(g_cnt_r, 0, (GD_TABLE_MAX + 1) * 8, ctx->stream));
		MDB_HIP(ctx, hipMemsetAsync(g_first_r, 0xFF, (GD_TABLE_MAX + 1) * 4, ctx->stream));
		const uint64_t chunks = (n_r + 2 * GD_THREADS - 1) / (2 * GD_THREADS);
		MDB_LAUNCH_LDS(ctx, "group_direct", k_group_direct<false>, (uint32_t)(chunks < resident ? chunks : resident), GD_THREADS, lds, b);
	}
	MDB_LAUNCH(ctx, "group_direct_emit", k_group_direct_emit, (range + 1 + 255) / 256, 256, a, (const unsigned long long *)g_cnt_r, kbits, rec, rec_n,
		   d_joined);
	uint32_t *h32 = (uint32_t *)ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h32, ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h32[0] & 1024u)
		ctx->sr_valid = 0;	/* the sample missed a value outside its range: not to be reused */
	if (h32[0] & 1024u)
		return 1;	/* a value outside the window: the partitioned path */
	const uint64_t G = h32[1];
	const uint64_t joined = (uint64_t)h32[2] | ((uint64_t)h32[3] << 32);
	if (G > cap)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups", (unsigned long long)cap,
				   (unsigned long long)G);
	*out_groups = G;
	if (out_joined)
		*out_joined = joined;
	if (G == 0)
		return 0;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
	rc = mdb_order_records_by_rowid(ctx, rec, G, n, kbits, out_first, out_count);
	if (rc)
		return rc < 0 ? rc : -MIDORIDB_INTERNAL;
	MDB_LAUNCH(ctx, "group_direct_counts", k_group_direct_counts, (uint32_t)((G + 255) / 256), 256, out_count, G, (const unsigned long long *)a.g_cnt,
		   (const unsigned long long *)g_cnt_r);
	rc = mdb_dev_sync(ctx);
	if (rc)
		return rc;
	if (out_key) {
		rc = mdb_dev_gather64(ctx, keys, NULL, out_first, G, out_key, NULL);
		if (!rc)
			rc = mdb_dev_sync(ctx);
	}
	return rc;
}

/* ------------------------------------------------------------------ GROUP BY with few distinct values, any value range
 *
 * The direct tables above need the values themselves to be close together.  A thousand customer ids scattered over the
 * int64 range are the same shape - few groups, 10^5 equal rows each - and cost 3 ms per 10^8 rows through the partitioned
 * path (a thousand hot leaves).  When a sample of the column holds few distinct values, every workgroup aggregates its
 * share of the rows in an LDS hash table keyed by the hashed value (equal values of a wave are merged first, so a hot
 * value costs one table update per wave, not one per row) and merges its table into a global one; a workgroup table that
 * fills up (the sample was wrong about the column) raises a flag and the partitioned path takes over. */
#define GH_SLOTS 4093u		/* per workgroup (prime): 12 B/slot + ... = 64 KiB, two workgroups per CU */
#define GH_MAX_FILL 2800u	/* distinct values a workgroup table may take */
#define GH_GSLOTS 16381u	/* global table (prime) */
#define GH_SAMPLE_MAX 1200u	/* distinct values among the 4096 sampled keys up to which the path is tried */
#define GH_MERGE_ROUNDS 16

__global__ __launch_bounds__(1024) void k_key_sample_distinct(const int64_t *__restrict__ keys, const uint64_t *__restrict__ nullbits, uint64_t n,
							       uint32_t *out)
{
	/* number of distinct values among GC_NARROW_SAMPLE evenly spaced non-NULL keys (an LDS set) */
	__shared__ unsigned long long s_set[8192];
	__shared__ uint32_t s_n, s_zero;
	for (uint32_t i = threadIdx.x; i < 8192; i += 1024)
		s_set[i] = 0ull;
	if (threadIdx.x == 0)
		s_n = s_zero = 0;
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < GC_NARROW_SAMPLE; t += 1024) {
		const uint64_t i = gc_sample_pos(t, n);
		if (nullbits && mdb_bit_is_set(nullbits, i))
			continue;
		const uint64_t hv = mdb_fmix64((uint64_t)keys[i]);
		if (hv == 0) {
			if (atomicExch(&s_zero, 1u) == 0)
				atomicAdd(&s_n, 1u);
			continue;
		}
		uint32_t s = (uint32_t)(((hv >> 32) * 8192ull) >> 32);
		for (;;) {
			const unsigned long long old = atomicCAS(&s_set[s], 0ull, (unsigned long long)hv);
			if (old == 0ull) {
				atomicAdd(&s_n, 1u);
				break;
			}
			if (old == hv)
				break;
			s = (s + 1) & 8191u;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0)
		out[0] = s_n;
}

struct gh_args {
	const int64_t *keys;
	const uint64_t *nullbits;
	uint64_t n;
	uint32_t null_group;
	unsigned long long *g_key;	/* [GH_GSLOTS] hashed value, 0 = empty */
	unsigned long long *g_cnt;	/* [GH_GSLOTS + 2]: + the value whose hash is 0, + the NULL group */
	uint32_t *g_first;		/* [GH_GSLOTS + 2] */
	uint32_t *status;		/* bit 11: a table filled up */
};

__global__ __launch_bounds__(GD_THREADS) void k_group_hashed(gh_args a)
{
	__shared__ unsigned long long s_key[GH_SLOTS];
	__shared__ uint32_t s_cnt[GH_SLOTS + 2];	/* [GH_SLOTS] hash-0 value, [GH_SLOTS + 1] NULL group */
	__shared__ uint32_t s_first[GH_SLOTS + 2];
	__shared__ uint32_t s_fill, s_bad;
	for (uint32_t i = threadIdx.x; i < GH_SLOTS + 2; i += GD_THREADS) {
		if (i < GH_SLOTS)
			s_key[i] = 0ull;
		s_cnt[i] = 0;
		s_first[i] = 0xFFFFFFFFu;
	}
	if (threadIdx.x == 0)
		s_fill = s_bad = 0;
	__syncthreads();
	for (uint64_t row0 = (uint64_t)blockIdx.x * (2 * GD_THREADS); row0 < a.n; row0 += (uint64_t)gridDim.x * (2 * GD_THREADS)) {
		if (s_bad)
			break;		/* (uniform enough: read by every thread at the top of a round; a late reader only does one more round) */
		const uint64_t i0 = row0 + 2 * (uint64_t)threadIdx.x;
		int64_t k[2] = { 0, 0 };
		if (i0 + 1 < a.n) {
			const longlong2 q = *reinterpret_cast<const longlong2 *>(a.keys + i0);
			k[0] = q.x;
			k[1] = q.y;
		} else if (i0 < a.n) {
			k[0] = a.keys[i0];
		}
#pragma unroll
		for (int u = 0; u < 2; u++) {
			const uint64_t row = i0 + (uint64_t)u;
			const bool valid = row < a.n;
			const bool isnull = valid && a.nullbits && mdb_bit_is_set(a.nullbits, row);
			if (a.nullbits) {
				const uint64_t nm = __ballot(isnull);
				if (nm && a.null_group && mdb_lane() == (uint32_t)__ffsll((long long)nm) - 1u) {
					atomicAdd(&s_cnt[GH_SLOTS + 1], (uint32_t)__popcll(nm));
					atomicMin(&s_first[GH_SLOTS + 1], (uint32_t)row);
				}
			}
			const bool act = valid && !isnull;
			const uint64_t hv = act ? mdb_fmix64((uint64_t)k[u]) : 0ull;
			/* equal values of the wave are merged: up to GH_MERGE_ROUNDS leaders update the table for all lanes that
			 * hold their value; whoever is left (many distinct values in the wave: little contention) goes alone */
			uint64_t pending = __ballot(act);
			uint32_t mult = 1, first_row = (uint32_t)row;
			bool mine_todo = act;
			for (int round = 0; round < GH_MERGE_ROUNDS && pending; round++) {
				const int leader = __ffsll((long long)pending) - 1;
				const uint32_t llo = (uint32_t)__shfl((int)(uint32_t)hv, leader, MDB_WAVE);
				const uint32_t lhi = (uint32_t)__shfl((int)(uint32_t)(hv >> 32), leader, MDB_WAVE);
				const bool same = mine_todo && (uint32_t)hv == llo && (uint32_t)(hv >> 32) == lhi;
				const uint64_t grp = __ballot(same);
				if (same) {
					if ((int)mdb_lane() == leader) {
						mult = (uint32_t)__popcll(grp);		/* the leader holds the smallest row of its group (rows grow with the lane) */
					} else {
						mine_todo = false;
					}
				}
				pending &= ~grp;	/* (the leader stays in the loop: its value cannot come up again, and all leaders then
							 * update the table together instead of one after the other) */
				if (__popcll(grp) < 3)
					break;		/* the wave's values are diverse: merging more leaders costs more than the atomics it saves */
			}
			if (mine_todo) {
				uint32_t s;
				if (hv == 0) {
					s = GH_SLOTS;
				} else {
					s = leaf_slot(hv, GH_SLOTS);
					const uint32_t step = leaf_step(hv, GH_SLOTS);
					uint32_t probe = 0;
					for (;;) {
						const unsigned long long old = atomicCAS(&s_key[s], 0ull, (unsigned long long)hv);
						if (old == hv)
							break;
						if (old == 0ull) {
							if (atomicAdd(&s_fill, 1u) >= GH_MAX_FILL)
								s_bad = 1;
							break;
						}
						if (++probe >= GH_SLOTS) {
							s_bad = 1;
							s = 0xFFFFFFFFu;
							break;
						}
						s += step;
						if (s >= GH_SLOTS)
							s -= GH_SLOTS;
					}
				}
				if (s != 0xFFFFFFFFu) {
					atomicAdd(&s_cnt[s], mult);
					atomicMin(&s_first[s], first_row);
				}
			}
		}
	}
	__syncthreads();
	if (s_bad) {
		if (threadIdx.x == 0)
			mdb_raise(a.status, 2048u);
		return;
	}
	/* merge into the global table */
	for (uint32_t s = threadIdx.x; s < GH_SLOTS + 2; s += GD_THREADS) {
		const uint32_t c = s_cnt[s];
		if (!c)
			continue;
		uint32_t g;
		if (s >= GH_SLOTS) {
			g = GH_GSLOTS + (s - GH_SLOTS);
		} else {
			const uint64_t hv = s_key[s];
			g = leaf_slot(hv, GH_GSLOTS);
			const uint32_t step = leaf_step(hv, GH_GSLOTS);
			uint32_t probe = 0;
			for (;;) {
				const unsigned long long old = atomicCAS(&a.g_key[g], 0ull, (unsigned long long)hv);
				if (old == 0ull || old == hv)
					break;
				if (++probe >= GH_GSLOTS) {
					mdb_raise(a.status, 2048u);
					g = 0xFFFFFFFFu;
					break;
				}
				g += step;
				if (g >= GH_GSLOTS)
					g -= GH_GSLOTS;
			}
		}
		if (g != 0xFFFFFFFFu) {
			atomicAdd(&a.g_cnt[g], (unsigned long long)c);
			atomicMin(&a.g_first[g], s_first[s]);
		}
	}
}

__global__ void k_group_hashed_emit(gh_args a, uint32_t kbits, unsigned long long *rec, uint32_t *rec_n)
{
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= GH_GSLOTS + 2 || !a.g_cnt[g])
		return;
	rec[atomicAdd(rec_n, 1u)] = ((unsigned long long)a.g_first[g] << (64 - kbits)) | (unsigned long long)(g + 1);
}

/* 0 = done, 1 = not applicable, < 0 = error */
int group_hashed_try(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, bool null_group, uint32_t *out_first,
			    int64_t *out_count, uint64_t cap, uint64_t *out_groups)
{
	if (n < GD_MIN_ROWS || n >= 0xFFFFFFFFull || ((uintptr_t)keys & 15) || !out_first || !out_count)
		return 1;	/* (mdb_dev_explain_group_count passes no outputs: whether a column has few distinct values is no statistic a catalog hands over) */
	/* distinct values in a sample of the column, remembered like the range sample */
	uint32_t distinct;
	if (ctx->gh_keys == keys && ctx->gh_n == n && ++ctx->gh_uses < GC_HINT_USES) {
		distinct = ctx->gh_distinct;
	} else {
		uint32_t *d = ctx->d_status + 9;
		MDB_LAUNCH(ctx, "key_sample_distinct", k_key_sample_distinct, 1, 1024, keys, nullbits, n, d);
		uint32_t *h = (uint32_t *)ctx->h_pinned;
		MDB_HIP(ctx, hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		distinct = h[0];
		ctx->gh_keys = keys;
		ctx->gh_n = n;
		ctx->gh_distinct = distinct;
		ctx->gh_uses = 0;
	}
	if (distinct > GH_SAMPLE_MAX)
		return 1;
	uint32_t kbits = 0;
	const size_t order_bytes = mdb_order_records_arena_bytes(GH_GSLOTS + 2, n, &kbits);
	if (!order_bytes)
		return 1;
	int rc = mdb_arena_begin(ctx, order_bytes + 4 * mdb_align_up((GH_GSLOTS + 2) * 8) + 8192);
	if (rc)
		return rc;
	gh_args a;
	memset(&a, 0, sizeof(a));
	a.keys = keys;
	a.nullbits = nullbits;
	a.n = n;
	a.null_group = null_group ? 1u : 0u;
	a.g_key = (unsigned long long *)mdb_arena_take(ctx, GH_GSLOTS * 8);
	a.g_cnt = (unsigned long long *)mdb_arena_take(ctx, (GH_GSLOTS + 2) * 8);
	a.g_first = (uint32_t *)mdb_arena_take(ctx, (GH_GSLOTS + 2) * 4);
	a.status = ctx->d_status;
	unsigned long long *rec = (unsigned long long *)mdb_arena_take(ctx, (GH_GSLOTS + 2) * 8);
	if (!a.g_key || !a.g_cnt || !a.g_first || !rec)
		return -MIDORIDB_INTERNAL;
	uint32_t *rec_n = ctx->d_status + 1;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 8, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_key, 0, GH_GSLOTS * 8, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_cnt, 0, (GH_GSLOTS + 2) * 8, ctx->stream));
	MDB_HIP(ctx, hipMemsetAsync(a.g_first, 0xFF, (GH_GSLOTS + 2) * 4, ctx->stream));
	const uint64_t chunks = (n + 2 * GD_THREADS - 1) / (2 * GD_THREADS);
	const uint32_t resident = 2u * (uint32_t)ctx->num_cus;
	MDB_LAUNCH(ctx, "group_hashed", k_group_hashed, (uint32_t)(chunks < resident ? chunks : resident), GD_THREADS, a);
	MDB_LAUNCH(ctx, "group_hashed_emit", k_group_hashed_emit, (GH_GSLOTS + 2 + 255) / 256, 256, a, kbits, rec, rec_n);
	uint32_t *h32 = (uint32_t *)ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h32, ctx->d_status, 8, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h32[0] & 2048u) {
		ctx->gh_distinct = 0xFFFFFFFFu;	/* the sample was wrong about this column: not tried again while it is remembered */
		return 1;
	}
	const uint64_t G = h32[1];
	if (G > cap)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups", (unsigned long long)cap,
				   (unsigned long long)G);
	*out_groups = G;
	if (G == 0)
		return 0;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
	rc = mdb_order_records_by_rowid(ctx, rec, G, n, kbits, out_first, out_count);
	if (rc)
		return rc < 0 ? rc : -MIDORIDB_INTERNAL;
	MDB_LAUNCH(ctx, "group_direct_counts", k_group_direct_counts, (uint32_t)((G + 255) / 256), 256, out_count, G, (const unsigned long long *)a.g_cnt,
		   (const unsigned long long *)NULL);
	return mdb_dev_sync(ctx);
}
