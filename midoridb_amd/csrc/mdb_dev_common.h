/*
 * mdb_dev_common.h - internals shared by the HIP translation units of the device layer
 * (context, scratch arena, launch/profiling helpers, wave64 primitives).  gfx950 only.
 */
#ifndef MDB_DEV_COMMON_H
#define MDB_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>
#include <unordered_map>
#include "mdb_dev.h"

#define MDB_WAVE 64

struct mdb_prof_rec {
	int name_id;
	hipEvent_t start, stop;
};

/* What the join / GROUP BY operators remember about key columns (sampled ranges, key forms, duplicate flags, layouts that
 * overflowed): verdicts keyed by the columns' device addresses and lengths, each verified on the device when it is used - a
 * stale one costs a failed attempt, never a result.  The live set belongs to ONE pair of key columns (memo_key); the context
 * keeps the sets of the last MDB_MEMO_SLOTS pairs (mdb_memo_switch, mdb_dev_core.hip), so queries that alternate over several
 * table pairs do not evict each other's verdicts and pay the key sample + its host sync on every call. */
struct mdb_col_memo {
	/* outcome of the last narrow-form decision, keyed by the key columns it was made for: a repeated query over the same
	 * columns skips the sampling kernel and its sync (every key is still verified while it is partitioned) */
	const void *nh_kl, *nh_kr;
	uint64_t nh_nl, nh_nr;
	int nh_result;			/* -1 nothing remembered, 0 wide, 1 narrow */
	int64_t nh_base;		/* the window centre that went with "narrow" */
	uint32_t nh_kbits;		/* ... and the compact window the sample offered (0 = none): [nh_lo, nh_lo + 2^nh_kbits) */
	int64_t nh_lo;
	bool nh_r_based;		/* ... and the compact window was taken from the right table's keys alone */
	int64_t sr_rlo, sr_rhi;		/* the right table's sampled extremes (last sample) */
	bool nh_prunable;		/* ... or at least not all of it (min-max pruning worth recording the right table's range) */
	bool nh_by_span;		/* ... because its sampled SPAN is small (min-max pruning does the work, no bitmap) */
	bool nh_selective;		/* ... and whether the right table looked like it covers a small part of the left table's keys */
	uint64_t sr_span_l, sr_span_r;	/* spans of the two tables' sampled keys (last sample) */
	/* range of the last key sample (smallest / largest of 2 x 4096 evenly spaced keys), by the columns it was taken from */
	const void *sr_kl, *sr_kr;
	uint64_t sr_nl, sr_nr;
	int64_t sr_lo, sr_hi;
	int sr_valid;
	const void *gh_keys;		/* distinct values among the sampled keys of a column (group_hashed_try) */
	uint64_t gh_n;
	uint32_t gh_distinct;
	int gh_uses;
	int sr_uses, nh_uses;		/* remembered verdicts expire after a few uses: the same buffer may hold other data by then */
	int nh_distrust;		/* > 0: a remembered "narrow" just proved wrong (buffer reused for other data): sample again for a while */
	const void *ex_keys;		/* left key column (and the two row counts) whose histogram-free partition layout overflowed last time: */
	uint64_t ex_nl, ex_nr;		/* the exact layout at once, for a few uses (a failed attempt costs a whole partition pass) */
	int ex_uses;
	const void *pw_bad_keys;	/* right key column (and row count) the one-level unique-key join (join_pairs_unique_wide) gave up on */
	uint64_t pw_bad_n;
	const void *jk_dup_l, *jk_dup_r;	/* key columns whose join had COUNTs above 1 (mdb_dev_join_keys asks for the COUNT column at once) */
	uint64_t jk_dup_nl, jk_dup_nr;
	int jk_dup_uses;
	const void *jp_bad_l, *jp_bad_r;	/* key columns mdb_dev_join_payload found not to be an every-left-row-has-its-partner join */
	uint64_t jp_bad_nl, jp_bad_nr;
	int jp_bad_skips;
	const void *pu_dup_keys;	/* right key column that the unique-key join found duplicates in (not tried again) */
	uint64_t pu_dup_n;
	const void *pu_dupl_keys;	/* ... and the left key column that did (both: a true N:M join, the general path) */
	uint64_t pu_dupl_n;
	int pu_dup_skips;
	bool r32_ok;			/* the last join over (r32_kl, r32_nl, r32_kr, r32_nr) fitted every COUNT(*) into a 4-byte group record */
	const void *r32_kl, *r32_kr;
	uint64_t r32_nl, r32_nr;
	const void *lw_bad_keys;	/* the left key column (and the two row counts) for which a 16-bit row count of the one-level direct leaves */
	uint64_t lw_bad_nl, lw_bad_nr;	/* (k_leaf_wide) overflowed last: two levels for these columns */
	const void *l4_bad_keys;	/* ... and the one for which k_leaf_wide4's 5-bit / 4-bit counts did (more than 31 right or 15 left rows of a key): */
	uint64_t l4_bad_nl, l4_bad_nr;	/* k_leaf_wide's 16-bit counts at once */
	int lw_bad_uses, l4_bad_uses;	/* both verdicts serve MDB_BAD_LEAF_USES calls, then the fast form is tried again: the columns go by address and length,
					 * and a caller's allocator hands the same addresses out for other data */
	int keyed_distrust;		/* > 0: a COUNT(*) did not fit a keyed group record lately - plain records for the next operators */
	const void *lg_kl, *lg_kr;	/* the last join over (lg_kl, lg_nl, lg_kr, lg_nr) delivered lg_groups groups: when that is under a quarter of the */
	uint64_t lg_nl, lg_nr, lg_groups;	/* left rows, most left rows find no partner whatever the tables' sizes and ranges say (a right table */
	bool lg_valid;			/* of few distinct keys spread over the left table's range): the next join filters them early */
	uint32_t lg_nextra;		/* (further right tables of that join: the three-table operator remembers too) */
	uint64_t lg_joined;		/* ... and lg_joined joined rows: nearly every left row a group of COUNT 1 -> the groups as one bit per row (k_leaf_wide12) */
	int dn_distrust;		/* calls for which that form is not tried (its exception list overflowed) */
};

#define MDB_MEMO_SLOTS 8
#define MDB_BAD_LEAF_USES 32
struct mdb_memo_key {
	const void *kl, *kr;
	uint64_t nl, nr;
	bool operator==(const mdb_memo_key &o) const { return kl == o.kl && kr == o.kr && nl == o.nl && nr == o.nr; }
};

struct mdb_dev_ctx : mdb_col_memo {
	int device;
	int num_cus;			/* compute units of the device (256 on MI355X) */
	hipStream_t stream;		/* stream every operator launches on */
	bool own_stream;
	hipStream_t aux_stream;		/* second stream: the two tables of a join are partitioned concurrently */
	hipEvent_t ev_fork, ev_join;
	bool overlap;			/* false: everything on the main stream (isolated per-kernel timing) */
	int last_semijoin;		/* ... and dropped left rows through the right table's key bitmap (0 no; else 1 + log2 values per bit) */
	int last_narrow;		/* the last join / GROUP BY operator ran in the narrow form */
	bool last_left_dups_known, last_left_dups;	/* ... through leaf kernels that tell whether a key with partners has several LEFT rows, and whether one has */
	int last_pairs_identity;	/* the last mdb_dev_join_pairs: every left row joined exactly one right row - its left vector is 0, 1, 2 ... */
	int narrow_mode;		/* 32-bit hashes for int32-range join keys: 0 never, 1 sampled + verified (default), 2 always try */
	int unordered_no_counts;	/* set by mdb_dev_join_keys around its call of the any-order operator: group keys only, no COUNT column */
	/* the caller's MDB_KEYS_MAY_ALIAS / MDB_COUNTS_OPTIONAL of the current join + GROUP BY call, and what became of them (mdb_dev_last_plan) */
	bool key_alias_ok, counts_optional;
	uint32_t pl_keys_left, pl_counts_one, pl_payload_tables;
	void *pending_op;		/* state of a begun-but-unfinished split operator (mdb_dev_join.hip) */
	bool guess_remembered;		/* the last narrow-form decision came from the memo, not from a sample of the data */
	/* mdb_dev_call_stats(): the caller's statistics of the key columns of the calls that follow */
	bool cs_on, cs_has_r;
	const void *cs_kl, *cs_kr;
	struct mdb_dev_col_stats cs_l, cs_r;
	/* mdb_dev_last_plan(): what the current / last operator did beyond the last_* words */
	uint32_t pl_retries, pl_samples, pl_from_stats, pl_key_bits, pl_payload_form, pl_group_form, pl_bits, pl_small_form;
	int pl_depth;			/* operators that call operators: the outermost one's entry clears the counters (mdb_plan_scope) */
	/* mdb_dev_counters(): running totals since the context was created (what a slow call paid for) */
	uint64_t ct_calls, ct_retries, ct_samples, ct_arena_grows, ct_alloc_misses;
	/* mdb_dev_explain_*(): the operator's own decision code runs and stops in front of its first launch (a context without a device) */
	struct mdb_dev_plan_info *explain;
	bool explain_as_sample;		/* ... as if the statistics were what a key sample found: the forms only a catalog's promise opens are not taken */
	mdb_memo_key memo_key;		/* the key-column pair the live mdb_col_memo belongs to */
	std::vector<std::pair<mdb_memo_key, mdb_col_memo>> memo_lru;	/* the other pairs' sets, most recently used last */
	/* mdb_dev_alloc / mdb_dev_free recycle buffers (stream-ordered reuse on the context's stream): a query
	 * allocates dozens of temporaries and hipMalloc/hipFree cost 0.1-0.3 ms each */
	std::unordered_map<void *, size_t> live;		/* buffers handed out -> size */
	std::unordered_map<void *, uint32_t> holders;		/* ... -> holders beyond the first (mdb_dev_retain) */
	std::vector<std::pair<void *, size_t>> cache;		/* released buffers kept for reuse */
	size_t cache_bytes;
	char err[512];
	/* scratch arena (grow-only, bump allocated per operator) */
	char *arena;
	size_t arena_cap;
	size_t arena_off;
	/* device-side status words: [0] = leaf hash-table overflow flag, [1..] scratch */
	uint32_t *d_status;
	/* pinned host mirror for small read-backs */
	uint64_t *h_pinned;
	/* profiling */
	bool prof_on;
	std::vector<std::string> prof_names;
	std::vector<std::vector<const void *>> prof_kernels;	/* per name: the kernel functions (template instances) launched under it */
	std::vector<mdb_prof_rec> prof_recs;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
	size_t prof_pool_used;
};

int mdb_set_err(mdb_dev_ctx *ctx, int code, const char *fmt, ...);
/* the value of an MDB_* environment knob (NULL: not set) - the library's ONE reader of the environment, kept per process (mdb_dev_core.hip) */
extern "C" const char *mdb_knob(const char *name);

/* first statement of every public join / GROUP BY operator: what mdb_dev_last_plan() counts starts at the OUTERMOST operator's entry */
struct mdb_plan_scope {
	mdb_dev_ctx *c;
	explicit mdb_plan_scope(mdb_dev_ctx *ctx) : c(ctx)
	{
		if (c && c->pl_depth++ == 0)
			c->pl_retries = c->pl_samples = c->pl_from_stats = c->pl_key_bits = c->pl_payload_form = c->pl_group_form = c->pl_bits = c->pl_small_form = c->pl_keys_left = c->pl_counts_one = c->pl_payload_tables = 0;
	}
	~mdb_plan_scope()
	{
		if (c && --c->pl_depth == 0) {
			c->ct_calls++;
			c->ct_retries += c->pl_retries;
			c->ct_samples += c->pl_samples;
		}
	}
};

/* make the live memo the one of the key-column pair (kl, nl, kr, nr) - kr = NULL for a one-column operator (GROUP BY): the set
 * of the pair used before is put aside, the pair's own set (or an empty one) comes back (mdb_dev_core.hip) */
void mdb_memo_switch(mdb_dev_ctx *ctx, const void *kl, uint64_t nl, const void *kr, uint64_t nr);

#define MDB_HIP(ctx, call)                                                                         \
	do {                                                                                       \
		hipError_t e__ = (call);                                                           \
		if (e__ != hipSuccess)                                                             \
			return mdb_set_err((ctx), -MIDORIDB_INTERNAL, "%s failed: %s (%s:%d)", #call, \
					   hipGetErrorString(e__), __FILE__, __LINE__);            \
	} while (0)

/* cached device allocations (mdb_dev_core.hip) */
int mdb_cached_alloc(mdb_dev_ctx *ctx, size_t bytes, void **dptr);
int mdb_cached_free(mdb_dev_ctx *ctx, void *dptr);

/* scratch arena */
int mdb_arena_begin(mdb_dev_ctx *ctx, size_t total_bytes);
void *mdb_arena_take(mdb_dev_ctx *ctx, size_t bytes);
/* behind the 128 status words of ctx->d_status: a block of counters that one memset clears together with them at the start of an
 * operator call - the first-level cursors of both tables (2 x 4096 words) and the ordering ranges' fills (2048 words) - instead of a
 * fill command each (5 us apiece between two kernels) */
#define MDB_ZERO_BLK_OFF 128u
#define MDB_ZERO_BLK_SLOT 4096u
#define MDB_ZERO_BLK_WORDS (2u * MDB_ZERO_BLK_SLOT + 2048u)
static inline size_t mdb_align_up(size_t x) { return (x + 255) & ~(size_t)255; }

/* run the following launches on the auxiliary stream (after everything queued so far on the main
 * stream), and come back; between the two calls ctx->stream IS the auxiliary stream */
int mdb_aux_begin(mdb_dev_ctx *ctx, hipStream_t *saved_main);
int mdb_aux_end(mdb_dev_ctx *ctx, hipStream_t saved_main);	/* back to the main stream; marks the end of the aux work */
int mdb_aux_join(mdb_dev_ctx *ctx);				/* main stream waits for the marked aux work */

/* profiling hooks around one launch */
void mdb_prof_begin(mdb_dev_ctx *ctx, const char *name, const void *kernel);
void mdb_prof_end(mdb_dev_ctx *ctx);

#define MDB_LAUNCH(ctx, name, kernel, grid, block, ...)                                            \
	do {                                                                                       \
		mdb_prof_begin((ctx), (name), (const void *)(kernel));                             \
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__); \
		mdb_prof_end((ctx));                                                               \
		hipError_t le__ = hipGetLastError();                                               \
		if (le__ != hipSuccess)                                                            \
			return mdb_set_err((ctx), -MIDORIDB_INTERNAL, "launch %s failed: %s", (name), \
					   hipGetErrorString(le__));                               \
	} while (0)

/* the same with dynamic LDS */
#define MDB_LAUNCH_LDS(ctx, name, kernel, grid, block, lds_bytes, ...)                             \
	do {                                                                                       \
		mdb_prof_begin((ctx), (name), (const void *)(kernel));                             \
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (ctx)->stream, __VA_ARGS__); \
		mdb_prof_end((ctx));                                                               \
		hipError_t le__ = hipGetLastError();                                               \
		if (le__ != hipSuccess)                                                            \
			return mdb_set_err((ctx), -MIDORIDB_INTERNAL, "launch %s failed: %s", (name), \
					   hipGetErrorString(le__));                               \
	} while (0)

/* ------------------------------------------------------------------ device helpers */
#ifdef __HIPCC__

/* murmur3 fmix64: a bijection on 64-bit words (so the hashed value identifies the key),
 * fmix64(0) == 0. */
__host__ __device__ static inline uint64_t mdb_fmix64(uint64_t k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ULL;
	k ^= k >> 33;
	return k;
}

/* murmur3 fmix32: a bijection on 32-bit words, fmix32(0) == 0.  The narrow form of the join (every key of both
 * tables inside the int32 range) partitions and compares 32-bit hashes. */
__host__ __device__ static inline uint32_t mdb_fmix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x85EBCA6Bu;
	h ^= h >> 13;
	h *= 0xC2B2AE35u;
	h ^= h >> 16;
	return h;
}

/* The same finaliser on k-bit words (8 <= k <= 32): x ^= x >> s and a multiplication by an odd constant modulo 2^k are
 * both bijections of [0, 2^k), so mixk is one, and mixk(0) == 0.  The compact narrow form uses it on key - window base: a
 * key column whose values span fewer than 2^k values hashes into k bits, the radix partition consumes the top bits and what
 * is left below them is small enough to INDEX a per-leaf table in LDS directly (mdb_dev_join.hip, k_leaf_direct). */
__host__ __device__ static inline uint32_t mdb_mixk(uint32_t x, uint32_t k)
{
	const uint32_t mask = k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
	const uint32_t s = (k + 1) >> 1;
	x ^= x >> s;
	x = (x * 0x85EBCA6Bu) & mask;
	x ^= x >> s;
	x = (x * 0xC2B2AE35u) & mask;
	x ^= x >> s;
	return x;
}

/* inverse of mdb_mixk: x ^= x >> s is an involution on k-bit words (2 s >= k), the multipliers have inverses modulo 2^32
 * and therefore modulo 2^k */
__host__ __device__ static inline uint32_t mdb_unmixk(uint32_t x, uint32_t k)
{
	const uint32_t mask = k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
	const uint32_t s = (k + 1) >> 1;
	x ^= x >> s;
	x = (x * 0x7ED1B41Du) & mask;		/* 0xC2B2AE35^-1 */
	x ^= x >> s;
	x = (x * 0xA5CB9243u) & mask;		/* 0x85EBCA6B^-1 */
	x ^= x >> s;
	return x;
}

/* inverse of mdb_fmix64 (modular inverses of the two odd multipliers; x ^= x >> 33 is an involution
 * on 64-bit words because 2*33 >= 64). */
__host__ __device__ static inline uint64_t mdb_fmix64_inv(uint64_t k)
{
	k ^= k >> 33;
	k *= 0x9cb4b2f8129337dbULL;
	k ^= k >> 33;
	k *= 0x4f74430c22a54005ULL;
	k ^= k >> 33;
	return k;
}

__device__ static inline uint32_t mdb_lane(void) { return threadIdx.x & (MDB_WAVE - 1); }
__device__ static inline uint64_t mdb_lanemask_lt(void) { return (1ull << mdb_lane()) - 1ull; }

/* inclusive scan across the 64 lanes of a wave: data-parallel-primitive moves inside the vector ALU (shifts by 1, 2, 4, 8 inside
 * every row of 16 lanes, then the last lane of a row broadcast to the next row, then lane 31 to the upper half) - six adds, no
 * trip through the LDS crossbar (six ds_bpermute round trips of ~60 cycles each before) */
__device__ static inline uint32_t mdb_wave_incl_scan(uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
	return v;
}

/* Block-wide exclusive scan of one value per thread.  blockDim.x must be a multiple of 64 and
 * <= 1024; `tmp` is LDS scratch of at least 17 words.  Returns the exclusive prefix, *total the
 * block sum.  Contains __syncthreads(): call from uniform control flow.
 * Every wave scans the (at most 16) wave totals itself: two barriers and no serial walk by one thread (which cost 2 000 -
 * 4 000 cycles per call with 16 waves - a quarter of a partition tile's fixed costs). */
__device__ static inline uint32_t mdb_block_excl_scan(uint32_t v, uint32_t *tmp, uint32_t *total)
{
	const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, lane = mdb_lane();
	const uint32_t incl = mdb_wave_incl_scan(v);
	__syncthreads();	/* protect tmp against a previous use */
	if (lane == MDB_WAVE - 1)
		tmp[wave] = incl;
	__syncthreads();
	const uint32_t pi = mdb_wave_incl_scan(lane < nwaves ? tmp[lane] : 0u);
	*total = (uint32_t)__shfl((int)pi, (int)nwaves - 1, MDB_WAVE);
	const uint32_t before = (uint32_t)__shfl((int)pi, wave ? (int)wave - 1 : 0, MDB_WAVE);
	return incl - v + (wave ? before : 0u);
}

/* Raise flag bits in a device status word.  The word is looked at first: a condition that holds for every row of a
 * table (duplicate keys under the unique-key join, wide keys under the narrow form) would otherwise send one global
 * atomic per row to a single address - 10^8 of them took 18 ms. */
__device__ static inline void mdb_raise(uint32_t *status, uint32_t bits)
{
	if ((*(volatile const uint32_t *)status & bits) != bits)
		atomicOr(status, bits);
}

__device__ static inline bool mdb_bit_is_set(const uint64_t *bits, uint64_t i)
{
	return (bits[i >> 6] >> (i & 63)) & 1ull;
}

#endif /* __HIPCC__ */
#endif /* MDB_DEV_COMMON_H */
