/*
 * mdb_dev_join_internal.h - what the translation units of the join / GROUP BY operators share: the leaf-table helpers, the
 * argument block of the fused operator's leaf kernels, the retry codes of its driver, and the host functions that cross the files
 *
 *   mdb_dev_join.hip     fused join + GROUP BY key + COUNT(*) (leaf kernels, planning, retries, split and N-way forms), plain GROUP BY driver
 *   mdb_dev_order.hip    ordering of the group records by first row id
 *   mdb_dev_groupby.hip  plain GROUP BY fast paths (small value range, few distinct values)
 *   mdb_dev_pairs.hip    materialising INNER JOIN
 */
#ifndef MDB_DEV_JOIN_INTERNAL_H
#define MDB_DEV_JOIN_INTERNAL_H

#include <stdlib.h>
#include "mdb_dev_internal.h"

/* ------------------------------------------------------------------ shared leaf helpers */

#define LEAF_THREADS 512		/* pairs-join leaf kernels */
#define GC_THREADS 1024		/* group-count leaf kernel: 16 waves x 2 workgroups = 32 waves/CU hide the LDS probe latency */
#define GC_EMIT_ITERS ((GC_SLOTS + 1 + GC_THREADS - 1) / GC_THREADS)	/* table slots visited per thread */
#define GC_REC_CHUNK 16384u	/* record-list slots a workgroup reserves at a time (one global atomic per chunk) */
#define LEAF_BATCH 2		/* keys loaded per thread before the first is consumed */
#define GC_SLOTS 3833u		/* group-count table (prime, for double hashing): 20 B/slot -> 75 KiB, two workgroups per CU */
#define GC_TARGET 1536u		/* average build keys per leaf (load factor ~0.4) */
#define PJ_SLOTS 2039u		/* pairs table (prime) */
#define PJ_TARGET 640u
#define PJ_CHUNK 2048u		/* right-side rows staged per sweep in the emit kernel */

__device__ static inline uint32_t leaf_slot(uint64_t hv, uint32_t slots)
{
	const uint32_t x = (uint32_t)hv * 0x9E3779B1u;
	return (uint32_t)(((uint64_t)x * slots) >> 32);
}

/* Double hashing: the probe step comes from the other half of the hashed key, in [1, slots - 1]; the table
 * sizes are prime, so every step visits all slots.  Linear probing clusters: at load 0.4 the longest of the
 * 64 probe chains a wave waits for was ~2x longer, and the wave pays the longest. */
__device__ static inline uint32_t leaf_step(uint64_t hv, uint32_t slots)
{
	const uint32_t y = (uint32_t)(hv >> 32) * 0x85EBCA6Bu;
	return 1u + (uint32_t)(((uint64_t)y * (slots - 1)) >> 32);
}

/* insert-or-find hv (hv != 0); returns the slot or 0xFFFFFFFF when the table is full; *created = this call
 * claimed the slot (exactly one caller per distinct key sees true) */
__device__ static inline uint32_t leaf_insert(unsigned long long *keys, uint32_t slots, uint64_t hv, bool *created = nullptr)
{
	uint32_t s = leaf_slot(hv, slots);
	const uint32_t step = leaf_step(hv, slots);
	for (uint32_t probe = 0; probe < slots; probe++) {
		const unsigned long long old = atomicCAS(&keys[s], 0ull, (unsigned long long)hv);
		if (old == 0ull || old == hv) {
			if (created)
				*created = old == 0ull;
			return s;
		}
		s += step;
		if (s >= slots)
			s -= slots;
	}
	return 0xFFFFFFFFu;
}

/* find hv (hv != 0) after the build phase; 0xFFFFFFFF = absent */
__device__ static inline uint32_t leaf_find(const unsigned long long *keys, uint32_t slots, uint64_t hv)
{
	uint32_t s = leaf_slot(hv, slots);
	const uint32_t step = leaf_step(hv, slots);
	for (uint32_t probe = 0; probe < slots; probe++) {
		const unsigned long long cur = keys[s];
		if (cur == hv)
			return s;
		if (cur == 0ull)
			return 0xFFFFFFFFu;
		s += step;
		if (s >= slots)
			s -= slots;
	}
	return 0xFFFFFFFFu;
}

#define GC_MAX_EXTRA 2		/* right tables beyond the first one (mdb_dev_join_group_count_multi: up to 3 right tables) */

struct gc_args {
	const uint64_t *hv_l;
	const uint32_t *rid_l;
	const uint32_t *off_l;		/* exact leaf offsets, or ... */
	const uint32_t *cnt_l;		/* ... rows per leaf of the fixed-capacity (fast) layout */
	uint32_t cap_l;			/* 0 = exact offsets */
	const uint64_t *hv_r;		/* NULL: plain GROUP BY over the left stream */
	const uint32_t *off_r;
	const uint32_t *cnt_r;
	uint32_t cap_r;
	int64_t *dense_cnt;		/* dense mode: [n_l], zeroed: COUNT(*) written at the group's first L position */
	uint32_t dense_n;		/* ... = n_l: a first row id is checked against it before it indexes dense_cnt[] - a fixed-capacity region that
					 * overflowed (status bit 1: the operator is redone) holds slots nobody wrote, with whatever an earlier call left */
	unsigned long long *rec;	/* record mode: one 64-bit record per group, (first << (64 - kbits)) | COUNT(*) */
	uint32_t *rec_count;		/* record mode: list slots handed out so far (the list has zero-filled gaps) */
	uint32_t *rec_valid;		/* record mode: number of real records (= groups) */
	uint32_t rec_cap;		/* record mode: capacity of the list */
	uint32_t kbits;			/* record mode: bits of a left row id (0 = dense mode) */
	unsigned long long *joined;	/* sum of all counts */
	uint32_t *status;		/* bit 0: a leaf table overflowed */
	uint32_t nleaves;
	uint32_t heavy_l, heavy_r;	/* rows of a side from which a leaf counts as hot (>= GC_HEAVY and >= 8x the side's average leaf) */
	uint32_t narrow;		/* narrow form: hv_l[i] = hash32 << 32 | row id (rid_l unused), hv_r = array of 4-byte hash32 */
	uint32_t rec32;			/* direct-address leaves: the records are written as 4-byte words (first row id << (32 - kbits)) | COUNT(*) - the
					 * caller has seen, for these very columns, that every COUNT fits; one that does not raises status bit 9 and
					 * the operator is redone with 8-byte records */
	uint32_t keyed_cbits;		/* direct-address leaves, != 0: KEYED group records - (first row id, hashed key, COUNT(*)) with COUNT in
					 * the low keyed_cbits bits and the key_bits-wide hashed key above it: the ordering kernel decodes the
					 * group key from the record instead of gathering it from the key column (selective joins: the groups'
					 * first rows are scattered over the left table, every gathered key costs a 128-byte line) */
	uint32_t merge_all;		/* plain GROUP BY: the key sample held duplicates (some 10^4 - 10^5 distinct values): merge equal
					 * values per wave in every leaf, not only in the oversize ones */
	/* further right tables joined on the SAME key (A JOIN B ON a = b JOIN C ON a = c ... GROUP BY a: BASELINE configs[4]) -
	 * direct-address leaves only: partitioned exactly like the right table (4-byte words, fixed-capacity leaves), counted into
	 * LDS arrays of their own; a key's right count becomes the PRODUCT of its counts in all right tables */
	/* k_leaf_wide4, rg_rec != NULL: the group records are written straight into the ordering kernel's ranges of 2^rg_shift first row ids -
	 * region r of rg_cap records at rg_rec + r * rg_cap, its fill in rg_cnt[r] (zeroed by the caller) - instead of one list that two scatter
	 * levels then partition by row id; a range that outgrows its region raises flag 8192 */
	unsigned long long *rg_rec;
	uint32_t *rg_cnt;
	uint32_t rg_cap, rg_shift, rg_n;
	/* k_leaf_wide12, dn_bits != NULL (nearly every left row a group of COUNT 1 - remembered from the last call over these columns): no
	 * record per group.  One bit per LEFT row, set by the caller, cleared here for every row that is no group's first row (no partner, or
	 * not the first of its key); groups whose COUNT is not 1 are appended to dn_exc as first row << 32 | COUNT (mdb_dev_dense.hip expands) */
	unsigned int *dn_bits;
	unsigned long long *dn_exc;
	uint32_t dn_exc_cap;
	uint32_t *dn_cnt;		/* [0] bits cleared, [1] exceptions */
	uint32_t dn_pilot;		/* != 0 (dn_bits = dn_exc = NULL): the pilot - the first dn_pilot digits only, nothing but the counters is written */
	uint32_t nextra;
	const uint32_t *hv_x[GC_MAX_EXTRA];
	const uint32_t *cnt_x[GC_MAX_EXTRA];
	uint32_t cap_x[GC_MAX_EXTRA];
};

/* narrow word -> the 64-bit value the leaf tables work with (both halves = the 32-bit hash, so that the slot and the
 * probe step still come from different multipliers; 0 only for the key whose hash is 0) */
__device__ static inline uint64_t gc_narrow_hv(uint64_t w)
{
	const uint32_t h = (uint32_t)(w >> 32);
	return ((uint64_t)h << 32) | h;
}

__device__ static inline void gc_leaf_range(const uint32_t *off, const uint32_t *cnt, uint32_t cap, uint32_t leaf, uint32_t *b,
					    uint32_t *e)
{
	if (cap) {
		const uint32_t c = cnt[leaf];
		*b = leaf * cap;
		*e = *b + (c < cap ? c : cap);
	} else {
		*b = off[leaf];
		*e = off[leaf + 1];
	}
}

#define LW_THREADS 1024
#define LW_MIN_REM 6u		/* tables of 64 entries at least (key windows from 2^15 values: that few values per first-level region - their
				 * number varies by 13 % - need the looser regions of mdb_part_filter.loose) */
#define LW_EMIT_REM 11u		/* from here on every thread owns at least one 32-bit word of halves in the emit pass */
#define LW_MAX_REM 14u
#define LW_UNROLL 4

__device__ static inline unsigned long long lw_block_sum(unsigned long long v, unsigned long long *s_red)
{
#pragma unroll
	for (int o = 32; o; o >>= 1)
		v += __shfl_down(v, o, MDB_WAVE);
	__syncthreads();	/* protect s_red against a previous use */
	if (mdb_lane() == 0)
		s_red[threadIdx.x >> 6] = v;
	__syncthreads();
	unsigned long long t = 0;
#pragma unroll
	for (int w = 0; w < LW_THREADS / 64; w++)
		t += s_red[w];
	return t;
}

/* internal return codes of the operator's attempts (never leave the library): ONE enum, so that no two can share a value
 * (round 5 gave GC_RETRY_NODENSE the value of GC_NOT_SERVED and an unserved three-table operator was retried six times) */
enum gc_internal_rc {
	GC_RETRY_EXACT = 1000,	/* internal: a fast-layout leaf overflowed, redo with exact histograms */
	GC_RETRY_DENSE,	/* internal: a COUNT(*) does not fit a group record, redo with the dense ordering */
	GC_RETRY_BUILD_L,	/* internal: the right side's distinct keys overflowed a leaf table, redo building on the left side */
	GC_RETRY_WIDE,	/* internal: a key outside the int32 range met the narrow form, redo with 64-bit hashes */
	GC_RETRY_PLAIN,	/* internal: a key outside the compact window (the sample missed the column's extremes): redo in the plain narrow form */
	GC_RETRY_UNKEYED,	/* internal: a COUNT(*) does not fit a keyed group record (ctx->keyed_distrust is set): redo with plain records */
	GC_RETRY_REC64,	/* internal: 4-byte group records were written on a remembered verdict that no longer holds: redo with 8-byte ones */
	GC_RETRY_TWO_LEVEL,	/* internal: a 16-bit row count of k_leaf_wide overflowed (ctx->lw_bad_* remember the columns): redo with two levels */
	GC_RETRY_NODENSE,	/* internal: the bit-per-row form of the groups met more groups of COUNT != 1 than its list holds (ctx->dn_distrust is set) */
	GC_EXPLAINED,	/* internal: mdb_dev_explain_*() - the plan is written, nothing was launched */
	GC_NOT_SERVED,	/* internal: further right tables, but the operator did not take the two-level direct-address form (or a product of counts overflowed, or a hot leaf): the caller chains two-table operators instead */
};
#define GC_ST_LEFT_DUPS 32768u	/* status bit 15: a key that has partners has several rows in the LEFT table (raised by the direct-address leaf kernels) */
/* words of ctx->d_status the fused operator uses beyond [0..9] (flags, record-list length, joined rows, NULL-group stats, records):
 * [10..21] the key sample's six 8-byte extremes (before the operator starts), [16..17] the right table's smallest / largest
 * key - window base (min-max pruning, while it runs) */
#define GC_ST_MINMAX 16
#define GC_ST_MINMAX64 40	/* [40..43] the right table's smallest / largest key as two signed 64-bit words (min-max pruning, 64-bit form) */
#define GC_ST_WINDOW 20	/* [20..21] 0 and 2^key_bits - 1: the whole window as a pruning range (further right tables drop what lies outside) */

struct gc_window {
	uint32_t kbits;		/* 0 = no compact window */
	int64_t lo;
	bool selective;		/* the right table's sampled keys cover less than a quarter of the left table's sampled key range, or the right
				 * table has less than a quarter of the left table's rows: most left rows will find no partner (semi-join filter) */
	bool by_span;		/* ... the former: min-max pruning at the first level will drop them, no bitmap needed */
	bool prunable;		/* the right table's sampled keys cover less than 7/8 of the left table's sampled range: worth recording the
				 * right table's exact range for min-max pruning */
	bool r_based;		/* the compact window covers the RIGHT table's sampled keys only (by_span, unsplit call): the left rows outside
				 * it are exactly the ones min-max pruning drops - fewer key bits, hence fewer and larger leaves */
	bool fast1;		/* plain GROUP BY, duplicates in the key sample: the fixed-capacity layout only if the ONE-level form applies */
};

#define TINY_ROWS (GC_THREADS * LEAF_BATCH)	/* rows per table up to which ONE workgroup does the whole operator in LDS */
#define GC_NARROW_MIN_ROWS (1u << 20)
#define GC_NARROW_SAMPLE 4096u
#define GC_HINT_USES 32		/* a remembered sample / verdict serves this many calls, then the data is looked at again (one tiny
				 * kernel + sync - ~50 us, a tenth of a 10^7-row query - in 32 calls; a buffer refilled with another
				 * distribution runs that long on a plan that is exact but may not be the best; a hint the data
				 * contradicts is noticed by the kernels at once) */

/* position of sample t: pseudo-random, not evenly spaced - generated or periodic data (an affine sequence, a table sorted by
 * a low-cardinality column) looks very different at a fixed stride than it is */
__device__ static inline uint64_t gc_sample_pos(uint32_t t, uint64_t n)
{
	return mdb_fmix64(0x9E3779B97F4A7C15ull * (uint64_t)(t + 1)) % n;
}

/* ---- host functions that cross the files (definitions: see the list at the top) */
uint64_t gc_rec_capacity(mdb_dev_ctx *ctx, uint64_t n_l);
bool order_bits(uint64_t n_l, uint32_t *kbits, int *sb1, int *sb2);
/* ranges of 2^ORDER_RANGE_BITS first row ids, ORDER_RANGE_CAP records each, as k_order_leaf_sparse ranks them: whether records of a table of
 * n_l rows expected to number `groups` may be written into them by the leaf kernel itself, and the ordering of such ranges */
#define ORDER_RANGE_BITS 16u
#define ORDER_RANGE_CAP 8192u
bool order_ranges_apply(uint64_t n_l, uint32_t kbits, uint64_t groups, uint32_t *nranges);
int order_presorted(mdb_dev_ctx *ctx, const unsigned long long *regions, const uint32_t *counts, uint32_t nranges, uint32_t kbits, uint32_t *out_first,
		    int64_t *out_count, const int64_t *keys, int64_t *out_key, bool keys32, uint32_t keyed_cbits, uint32_t key_bits, int64_t key_lo,
		    uint64_t early_cap = 0);
uint32_t order_digits0(uint64_t n_l, uint32_t kbits, int sb1);
int order_records(mdb_dev_ctx *ctx, const unsigned long long *rec, uint64_t list_len, uint64_t n_l, uint32_t kbits, int sb1,
			 int sb2, uint32_t *out_first, int64_t *out_count, uint32_t *out_val32, const int64_t *keys, int64_t *out_key,
			 bool keys32 = false, bool rec32 = false, uint32_t keyed_cbits = 0, uint32_t key_bits = 0, int64_t key_lo = 0,
			 bool in32 = false /* the list already holds 4-byte records */, uint64_t n_rec = 0 /* records in the list (0: unknown) */);
size_t order_records_arena_bytes(uint64_t cap, uint64_t n_l, uint32_t kbits, int sb1, int sb2);
int gc_sample_range(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			   const uint64_t *null_r, uint64_t n_r, bool fresh, int64_t *lo, int64_t *hi, bool keys32 = false);
void gc_narrow_note(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const int64_t *keys_r, uint64_t n_r, bool narrow,
			   int64_t base = 0, uint32_t key_bits = 0, int64_t key_lo = 0);
int gc_narrow_guess(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			   const uint64_t *null_r, uint64_t n_r, bool *narrow, int64_t *base, gc_window *win = nullptr, bool keys32 = false,
			   bool prune_ok = false /* unsplit call: min-max pruning can run */);
int tiny_group_count(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			    const uint64_t *null_r, uint64_t n_r, bool has_r, bool null_group, int64_t *out_key, int64_t *out_count,
			    uint32_t *out_first, uint64_t cap, uint64_t *out_groups, uint64_t *out_joined);
int group_hashed_try(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, bool null_group, uint32_t *out_first,
			    int64_t *out_count, uint64_t cap, uint64_t *out_groups);
int group_direct_try(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, const int64_t *keys_r,
			    const uint64_t *null_r, uint64_t n_r, bool null_group, int64_t *out_key, uint32_t *out_first, int64_t *out_count,
			    uint64_t cap, uint64_t *out_groups, uint64_t *out_joined);
bool ld_disabled(void);
/* mdb_dev_leaf_wide.hip: one workgroup per first-level digit */
int leaf_wide_launch(mdb_dev_ctx *ctx, const gc_args &a, uint32_t nleaves, uint32_t rem, uint32_t shift, uint32_t nsub, bool has_r, bool r16);
int leaf_wide4_launch(mdb_dev_ctx *ctx, const gc_args &a, uint32_t nleaves, uint32_t rem, uint32_t shift, uint32_t nsub);
bool leaf_wide12_fits(const mdb_dev_ctx *ctx, uint64_t n_l, uint64_t n_r, uint64_t n_x);
int leaf_wide12_launch(mdb_dev_ctx *ctx, const gc_args &a, uint32_t nleaves, uint32_t rem, uint32_t nsub, int nextra);

#endif /* MDB_DEV_JOIN_INTERNAL_H */
