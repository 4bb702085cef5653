/*
 * mdb_dev_internal.h - host-side entry points shared between the device-layer translation units.
 */
#ifndef MDB_DEV_INTERNAL_H
#define MDB_DEV_INTERNAL_H

#include "mdb_dev_common.h"

/* ---- scan (mdb_dev_core.hip) ----------------------------------------------------------------
 * In-place exclusive prefix sum over data[0..len) (uint32).  data[len-1]'s exclusive prefix is the
 * last value written; callers that need the grand total append one zero element.  `block_sums`
 * is scratch of mdb_scan_scratch_words(len) uint32 words. */
#define MDB_SCAN_CHUNK 4096u
#define MDB_SCAN_SMALL 16384u	/* up to here the scan is one single-workgroup launch */
static inline size_t mdb_scan_scratch_words(uint64_t len) { return (size_t)((len + MDB_SCAN_CHUNK - 1) / MDB_SCAN_CHUNK) + 1; }
int mdb_scan_u32_inplace(mdb_dev_ctx *ctx, uint32_t *data, uint64_t len, uint32_t *block_sums);
/* dst[0..n] = exclusive prefix sums of src[0..n), dst[n] = total; one single-workgroup launch (n <= MDB_SCAN_FROM_MAX) */
#define MDB_SCAN_FROM_MAX 65536u
int mdb_scan_u32_small_from(mdb_dev_ctx *ctx, const uint32_t *src, uint32_t n, uint32_t *dst);

/* ---- radix partition (mdb_dev_partition.hip) ------------------------------------------------ */
#define MDB_TILE 4096u		/* elements per partition tile */
#define MDB_MAX_RADIX_BITS 9	/* up to 512-way per level */

/* one tile of a partition level's input (mdb_dev_partition.hip) */
struct mdb_tile_desc {
	uint32_t start;		/* first input element of the tile */
	uint32_t len;		/* elements in the tile (0 = unused tile) */
	uint32_t hbase;		/* histogram index of (segment, digit 0, this tile) */
	uint32_t nt;		/* tiles in this tile's segment = histogram stride between digits */
	uint32_t seg;		/* index of the segment (parent partition) the tile belongs to */
};

enum mdb_digit_mode {
	MDB_DIGIT_RADIX = 0,	/* digit = bit field of the hashed key (MSD levels) */
	MDB_DIGIT_MOD = 1,	/* digit = low32(hash) mod n_dest (multi-GPU destination) */
};

struct mdb_part_result {
	uint64_t *hv;		/* hashed keys grouped by leaf (or original keys when inverse_out) */
	uint32_t *rid;		/* row ids, same order (NULL when not requested) */
	uint32_t *leaf_off;	/* exact form: nleaves + 1 offsets into hv/rid (NULL in the fast form) */
	uint32_t *leaf_cnt;	/* fast form: rows of each leaf, leaf i lives at [i * leaf_cap, i * leaf_cap + leaf_cnt[i]) */
	uint32_t leaf_cap;	/* 0 = exact form */
	uint32_t nleaves;
	uint32_t bits_total;
	bool w32;		/* hv holds 4-byte words (narrow form without row ids) */
	bool w16;		/* ... 2-byte words: the hash bits below the first level's digit (mdb_part_filter.out16) */
	uint64_t *pay[2];	/* mdb_part_filter.npay: the payload cells, laid out like hv (the cell of hv[i] is pay[c][i]) */
	uint32_t nsub;		/* != 0: first level only (mdb_part_filter.level0_only) - nleaves = 2^bits1 digits, digit d's rows lie in nsub
				 * regions: region r = d * nsub + s at [r * leaf_cap, r * leaf_cap + min(count, leaf_cap)), its count at
				 * leaf_cnt[s * nleaves + d] */
};

/* bytes of arena needed by mdb_partition_table() */
size_t mdb_partition_arena_bytes(uint64_t n, int bits1, int bits2, bool want_rid, bool fast, int npay = 0 /* mdb_part_filter.npay */);

/* Partition one key column into 2^(bits1+bits2) leaves by the top bits of fmix64(key), dropping
 * NULL keys.  stable = keep input order inside every leaf (slower ballot ranking; requires want_rid),
 * otherwise the order inside a leaf is unspecified.  fast = skip the second level's histogram pass: every
 * leaf gets a fixed-capacity region and runs are placed with atomic cursors; if a leaf overflows, bit 1 of
 * ctx->d_status[0] is set and the caller must redo the operator with fast = false.  All temporaries and
 * outputs are carved from the arena (caller has called mdb_arena_begin with enough room).  No host sync.
 * narrow (want_rid must be false): every key is expected within 2^31 of narrow_base - a key outside raises bit 7 of
 * ctx->d_status[0] and the caller redoes the operator wide.  1: hv[i] = fmix32(key) << 32 | row id, 2: hv is an array of 4-byte words fmix32(key) - only in the
 * two-level fast layout, see mdb_partition_w32_applies(). */
/* Semi-join filter of the second partition level: a bitmap of the hashed key values the OTHER table holds, one bit per
 * 2^coarse adjacent values, laid out by first-level digit: digit d owns words [d * words, d * words + words).  The
 * second level's workgroup loads its digit's slice into LDS (into the staging buffer, before it is needed) and drops the
 * rows whose bit is clear before they are ranked and written. */
struct mdb_part_filter {
	const uint32_t *bits;	/* NULL: no bitmap */
	uint32_t words;		/* per first-level digit: a power of two, 4 ... 8192 (32 KiB) */
	uint32_t shift;		/* bit index (before masking to the slice) = hash32 >> shift */
	/* min-max pruning at the FIRST level (compact narrow form): the right table's call passes minmax_out (two words the caller
	 * initialised to 0xFFFFFFFF, 0: smallest / largest key - window base seen), the left table's call, later on the same
	 * stream, passes the same words as range_in and drops the rows outside */
	uint32_t *minmax_out;
	uint32_t *minmax_tiles;	/* with minmax_out: scratch of mdb_part_minmax_words(n) words (a pair per first-level tile, reduced after the launch) */
	const uint32_t *range_in;
	/* the same in the 64-bit form (mdb_partition_table with narrow = 0): the right table (no row ids) leaves [lo, hi] of its keys as
	 * two signed 64-bit words in minmax64_out (scratch: minmax64_tiles, mdb_part_minmax_words(n) * 2 uint32 words), the left table
	 * (with row ids) reads them through range64_in */
	long long *minmax64_out;
	unsigned long long *minmax64_tiles;
	const long long *range64_in;
	/* narrow = 1 (hash | row id words), histogram-free layout, one level (level0_only) or two: up to two 8-byte payload columns of the
	 * table travel with the rows (mdb_part_result.pay) - a join that carries the right table's payload to the leaf instead of
	 * gathering it afterwards */
	const void *pay_in[2];
	int npay;
	/* first level, any form: keep only the rows whose KEY lies in [keep_lo, keep_hi] (keep_on) - the other table's global key
	 * range, known before an exchange (mdb_dev_partition_by_dest_pruned) */
	bool keep_on;
	int64_t keep_lo, keep_hi;
	bool own_on;		/* ... and verify that every key of THIS table lies in [own_lo, own_hi] (a promised range: status bit 10 otherwise) */
	int64_t own_lo, own_hi;
	bool loose;		/* with level0_only: regions of 1.5 x (instead of 1.25 x) the average size - a column of some 10^4 - 10^5 distinct
				 * values puts a few dozen of them, with thousands of rows each, into a region */
	bool out16;		/* with level0_only, right side of the compact narrow form: 2-byte words (mdb_part_result.w16) when the hash bits
				 * below the digit fit 16 bits */
	bool level0_only;	/* stop after the histogram-free first level (bits2 = 0): the consumer reads the digits' sub-regions itself
				 * (mdb_part_result.nsub; k_leaf_wide in mdb_dev_join.hip) */
	uint32_t region_cap;	/* with level0_only, != 0: words per first-level region (a multiple of 64) instead of 1.25 x the average + 1024 */
	uint32_t *cursor0_ext;	/* != NULL: the first level's region cursors live here (cursor0_ext_words of them at least), ALREADY ZERO - the caller's
				 * one memset covers them (ctx->d_status + MDB_ZERO_BLK_OFF) - instead of an arena block and a memset of its own */
	uint32_t cursor0_ext_words;
	bool expect_pruned;	/* with range_in: the caller expects most rows to be dropped (key sample): the second level's grid is then
				 * sized by the tiles that exist (a 4-byte read-back + synchronisation) instead of by the table */
};
static inline size_t mdb_part_minmax_words(uint64_t n) { return (size_t)((n + MDB_TILE - 1) / MDB_TILE + 8) * 2; }

int mdb_partition_table(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n,
			int bits1, int bits2, bool want_rid, bool stable, bool fast, mdb_part_result *out, int narrow = 0,
			bool keys32 = false,	/* keys32: `keys` points to int32 values */
			int64_t narrow_base = 0,	/* narrow: the keys are taken relative to this value (centre of their 2^32 window) */
			uint32_t narrow_kbits = 0,	/* compact narrow form: keys in [narrow_base, narrow_base + 2^narrow_kbits), hash32 =
							 * mdb_mixk(key - narrow_base, narrow_kbits) << (32 - narrow_kbits): below the
							 * bits1 + bits2 partition bits only narrow_kbits - bits1 - bits2 bits tell keys apart */
			const struct mdb_part_filter *flt = NULL);	/* semi-join filter at the second level (left side of a join in the
									 * compact narrow form only), see struct mdb_part_filter */

/* arena bytes of a first-level-only partition (mdb_part_filter.level0_only) of n rows by bits1 bits */
size_t mdb_partition_level0_arena_bytes(uint64_t n, int bits1, bool loose = false /* mdb_part_filter.loose */, int npay = 0 /* mdb_part_filter.npay */);

/* whether narrow = 2 is available for a table of n rows */
bool mdb_partition_w32_applies(uint64_t n, int bits1, int bits2, bool fast);

/* fast = fixed-capacity regions + cursors (a region overflow sets bit 1 of ctx->d_status[0]: the caller must check it
 * after the consumer kernel and redo with fast = false, the exact histogram layout) */
/* digits0_used: how many of the 2^bits1 first-level digits can occur at all (0 = every one) - sizes the fast regions */
/* digit0_rows: the words are row ids and a first-level digit spans that many of them (0: not so) - sizes the fast first-level regions for
 * the most a digit can hold instead of the average */
size_t mdb_partition_raw_arena_bytes(uint64_t n, int bits1, int bits2, uint32_t leaf_cap, bool fast, uint32_t digits0_used, uint64_t digit0_rows = 0);
int mdb_partition_raw(mdb_dev_ctx *ctx, const uint64_t *hv, uint64_t n, int bits1, int bits2, uint32_t leaf_cap, bool fast,
		      uint32_t digits0_used, mdb_part_result *out, bool zero_is_gap = true,	/* zero words are gaps of a chunked list */
		      int fold32 = 0,	/* 1: the 8-byte records are folded into 4-byte words by the first level; 2: `hv` already holds 4-byte words */	/* the first level folds every 8-byte word w into the 4-byte word (w >> 32) | (uint32_t)w - the caller knows
					 * that the two parts share no bit - and everything after it moves 4-byte words (out->w32; two fast levels only) */
		      uint64_t digit0_rows = 0);

/* one stable least-significant-digit radix pass over (key, row id) pairs: digit = (key >> shift) & (2^bits - 1),
 * bits <= 8.  hist = scratch of mdb_sort_pass_hist_words(n) uint32, scan_tmp = mdb_scan_scratch_words(of that). */
size_t mdb_sort_pass_hist_words(uint64_t n);
int mdb_sort_pass(mdb_dev_ctx *ctx, const uint64_t *key_in, const uint32_t *rid_in, uint64_t n, uint32_t shift, uint32_t bits,
		  uint64_t *key_out, uint32_t *rid_out, uint32_t *hist, uint32_t *scan_tmp);

/* One histogram-free radix level over 4-byte words that lie in caller-described tiles: word w of tile t belongs to segment
 * tiles[t].seg and goes to child seg * 2^bits + ((w >> shift) & (2^bits - 1)); child c owns words_out[c * cap, c * cap + cap),
 * cursor[c] (zeroed by the call) counts its words; bit 1 of ctx->d_status[0] is raised when a child overflows.  Tile starts
 * must be multiples of 4 words.  No host sync. */
int mdb_partition_words_level(mdb_dev_ctx *ctx, const uint32_t *words_in, const mdb_tile_desc *tiles, uint32_t ntiles, int bits, uint32_t shift,
			      uint32_t *words_out, uint32_t *cursor, uint32_t nchild, uint32_t cap,
			      uint32_t out16_shift = 0 /* != 0: the children receive 2-byte words (uint16_t)(word >> out16_shift); cap counts them */);

/* ---- the sharded join + GROUP BY with first-level regions on the wire (mdb_dev_shard.hip) ------------------------------
 *
 * Every rank partitions ITS rows of a table ONCE with the histogram-free first level of the compact narrow form (512 digits,
 * 8 per-XCD sub-regions each, fixed capacity, 4- or 2-byte words = the k-bit hash of key - window base); the digit's top
 * bits are the destination rank, so the regions of one destination are one contiguous block of a size every rank knows
 * beforehand: the blocks ARE the all-to-all, no counts have to reach a host before it is posted.  The receiver joins the
 * regions it got from all ranks - straight from them when the hash bits below the digit index an LDS table (one level), or
 * after one more partition level of its own - and emits (key, COUNT) pairs; no row ids travel and no result ordering runs
 * (across ranks SQL leaves the order open anyway). */
#define MDB_SHARD_MAX_TABS 4
struct mdb_shard_plan {
	uint32_t world, rank;
	uint32_t dbits;			/* first-level digit bits: 9 (the join's own first level), or 12 (the wide fan-out form) */
	uint32_t D, Dp, nsub;		/* first-level digits, digits per destination, sub-regions per digit */
	uint32_t kbits;			/* key window [key_lo, key_lo + 2^kbits) */
	int64_t key_lo;
	uint32_t ntab;			/* tables: [0] the left one, [1] the right one, [2..] further right tables joined on the same key */
	uint32_t cap[MDB_SHARD_MAX_TABS];	/* words per first-level region of each table */
	uint32_t wbytes;		/* bytes per word on the wire: 4, or 2 when the hash bits below the digit fit */
	int b2;				/* receiver: bits of its own partition level (0: the regions are joined as they are) */
	uint32_t rem;			/* key bits that index the leaf tables */
	uint64_t block_words[MDB_SHARD_MAX_TABS];	/* words per destination block = Dp * nsub * cap */
	uint32_t leaf_cap[MDB_SHARD_MAX_TABS];		/* b2 > 0: words per leaf region of the receiver's level */
	bool right_only;		/* GROUP BY of ONE table (it plays the right table; the left one has no rows): a slot's group is its first row, COUNT = its rows */
	uint64_t l_rel_hi;		/* the left table keeps the rows with key - key_lo in [0, l_rel_hi] (= the right table's range) */
};
/* 0 = plan made; 1 = this shape is not served (window too wide, world not a power of two ...): the caller takes another path */
int mdb_shard_plan_make(uint32_t world, uint32_t rank, uint32_t ntab, const uint64_t *n_max /* [ntab]: the largest shard of each table */,
			int64_t l_lo, int64_t l_hi, int64_t r_lo, int64_t r_hi, mdb_shard_plan *plan);
size_t mdb_shard_arena_bytes(const mdb_shard_plan *plan);
/* sender: table 0 = left (rows outside the right table's range are dropped), 1 = right (a key outside the window is reported),
 * 2.. = further right tables (rows outside the window are dropped: they join nothing); *regions = the region buffer
 * (world * block_words[side] words of wbytes bytes, destination-major), *cursors = D * nsub region counters (sub-major).
 * The caller has begun the arena and cleared ctx->d_status[0..15].  No host sync. */
int mdb_shard_partition(mdb_dev_ctx *ctx, const mdb_shard_plan *plan, int side, const int64_t *keys, const uint64_t *nulls, uint64_t n,
			const void **regions, const uint32_t **cursors);
/* receiver: recv[x] = world blocks of block_words[x] words (source-major), cnt[x] = world cursor arrays of D * nsub counters;
 * out_key / out_count (capacity cap): the groups; d_status[1] = their number, d_status[2..3] = joined rows (u64), flags in
 * d_status[0] (bit 1 a region overflowed, bit 3 cap too small, bit 7 a right key outside the window).  No host sync. */
int mdb_shard_join(mdb_dev_ctx *ctx, const mdb_shard_plan *plan, const void *const *recv, const uint32_t *const *cnt, int64_t *out_key,
		   int64_t *out_count, uint64_t cap, void *const *arrived = NULL /* [ntab] hipEvent_t or NULL: table x's blocks and counters
		   have arrived when arrived[x] has happened - the context's stream waits for it right before the first kernel that reads
		   table x, so the receiver's level over one table runs while the next table is still on the wire */);
/* one 4096-digit pass over a key column of the compact narrow form (mdb_dev_shard.hip): 2-byte words, or 4-byte row words with run headers */
uint32_t mdb_scatter4096_cap(const mdb_dev_ctx *ctx, uint64_t n, bool row_words);
size_t mdb_scatter4096_arena_bytes(const mdb_dev_ctx *ctx, uint64_t n, bool row_words);
int mdb_scatter4096(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nulls, uint64_t n, int64_t key_lo, uint32_t kbits, bool report, uint32_t rel_hi,
		    uint32_t cap, bool row_words, const char *name, void **regions, uint32_t **cursors);

/* ---- ordering of (row id, payload) records (mdb_dev_join.hip) ----------------------------------
 * rec[i] = (row id << (64 - kbits)) | payload (payload >= 1; zero words are gaps), kbits = bits of a row id as
 * returned by mdb_order_records_arena_bytes(); delivers out_first[] = row ids ascending, out_count[] = payloads. */
size_t mdb_order_records_arena_bytes(uint64_t cap, uint64_t n_rows, uint32_t *kbits_out);
int mdb_order_records_by_rowid(mdb_dev_ctx *ctx, const unsigned long long *rec, uint64_t list_len, uint64_t n_rows, uint32_t kbits,
			       uint32_t *out_first, int64_t *out_count);

/* (a, b) pairs of 32-bit ids (a < na, b < nb, unique as pairs; at least 2^18 of them) into ascending (a, b) order
 * (mdb_dev_sort.hip).  0 = done, 1 = not applicable (few pairs, or a's too unevenly spread): the caller sorts another way. */
int mdb_sort_pairs(mdb_dev_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t n, uint64_t na, uint64_t nb, uint32_t *out_a,
		   uint32_t *out_b);

/* (key, COUNT) groups -> every key written COUNT times (mdb_dev_pairs.hip): *out = a device array of `joined` keys allocated with
 * mdb_dev_alloc.  joined = the sum of the counts (< 2^32).  Synchronises. */
int mdb_expand_keys_by_count(mdb_dev_ctx *ctx, const int64_t *key, const int64_t *count, uint64_t groups, uint64_t joined, int64_t **out);

/* group i = row i, COUNT 1 (mdb_dev_dense.hip): a GROUP BY over a column whose statistics say MDB_COL_DISTINCT */
int mdb_group_identity(mdb_dev_ctx *ctx, uint64_t n, uint32_t *out_first, int64_t *out_count);

/* choose level bits so that the average leaf holds about `target` keys */
void mdb_choose_bits(uint64_t n, uint32_t target, int *bits1, int *bits2);

#endif /* MDB_DEV_INTERNAL_H */
