/*
 * mdb_dev_rowjoin.h - the row-order payload join (mdb_dev_rowjoin.hip) as mdb_dev_join_payload (mdb_dev_pairs.hip) calls it.
 */
#ifndef MDB_DEV_ROWJOIN_H
#define MDB_DEV_ROWJOIN_H

#include "mdb_dev_internal.h"

/* whether the form serves a join of n_l x n_r rows over a compact key window of 2^kbits values: windows of up to 2^27 values, key and
 * payload columns without NULL keys and 16-byte aligned */
bool mdb_rowjoin_serves(uint64_t n_l, uint64_t n_r, uint32_t kbits, const void *keys_l, const void *null_l, const void *keys_r, const void *null_r,
			const void *const *pay_in, void *const *out, int npay);
uint32_t mdb_rowjoin_dbits(uint32_t kbits);
size_t mdb_rowjoin_arena_bytes(uint64_t n_l, uint64_t n_r, uint32_t kbits, int npay);
/* queues everything on ctx->stream (the arena begun, ctx->d_status cleared by the caller); no host sync.  Afterwards d_status[0] holds
 * the flags (4 a left row without partner, 32 duplicate right key, 128 key outside the window), d_status[2..3] the joined rows (u64) */
int mdb_rowjoin_run(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const int64_t *keys_r, uint64_t n_r, const void *const *pay_in,
		    int64_t win_lo, uint32_t kbits, int npay, void *const *out);

/* ... of ONE left key column with several right tables on that key (mdb_dev_join_payload_multi): the left table is sorted once, one leaf
 * launch and one placement pass serve every (right table, payload column) pair - at most four.  d_status[2..3] then holds the (left row,
 * payload column) pairs served: the columns x the left rows when every left row found its partner in every table */
struct mdb_rowjoin_right {
	const int64_t *keys;
	uint64_t n;
	int npay;		/* 1 or 2 */
	const void *pay_in[2];
	void *out[2];
};
size_t mdb_rowjoin_arena_bytes_multi(uint64_t n_l, const struct mdb_rowjoin_right *rt, int nrt, uint32_t kbits);
int mdb_rowjoin_run_multi(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const struct mdb_rowjoin_right *rt, int nrt, int64_t win_lo, uint32_t kbits);


/* GROUP BY key + COUNT(*) of one NULL-free key column through the tile sort: 0 = done (groups in first-row order), 1 = not served
 * (*outside: a key lay outside the window), < 0 = error.  Synchronises. */
int mdb_group_count_tiled(mdb_dev_ctx *ctx, const int64_t *keys, uint64_t n, int64_t win_lo, uint32_t kbits, uint32_t *out_first, int64_t *out_count,
			  uint64_t cap, uint64_t *out_groups, bool *outside);

/* a key that is the composite value of up to four columns (GROUP BY k2, k3: mdb_dev_sort.hip, group_multi_packed) - most significant first:
 * field c = [NULL flag, where the column has a NULL bitmap | image of the value - lo[c]], kb[c] value bits; an image outside
 * [lo[c], lo[c] + span[c]] is a key outside the window.  The band sort reads the columns itself: no composite column is written. */
#define MDB_BG_COMP_MAX 4
struct mdb_bg_comp {
	const uint64_t *values[MDB_BG_COMP_MAX];	/* 16-byte aligned */
	const uint64_t *nullbits[MDB_BG_COMP_MAX];	/* or NULL */
	uint64_t lo[MDB_BG_COMP_MAX], span[MDB_BG_COMP_MAX];
	uint32_t kb[MDB_BG_COMP_MAX];
	int32_t is_double[MDB_BG_COMP_MAX], desc[MDB_BG_COMP_MAX];
	int32_t nkeys;
};

/* ... through the band sort (mdb_dev_bandgroup.hip): windows of 2^18 ... 2^25 values, tables of 2^21 rows and more; same contract
 * (comp != NULL: the key is that composite value, keys = its first column, win_lo = 0, kbits = the bits of all fields) */
int mdb_group_count_banded(mdb_dev_ctx *ctx, const int64_t *keys, uint64_t n, int64_t win_lo, uint32_t kbits, uint32_t *out_first, int64_t *out_count,
			   uint64_t cap, uint64_t *out_groups, bool *outside, const struct mdb_bg_comp *comp = NULL);

/* ... through per-workgroup LDS tables (mdb_dev_groupby.hip): composite values of at most 14 bits; 1 = not served */
int mdb_group_count_direct_comp(mdb_dev_ctx *ctx, const struct mdb_bg_comp *comp, uint64_t n, uint32_t bits, uint32_t *out_first, int64_t *out_count,
				uint64_t cap, uint64_t *out_groups);

/* ---- groups of nearly unique keys as one bit per row + exceptions (mdb_dev_dense.hip) */
size_t mdb_dense_arena_bytes(uint64_t n);
int mdb_dense_bits_begin(mdb_dev_ctx *ctx, uint64_t n, unsigned long long **bits);
int mdb_dense_emit(mdb_dev_ctx *ctx, const unsigned long long *bits, uint64_t n, const unsigned long long *exc, uint32_t n_exc, uint32_t *out_first,
		   int64_t *out_count, const int64_t *keys = NULL, bool keys32 = false, int64_t *out_key = NULL);

#endif
