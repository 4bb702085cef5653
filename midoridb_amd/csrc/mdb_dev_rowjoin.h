/*
 * mdb_dev_rowjoin.h - the row-order payload join (mdb_dev_rowjoin.hip) as mdb_dev_join_payload (mdb_dev_pairs.hip) calls it.
 */
#ifndef MDB_DEV_ROWJOIN_H
#define MDB_DEV_ROWJOIN_H

#include "mdb_dev_internal.h"

/* whether the form serves a join of n_l x n_r rows over a compact key window of 2^kbits values (left keys and result columns
 * 16-byte aligned) */
bool mdb_rowjoin_serves(uint64_t n_l, uint64_t n_r, uint32_t kbits, const void *keys_l, const void *null_l, void *const *out, int npay);
/* whether the right table takes the tile sort as well (otherwise the caller partitions it with mdb_partition_table: two levels) */
bool mdb_rowjoin_tiles_right(const void *keys_r, const void *null_r, const void *const *pay_in, int npay);
/* arena bytes (beyond the right table's partition when the caller makes it: n_r_tiled = 0) */
size_t mdb_rowjoin_arena_bytes(uint64_t n_l, uint64_t n_r_tiled, uint32_t kbits, int npay);
/* pr = the caller's partition of the right table, or NULL: tile-sorted here */
int mdb_rowjoin_run(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const int64_t *keys_r, uint64_t n_r, const void *const *pay_in,
		    int64_t win_lo, uint32_t kbits, const mdb_part_result *pr, uint32_t rem_r, int npay, void *const *out);

#endif
