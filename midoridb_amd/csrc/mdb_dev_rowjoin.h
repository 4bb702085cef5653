/*
 * mdb_dev_rowjoin.h - the row-order payload join (mdb_dev_rowjoin.hip) as mdb_dev_join_payload (mdb_dev_pairs.hip) calls it.
 */
#ifndef MDB_DEV_ROWJOIN_H
#define MDB_DEV_ROWJOIN_H

#include "mdb_dev_internal.h"

/* whether the form serves a join of n_l x n_r rows over a compact key window of 2^kbits values (left keys and result columns
 * 16-byte aligned) */
bool mdb_rowjoin_serves(uint64_t n_l, uint64_t n_r, uint32_t kbits, const void *keys_l, const void *null_l, void *const *out, int npay);
/* arena bytes beyond the right table's partition */
size_t mdb_rowjoin_arena_bytes(uint64_t n_l, uint32_t kbits);
int mdb_rowjoin_run(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, int64_t win_lo, uint32_t kbits,
		    const mdb_part_result *pr, uint32_t rem_r, int npay, void *const *out);

#endif
