/*
 * mdb_dist.h - C-ABI of the multi-GPU form of the join / GROUP BY path: one process per GPU, the two key
 * columns hash-partitioned by destination GPU, exchanged with one uneven all-to-all per table over xGMI
 * (RCCL), joined locally.  Groups are disjoint across ranks - both tables are partitioned by
 * the join key - so no reduction or gather is on the data path.
 *
 * What it replaces: nothing upstream is distributed; the reference's one entry point is query_execute()
 * (reference src/engine/query.c:35-106) over the nested-loop join of src/engine/executor_select.c:1076-1149.
 * This is how that same entry point shards: a host process per GPU opens its database, loads ITS rows of
 * every table, and - with MIDORIDB_WORLD_SIZE / MIDORIDB_RANK / MIDORIDB_DIST_ID_FILE in the environment -
 * query_execute() of the north-star shape (JOIN ... ON l = r [JOIN ...] GROUP BY that key, COUNT(*)) runs
 * through mdb_dist_join_group_count(): every rank returns the groups whose keys hash to it, SELECT COUNT(*)
 * over such a join returns the global count on every rank.  Hosts that manage their own buffers call the
 * functions below directly (bench.py does, through ctypes).
 *
 * Plain pointers and sizes only.  Every function returns MIDORIDB_OK (0) or a negative code; the text of
 * the last error is mdb_dist_last_error().  All "dptr" arguments are device pointers on the context's GPU.
 */
#ifndef MDB_DIST_H
#define MDB_DIST_H

#include <stddef.h>
#include <stdint.h>
#include "mdb_dev.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdb_dist mdb_dist;

#define MDB_DIST_ID_BYTES 128		/* = NCCL_UNIQUE_ID_BYTES */

/* ------------------------------------------------------------------ rendezvous
 * Rank 0 creates the communicator id, the host program ships the 128 bytes to the other ranks however it
 * likes (MPI, a socket, a file, the launcher's store) and every rank calls mdb_dist_init() with it. */
int mdb_dist_unique_id(void *id_out /* MDB_DIST_ID_BYTES */);
/* The same through a file every rank can see (one node: a path in /tmp or /dev/shm).  Safe against leftovers of earlier
 * runs: every rank r > 0 announces itself with a fresh nonce (path.hello.<r>), rank 0 removes any old id file, writes the id
 * together with the nonces it has seen (atomically: temp file + rename), and a rank only accepts a file that carries its own
 * nonce - a stale id can never be taken for this run's.  Rank 0 returns once every rank has taken the id; all wait at most
 * timeout_s seconds. */
int mdb_dist_id_via_file(const char *path, int world, int rank, double timeout_s, void *id_out);

/* One communicator pair (key transfers; the tiny count exchanges have their own, so that they never queue
 * behind a transfer in flight) over RCCL, for the GPU of `ctx`.  Collective: every rank calls it. */
int mdb_dist_init(mdb_dev_ctx *ctx, int world, int rank, const void *id, mdb_dist **out);
void mdb_dist_destroy(mdb_dist *d);
int mdb_dist_world(const mdb_dist *d);
int mdb_dist_rank(const mdb_dist *d);
const char *mdb_dist_last_error(const mdb_dist *d);

/* ------------------------------------------------------------------ transport plug-in
 * The exchange needs three operations; RCCL provides them by default.  A host with its own fabric (MPI,
 * a test harness that moves the bytes through host memory) passes them here instead of an id. */
struct mdb_dist_transport {
	void *self;
	/* per peer `n` 64-bit counters: send[world * n] -> recv[world * n], HOST arrays, blocking */
	int (*counts)(void *self, const uint64_t *send, uint64_t *recv, int n);
	/* uneven all-to-all of DEVICE buffers, counts and displacements in elements of elem_bytes bytes, ordered on
	 * `stream` (a hipStream_t): it may return before the bytes have moved, later work on `stream` sees them */
	int (*alltoallv)(void *self, const void *d_send, const size_t *sendcounts, const size_t *sdispls, void *d_recv,
			 const size_t *recvcounts, const size_t *rdispls, size_t elem_bytes, void *stream);
	/* vals[n] (HOST) summed over the ranks, in place, blocking */
	int (*allreduce_sum_u64)(void *self, uint64_t *vals, int n);
	void (*destroy)(void *self);
};
int mdb_dist_init_transport(mdb_dev_ctx *ctx, int world, int rank, const struct mdb_dist_transport *t, mdb_dist **out);

/* ------------------------------------------------------------------ options */
enum mdb_dist_wire {
	MDB_WIRE_AUTO = 0,	/* per call: 4-byte keys when the column statistics of BOTH tables on EVERY rank fit 32 bits
				 * (mdb_dev_key_range + one tiny exchange; two extra read passes per call).  The same statistics give
				 * every rank the two tables' GLOBAL key ranges: rows outside the other table's range join nothing on any
				 * GPU and are dropped before the shuffle (min-max pruning: a fact table whose dimension covers a
				 * sixteenth of its key range sends a sixteenth of its rows) */
	MDB_WIRE_64 = 1,	/* always 8-byte keys */
	MDB_WIRE_32 = 2,	/* the caller knows (catalog statistics) that every key fits 32 bits; a key that does not is
				 * reported as an error by the partition kernel, never truncated */
};
int mdb_dist_set_wire(mdb_dist *d, int mode);
/* 1 when the last exchange shipped 4-byte keys */
int mdb_dist_last_wire32(const mdb_dist *d);
/* Catalog statistics for the next calls with MDB_WIRE_32 / MDB_WIRE_64 (MDB_WIRE_AUTO measures them itself): the GLOBAL
 * [smallest, largest] key of the left and of the right table - the same on every rank.  Rows outside the OTHER table's range
 * are dropped before the shuffle; a key outside the range promised for its OWN table is reported as an error by the
 * partition kernel (supersets are fine, stale statistics are caught).  NULL, NULL: forget them. */
int mdb_dist_set_key_ranges(mdb_dist *d, const int64_t left[2], const int64_t right[2]);
/* 1 when the last call pruned the tables by each other's key range before the shuffle */
int mdb_dist_last_pruned(const mdb_dist *d);
/* 1 when the last mdb_dist_join_group_count() shipped first-level partition regions instead of keys (both global key ranges
 * known, the right table's spanning at most 2^30 values, world a power of two up to 8, caller-provided output buffers): each
 * table is partitioned once - by the join's own first level, whose digit's top bits are the destination - the all-to-alls
 * are posted without a count reaching a host, and the receiver joins what arrived without hashing it again; groups come out
 * in leaf order.  MDB_DIST_FUSED=0 keeps the key-by-destination path. */
int mdb_dist_last_fused(const mdb_dist *d);

/* What the last regions-on-the-wire call planned - the shape every rank derived from the same agreed numbers (world size, the
 * largest shard of each table, the global key ranges): which branch of the layout a workload lands on at world 2, 4 or 8 can be
 * asserted by tests and printed by bench.py before an 8-GPU node is at hand.  Returns 1 when no such call has planned yet. */
struct mdb_dist_plan_info {
	uint32_t world, tables;
	uint32_t digit_bits;		/* first-level digits on the sender: 9 (512) or 12 (4096, the wide fan-out form) */
	uint32_t digits_per_rank;	/* 2^digit_bits / world */
	uint32_t key_bits;		/* the key window holds 2^key_bits values */
	uint32_t receiver_bits;		/* bits of the receiver's own partition level; 0 = digits are joined straight from the regions */
	uint32_t leaf_bits;		/* key bits that index a leaf's LDS tables */
	uint32_t word_bytes;		/* bytes per row on the wire: 2 or 4 */
	uint32_t completed;		/* 1: the call was answered by this path (mdb_dist_last_fused) */
	uint32_t region_words[4];	/* capacity of one first-level region, per table */
	uint64_t block_bytes[4];	/* bytes one rank sends to ONE peer for table x (fixed-size blocks: slack included) */
	uint64_t bytes_per_peer;	/* all tables' blocks + region counters: what crosses ONE xGMI link per direction and call */
};
int mdb_dist_last_plan(const mdb_dist *d, struct mdb_dist_plan_info *out);
/* The plan a call WOULD make at `world` ranks (1, 2, 4, 8) for `tables` tables whose largest shards hold rows_per_rank[t] rows and whose
 * global key ranges are left[] / right[]: a pure host computation (no handle, no GPU) - capacity planning, and what bench.py prints
 * beside a one-GPU measurement: bytes per xGMI link and call at 2, 4 and 8 GPUs.  Returns 1 when the shape is not served by the
 * regions-on-the-wire path at that world size (the key-by-destination path would run). */
int mdb_dist_plan_preview(int world, int tables, const uint64_t *rows_per_rank, const int64_t left[2], const int64_t right[2],
			  struct mdb_dist_plan_info *out);
/* Where a regions-on-the-wire call spends its time on this rank (measurement aid: HIP events on the operator's and the transfer
 * stream; off by default).  ms[0] = first level of every table (sender), ms[1] = time the receiver's first kernel waited for
 * the last block to arrive after the sender's passes were done (the wire, as far as it is not hidden), ms[2] = receiver (region
 * descriptors, own level if any, leaves), ms[3] = the whole device pipeline. */
#define MDB_DIST_PHASES 4
int mdb_dist_set_phase_timing(mdb_dist *d, int on);
int mdb_dist_last_phases(const mdb_dist *d, double *ms /* [MDB_DIST_PHASES] */);

/* ------------------------------------------------------------------ the sharded north-star operator
 *
 * SELECT l.key, COUNT(*) FROM L INNER JOIN R ON l.key = r.key GROUP BY l.key over tables whose rows are spread
 * over the ranks in any way: keys_l / keys_r are THIS rank's rows.  Per table: partition by destination
 * (mdb_dev_partition_by_dest: dest = hash(key) mod world, NULL keys dropped - a NULL key never joins,
 * reference executor_select.c:557-579), counts exchange, uneven all-to-all; table L's transfer overlaps table
 * R's partitioning, table R's the local hashing + radix partitioning of the received L
 * (mdb_dev_join_group_count_begin / _finish).  out_key / out_count (capacity cap; the rows received for L are
 * always enough): the groups whose key hashes to this rank, in order of first occurrence in the received
 * stream.  *out_groups = their number, *out_joined = joined rows on this rank (sum of its counts).
 * Collective and synchronous. */
int mdb_dist_join_group_count(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			      const uint64_t *null_r, uint64_t n_r, int64_t *out_key, int64_t *out_count, uint64_t cap,
			      uint64_t *out_groups, uint64_t *out_joined);
/* The same with outputs allocated by the call, once the number of received left rows is known (*out_key, *out_count
 * and, when out_first != NULL, *out_first: device buffers to release with mdb_dev_free; out_first[g] = position of the
 * group's first row in the left stream the local join saw).  MDB_DIST_LEFT_IN_PLACE: the left rows are already on the
 * rank their keys hash to - the groups a previous call returned - so only the right table is exchanged and out_first
 * indexes keys_l itself: this is how query_execute() chains the operator over further tables joined on the same key
 * (reference shape: A JOIN B ON a = b JOIN C ON a = c ... GROUP BY a). */
#define MDB_DIST_LEFT_IN_PLACE 1u
/* the groups must land where their KEY hashes to (mdb_dev_partition_by_dest's hash) because a later MDB_DIST_LEFT_IN_PLACE call
 * will send another table after them: the regions-on-the-wire path (mdb_dist_last_fused), whose placement follows the hash of
 * the key's offset in this call's window, is not taken */
#define MDB_DIST_PLACE_BY_KEY_HASH 2u
int mdb_dist_join_group_count_alloc(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
				    const uint64_t *null_r, uint64_t n_r, uint32_t flags, int64_t **out_key, int64_t **out_count,
				    uint32_t **out_first, uint64_t *out_groups, uint64_t *out_joined);
/* One left table and 2 ... 3 right tables, all joined on ONE key - SELECT l.key, COUNT(*) FROM L JOIN R0 ON l.key = r0.key JOIN R1 ON
 * l.key = r1.key ... GROUP BY l.key (reference: recursive join executor_select.c:1151-1280 + GROUP BY :1526-1588; BASELINE
 * configs[4]) - as ONE exchange: every table is partitioned once with the same window hash (mdb_dist_last_fused), every rank
 * joins the regions of all tables it received and multiplies the right tables' counts per key.  keys_r / null_r / n_r: HOST
 * arrays of n_right entries.  Returns MIDORIDB_OK (*out_key / *out_count allocated by the call: mdb_dev_free), 1 when this
 * shape is not served - the global key ranges are not known (neither MDB_WIRE_AUTO nor mdb_dist_set_key_ranges) or too wide,
 * skewed keys overflowed a region ... - on EVERY rank alike, and the caller chains mdb_dist_join_group_count_alloc calls
 * (MDB_DIST_PLACE_BY_KEY_HASH, then MDB_DIST_LEFT_IN_PLACE), or a negative error code.  Collective and synchronous. */
int mdb_dist_join_group_count_multi_alloc(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, int n_right,
					  const int64_t *const *keys_r, const uint64_t *const *null_r, const uint64_t *n_r, int64_t **out_key,
					  int64_t **out_count, uint64_t *out_groups, uint64_t *out_joined);

/* GROUP BY key + COUNT(*) of ONE sharded key column (reference src/engine/executor_select.c:1526-1588 over rows spread across the
 * ranks) as (key, COUNT) pairs in unspecified order, every key on exactly one rank: each rank's single partition pass writes 2- or
 * 4-byte words into first-level regions, the regions travel (no rows, no row ids), the receiver counts per leaf slot.  For
 * statements that select nothing but the group key and COUNT(*) and ask for no order.  Collective.  Outputs allocated by the call
 * (mdb_dev_free).  Returns 1 - the same on every rank - when it does not serve the column (global key range unknown - see
 * mdb_dist_set_key_ranges, whose RIGHT range is the column's here - or wider than 2^30 values, a NULL bitmap passed, skew): the
 * caller then exchanges the rows (mdb_dist_shuffle_rows) and groups locally. */
int mdb_dist_group_count_keys_alloc(mdb_dist *d, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t **out_key,
				    int64_t **out_count, uint64_t *out_groups);
/* rows of L this rank received in the last call (what `cap` has to cover), 0 before the first */
uint64_t mdb_dist_last_received_left(const mdb_dist *d);

/* ------------------------------------------------------------------ sharded joins that materialise rows
 *
 * What the reference's one entry point does for ANY join (_join_nested_loop_tbl2tbl, reference
 * src/engine/executor_select.c:1076-1149; recursive form :1151-1280) when the tables are spread over the ranks: the rows
 * of a table (or of a joined tuple stream) are re-distributed so that every row lands on the rank its join key hashes to
 * (dest = the same hash mdb_dev_partition_by_dest uses), payload columns travelling with the key; the join itself is then
 * the single-GPU operator on what arrived.  BASELINE configs[3] (10^9-row INNER JOIN over 8 GPUs) and configs[4]
 * (three-way join with DOUBLE payload) run through these two functions; query_execute() in sharded mode uses
 * mdb_dist_shuffle_rows() for every join, GROUP BY and DISTINCT whose rows are not yet where their key hashes to.
 *
 * A column of the stream: 8-byte cells (INT64 values or the bits of a DOUBLE - moved as opaque words), optional NULL
 * bits, read for stream position k at row rid[k] (rid == NULL: row k) - the tuple-stream convention of mdb_dev.h. */
struct mdb_dist_col {
	const void *values;
	const uint64_t *nullbits;	/* or NULL */
	const uint32_t *rid;		/* or NULL = identity */
};
#define MDB_DIST_SHUFFLE_MAX_COLS 64
/* rows whose key is NULL are not dropped but all sent to one rank (GROUP BY / DISTINCT: NULL keys form one group,
 * reference executor_select.c:1477-1482); without it they stay home and vanish, as a NULL key joins nothing (:557-579) */
#define MDB_DIST_KEEP_NULL_KEYS 1u
/* return as soon as the transfers are posted: the outputs may be passed to operators on the context's stream only after
 * mdb_dist_wait_transfers() - lets the next table's partitioning overlap this table's transfers */
#define MDB_DIST_NO_WAIT 2u

/* keys[n] (with key_nulls) are the stream's partitioning key; cols[0..ncols) the columns to carry (the key column itself
 * among them when the caller wants it back).  out_values[c] / out_nullbits[c]: device buffers allocated by the call
 * (mdb_dev_free), *out_n rows each, in the order (source rank, position in that rank's send order); out_nullbits[c] is NULL
 * when column c has no NULL bits on ANY rank.  At most MDB_DIST_SHUFFLE_MAX_COLS columns.  Collective; one host
 * synchronisation (the counts).  A failure on one rank (allocation, a bad argument) is agreed on with the counts: every rank
 * returns an error, none is left waiting in a collective. */
int mdb_dist_shuffle_rows(mdb_dist *d, const int64_t *keys, const uint64_t *key_nulls, uint64_t n, uint32_t flags,
			  const struct mdb_dist_col *cols, int ncols, void **out_values, uint64_t **out_nullbits, uint64_t *out_n);
int mdb_dist_wait_transfers(mdb_dist *d);
/* Every rank's rows to EVERY rank (the small side of a join without an equi-join key - FROM A, B or a general ON expression,
 * reference executor_select.c:1096-1141, optimiser_select.c:395-464: nothing says which rank a row's partners live on, so the table is
 * replicated and every rank pairs its own rows of the other side with all of it).  cols as above; out_values[c] / out_nullbits[c]:
 * the rows of all ranks in rank order (*out_n of them, the same on every rank), allocated by the call.  Collective and synchronous;
 * a failure on one rank is agreed on before anything is posted. */
int mdb_dist_broadcast_rows(mdb_dist *d, uint64_t n, const struct mdb_dist_col *cols, int ncols, void **out_values, uint64_t **out_nullbits,
			    uint64_t *out_n);

/* SELECT ... FROM L INNER JOIN R ON l.key = r.key over sharded tables: both tables are shuffled by their key (L's transfers
 * overlap R's partitioning), joined locally with mdb_dev_join_pairs and projected with mdb_dev_gather_cols.  Output (device
 * buffers allocated by the call, *out_rows rows each - the joined rows whose key hashes to this rank, in (received left
 * row, received right row) order): *out_key = the join key of every joined row (both sides hold the same value; pass
 * NULL when not wanted), out_l[c] / out_l_nulls[c] = column c of cols_l, out_r[c] / out_r_nulls[c] = column c of cols_r.
 * With no payload column on either side (BASELINE configs[3]: the join of two key columns) and the global key ranges known
 * (MDB_WIRE_AUTO, or mdb_dist_set_key_ranges), no row has to be identified: the tables travel as first-level partition
 * regions (mdb_dist_last_fused), every key's partners are counted and the key written COUNT times - in leaf order.
 * Collective and synchronous. */
int mdb_dist_join_pairs(mdb_dist *d, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const struct mdb_dist_col *cols_l,
			int ncols_l, const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, const struct mdb_dist_col *cols_r,
			int ncols_r, int64_t **out_key, void **out_l, uint64_t **out_l_nulls, void **out_r, uint64_t **out_r_nulls,
			uint64_t *out_rows);

/* helpers for hosts without a collective library of their own */
/* every rank's n (<= 8) 64-bit values to every rank: all[p * n + i] = value i of rank p (HOST arrays, blocking, collective) */
int mdb_dist_allgather_u64(mdb_dist *d, const uint64_t *mine, int n, uint64_t *all);
int mdb_dist_allreduce_sum_u64(mdb_dist *d, uint64_t *vals, int n);
/* every rank's n bytes (HOST) to every rank: *all = malloc'd concatenation in rank order (free()), counts[p] = rank p's bytes
 * (counts: world entries).  For small variable-length payloads - query_execute() in sharded mode announces the new entries of the
 * ranks' string dictionaries with it, so that VARCHAR cells can travel as ids every rank agrees on. */
int mdb_dist_allgather_bytes(mdb_dist *d, const void *mine, uint64_t n, void **all, uint64_t *counts);
int mdb_dist_barrier(mdb_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* MDB_DIST_H */
