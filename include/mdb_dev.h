/*
 * mdb_dev.h - C-ABI of the MI355X (gfx950) device layer of libmidoridb_amd.so.
 *
 * This is the thin boundary between the host executor (plain C, the reference's
 * language) and the hand-written HIP kernels.  Plain pointers and sizes only; no
 * C++ or torch types.  Every function returns MIDORIDB_OK (0) or a negative code
 * (mdb_error.h); HIP failures map to -MIDORIDB_INTERNAL with the text available
 * from mdb_dev_last_error().  Nothing here ever exit()s.
 *
 * Each operator replaces one loop of the reference's SELECT executor
 * (reference src/engine/executor_select.c); the file:line of the code it replaces
 * is cited per function.  All "dptr" arguments are DEVICE pointers; tables are
 * device-resident columns:
 *
 *     values   : int64_t[n] (CT_INTEGER) or double[n] (CT_DOUBLE), 8 bytes per row
 *     nullbits : uint64_t[(n+63)/64], bit (i & 63) of word (i >> 6) set  <=>  row i is NULL
 *                (same polarity as the reference's row->null_bitmap, executor_select.c:384);
 *                a NULL pointer means "column has no NULLs"
 *
 * which replaces the reference's 4 KiB row-store datablocks
 * (reference include/primitive/row.h:15-28, datablock.h:7-13) on this path.
 *
 * A "tuple stream" of length n is the device analogue of the reference's early
 * materialisation table: tuple k of a joined stream refers to base rows through
 * row-id vectors (uint32_t[n], one per FROM table); rid == NULL means identity.
 */
#ifndef MDB_DEV_H
#define MDB_DEV_H

#include <stddef.h>
#include <stdint.h>
#include "mdb_error.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdb_dev_ctx mdb_dev_ctx;

/* ------------------------------------------------------------------ context */

/* Create a context on HIP device `device`.  `stream` is the hipStream_t to launch on:
 * NULL = the device's default stream (what torch.cuda.current_stream().cuda_stream is
 * unless the caller switched streams), MDB_STREAM_OWN = a private non-blocking stream. */
#define MDB_STREAM_OWN ((void *)(intptr_t)-1)
int mdb_dev_ctx_create(int device, void *stream, mdb_dev_ctx **out);
void mdb_dev_ctx_destroy(mdb_dev_ctx *ctx);
int mdb_dev_ctx_set_stream(mdb_dev_ctx *ctx, void *stream);
const char *mdb_dev_last_error(mdb_dev_ctx *ctx);
int mdb_dev_sync(mdb_dev_ctx *ctx);
/* Number of visible HIP devices (no context needed; <0 on error). */
int mdb_dev_device_count(void);
/* Pre-size the internal scratch arena (bytes).  Operators grow it on demand; growing
 * synchronises and reallocates, so benchmarks call this (or one warm-up) first. */
int mdb_dev_reserve(mdb_dev_ctx *ctx, size_t bytes);
/* When on, the two tables of a join are partitioned concurrently on two HIP streams.  Default off:
 * measured on MI355X it does not shorten the single-GPU pipeline (every kernel already fills the chip). */
int mdb_dev_set_overlap(mdb_dev_ctx *ctx, int on);
/* Join / GROUP BY keys that all lie inside the int32 range (the reference's integers are 32-bit in practice: literals
 * through atoi `midorisql.l:85`, compares through `int` `executor_select.c:462`) - or, more generally, inside any
 * 2^32-wide window of the int64 range (surrogate keys that start at 10^12, timestamps of one year ...; the window is
 * centred on a sample of both columns) - are partitioned and compared as 32-bit hashes of their offset in the window,
 * the row id sharing the 8-byte word.  Verified on the device for every key, never assumed: a key outside the
 * window makes the operator redo its work with 64-bit hashes; results are identical either way.
 * mode 0: never; 1 (default): when a sample of both key columns fits, tables of 2^20 rows or more; 2: always try
 * (the plain int32 window, no sampling). */
int mdb_dev_set_narrow_keys(mdb_dev_ctx *ctx, int mode);
/* 1 when the last completed join / GROUP BY operator of this context ran in the narrow form (for byte accounting) */
int mdb_dev_last_join_narrow(mdb_dev_ctx *ctx);
/* Pruning of the left table in the compact narrow form (unsplit calls): the right table is partitioned first.
 * Min-max pruning: its first partition level records the exact range of its keys and the left table's first level drops
 * every row outside (MDB_MINMAX_PRUNE=0 turns it off).  Semi-join filter: when the key sample says that the right table
 * has less than a quarter of the left table's rows without covering a small part of its range, its hashed keys also
 * become a bitmap and the left table's second level drops the rows whose bit is clear (MDB_SEMIJOIN=0 turns it off).
 * Both are exact in effect (a row is only dropped when no right row can have its key): results are identical, only the
 * bytes moved change.
 * -> of the last completed join / GROUP BY operator: bit 8 = min-max pruning ran; low byte = 0 without the bitmap, else 1 +
 * log2(adjacent hashed values per bitmap bit); bit 9 = the tables were partitioned ONCE (key windows of 2^15 ... 2^23
 * values: one 9-bit level, direct-address leaf tables of 2^(k - 9) entries with 16-bit row counts; MDB_ONE_LEVEL=0 turns it
 * off; a key with 2^16 or more rows sends the operator back to two levels; joins with up to 31 right and 15 left rows per key keep 4 bytes
 * per key value in the leaf - two workgroups per CU -, others take the 16-bit counts on the same partitioned tables: MDB_LEAF4=0); bit 10 = several right tables were counted in one
 * pass (mdb_dev_join_group_count_multi did not chain two-table operators); bit 11 = the operator ran without MDB_ORDER_FIRST
 * (any group order); bit 12 = the ordered operator took ONE 4096-digit pass per table (key windows of 2^24 ... 2^27 values, from
 * 2^24 rows in all, at most 2^27 left rows, no min-max pruning to be had: the left table's rows travel as 4-byte words that name
 * their place in a 32 768-row tile, one workgroup joins a digit of up to 2^15 key values; MDB_WIDE12=0 turns it off,
 * MDB_WIDE12_MIN=<rows> moves the threshold; more than 31 right or 15 left rows of one key send the operator back to two levels); bit 13 = the
 * one-level join wrote its group records straight into the ordering kernel's ranges of 2^16 row ids (from the second call over the same columns
 * on, when the remembered group count is small enough; MDB_ORDER_RANGES=0 turns it off) - the ordering kernel is then launched right behind the
 * leaf kernel, before the host has seen the group count, its writes bounded by `cap`: one sync per call (MDB_ORDER_EARLY=0: two). */
int mdb_dev_last_join_filter(mdb_dev_ctx *ctx);
/* 1 when the last mdb_dev_join_pairs() matched EVERY left row with exactly one right row (unique right keys, no left row
 * without a partner - the primary-key join of BASELINE configs[1]): out_l is then 0, 1, 2 ... and the left table's columns
 * of the joined stream are its columns as they stand - a caller need not gather them through out_l. */
int mdb_dev_last_pairs_identity(mdb_dev_ctx *ctx);

/* ---- catalog statistics instead of key samples (round 5).  A caller that KNOWS its key columns - query_execute() keeps the smallest /
 * largest value of every column as rows are ingested (csrc/mdb_store.c) - hands that to the operators: between
 *	mdb_dev_call_stats(ctx, keys_l, &stats_l, keys_r, &stats_r);	(keys_r / stats_r NULL for a one-column operator)
 * and
 *	mdb_dev_call_stats(ctx, NULL, NULL, NULL, NULL);
 * every join / GROUP BY operator called with exactly these key-column pointers takes its key windows, its pruning and narrow-form
 * decisions from the statistics: no sampling kernel, no host synchronisation for it, nothing remembered by address, and the first query
 * over a column costs what the twentieth does.  The range may be a superset of the column's (rows deleted since, a filtered copy of the
 * column): every key is still verified on the device.  min > max: the column holds no non-NULL value.  Raw callers that pass nothing
 * keep the sampled decisions. */
struct mdb_dev_col_stats {
	int64_t min, max;	/* smallest / largest non-NULL value (a superset range is fine) */
	uint64_t rows, nulls;	/* rows of the column, NULLs among them (exact when MDB_COL_DISTINCT is set: "the column holds every value of
				 * [min, max]" is rows - nulls == max - min + 1) */
	uint64_t flags;		/* MDB_COL_* */
};
/* No non-NULL value occurs twice in the column - MEASURED over all its rows (mdb_dev_distinct_scan; the store measures at ingest, follows
 * appended rows, and drops the verdict with an UPDATE: the reference carries UNIQUE / PRIMARY KEY per column, include/primitive/column.h:41-46,
 * set by src/engine/executor_create.c:29-58, and never enforces them - so this flag is never taken from DDL).  With it the operators run
 * the forms that need unique keys without a pilot launch and without remembering a column by its address: a join + GROUP BY of two such
 * columns leaves its groups as one bit per left row from the first statement on (verified like every promise: counts that do not add up
 * send the call to the record form), a GROUP BY over one is the identity - the one form that TRUSTS a statistic (verifying it would be the
 * scan that measured it): set the flag only from a measurement. */
#define MDB_COL_DISTINCT 1ull
int mdb_dev_call_stats(mdb_dev_ctx *ctx, const void *keys_l, const struct mdb_dev_col_stats *l, const void *keys_r, const struct mdb_dev_col_stats *r);

/* What the last join / GROUP BY operator of this context did (for tests, EXPLAIN-like output and byte accounting) */
struct mdb_dev_plan_info {
	uint32_t key_form;	/* 0: 64-bit hashes; 1: 32-bit hashes of a 2^32-wide window; 2: k-bit hashes of a compact window, direct-address leaves */
	uint32_t key_bits;	/* key_form 2: the window holds 2^key_bits values */
	uint32_t levels;	/* partition levels of the final attempt (1 or 2) */
	uint32_t digits;	/* first-level digits: 512, or 4096 (one 4096-digit pass per table) */
	uint32_t minmax_pruned;	/* the left table's first level dropped the rows outside the right table's key range */
	uint32_t semijoin;	/* 0, or 1 + log2(values per bit of the right table's key bitmap) */
	uint32_t any_order;	/* ran without row ids and ordering */
	uint32_t ranged_order;	/* group records written straight into the ordering kernel's ranges */
	uint32_t multi_one_pass;	/* several right tables counted in one pass */
	uint32_t retries;	/* times the operator was redone (0: the first plan held) */
	uint32_t samples;	/* key-sample kernels (each with a host synchronisation) the call launched */
	uint32_t from_stats;	/* 1: windows and ranges came from mdb_dev_call_stats() */
	uint32_t payload_form;	/* last mdb_dev_join_payload: 0 not served, 1 one level (cells in the leaf's LDS), 2 two levels, 3 row order (tile sort) */
	uint32_t group_form;	/* last mdb_dev_group_count: 0 the partitioned path (or one of its small-input forms), 1 band sort (4-byte row words), 2 tile sort,
				 * 3 the identity (MDB_COL_DISTINCT: group i = row i) */
	uint32_t arena_mib;	/* mdb_dev_explain_* only: MiB of scratch arena the plan asks for (mdb_dev_reserve) */
	uint32_t small_form;	/* 0, or the operator answered before any partition level: 1 one workgroup in LDS (at most 2048 rows per table), 2 per-workgroup
				 * LDS tables (both key columns inside one window of at most 4096 values) */
	uint32_t keys_are_left_column;	/* MDB_KEYS_MAY_ALIAS was given and every left row is a group: out_key was not written - the left key column holds the keys */
	uint32_t counts_all_one;	/* MDB_COUNTS_OPTIONAL was given and every COUNT(*) is 1: out_count was not written */
	uint32_t groups_as_bits;	/* the groups left the leaf kernel as one bit per row + exceptions (nearly unique keys / nearly every left row a group of
				 * COUNT 1), not as a record each: 1 decided by a pilot launch, 2 by what the last call over the columns delivered, 3 by the
				 * caller's statistics (MDB_COL_DISTINCT on both key columns, the right one holding every value of its range) */
	uint32_t payload_tables;	/* right tables the last payload join served in ONE call: 1 mdb_dev_join_payload, 2 ... mdb_dev_join_payload_multi (the left
				 * table sorted once, one leaf launch, one placement pass); 0 not served */
};
int mdb_dev_last_plan(mdb_dev_ctx *ctx, struct mdb_dev_plan_info *out);
/* The MDB_* environment knobs are read once per process and kept: a process that changes one while it runs calls this. */
void mdb_dev_reload_knobs(void);
/* Plans as data: what mdb_dev_join_group_count (further_tables > 0: mdb_dev_join_group_count_multi) / mdb_dev_group_count WOULD do for key
 * columns with these statistics - the operators' own decision code, run up to its first launch on a context without a device: a pure host
 * computation (no GPU needed), so that a change of a heuristic shows as a diff of tests/golden/plans.json.  as_sample != 0: the numbers are
 * treated as what a key sample found on a first call of a raw caller (from_stats 0, samples 1; the forms that only a catalog's promise
 * opens are not taken: groups_as_bits 1 = "a pilot launch decides").  The sharded operator's plan: mdb_dist_plan_preview (mdb_dist.h). */
struct mdb_dev_explain_request {
	struct mdb_dev_col_stats left, right;	/* rows = the tables' rows; right is ignored by mdb_dev_explain_group_count */
	uint32_t further_tables;		/* further right tables joined on the same key (0 ... 2) */
	uint64_t further_rows[2];
	uint32_t left_nulls_bitmap;		/* != 0: the left key column comes with a NULL bitmap */
	uint32_t as_sample;
	uint32_t num_cus;			/* 0: 256 (MI355X) */
};
int mdb_dev_explain_join_group_count(const struct mdb_dev_explain_request *rq, struct mdb_dev_plan_info *out);
int mdb_dev_explain_group_count(const struct mdb_dev_explain_request *rq, struct mdb_dev_plan_info *out);
int mdb_dev_explain_join_payload(const struct mdb_dev_explain_request *rq, int cells /* 1 or 2 payload columns */, struct mdb_dev_plan_info *out);

/* Running totals since the context was created: what a call that took longer than its neighbours paid for.  Read before and after a timed
 * loop (bench.py: `retries_in_timed_steps`), they cost the loop nothing. */
struct mdb_dev_counters {
	uint64_t operator_calls;	/* outermost join / GROUP BY operator calls */
	uint64_t retries;		/* times an operator was redone inside them (a plan that did not hold) */
	uint64_t samples;		/* key-sample kernels, each with a host synchronisation */
	uint64_t arena_grows;		/* the scratch arena was released and allocated larger (a device synchronisation + hipFree + hipMalloc) */
	uint64_t alloc_misses;		/* mdb_dev_alloc() calls no released buffer could serve (hipMalloc) */
};
int mdb_dev_counters(mdb_dev_ctx *ctx, struct mdb_dev_counters *out);
size_t mdb_dev_arena_bytes(mdb_dev_ctx *ctx);

/* ------------------------------------------------------------------ memory
 * The join / GROUP BY operators remember what they learned about a key column (sampled range, key form, duplicate flags) by
 * the column's device address and length, for a few uses.  Every remembered verdict is verified on the device - a stale one
 * costs a failed attempt, never a result -, and the library drops what it knows about a buffer that is released
 * (mdb_dev_free), uploaded into (mdb_dev_h2d) or updated (mdb_dev_scatter_set64).  A caller that rewrites a column in place
 * by other means (its own kernels) pays at most a few slower calls. */
int mdb_dev_alloc(mdb_dev_ctx *ctx, size_t bytes, void **dptr);
int mdb_dev_free(mdb_dev_ctx *ctx, void *dptr);
/* bytes the allocator holds behind a buffer of mdb_dev_alloc (it hands out released buffers of up to twice the size asked for),
 * 0 for a pointer it does not know */
size_t mdb_dev_alloc_size(mdb_dev_ctx *ctx, const void *dptr);
/* A buffer of mdb_dev_alloc with several READERS: mdb_dev_retain() adds a holder, every holder ends with mdb_dev_free(), the last one
 * releases the buffer; mdb_dev_holders() = holders beyond the first.  query_execute() uses it to hand a table's device column to a
 * device-resident result without a copy (SELECT * over a join in the left table's row order: the left table's columns ARE result
 * columns): a holder must not write; the store copies a column that has other holders before an UPDATE writes into it. */
int mdb_dev_retain(mdb_dev_ctx *ctx, const void *dptr);
unsigned mdb_dev_holders(mdb_dev_ctx *ctx, const void *dptr);
int mdb_dev_memset(mdb_dev_ctx *ctx, void *dptr, int byte, size_t bytes);
/* Page-locked host memory for result columns (D2H at PCIe rate instead of through the driver's staging copy).
 * Process-wide pool, independent of any context: buffers are recycled by later results and may be freed after the
 * context is gone (query_free() after database_close()).  Small requests fall back to malloc. */
void *mdb_dev_host_alloc(size_t bytes);
void mdb_dev_host_free(void *p);
int mdb_dev_h2d(mdb_dev_ctx *ctx, void *dptr, const void *host, size_t bytes);	/* synchronous */
int mdb_dev_d2h(mdb_dev_ctx *ctx, void *host, const void *dptr, size_t bytes);	/* synchronous */

/* ------------------------------------------------------------------ profiling */
/* When enabled, every kernel launch is bracketed by HIP events on the context's
 * stream; mdb_dev_prof_read() synchronises and reports per-kernel launch counts and
 * total milliseconds since the last mdb_dev_prof_reset(). */
#define MDB_DEV_PROF_MAX 32
struct mdb_dev_prof_entry {
	char name[48];
	uint32_t launches;
	double total_ms;
};
int mdb_dev_prof_enable(mdb_dev_ctx *ctx, int on);
int mdb_dev_prof_reset(mdb_dev_ctx *ctx);
int mdb_dev_prof_read(mdb_dev_ctx *ctx, struct mdb_dev_prof_entry *out, int cap, int *n_out);
/* out (capacity cap) = the symbols of the kernels launched under profiler name `name` since profiling was enabled - the
 * template instances an external profiler (rocprofv3) lists them under -, newline-separated: ties the live per-kernel
 * timings to the rows of a rocprofv3 summary without anybody spelling a mangled name */
int mdb_dev_prof_symbols(mdb_dev_ctx *ctx, const char *name, char *out, size_t cap);

/* ------------------------------------------------------------------ scan + filter
 *
 * Replaces proc_where_clause() + eval_row_cond()/eval_cmp()/eval_isxnull()
 * (reference executor_select.c:1435-1463, 1027-1074, 865-966): evaluate a boolean
 * predicate per tuple and keep the tuples for which it is true, preserving order.
 * Semantics kept from the reference: a comparison with a NULL operand is false
 * (executor_select.c:557-579, 629-631); AND/OR/XOR fold left to right on plain
 * booleans (:1041-1058); IS [NOT] NULL tests the NULL bit (:965).  Integer
 * comparisons are full 64-bit here (the reference truncates to 32 bits - SURVEY
 * 8a D5; identical on values in [-2^31, 2^31)).
 *
 * The predicate is a postfix program over a boolean stack.
 */
enum mdb_pred_op {
	MDB_P_CMP_COL_CONST = 1,	/* push (col <cmp> imm); a=slot, cmp, type, imm           */
	MDB_P_CMP_CONST_COL = 2,	/* push (imm <cmp> col)                                    */
	MDB_P_CMP_COL_COL   = 3,	/* push (col a <cmp> col b)                                */
	MDB_P_ISNULL        = 4,	/* push (col a IS NULL) ^ cmp  (cmp = 1 for IS NOT NULL)   */
	MDB_P_CONST         = 5,	/* push imm != 0                                           */
	MDB_P_AND           = 6,
	MDB_P_OR            = 7,
	MDB_P_XOR           = 8,
	MDB_P_IN_BITS       = 9,	/* push (bit <col a's value> of the device bit table at imm is set) ^ cmp, false for a NULL cell and for
					 * a value beyond the table's b 64-bit words: set membership of small non-negative ids - LIKE over a
					 * VARCHAR column, whose cells are dictionary ids (the pattern is matched once per distinct string) */
};
/* comparison codes = the reference's enum ast_comparison_type (include/parser/ast.h:71-84) */
enum mdb_cmp { MDB_CMP_LT = 1, MDB_CMP_GT = 2, MDB_CMP_NE = 3, MDB_CMP_EQ = 4, MDB_CMP_LE = 5, MDB_CMP_GE = 6 };
enum mdb_valtype { MDB_T_INT64 = 0, MDB_T_DOUBLE = 1 };

struct mdb_pred_insn {
	int32_t op;		/* enum mdb_pred_op */
	int32_t cmp;		/* enum mdb_cmp (or negation flag for MDB_P_ISNULL) */
	int32_t type;		/* enum mdb_valtype */
	int32_t a, b;		/* column slots */
	int32_t pad;
	int64_t imm;		/* int64 value or the bits of a double */
};

#define MDB_PRED_MAX_INSNS 64
#define MDB_PRED_MAX_SLOTS 16

struct mdb_col_binding {
	const void *values;		/* dptr: 8-byte values of the base column */
	const uint64_t *nullbits;	/* dptr or NULL */
	const uint32_t *rid;		/* dptr: row-id vector of the tuple stream for this column's table, or NULL = identity */
};

/* out_sel (dptr, capacity n) receives the ascending tuple positions that pass;
 * *out_count (host) their number.  Synchronises. */
int mdb_dev_filter(mdb_dev_ctx *ctx, const struct mdb_pred_insn *prog, int n_insns,
		   const struct mdb_col_binding *cols, int n_cols, uint64_t n,
		   uint32_t *out_sel, uint64_t *out_count);

/* Scan + WHERE + projection of ONE table in one call (proc_from_clause_table + proc_where_clause +
 * proc_select_clause, reference executor_select.c:1282-1343, 1435-1463, 1369-1433) in ONE pass: the kernel that evaluates
 * the predicate also writes the surviving rows of every projected column at their final positions (decoupled look-back
 * over the row blocks) - no bitmap pass, no scan, no selection vector.  proj[c].values / nullbits: base columns of the
 * scanned table (no row-id vectors); *proj[c].out_values / *proj[c].out_nullbits: device buffers allocated by the call
 * with room for all n rows, of which the first *out_count hold the result (release with mdb_dev_free; out_nullbits
 * stays NULL for a column without NULL bitmap).  Synchronises. */
struct mdb_project_col {
	const void *values;
	const uint64_t *nullbits;
	void **out_values;
	uint64_t **out_nullbits;
};
int mdb_dev_filter_project(mdb_dev_ctx *ctx, const struct mdb_pred_insn *prog, int n_insns, const struct mdb_col_binding *cols, int n_cols,
			   uint64_t n, const struct mdb_project_col *proj, int n_proj, uint64_t *out_count);

/* ------------------------------------------------------------------ gather / projection
 *
 * Replaces cpy_cols()/_merge_rows() (reference executor_select.c:340-438) and the
 * column re-pack of proc_select_clause()/table_rem_column() (:1369-1433,
 * src/primitive/column.c:146-243): late materialisation of one output column.
 * dst[k] = src[idx ? idx[k] : k], NULL bits carried.  dst_nullbits may be NULL when
 * src_nullbits is NULL. */
int mdb_dev_gather64(mdb_dev_ctx *ctx, const void *src, const uint64_t *src_nullbits,
		     const uint32_t *idx, uint64_t n, void *dst, uint64_t *dst_nullbits);
/* The projection of a whole result in ONE launch: column c of the output is dst[c][k] = src[c][rid[c] ? rid[c][k] : k]
 * for k < n, NULL bits carried like mdb_dev_gather64.  Columns that share a row-id vector (the columns of one FROM
 * table) read it once, and a thread's gathers of all columns are in flight together - against one launch per column
 * (four for BASELINE configs[1]: 0.51 of 1.2 ms).  At most MDB_GATHER_MAX_COLS columns over at most
 * MDB_GATHER_MAX_RIDS distinct row-id vectors per call. */
#define MDB_GATHER_MAX_COLS 16
#define MDB_GATHER_MAX_RIDS 8
struct mdb_gather_col {
	const void *src;		/* 8-byte values of the base column */
	const uint64_t *src_nullbits;	/* or NULL */
	const uint32_t *rid;		/* row-id vector of the column's table in the tuple stream, or NULL = identity */
	void *dst;			/* n 8-byte values */
	uint64_t *dst_nullbits;		/* (n + 63) / 64 words; required when src_nullbits != NULL */
};
int mdb_dev_gather_cols(mdb_dev_ctx *ctx, const struct mdb_gather_col *cols, int ncols, uint64_t n);

/* DOUBLE equi-join keys (cmp_double_value_to_value, reference executor_select.c:440-460, compares with IEEE `==`):
 * dst[k] = the key of row idx[k] (idx == NULL: row k) as a word that the join operators can compare bit for bit -
 * -0.0 becomes +0.0 (they are equal upstream) and a NaN row gets its bit in dst_nullbits set (NaN equals nothing
 * upstream, exactly like a NULL key, :557-579); the source's NULL bits are carried over.  dst_nullbits is required:
 * (n + 63) / 64 words. */
int mdb_dev_double_join_keys(mdb_dev_ctx *ctx, const double *src, const uint64_t *src_nullbits, const uint32_t *idx, uint64_t n,
			     int64_t *dst, uint64_t *dst_nullbits);
/* dst[k] = src[idx[k]] for uint32 row-id vectors (composition of tuple streams). */
int mdb_dev_gather32(mdb_dev_ctx *ctx, const uint32_t *src, const uint32_t *idx, uint64_t n, uint32_t *dst);
/* out[i] = table[cells[i]] (0 for a cell outside [0, table_n)): 8-byte ids translated through a device table - how VARCHAR cells
 * (ids of one process's string dictionary, reference src/primitive/column.c:255-293 keeps heap pointers) become ids every rank agrees
 * on before they cross xGMI, and local ids again after.  In place when out == cells.  Asynchronous on the context's stream. */
int mdb_dev_map_ids(mdb_dev_ctx *ctx, const int64_t *cells, uint64_t n, const int64_t *table, uint64_t table_n, int64_t *out);
int mdb_dev_iota32(mdb_dev_ctx *ctx, uint32_t *dst, uint64_t n);

/* ------------------------------------------------------------------ UPDATE ... SET col = literal
 *
 * Device half of set_field_to_value() (reference src/engine/executor_update.c:394-433): for every
 * selected row (idx[k], or all rows 0..n-1 when idx == NULL) dst[row] = value_bits, and the row's bit
 * in dst_nullbits is set (set_null != 0; the cell keeps its old bytes, as upstream) or cleared.
 * dst_nullbits may be NULL when set_null == 0 (a column that has never held a NULL). */
int mdb_dev_scatter_set64(mdb_dev_ctx *ctx, void *dst, uint64_t *dst_nullbits, const uint32_t *idx, uint64_t n,
			  int64_t value_bits, int set_null);

/* ------------------------------------------------------------------ ORDER BY
 *
 * The reference's grammar and semantic phase accept ORDER BY (src/parser/midorisql.y:183-191,
 * src/parser/semantic_select.c:1895-1928) but executor_run_select_stmt() never looks at the node
 * (SURVEY.md 8a D7); this is the device operator of the extension (8f row 4).
 * perm_out[k] = position in the stream (0..n-1) of the row that comes k-th when the stream is sorted by
 * keys[0], then keys[1], ...; ties keep the stream order (stable).  Row i of the stream reads
 * values[rid ? rid[i] : i]; NULL sorts before every value (ASC: first, DESC: last); DOUBLE keys follow
 * IEEE order with -0.0 before +0.0.  Synchronous. */
#define MDB_SORT_MAX_KEYS 16
struct mdb_sort_key {
	const void *values;		/* int64_t[] or double[] */
	const uint64_t *nullbits;	/* or NULL */
	const uint32_t *rid;		/* or NULL = identity */
	int32_t type;			/* enum mdb_valtype */
	int32_t desc;			/* 0 ASC, 1 DESC */
};
int mdb_dev_sort_perm(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *perm_out);
/* ORDER BY ... LIMIT: perm_out[0..min(k, n)) = the first k entries of the permutation mdb_dev_sort_perm() would deliver,
 * without sorting the table: a threshold from a sample of the first key, one filter pass that keeps the rows at or before
 * it, the stable sort of those candidates only (*out_candidates, when not NULL: how many rows were sorted - n when the
 * operator fell back to the full sort: small tables, k above an eighth of the rows, a first key with few distinct
 * values).  Synchronous. */
int mdb_dev_topk_perm(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint64_t k, uint32_t *perm_out,
		      uint64_t *out_candidates);

/* SELECT DISTINCT (midorisql.y:203, equally parsed-but-ignored upstream): out_sel[0..*out_count) = ascending
 * stream positions of the FIRST occurrence of every distinct combination of the key columns (NULL equals NULL,
 * DOUBLE compared by bits; `desc` is ignored).  out_sel has room for n entries.  Synchronous. */
int mdb_dev_distinct_sel(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *out_sel,
			 uint64_t *out_count);

/* GROUP BY col_1, ..., col_k + COUNT(*) with composite-key semantics (rows are one group when they agree on every
 * column, NULL = NULL): the reference instead applies its single-field loop once per field
 * (executor_select.c:1537-1541), which is not a grouping by the combination - see DESIGN.md 2.  Output as
 * mdb_dev_group_count(): out_first[g] = stream position of the group's first row (ascending), out_count[g].
 * Columns whose value ranges fit 63 bits together are ONE key for mdb_dev_group_count's forms (round 6): built from the
 * columns as they are loaded where the composite has at most 14 or 18 ... 25 bits and there is no row-id vector, written
 * as a composite column otherwise; wider combinations, or more than four columns: a stable sort of the stream and its
 * run heads.  mdb_dev_distinct_sel() above takes the same road. */
int mdb_dev_group_count_multi(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *out_first,
			      int64_t *out_count, uint64_t cap, uint64_t *out_groups);

/* ------------------------------------------------------------------ INNER JOIN (materialising)
 *
 * Replaces _join_nested_loop_tbl2tbl() for ON l = r (reference
 * executor_select.c:1076-1149) and, applied twice, the intended semantics of the
 * recursive join (:1151-1280; the reference's own tbl2mat is defective - SURVEY 8a D2).
 *
 * keys_l[n_l], keys_r[n_r]: 8-byte join keys (compared bit-wise; INT64 or DOUBLE bits)
 * with optional NULL bits; a NULL key never matches (:557-579).  Output: the pairs
 * (pos_l, pos_r) with keys_l[pos_l] == keys_r[pos_r], ordered by (pos_l, pos_r) -
 * the reference's left-major / right-minor emission order.  *out_l / *out_r are
 * device arrays allocated by the call (free with mdb_dev_free); *out_count their
 * length.  Synchronises.
 */
int mdb_dev_join_pairs(mdb_dev_ctx *ctx,
		       const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
		       const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r,
		       uint32_t **out_l, uint32_t **out_r, uint64_t *out_count);

/* The same join when its ONLY output is the join key column - SELECT * over two key columns (BASELINE configs[3]), SELECT id_a,
 * id_b ... ON id_a = id_b: both sides hold the same value in every joined row, so no row has to be identified.  The any-order
 * join + GROUP BY pipeline (first-level regions of 2-byte words, no row ids, no ordering sort) counts every key's partners and the
 * key is written COUNT times: *out_key = device array allocated by the call (mdb_dev_free) of *out_rows joined keys, in
 * UNSPECIFIED order (equal keys adjacent) - for callers that ask for no order (mdb_database_groups_any_order, the sharded
 * mdb_dist_join_pairs).  10^8 x 10^8 unique keys: 1.2 ms where mdb_dev_join_pairs + the key gather take 3.7.  Synchronises. */
int mdb_dev_join_keys(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
		      const uint64_t *null_r, uint64_t n_r, int64_t **out_key, uint64_t *out_rows);
/* ... and in the REFERENCE's row order when no key that has partners occurs twice in the LEFT table (a primary key joined with another, or
 * with its foreign keys): every such key COUNT times at its left row's place - what mdb_dev_join_pairs + a gather of the key column deliver
 * (10^8 x 10^8 unique keys: 2.1 ms instead of 3.7).  Found out by running the ordered join + GROUP BY + COUNT(*) operator: J = G, or its
 * direct-address leaf kernels saw no key with several left rows.  *served = 0 and nothing allocated otherwise (remembered for these
 * columns): the caller takes mdb_dev_join_pairs.  *served = 2 (round 6): every left row joined exactly one right row - *out_key IS keys_l
 * (nothing allocated, nothing copied: do not free it).  Synchronises. */
int mdb_dev_join_keys_ordered(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			      const uint64_t *null_r, uint64_t n_r, int64_t **out_key, uint64_t *out_rows, int *served);

/* The join of a foreign key to a primary key with the right table's payload: when EVERY left row has exactly one partner (unique
 * right keys, referential integrity - verified on the device, not assumed) the joined rows are the left rows in their own order,
 * and all a projection needs of the right table are its payload cells in that order.  Up to two 8-byte payload columns of the
 * right table (pay_in[c]: n_r cells each, no NULL bitmap) travel through its one partition level beside the key and are written to
 * out[c][i] = cell of left row i's partner (caller buffers of n_l cells) - instead of a scatter of partner row ids, a compaction and a
 * random 8-byte gather per cell (BASELINE configs[1], 10^7 x 10^7 rows, SELECT *: 0.58 -> 0.4 ms).  Returns MIDORIDB_OK when served;
 * 1 when it is not such a join - a left row without partner (NULL keys included), duplicate right keys, keys outside every 2^24-value
 * window, fewer than 2^20 rows in all: nothing usable was written, mdb_dev_join_pairs answers - or a negative error code.  Synchronises.
 * (reference: _join_nested_loop_tbl2tbl + cpy_cols / _merge_rows, src/engine/executor_select.c:1076-1149, 340-438) */
int mdb_dev_join_payload(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			 const uint64_t *null_r, uint64_t n_r, const void *const *pay_in, int npay, void *const *out);

/* ... of ONE left key column with several right tables on that key (SELECT * FROM A JOIN B ON A.k = B.k JOIN C ON A.k = C.k: BASELINE
 * configs[4]'s join-only form; reference: _join_nested_loop_tbl2tbl then _join_nested_loop_tbl2mat, src/engine/executor_select.c:1076-1232):
 * right[t].out[c][i] = payload cell c of left row i's partner in table t.  Left tables of 2^24 rows and more, at most four payload columns
 * over all tables, no NULL keys; [key_min, key_max] is the caller's bound on every key of every table (the catalog's ranges; at most 2^27
 * values) - verified on the device like every property of the join: MIDORIDB_OK when served, 1 when not (a key outside the bound, a left
 * row without partner in some table, a right key twice, a smaller table ...: nothing usable was written - the caller joins table by table
 * with mdb_dev_join_payload / mdb_dev_join_pairs), or a negative error code.  The left table is sorted once, one leaf launch and one
 * placement pass serve all the tables (round 6).  Nothing is sampled or remembered.  Synchronises. */
struct mdb_dev_payload_right {
	const int64_t *keys;
	const uint64_t *nulls;	/* NULL bitmap of the key column or NULL (a nullable right key column is not served) */
	uint64_t rows;
	int npay;		/* 1 or 2 */
	const void *pay_in[2];	/* rows cells of 8 bytes each, no NULL bitmap */
	void *out[2];		/* caller buffers of n_l cells */
};
int mdb_dev_join_payload_multi(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const struct mdb_dev_payload_right *right,
			       int nright, int64_t key_min, int64_t key_max);

/* Cross join (FROM A, B  ==  JOIN ... ON 1=1, reference optimiser_select.c:395-464):
 * all n_l * n_r pairs in (l, r) order, into caller buffers of that capacity. */
int mdb_dev_cross_pairs(mdb_dev_ctx *ctx, uint64_t n_l, uint64_t n_r, uint32_t *out_l, uint32_t *out_r);

/* ------------------------------------------------------------------ GROUP BY key + COUNT(*)
 *
 * Replaces proc_groupby_clause() + cmp_rows_col_mattbl() + inc_count_cols()
 * (reference executor_select.c:1526-1588, 1465-1524) for one group field:
 * groups of equal key (NULL keys form one group, :1477-1482), COUNT(*) per group.
 * flags & MDB_ORDER_FIRST: groups come out in first-occurrence order like the
 * reference's survivors; otherwise in unspecified order (still deterministic).
 *
 * out_first[g] = position of the group's first tuple, out_count[g] = COUNT(*).
 * Caller buffers of capacity `cap` groups (n is always enough); *out_groups = G.
 * Synchronises.
 */
#define MDB_ORDER_FIRST 1u
/* The caller can do without copies (round 6).  MDB_KEYS_MAY_ALIAS: when every left row turns out to be a group (*out_groups == n_l, in row order), the
 * group keys ARE the left key column - out_key is then NOT written and mdb_dev_last_plan().keys_are_left_column = 1: read keys_l instead (query_execute()
 * lets the result hold the table's buffer).  MDB_COUNTS_OPTIONAL: when every COUNT(*) is 1, out_count is NOT written and plan.counts_all_one = 1.
 * Both only where the operator knows it for free (the bit-per-row form of the 4096-digit join: two primary keys, a key and its foreign keys); at
 * 10^8 x 10^8 rows 0.3 ms of 1.6 each.  Without the flags every output is written, as before. */
#define MDB_KEYS_MAY_ALIAS 2u
#define MDB_COUNTS_OPTIONAL 4u
int mdb_dev_group_count(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n,
			uint32_t flags, uint32_t *out_first, int64_t *out_count, uint64_t cap,
			uint64_t *out_groups);

/* The same grouping as (key, COUNT(*)) PAIRS in unspecified order, for statements that select nothing but the group key and
 * COUNT(*) and ask for no order (mdb_database_groups_any_order): no row ids, no ordering sort - one partition pass of 2-byte
 * words and a counting pass (10^8 rows, 6.25 * 10^6 groups: 0.35 ms instead of 0.84).  out_key / out_count: caller buffers of
 * capacity cap.  Returns 1 - and writes nothing - when it does not serve the column (NULL keys present: pass nullbits only when
 * there are any; keys beyond every 2^30-value window; skew; fewer than 2^21 rows): mdb_dev_group_count() then answers. */
int mdb_dev_group_count_keys(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t *out_key,
			     int64_t *out_count, uint64_t cap, uint64_t *out_groups);

/* ------------------------------------------------------------------ fused north-star pipeline
 *
 * SELECT l.key, COUNT(*) FROM L INNER JOIN R ON l.key = r.key GROUP BY l.key
 * (reference tests/engine/executor_select.c:348-378; executor_select.c:1076-1149 +
 * 1526-1588) without materialising the joined rows: per distinct non-NULL key k
 * present on both sides, COUNT(*) = |{l: key=k}| * |{r: key=k}|.
 *
 * out_key[g], out_count[g] (and optionally out_first[g] = first L position of the
 * key, may be NULL): caller buffers of capacity `cap` (n_l is always enough).
 * With MDB_ORDER_FIRST the groups are in the reference's order (first occurrence in
 * L-major join order).  Without it - and without out_first - the order is unspecified and the operator may
 * skip everything that order costs: no row id travels through the partition levels and no ordering sort runs
 * (10^8 x 10^8 unique keys: 1.2 ms instead of 2.6; bit 11 of mdb_dev_last_join_filter() says this form ran).  *out_groups = G, *out_joined = number of joined rows
 * (sum of counts).  Synchronous: G, J and the overflow flags come back before the groups are ordered
 * (the ordering sort is sized by G), completion after it.  Size limit per call: about 7*10^8 left rows
 * (beyond that the tables must be sharded, see mdb_dev_partition_by_dest).
 */
int mdb_dev_join_group_count(mdb_dev_ctx *ctx,
			     const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
			     const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r,
			     uint32_t flags,
			     int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap,
			     uint64_t *out_groups, uint64_t *out_joined);

/* Chaining the fused operator over several tables joined on ONE key (A JOIN B ON a=b JOIN C ON a=c ... GROUP BY a,
 * BASELINE configs[4]): run it on (A, B), then on (the group keys it returned, C) with out_first, and combine -
 * a group of the second run at position k stems from group idx[k] of the first, so COUNT(*) = cnt1[idx[k]] * cnt2[k]
 * and its first left row is first1[idx[k]] (first1 == NULL: idx[k] itself).  *out_sum = sum of the combined counts
 * (= rows of the full join).  The joined rows of neither join are ever materialised.  Synchronises. */
int mdb_dev_combine_counts(mdb_dev_ctx *ctx, const int64_t *cnt1, const uint32_t *first1, const uint32_t *idx, const int64_t *cnt2,
			   uint64_t n, int64_t *out_cnt, uint32_t *out_first, uint64_t *out_sum);

/* The same shape in ONE operator: left table L and n_right (1 ... 3) right tables, all joined on one key -
 * SELECT l.key, COUNT(*) FROM L JOIN R0 ON l.key = r0.key JOIN R1 ON l.key = r1.key ... GROUP BY l.key (reference:
 * recursive join executor_select.c:1151-1280 + GROUP BY :1526-1588; BASELINE configs[4]).  COUNT(*) of a key = its rows in L
 * x its rows in R0 x its rows in R1 ...; outputs exactly as mdb_dev_join_group_count() (first-occurrence order with MDB_ORDER_FIRST - any order, cheaper, without it and without out_first; out_first =
 * first L position, *out_joined = rows of the full join).  Every table is partitioned once and the groups ordered once when the
 * keys take the compact narrow form; otherwise the call chains the two-table operator itself.  keys_r / null_r / n_r:
 * HOST arrays of n_right entries (null_r may be NULL, or hold NULLs).  Synchronous. */
int mdb_dev_join_group_count_multi(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, int n_right,
				   const int64_t *const *keys_r, const uint64_t *const *null_r, const uint64_t *n_r, uint32_t flags,
				   int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap, uint64_t *out_groups,
				   uint64_t *out_joined);

/* Split form for pipelines that receive the two tables at different times (the multi-GPU exchange):
 * _begin() hashes and partitions the LEFT table and returns without a host sync, so the work overlaps
 * whatever is still in flight (e.g. the right table's all-to-all); n_r_max is the expected bound of the right
 * table's size, for scratch sizing.  _finish() takes the right table and completes exactly like
 * mdb_dev_join_group_count() - also when the table turns out larger than announced (skewed keys sent this GPU
 * more than its share): the prepared work is then dropped and the operator runs once more on the real sizes.
 * No other operator may run on the context in between. */
int mdb_dev_join_group_count_begin(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
				   uint64_t n_r_max);
int mdb_dev_join_group_count_finish(mdb_dev_ctx *ctx, const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r,
				    uint32_t flags, int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap,
				    uint64_t *out_groups, uint64_t *out_joined);

/* The same operator over int32 key columns - what the other GPUs' keys look like on arrival when they crossed xGMI in
 * the 4-byte wire format (mdb_dev_partition_by_dest with keys32): the first partition level reads the 4-byte keys
 * directly, no widening pass.  No NULL bitmaps (NULL keys never leave their GPU); 8-byte aligned columns.
 * out_key[] is int64 like everywhere else. */
int mdb_dev_join_group_count_i32(mdb_dev_ctx *ctx, const int32_t *keys_l, uint64_t n_l, const int32_t *keys_r, uint64_t n_r,
				 uint32_t flags, int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap,
				 uint64_t *out_groups, uint64_t *out_joined);
int mdb_dev_join_group_count_begin_i32(mdb_dev_ctx *ctx, const int32_t *keys_l, uint64_t n_l, uint64_t n_r_max);
int mdb_dev_join_group_count_finish_i32(mdb_dev_ctx *ctx, const int32_t *keys_r, uint64_t n_r, uint32_t flags, int64_t *out_key,
					int64_t *out_count, uint32_t *out_first, uint64_t cap, uint64_t *out_groups,
					uint64_t *out_joined);

/* ------------------------------------------------------------------ multi-GPU shuffle support
 *
 * Hash-partition a key column by destination GPU for the all-to-all exchange
 * (SURVEY 8e): dest = hash(key) mod n_dest, NULL keys are dropped (they never
 * join).  out_keys (capacity n) receives the keys grouped by destination (order inside a
 * destination is unspecified); out_rid (capacity n, or NULL) the source row of every out_keys
 * entry, which is what carries payload columns and row identity through the exchange
 * (send column = mdb_dev_gather64(column, out_rid)); out_counts (HOST, n_dest entries) the
 * group sizes.  keys32 != 0: out_keys is an int32_t[] - the 4-byte wire format for key columns whose
 * statistics (mdb_dev_key_range) say that every key fits; it halves the bytes on xGMI, which is what bounds
 * the exchange; the receiver widens with mdb_dev_widen32to64.  Synchronises.
 */
int mdb_dev_partition_by_dest(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n,
			      uint32_t n_dest, int keys32, void *out_keys, uint32_t *out_rid, uint64_t *out_counts);
/* ... _pruned: rows whose key lies outside [keep_lo, keep_hi] are dropped as well - the other table's GLOBAL key range, which
 * the ranks know before the exchange (mdb_dev_key_range + one tiny exchange, or catalog statistics): a row outside it joins
 * nothing on any GPU, so it need not cross xGMI (min-max pruning before the shuffle; mdb_dist_join_group_count does this).
 * [own_lo, own_hi]: the range promised for THIS column (what the other table is pruned with): a key outside it is an error,
 * never a silently wrong result.  INT64_MIN / INT64_MAX: no bound. */
int mdb_dev_partition_by_dest_pruned(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, uint32_t n_dest, int keys32,
				     int64_t keep_lo, int64_t keep_hi, int64_t own_lo, int64_t own_hi, void *out_keys, uint32_t *out_rid,
				     uint64_t *out_counts);
/* column statistics: smallest / largest non-NULL key (min > max when there is none).  Synchronises. */
int mdb_dev_key_range(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t *out_min,
		      int64_t *out_max);
int mdb_dev_widen32to64(mdb_dev_ctx *ctx, const int32_t *src, uint64_t n, int64_t *dst);
/* "no key twice": every non-NULL key of rows [0, n) sets its bit (key - window_lo) in `seen` (a device bitmap of window_bits bits the caller
 * owns - cleared by the caller before the column's first rows, kept for the rows appended later); *out_twice = 1 when a bit was already set
 * (a value twice, among these rows or against the rows scanned into `seen` before) or a key lies outside the window (nothing is known then),
 * else 0.  One scattered atomic per row: ~1.3 ms per 10^8 rows, off the query path.  Synchronises. */
int mdb_dev_distinct_scan(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t window_lo, uint64_t window_bits,
			  uint32_t *seen, int *out_twice);

/* ------------------------------------------------------------------ synthetic data (bench / tests)
 * keys[i] = perm(i) mod modulus, perm = the bijection on [0, n) defined in
 * include/mdb_gen.h (affine map modulo a prime with cycle walking, seeded). */
int mdb_dev_gen_keys(mdb_dev_ctx *ctx, int64_t *keys, uint64_t n, uint64_t first_index, uint64_t domain,
		     uint64_t seed, uint64_t modulus);

/* payload columns of the synthetic tables: cell i = the (first_index + i)-th output of SplitMix64(seed),
 * kind 0: INT64 = z >> 33; kind 1: DOUBLE = (double)(z >> 11) * 2^-53 */
int mdb_dev_gen_payload(mdb_dev_ctx *ctx, void *out, uint64_t n, uint64_t first_index, uint64_t seed, int kind);

#ifdef __cplusplus
}
#endif
#endif /* MDB_DEV_H */
