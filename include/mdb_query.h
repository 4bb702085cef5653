/*
 * mdb_query.h - the public C API of libmidoridb_amd.so: a drop-in for the reference's
 * <engine/query.h> + <engine/database.h> on the SELECT path.
 *
 * The six entry points keep the reference's names, signatures, return conventions and
 * ownership rules, so a program written against MidoriDB (reference README.md:39-81,
 * tests/engine/executor_select.c) recompiles against this header unchanged:
 *
 *   database_open()        reference include/engine/database.h:26,  src/engine/database.c:10
 *   database_close()       reference include/engine/database.h:32,  src/engine/database.c:30
 *   query_execute()        reference include/engine/query.h:42,     src/engine/query.c:35
 *   query_cur_step()       reference include/engine/query.h:44-51,  src/engine/query.c:108
 *   query_column_int64()   reference include/engine/query.h:53-61,  src/engine/query.c:148
 *   query_free()           reference include/engine/query.h:63-69,  src/engine/query.c:169
 *
 * Behind them the executor is the MI355X path (include/mdb_dev.h); there is NO CPU
 * executor in this library: a SELECT fails with status ST_ERROR when no HIP device is
 * usable.  Struct layouts keep the reference's field order and sizes (callers allocate
 * struct database themselves, zero-initialised); the pointed-to objects are this
 * library's columnar tables instead of the reference's row-store.
 */
#ifndef MDB_QUERY_H
#define MDB_QUERY_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#include <pthread.h>
#include "mdb_error.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference include/engine/database.h:15-18 */
struct database {
	void *tables;			/* reference: struct hashtable *; here: the catalog */
	pthread_mutex_t mutex;
};

/* reference include/engine/query.h:15-22 */
enum query_output_status {
	ST_OK_WITH_RESULTS,		/* SELECT */
	ST_OK_EXECUTED,			/* CREATE, INSERT */
	ST_ERROR
};

/* reference include/engine/query.h:24-28.  `table` points at the columnar result, which BEGINS with the reference's
 * `struct table` layout (include/mdb_legacy.h: columns always, datablocks of rows for results of up to 4096 rows);
 * `cursor_blk` is non-NULL once stepping has started; `cursor_offset` is the current row. */
struct result_set {
	void *table;
	void *cursor_blk;
	size_t cursor_offset;
};

struct query_output_error {
	char message[1024];
};

/* reference include/engine/query.h:34-40 */
struct query_output {
	enum query_output_status status;
	struct result_set results;
	struct query_output_error error;
	size_t n_rows_aff;
};

int database_open(struct database *db);
void database_close(struct database *db);

/* Executes one SQL statement (SELECT / CREATE TABLE / INSERT ... VALUES).  Never returns an
 * error code: the returned object carries status + error.message; NULL only on allocation
 * failure (reference src/engine/query.c:44-46, 95-105). */
struct query_output *query_execute(struct database *db, char *query);

/* MIDORIDB_ROW (4) while a row is current, MIDORIDB_OK (0) at the end - for results of ANY
 * size (the reference's own cursor breaks past one 4 KiB datablock).  A result kept on the
 * device (mdb_database_results_on_device) is copied to the host by the first step: -MIDORIDB_INTERNAL when
 * that copy fails (no row is current; a `== MIDORIDB_ROW` loop ends as it does at the end of the result). */
int query_cur_step(struct result_set *res);
int64_t query_column_int64(struct result_set *res, int col_idx);
void query_free(struct query_output *output);

/* ------------------------------------------------------------------ extensions (not in the reference) */

/* Same as query_execute() but takes the parser's output instead of SQL text: the RPN token
 * strings of the reference grammar (src/parser/midorisql.y:517-528), one per line.  This is
 * the seam a MidoriDB build with its own bison/flex front end binds to. */
struct query_output *mdb_query_execute_rpn(struct database *db, const char *rpn_lines);

/* The reference only has query_column_int64(); for a DOUBLE column it returns the raw bits
 * (src/engine/query.c:162-166).  These add typed access, NULL tests and metadata. */
double query_column_double(struct result_set *res, int col_idx);
bool query_column_is_null(struct result_set *res, int col_idx);
/* VARCHAR column of the current row: the string (owned by the database, valid until database_close()); "" for a NULL
 * cell (upstream such a cell points at an empty string; query_column_is_null() tells them apart), NULL for another column type.  query_column_int64() of such a column returns the string's id in the database's
 * string dictionary (upstream: the bits of a heap pointer).  mdb_result_text_at(): the same for any row. */
const char *query_column_text(struct result_set *res, int col_idx);
const char *mdb_result_text_at(struct result_set *res, int col_idx, uint64_t row);
int query_column_count(struct result_set *res);
const char *query_column_name(struct result_set *res, int col_idx);	/* "T.col" / "COUNT(*)" */
int query_column_type(struct result_set *res, int col_idx);		/* reference enum COLUMN_TYPE values */
uint64_t query_row_count(struct result_set *res);
/* Whole result column at once (8-byte values, row order); NULL rows hold 0. */
const int64_t *query_column_data(struct result_set *res, int col_idx);
/* Device-pipeline milliseconds of the SELECT that produced `res`, and the rows its joins produced before
 * aggregation (after the WHERE conjuncts that were pushed below the joins; 0 when the query has no join). */
double query_exec_ms(struct result_set *res);
uint64_t query_joined_rows(struct result_set *res);

/* Bulk columnar ingest: append n rows; cols[c] = n 8-byte values for
 * column c of the table (INT64 values or double bits), nulls[c] = n bytes (1 = NULL) or NULL. */
int mdb_table_append_columns(struct database *db, const char *table, int ncols, uint64_t n,
			     const int64_t *const *cols, const uint8_t *const *nulls);
/* Fill a one-or-more-column INTEGER table with synthetic keys directly on the device (bench):
 * column c gets perm_{seed+c}(i) mod modulus[c] (include/mdb_gen.h), i in [0, n). */
int mdb_table_generate(struct database *db, const char *table, uint64_t n, uint64_t seed, const uint64_t *modulus);

/* Results that stay on the device: after mdb_database_results_on_device(db, 1) a SELECT returns as soon as its result columns
 * exist in HBM - no device-to-host copy inside query_execute() (for the north-star query at 10^8 rows that copy is 3/4 of
 * the call).  query_column_data_device() hands out the device column (8-byte cells, query_row_count() of them, READ-ONLY, valid until
 * query_free(); two result columns that hold the same values by construction - both key columns of `SELECT *` over an equi-join - may
 * be one buffer); the first query_cur_step() / query_column_data() / ... on such a result copies the columns to the host,
 * once, and from then on everything behaves as usual.  Such a result must be released (query_free) BEFORE database_close():
 * its columns live in the database's device context. */
int mdb_database_results_on_device(struct database *db, int on);
const void *query_column_data_device(struct result_set *res, int col_idx);	/* NULL when the column is not on the device */
/* ... and its NULL flags: a device bitmap of query_row_count() bits (bit set = the cell is NULL - the reference's polarity,
 * include/primitive/row.h - whose 8-byte cell then holds whatever the table stored), or NULL when no cell of the column is NULL
 * (or the column is not on the device).  The host path reports the same flags through query_column_is_null(). */
const uint64_t *query_column_nulls_device(struct result_set *res, int col_idx);

/* Group order: the reference emits the groups of a GROUP BY in the order their first row occurs, and so does this library by
 * default.  SQL promises no order without ORDER BY: after mdb_database_groups_any_order(db, 1) a join + GROUP BY + COUNT(*)
 * may return its groups in any order, and then carries no row ids and sorts nothing (mdb_dev_join_group_count without
 * MDB_ORDER_FIRST: 0.56 instead of 0.72 ms for the README query at 10^8 rows, 1.4 instead of 2.6 ms with unique keys).
 * SELECT COUNT(*) over a join, whose result has no order to keep, always runs that way. */
int mdb_database_groups_any_order(struct database *db, int on);
/* Join elimination (round 6): tables of this database's SELECT statements so far that were NOT joined at all, because the catalog's measured
 * statistics said that every row of the other side finds exactly one partner in them (their key column holds no value twice, no NULL, and every
 * value of its range, which covers the other side's) and the statement read nothing of them but that key.  MDB_JOIN_ELIMINATION=0: never. */
unsigned long long mdb_database_joins_eliminated(struct database *db);

/* mdb_table_generate() for one shard of a table spread over several processes: this process holds rows
 * [first_index, first_index + n) of a table of `domain` rows.  INTEGER column c = perm_{seed+c}(i) mod modulus[c] as above;
 * a DOUBLE column c = (double)(splitmix64(seed + c, i) >> 11) * 2^-53 (SURVEY.md 8d C5 payload). */
int mdb_table_generate_shard(struct database *db, const char *table, uint64_t n, uint64_t first_index, uint64_t domain, uint64_t seed,
			     const uint64_t *modulus);

/* Sharded mode set up by the host program instead of the environment (MIDORIDB_WORLD_SIZE / MIDORIDB_RANK /
 * MIDORIDB_DIST_ID_FILE): mdb_database_device() returns the database's device context (created on first use; NULL
 * without a usable HIP device), the host builds an exchange handle for it - mdb_dist_init() with a communicator id it
 * shipped itself, or mdb_dist_init_transport() over its own fabric (include/mdb_dist.h) - and hands it over; the database
 * owns it from then on (destroyed by database_close()).  Every later SELECT is collective. */
struct mdb_dev_ctx;
struct mdb_dist;
struct mdb_dev_ctx *mdb_database_device(struct database *db);
int mdb_database_set_dist(struct database *db, struct mdb_dist *dist);

#ifdef __cplusplus
}
#endif
#endif /* MDB_QUERY_H */
