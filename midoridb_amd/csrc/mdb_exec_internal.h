/*
 * mdb_exec_internal.h - what the files of the executor share (mdb_exec.c: SELECT lowering; mdb_exec_resolve.c: plan normalisation and
 * the semantic checks; mdb_exec_pred.c: predicate programs; mdb_exec_shard.c: sharded mode; mdb_exec_tail.c: HAVING / DISTINCT /
 * ORDER BY / LIMIT; mdb_exec_dml.c: CREATE / INSERT / DELETE / UPDATE).  Nothing here is part of the library's interface.
 */
#ifndef MDB_EXEC_INTERNAL_H
#define MDB_EXEC_INTERNAL_H

#include "mdb_host.h"
#include <time.h>

#define ERR(...) snprintf(err, errlen, __VA_ARGS__)

#pragma GCC visibility push(hidden)

/* ------------------------------------------------------------------ device-side execution state */

struct dbuf_list {
	void **p;
	int n, cap;
};

struct where_split;

struct exec {
	struct mdb_catalog *cat;
	mdb_dev_ctx *dev;
	struct mdb_select *s;
	char *err;
	size_t errlen;
	struct dbuf_list bufs;
	uint32_t *rid[MDB_MAX_TABS];	/* per FROM table: row-id vector of the current stream or NULL = identity */
	bool have_stream;		/* false until the first table is in the stream */
	uint64_t n;			/* stream length */
	int64_t *d_count;		/* COUNT(*) column of the stream (after GROUP BY), device */
	bool fused;			/* north-star plan: the stream is (d_fused_key, d_count), no row ids */
	int64_t *d_fused_key;
	uint64_t joined_rows;
	/* per joined table: its equi-join key column holds, in every tuple of the stream, the value of an earlier table's column
	 * (INT64-represented types: the join compared all 64 bits; never NULL - a NULL key joins nothing): the projection reads
	 * that column, through the earlier table's row ids (ascending after a join: near-sequential reads instead of a random gather) */
	int same_col[MDB_MAX_TABS], same_as_tbl[MDB_MAX_TABS], same_as_col[MDB_MAX_TABS];
	/* sharded mode (cat->dist): a FROM table whose rows were exchanged is read through a SHADOW table - the same schema over the
	 * columns this rank received - that stands in s->tabs[t].t for the rest of the statement (orig_tab[] puts the catalog's
	 * tables back at the end).  part[]: the fields whose value the current stream is hash-partitioned by (all equal in every
	 * tuple: the equi-join keys tied together so far); need[t][c]: the statement reads column c of table t */
	struct mdb_table *shadow[MDB_MAX_TABS], *orig_tab[MDB_MAX_TABS];
	const struct mdb_expr *part[2 * MDB_MAX_TABS];
	int npart;
	bool promised;		/* the exchange handle holds this statement's key ranges (shard_promise_ranges) */
	bool dict_synced;	/* the ranks' string dictionaries were made known to each other for this statement (shard_dict_sync) */
	bool need[MDB_MAX_TABS][MDB_MAX_COLS];
	const struct where_split *ws;		/* the WHERE conjuncts pushed down to single tables (general plan; NULL: none are) */
	int joins_eliminated;			/* tables that were not joined at all: the catalog said every row of the stream has exactly one partner (join_next_table) */
	bool joined_ahead[MDB_MAX_TABS];	/* table t was joined together with an earlier table on the same key (join_with_payload_multi) */
};

struct pred_prog {
	struct mdb_pred_insn insn[MDB_PRED_MAX_INSNS];
	int n;
	struct mdb_col_binding cols[MDB_PRED_MAX_SLOTS];
	int slot_tbl[MDB_PRED_MAX_SLOTS], slot_col[MDB_PRED_MAX_SLOTS];
	int ncols;
};

#define PUSH_MAX 16
#define PUSH_TABS MDB_MAX_TABS
struct where_split {
	const struct mdb_expr *push[PUSH_TABS][PUSH_MAX];
	int npush[PUSH_TABS];
	const struct mdb_expr *residual[64];
	int nresidual;
};

extern __thread const struct mdb_strdict *stmt_dict;	/* the statement's string dictionary (VARCHAR literals) */

bool field_eq(const struct mdb_expr *a, const struct mdb_expr *b);
int resolve_expr(struct mdb_select *s, struct mdb_expr *e, char *err, size_t errlen);
bool is_having_clause(const char *clause);
int operand_type(const struct mdb_expr *o, const struct mdb_expr *other, bool dml);
int check_predicate_x(const struct mdb_expr *e, const char *clause, bool dml, char *err, size_t errlen);
int check_predicate(const struct mdb_expr *e, const char *clause, char *err, size_t errlen);
bool expr_has_count(const struct mdb_expr *e);
int fields_in_select_list(const struct mdb_select *s, const struct mdb_expr *e, const char *clause, char *err, size_t errlen);
int resolve_select(struct mdb_catalog *cat, struct mdb_select *s, char *err, size_t errlen);
int dev_fail(struct exec *x, const char *what);
int track(struct exec *x, void *p);
void *dalloc(struct exec *x, size_t bytes);
void free_all(struct exec *x);
int stream_select(struct exec *x, int ntabs_in_stream, const uint32_t *sel, uint64_t n_new);
int stream_apply_sel(struct exec *x, int ntabs_in_stream, const uint32_t *sel, uint64_t n_new);
int stream_column(struct exec *x, const struct mdb_expr *f, const int64_t **vals, const uint64_t **nulls);
int double_join_keys(struct exec *x, const struct mdb_column *col, const uint32_t *rid, uint64_t n, const void **vals, const uint64_t **nulls);
void mark_needed(struct exec *x, const struct mdb_expr *e);
void mark_needed_all(struct exec *x);
bool in_part(const struct exec *x, const struct mdb_expr *f);
int shard_dict_sync(struct exec *x);
int shard_ids(struct exec *x, const int64_t *cells, uint64_t n, bool to_common, bool in_place, const int64_t **out);
#define SHARD_BROADCAST (1u << 30)	/* shard_rows flag: every rank's rows to every rank (mdb_dist_broadcast_rows) instead of rows by key */
int shard_rows(struct exec *x, const int *tabs, int nt, uint32_t *const *rid_of, uint64_t n, const int64_t *kv, const uint64_t *kn, uint32_t flags, uint64_t *n_out);
int shard_stream(struct exec *x, int nt, const struct mdb_expr *f, uint32_t flags);
int shard_promise_ranges(struct exec *x, const struct mdb_expr *fl, const struct mdb_expr *fr);
void shard_cleanup(struct exec *x);
void bind_operand(struct exec *x, const struct mdb_expr *f, const void **values, const uint64_t **nullbits, const uint32_t **rid);
int pred_slot(struct exec *x, struct pred_prog *p, const struct mdb_expr *f);
int pred_emit(struct pred_prog *p, int op, int cmp, int type, int a, int b, int64_t imm);
int64_t lit_bits_for(const struct mdb_expr *v, int coltype);
bool const_cmp(int op, const struct mdb_expr *l, const struct mdb_expr *r);
int pred_compile(struct exec *x, struct pred_prog *p, const struct mdb_expr *e);
int stream_filter(struct exec *x, int ntabs_in_stream, const struct mdb_expr *e);
void collect_conjuncts(struct mdb_expr *e, struct mdb_expr **out, int *n, int cap);
uint64_t expr_tables(const struct mdb_expr *e);
bool where_split(const struct mdb_select *s, struct where_split *w);
int table_filter(struct exec *x, int t, const struct mdb_expr *const *conj, int nconj, const uint32_t **sel, uint64_t *m);
int table_column(struct exec *x, int t, const struct mdb_expr *key, const uint32_t *sel, uint64_t m, const void **vals, const uint64_t **nulls);
int fused_operand(struct exec *x, int t, const struct mdb_expr *key, const struct mdb_expr *const *conj, int nconj, const void **vals, const uint64_t **nulls, uint64_t *n);
int join_with_payload(struct exec *x, int t, const struct mdb_expr *kr, const int64_t *vl, const uint64_t *nl, const void *vr, const uint64_t *nr, uint64_t r_rows);
int join_next_table(struct exec *x, int t, const struct mdb_expr *const *pconj, int npconj);
int result_column_to_host(mdb_dev_ctx *dev, struct mdb_result *res, int c, const void *d_vals, const uint64_t *d_nulls);
double now_ms(void);
int fused_chain(struct mdb_select *s, const struct mdb_expr **keys, bool count_only);
bool keys_only_join(const struct mdb_select *s, const struct mdb_catalog *cat, int has_count, const int *key_tbl, const int *key_col, const int *src, int ncols, const struct mdb_expr **kj);
int select_tail(struct exec *x, int has_count);
int dml_value_on_left(const struct mdb_expr *e);
int dml_check_values(const struct mdb_expr *e, char *err, size_t errlen);
int dml_select_rows(struct mdb_catalog *cat, struct mdb_dml *d, struct exec *x, struct mdb_select *s, struct mdb_from_tab *tab, bool complement, struct mdb_table **out_t, const uint32_t **sel, uint64_t *m, char *err, size_t errlen);

#pragma GCC visibility pop
/* the catalog's statistics of an operator call's key columns (mdb_exec.c) */
void op_stats_begin(struct exec *x, const struct mdb_expr *fl, const void *pl, const struct mdb_expr *fr, const void *pr);
void op_stats_end(struct exec *x);
uint64_t op_groups_bound(struct exec *x, const struct mdb_expr *f, uint64_t rows);

#endif
