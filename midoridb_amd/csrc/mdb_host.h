/*
 * mdb_host.h - internals of the host side (plain C, like the reference): SQL/RPN front end,
 * statement plans, columnar tables with device mirrors, the executor that lowers a SELECT onto
 * the mdb_dev_* C-ABI, and the result set behind query_cur_step()/query_column_int64().
 */
#ifndef MDB_HOST_H
#define MDB_HOST_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include "mdb_error.h"
#include "mdb_dev.h"
#include "mdb_dist.h"
#include "mdb_legacy.h"
#include "mdb_query.h"

/* ------------------------------------------------------------------ RPN token queue
 * Same vocabulary as the reference grammar's emit() (reference src/parser/midorisql.y:517-528,
 * SURVEY.md Appendix A). */
struct mdb_rpn {
	char **tok;
	int n, cap;
};
int mdb_rpn_push(struct mdb_rpn *r, const char *tok);
void mdb_rpn_free(struct mdb_rpn *r);
int mdb_rpn_from_text(const char *text, struct mdb_rpn *out);
int mdb_sql_parse(const char *sql, struct mdb_rpn *out, char *err, size_t errlen);

/* ------------------------------------------------------------------ catalog + columnar storage */
#define MDB_MAX_COLS 128		/* reference TABLE_MAX_COLUMNS (include/primitive/table.h:16) */
#define MDB_NAME_LEN 128
#define MDB_MAX_TABS 16			/* FROM tables of one SELECT (per-table state of the executor is sized by it) */

/* numeric values of the reference's enum COLUMN_TYPE (include/primitive/column.h:17-25) */
enum mdb_coltype {
	MDB_CT_VARCHAR = 0,
	MDB_CT_INTEGER = 1,
	MDB_CT_TINYINT = 2,
	MDB_CT_DOUBLE = 3,
	MDB_CT_DATE = 4,
	MDB_CT_DATETIME = 5,
};

struct mdb_column {
	char name[MDB_NAME_LEN];
	int type;			/* enum mdb_coltype.  INTEGER / DOUBLE / DATE / DATETIME (time_t) / TINYINT (0 | 1) cells are 8-byte values
					 * mirrored on the device; a VARCHAR cell is the 8-byte id of its string in the database's string
					 * dictionary (struct mdb_strdict; the reference keeps a heap pointer per cell, src/primitive/column.c:
					 * 255-293): equal strings have equal ids, so =, <>, IN, joins, GROUP BY and DISTINCT over VARCHAR
					 * columns run on the device as INT64 work */
	int precision;			/* VARCHAR(n): n bytes including the NUL (reference column.precision) */
	bool not_null;			/* NOT NULL / PRIMARY KEY (reference column.nullable == false, executor_create.c:37,53) */
	int64_t *data;			/* host copy, 8 bytes per row */
	uint64_t *nullbits;		/* host NULL bits (allocated with the column), bit set = NULL */
	uint64_t null_count;
	/* device mirror */
	void *d_data;
	uint64_t *d_nullbits;		/* NULL when null_count == 0 */
	/* catalog statistics of the device mirror: smallest / largest non-NULL value (st_lo > st_hi: none) as of table generation
	 * st_generation - 1 (0 = never computed; mdb_col_range) - what sharded joins promise the exchange instead of measuring per query and
	 * what every join / GROUP BY operator is handed instead of sampling its key columns (mdb_dev_call_stats) */
	uint64_t st_generation;
	int64_t st_lo, st_hi;
	/* "no non-NULL value twice" - MEASURED on the device over all rows (mdb_dev_distinct_scan), as of table generation dv_generation - 1
	 * (0 = not measured): at ingest, followed through appended rows (the bitmap of the values seen stays on the device while it is small),
	 * looked at again after an UPDATE / DELETE when somebody asks (mdb_col_distinct).  The reference keeps UNIQUE / PRIMARY KEY per
	 * column (include/primitive/column.h:41-46, set by src/engine/executor_create.c:29-58) and never enforces them: declared_unique only
	 * says the column is worth measuring whatever it looks like. */
	uint64_t dv_generation;
	bool dv_distinct;
	bool declared_unique;
	uint32_t *d_seen;		/* device bitmap: bit (value - seen_lo) of every non-NULL value of the rows measured (NULL: not kept) */
	int64_t seen_lo;
	uint64_t seen_bits;
};

struct mdb_table {
	char name[MDB_NAME_LEN];
	int ncols;
	struct mdb_column cols[MDB_MAX_COLS];
	uint64_t nrows, cap;
	uint64_t generation;		/* bumped by every mutation */
	uint64_t dev_generation;	/* generation the device mirror reflects (0 = none) */
	uint64_t dev_rows;
	uint64_t dev_cap;		/* rows the device buffers can hold (appends reuse the spare room) */
	bool device_only;		/* rows were generated on the device; there is no host copy */
};

/* The database's strings, each stored once: id = position + 1 (0 is what lies under a NULL cell, -1 the id of a literal
 * no cell holds - it equals nothing).  Open addressing over FNV-1a; strings are never removed (a DELETE leaves its
 * strings behind: ids must stay valid for the result sets and device mirrors that hold them). */
struct mdb_strdict {
	char **str;
	uint32_t *len;
	uint64_t n, cap;
	uint64_t *slot;			/* id or 0 */
	uint64_t nslots;
};
int64_t mdb_dict_intern(struct mdb_strdict *d, const char *s, size_t len);	/* id >= 1, or 0 when memory runs out */
int64_t mdb_dict_find(const struct mdb_strdict *d, const char *s, size_t len);	/* id or -1 */
const char *mdb_dict_str(const struct mdb_strdict *d, int64_t id);		/* NULL for an id the dictionary never gave out */
void mdb_dict_free(struct mdb_strdict *d);

struct mdb_catalog {
	struct mdb_table **tables;
	int n, cap;
	struct mdb_strdict dict;
	mdb_dev_ctx *dev;		/* created lazily by the first SELECT */
	int dev_rc;			/* sticky result of the lazy creation */
	/* sharded mode (MIDORIDB_WORLD_SIZE > 1 in the environment: one process per GPU, every process holds ITS rows of every
	 * table): the RCCL exchange handle, created with the device context (include/mdb_dist.h) */
	mdb_dist *dist;
	/* ... and the strings: a VARCHAR cell is the id of its string in THIS process's dictionary; cells that cross xGMI travel as ids of
	 * the ranks' COMMON dictionary `gdict`, which every rank builds from the same announcements in the same order (shard_dict_sync,
	 * mdb_exec_shard.c).  l2g[local id] = common id (0: not announced yet), g2l[common id] = local id; d_*: device copies */
	struct mdb_strdict gdict;
	int64_t *l2g, *g2l;
	uint64_t l2g_cap, g2l_cap;
	int64_t *d_l2g, *d_g2l;
	uint64_t d_l2g_n, d_g2l_n;	/* entries the device copies hold */
	uint64_t joins_eliminated;	/* tables of SELECT statements that were not joined at all: the catalog said every row has exactly one partner (mdb_exec.c) */
	bool groups_any_order;		/* mdb_database_groups_any_order(): GROUP BY over a join need not keep first-occurrence order */
	bool results_on_device;		/* mdb_database_results_on_device(): SELECT results stay in HBM until a consumer reads them */
};

/* dst[MDB_NAME_LEN] = src, cut to MDB_NAME_LEN - 1 characters, always terminated */
static inline void mdb_copy_name(char *dst, const char *src)
{
	const size_t n = strnlen(src, MDB_NAME_LEN - 1);
	memcpy(dst, src, n);
	dst[n] = 0;
}

struct mdb_table *mdb_catalog_find(struct mdb_catalog *cat, const char *name);
int mdb_catalog_add(struct mdb_catalog *cat, struct mdb_table *t);
void mdb_catalog_free(struct mdb_catalog *cat);
struct mdb_table *mdb_table_new(const char *name);
void mdb_table_free(struct mdb_table *t, mdb_dev_ctx *dev);
int mdb_table_add_column(struct mdb_table *t, const char *name, int type);
static inline bool mdb_type_is_int64(int type) { return type != MDB_CT_DOUBLE; }	/* how the 8-byte cell compares on the device */
/* time_t of a DATE ('%Y-%m-%d') / DATETIME ('%Y-%m-%d %H:%M:%S') literal, the reference's strptime + mktime
 * (include/primitive/column.h:27-28, src/engine/executor_insert.c:15-40); false when it does not parse */
bool mdb_parse_time(const char *quoted, int type, int64_t *out);
int mdb_table_reserve(struct mdb_table *t, uint64_t rows);
int mdb_table_sync_device(struct mdb_catalog *cat, struct mdb_table *t, char *err, size_t errlen);
int mdb_table_bulk_copy(struct mdb_catalog *cat, struct mdb_table *t, int ncols, uint64_t n, const int64_t *const *cols, bool *mirrored);
void mdb_table_bulk_mirrored(struct mdb_catalog *cat, struct mdb_table *t, uint64_t old_rows, uint64_t old_generation);
/* the value of an MDB_* environment knob, through the library's one reader of the environment (mdb_dev_core.hip: kept per process, mdb_dev_reload_knobs()) */
const char *mdb_knob(const char *name);
bool mdb_col_has_range(const struct mdb_column *col);
/* is the column known to hold no non-NULL value twice, as of now?  Measures (one pass on the device) when nothing is known for this generation and
 * the column is large enough to matter or declared UNIQUE / PRIMARY KEY; false when unknown or not measurable (key window too wide for a bitmap) */
bool mdb_col_distinct(struct mdb_catalog *cat, struct mdb_table *t, struct mdb_column *col);
int mdb_col_range(struct mdb_catalog *cat, struct mdb_table *t, struct mdb_column *col, int64_t *lo, int64_t *hi);	/* 0 ok, 1 no range for this type */
int mdb_catalog_device(struct mdb_catalog *cat, char *err, size_t errlen);

/* ------------------------------------------------------------------ statement plans */
enum mdb_expr_kind {
	MDB_EX_NAME = 1,	/* bare column name (before resolution) */
	MDB_EX_FIELD,		/* table.column */
	MDB_EX_INT,
	MDB_EX_FLOAT,
	MDB_EX_BOOL,
	MDB_EX_STRING,
	MDB_EX_NULL,
	MDB_EX_CMP,		/* op = comparison code 1..6, 2 kids */
	MDB_EX_LOGOP,		/* op = 0 AND, 1 OR, 2 XOR (reference enum ast_logop_type), 2 kids */
	MDB_EX_ISNULL,		/* op = negation, 1 kid */
	MDB_EX_ISIN,		/* op = negation, kid[0] = field, rest = values */
	MDB_EX_COUNT,		/* COUNT(*) / COUNT(x) */
	MDB_EX_LIKE,
	MDB_EX_ARITH,		/* + - * / % NEG: parsed, rejected by the executor like select-list math is ignored upstream */
	MDB_EX_ALIAS,		/* alias wrapper around kid[0] */
};

struct mdb_expr {
	int kind;
	int op;
	char tbl[MDB_NAME_LEN];
	char col[MDB_NAME_LEN];
	int64_t ival;
	double dval;
	char *sval;			/* STRING literal, quotes included (heap) */
	struct mdb_expr **kids;
	int nkids;
	/* resolution (FIELD): index into the plan's FROM tables and the column index there */
	int tbl_idx, col_idx, type;
};

struct mdb_from_tab {
	char name[MDB_NAME_LEN];
	char alias[MDB_NAME_LEN];
	struct mdb_table *t;
};

/* left-deep join list: tables[0] (x) tables[1] ON on[1] (x) tables[2] ON on[2] ...; on[i] == NULL
 * means the comma form (reference optimiser turns it into JOIN ... ON 1=1, optimiser_select.c:395-464) */
struct mdb_select {
	bool distinct, select_all;
	struct mdb_expr **sel;
	int nsel;
	char (*sel_alias)[MDB_NAME_LEN];	/* [nsel] or NULL: `expr AS name` of a select item ("" without one), filled by the resolver */
	struct mdb_from_tab *tabs;
	struct mdb_expr **on;
	int *join_type;
	int ntabs;
	struct mdb_expr *where;
	struct mdb_expr **group;
	int ngroup;
	/* clauses the reference parses and checks but never executes; executed here with SQL
	 * semantics as the 8f row 4 extension */
	struct mdb_expr *having;
	struct mdb_expr **order;	/* ORDER BY items (fields) */
	int *order_desc;		/* 0 ASC, 1 DESC (midorisql.y:175-177) */
	int norder;
	bool has_limit;
	int64_t limit_off, limit_cnt;	/* LIMIT cnt | LIMIT off, cnt (midorisql.y:193-196) */
};

struct mdb_create {
	char name[MDB_NAME_LEN];
	bool if_not_exists;
	int ncols;
	char colname[MDB_MAX_COLS][MDB_NAME_LEN];
	int coltype[MDB_MAX_COLS];
	int colprec[MDB_MAX_COLS];	/* VARCHAR length */
	bool notnull[MDB_MAX_COLS];	/* ATTR NOTNULL / ATTR PRIKEY in front of the COLUMNDEF (midorisql.y:468-472) */
	bool unique[MDB_MAX_COLS];	/* ATTR UNIQUEKEY / ATTR PRIKEY: never enforced (upstream neither) - the column's distinct-ness gets MEASURED */
	bool pending_notnull, pending_unique;	/* (builder state: attributes seen for the column being defined) */
};

struct mdb_insert {
	char name[MDB_NAME_LEN];
	int ncolnames;
	char colname[MDB_MAX_COLS][MDB_NAME_LEN];
	int ntuples, nvals;		/* nvals per tuple */
	struct mdb_expr ***vals;	/* [ntuples][nvals] literal expressions */
};

/* DELETE FROM t [WHERE ...] / UPDATE t SET col = literal [, ...] [WHERE ...]
 * (reference AST: ast_delete.c / ast_update.c; executors executor_delete.c:412-459, executor_update.c:460-503) */
struct mdb_assign {
	char col[MDB_NAME_LEN];
	struct mdb_expr *val;		/* literal: INT / FLOAT / NULL */
};

struct mdb_dml {
	char name[MDB_NAME_LEN];
	struct mdb_expr *where;		/* NULL = every row */
	struct mdb_assign *assign;	/* UPDATE only */
	int nassign;
};

enum mdb_stmt_kind { MDB_ST_SELECT = 1, MDB_ST_CREATE, MDB_ST_INSERT, MDB_ST_DELETE, MDB_ST_UPDATE };

struct mdb_stmt {
	int kind;
	struct mdb_select sel;
	struct mdb_create crt;
	struct mdb_insert ins;
	struct mdb_dml dml;
};

int mdb_plan_build(const struct mdb_rpn *rpn, struct mdb_stmt *out, char *err, size_t errlen);
void mdb_stmt_free(struct mdb_stmt *st);
void mdb_expr_free(struct mdb_expr *e);

/* ------------------------------------------------------------------ result set */
struct mdb_result {
	struct mdb_legacy_table legacy;	/* FIRST: what `results.table` points at begins with the reference's struct table (include/mdb_legacy.h) */
	struct mdb_legacy_list_head legacy_head;
	bool legacy_built;
	int ncols;
	char (*colname)[MDB_NAME_LEN];
	int *coltype;
	int *colprec;			/* the declared VARCHAR(n) of a VARCHAR result column (the legacy view's column.precision), else 0 */
	char *legacy_text;		/* the legacy view's VARCHAR cells: `precision` zero-filled bytes each, as upstream allocates them */
	int64_t **data;			/* host columns, 8-byte values (0 for NULL) */
	uint64_t **nullbits;		/* host NULL bits per column or NULL */
	uint64_t nrows;
	double exec_ms;			/* device pipeline wall time of the SELECT that produced it */
	uint64_t joined_rows;		/* rows produced by the join before aggregation (0 when no join) */
	const struct mdb_strdict *dict;	/* VARCHAR result cells are ids of this dictionary (the database's) */
	/* results kept on the device (mdb_database_results_on_device): d_data[c] / d_nullbits[c] are buffers of `dev` owned by the
	 * result; the host columns above are filled on first use (mdb_result_fetch) */
	mdb_dev_ctx *dev;
	void **d_data;
	uint64_t **d_nullbits;
	bool fetched;
};
void mdb_result_legacy_header(struct mdb_result *r);	/* mdb_legacy.c */
void mdb_result_legacy_rows(struct mdb_result *r);
void mdb_result_legacy_free(struct mdb_result *r);
int mdb_result_fetch(struct mdb_result *r);	/* host columns of a device-resident result (no-op otherwise) */
void mdb_result_free(struct mdb_result *r);

/* ------------------------------------------------------------------ executor */
int mdb_exec_create(struct mdb_catalog *cat, struct mdb_create *c, char *err, size_t errlen);
int mdb_exec_insert(struct mdb_catalog *cat, struct mdb_insert *ins, size_t *n_rows_aff, char *err, size_t errlen);
int mdb_exec_select(struct mdb_catalog *cat, struct mdb_select *s, struct mdb_result **out, char *err, size_t errlen);
int mdb_exec_delete(struct mdb_catalog *cat, struct mdb_dml *d, size_t *n_rows_aff, char *err, size_t errlen);
int mdb_exec_update(struct mdb_catalog *cat, struct mdb_dml *d, size_t *n_rows_aff, char *err, size_t errlen);

/* result column order of the reference (djb2 hashtable iteration) */
int mdb_reference_column_order(const char (*keys)[MDB_NAME_LEN], int nkeys, int *order_out);

#endif /* MDB_HOST_H */
