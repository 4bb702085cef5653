/*
 * ref_harness.c - TEST INFRASTRUCTURE ONLY (oracle). Never linked into the product.
 *
 * Thin driver around the *real* MidoriDB reference sources, compiled where they
 * lie under /root/reference by oracle/Makefile into oracle/_ref/libmidori_ref.so.
 *
 * The reference's flex/bison front end cannot be built in this image (no bison,
 * no flex), so nothing of src/parser/syntax.c, midorisql.y/.l or
 * src/engine/query.c (which calls syntax_parse) is compiled, and no stand-in for
 * them is written.  Instead this harness performs, call for call, the four steps
 * query_execute() performs after the parser (reference src/engine/query.c:63-87):
 *
 *     ast_build_tree(queue) -> semantic_analyse() -> optimiser_run() -> executor_run()
 *
 * feeding the queue with the RPN token strings the grammar's emit() would have
 * produced (reference src/parser/midorisql.y:517-528; vocabulary in SURVEY.md
 * Appendix A).  Everything behind the parser seam - AST builder, semantic
 * checks, optimiser, executor_select.c (the hot path) and the src/primitive
 * row store - is the unmodified reference.
 *
 * Results are read by walking the result table's datablocks directly (skip
 * deleted rows, stop a block at the first empty row), because the reference's
 * own query_cur_step() is unusable past one datablock.
 */
#include <compiler/common.h>
#include <datastructure/queue.h>
#include <datastructure/linkedlist.h>
#include <parser/ast.h>
#include <parser/semantic.h>
#include <engine/database.h>
#include <engine/query.h>
#include <engine/optimiser.h>
#include <engine/executor.h>
#include <primitive/table.h>
#include <primitive/row.h>
#include <primitive/column.h>
#include <primitive/datablock.h>
#include <lib/bit.h>

struct ref_ctx {
	struct database db;
	struct table *result;		/* result of the last successful SELECT */
	int64_t rows_affected;		/* query_output.n_rows_aff of the last statement */
	char err[1024];
};

void *ref_open(void)
{
	struct ref_ctx *c = calloc(1, sizeof(*c));
	if (!c)
		return NULL;
	if (database_open(&c->db) != MIDORIDB_OK) {
		free(c);
		return NULL;
	}
	return c;
}

static void drop_result(struct ref_ctx *c)
{
	if (c->result)
		table_destroy(&c->result);
	c->result = NULL;
}

void ref_close(void *h)
{
	struct ref_ctx *c = h;
	if (!c)
		return;
	drop_result(c);
	database_close(&c->db);
	free(c);
}

const char *ref_error(void *h)
{
	return ((struct ref_ctx *)h)->err;
}

/*
 * Run one statement given as RPN lines separated by '\n' (the parser's output
 * format).  Returns 0 = executed (DDL/DML), 1 = executed with results (SELECT),
 * <0 = error (message via ref_error).  Mirrors reference query.c:63-105.
 */
int ref_exec_rpn(void *h, const char *rpn)
{
	struct ref_ctx *c = h;
	struct queue queue = {0};
	struct ast_node *node = NULL;
	struct query_output *out = NULL;
	char *dup, *save = NULL;
	int ret = -1;

	c->err[0] = 0;
	drop_result(c);

	if (!queue_init(&queue)) {
		snprintf(c->err, sizeof(c->err), "queue_init failed");
		return -1;
	}
	dup = strdup(rpn);
	for (char *t = strtok_r(dup, "\n", &save); t; t = strtok_r(NULL, "\n", &save)) {
		if (!queue_offer(&queue, t, strlen(t) + 1)) {
			snprintf(c->err, sizeof(c->err), "queue_offer failed");
			goto out_queue;
		}
	}

	out = zalloc(sizeof(*out));
	node = ast_build_tree(&queue);
	if (!node) {
		snprintf(c->err, sizeof(c->err), "error while running syntax analysis on query");
		goto out_queue;
	}
	if (!semantic_analyse(&c->db, node, out->error.message, sizeof(out->error.message) - 1)) {
		snprintf(c->err, sizeof(c->err), "semantic: %s", out->error.message);
		ret = -2;
		goto out_ast;
	}
	if (optimiser_run(&c->db, node, out)) {
		snprintf(c->err, sizeof(c->err), "optimiser: %s", out->error.message);
		ret = -3;
		goto out_ast;
	}
	if (executor_run(&c->db, node, out)) {
		snprintf(c->err, sizeof(c->err), "executor: %s", out->error.message);
		ret = -4;
		goto out_ast;
	}
	c->rows_affected = (int64_t)out->n_rows_aff;
	if (node->node_type == AST_TYPE_SEL_SELECT) {
		c->result = out->results.table;
		ret = 1;
	} else {
		ret = 0;
	}
out_ast:
	ast_free(node);
out_queue:
	queue_free(&queue);
	free(dup);
	free(out);
	return ret;
}

/*
 * Bulk load through the reference's own storage API (table_insert_row,
 * reference src/primitive/row.c:26).  All columns must be 8-byte types
 * (CT_INTEGER / CT_DOUBLE / CT_DATE / CT_DATETIME).  vals is column-major
 * [ncols][nrows] of raw 8-byte values, nulls is column-major bytes (1 = NULL)
 * or NULL for "no NULLs".
 */
int ref_bulk_insert(void *h, const char *table_name, int ncols, int64_t nrows,
		    const int64_t *vals, const uint8_t *nulls)
{
	struct ref_ctx *c = h;
	struct table *t = database_table_get(&c->db, (char *)table_name);
	size_t rs;
	struct row *r;

	if (!t || t->column_count != ncols)
		return -1;
	for (int k = 0; k < ncols; k++)
		if (table_calc_column_space(&t->columns[k]) != 8)
			return -2;
	rs = table_calc_row_size(t);
	r = zalloc(rs);
	for (int64_t i = 0; i < nrows; i++) {
		memset(r, 0, rs);
		for (int k = 0; k < ncols; k++) {
			if (nulls && nulls[(size_t)k * nrows + i])
				bit_set(r->null_bitmap, k, sizeof(r->null_bitmap));
			else
				((int64_t *)r->data)[k] = vals[(size_t)k * nrows + i];
		}
		if (!table_insert_row(t, r, rs)) {
			free(r);
			return -3;
		}
	}
	free(r);
	return 0;
}

int ref_result_ncols(void *h)
{
	struct ref_ctx *c = h;
	return c->result ? c->result->column_count : -1;
}

const char *ref_result_colname(void *h, int i)
{
	struct ref_ctx *c = h;
	return c->result->columns[i].name;
}

int ref_result_coltype(void *h, int i)
{
	struct ref_ctx *c = h;
	return (int)c->result->columns[i].type;
}

/*
 * Walk the result table.  With vals == NULL only counts live rows.  Otherwise
 * writes row-major [nrows][ncols] raw 8-byte values (0 for NULL) and NULL flags.
 */
int64_t ref_result_fetch(void *h, int64_t *vals, uint8_t *nulls, int64_t cap_rows)
{
	struct ref_ctx *c = h;
	struct table *t = c->result;
	struct list_head *pos;
	size_t rs;
	int64_t n = 0;

	if (!t)
		return -1;
	if (!t->datablock_head)
		return 0;
	rs = table_calc_row_size(t);
	list_for_each(pos, t->datablock_head) {
		struct datablock *b = list_entry(pos, struct datablock, head);
		for (size_t i = 0; i < DATABLOCK_PAGE_SIZE / rs; i++) {
			struct row *r = (struct row *)&b->data[rs * i];
			size_t off = 0;
			if (r->flags.empty)
				break;
			if (r->flags.deleted)
				continue;
			if (vals && n < cap_rows) {
				for (int k = 0; k < t->column_count; k++) {
					bool isnull = bit_test(r->null_bitmap, k, sizeof(r->null_bitmap));
					int64_t v = 0;
					size_t sp = table_calc_column_space(&t->columns[k]);
					/* raw bytes whatever the NULL bit says, exactly like
					 * query_column_int64() (reference src/engine/query.c:162-166): the
					 * reference never clears the COUNT column's NULL bit
					 * (executor_select.c:324-338) and table_rem_column() does not shift
					 * the bitmap (src/primitive/column.c:146-211), so after a projection
					 * the bits no longer line up with the columns.  A NULL source cell
					 * reads as 0 because cpy_cols() skips the copy into the zeroed row
					 * (executor_select.c:384-387). */
					if (sp == 8)
						memcpy(&v, r->data + off, 8);
					else if (sp == 1)
						v = *(bool *)(r->data + off);
					vals[n * t->column_count + k] = v;
					nulls[n * t->column_count + k] = isnull;
					off += sp;
				}
			}
			n++;
		}
	}
	return n;
}

/*
 * A `struct table *` that did NOT come from this library - the product's legacy view of a result (include/mdb_legacy.h) - read
 * with the reference's own compiled struct layouts, list macros and helpers: the same walk as ref_result_fetch.  What the tests
 * use to prove that a consumer built against the reference's headers finds its rows behind `output->results.table`.
 */
int ref_foreign_ncols(void *table)
{
	return ((struct table *)table)->column_count;
}

const char *ref_foreign_colname(void *table, int i)
{
	return ((struct table *)table)->columns[i].name;
}

int ref_foreign_coltype(void *table, int i)
{
	return (int)((struct table *)table)->columns[i].type;
}

int ref_foreign_is_count(void *table, int i)
{
	return ((struct table *)table)->columns[i].is_count;
}

int ref_foreign_precision(void *table, int i)
{
	return ((struct table *)table)->columns[i].precision;
}

/* a VARCHAR cell read the way upstream's own code reads one (table_insert_row, src/primitive/row.c: memcpy of column->precision bytes from
 * the cell's pointer): copies `precision` bytes out and says whether everything behind the string's NUL is zero, as in a zalloc'd cell */
int ref_foreign_text_cell_ok(int64_t cell, int precision)
{
	char *copy = malloc(precision > 0 ? (size_t)precision : 1);
	int ok = 1, seen_nul = 0;
	if (!copy)
		return -1;
	memcpy(copy, (const char *)(uintptr_t)cell, (size_t)precision);
	for (int i = 0; i < precision; i++) {
		if (seen_nul && copy[i])
			ok = 0;
		if (!copy[i])
			seen_nul = 1;
	}
	free(copy);
	return ok && seen_nul;
}

const char *ref_foreign_name(void *table)
{
	return ((struct table *)table)->name;
}

/* rows as ref_result_fetch reads them; a VARCHAR cell (a pointer) is returned as the pointer's value: ref_foreign_text() reads it */
int64_t ref_foreign_fetch(void *table, int64_t *vals, uint8_t *nulls, int64_t cap_rows)
{
	struct table *t = table;
	struct list_head *pos;
	size_t rs;
	int64_t n = 0;

	if (!t || !t->datablock_head)
		return -1;
	rs = table_calc_row_size(t);
	list_for_each(pos, t->datablock_head) {
		struct datablock *b = list_entry(pos, struct datablock, head);
		for (size_t i = 0; i < DATABLOCK_PAGE_SIZE / rs; i++) {
			struct row *r = (struct row *)&b->data[rs * i];
			size_t off = 0;
			if (r->flags.empty)
				break;
			if (r->flags.deleted)
				continue;
			if (vals && n < cap_rows) {
				for (int k = 0; k < t->column_count; k++) {
					int64_t v = 0;
					size_t sp = table_calc_column_space(&t->columns[k]);
					if (sp == 8)
						memcpy(&v, r->data + off, 8);
					else if (sp == 1)
						v = *(bool *)(r->data + off);
					vals[n * t->column_count + k] = v;
					nulls[n * t->column_count + k] = bit_test(r->null_bitmap, k, sizeof(r->null_bitmap));
					off += sp;
				}
			}
			n++;
		}
	}
	return n;
}

const char *ref_foreign_text(int64_t cell)
{
	return (const char *)(uintptr_t)cell;
}

/* the reference's own cursor arithmetic (query_cur_step + query_column_int64, src/engine/query.c:108-168, restated: query.c itself is
 * left out of this build because it calls the parser) over a foreign result_set: rows until the cursor says "end" */
int64_t ref_foreign_cursor_walk(void *table, int col, int64_t *out, int64_t cap)
{
	struct table *t = table;
	struct datablock *blk = NULL;
	size_t off = 0, rs = table_calc_row_size(t);
	int64_t n = 0;

	if (list_is_empty(t->datablock_head))
		return 0;
	for (;;) {
		struct row *row;
		if (!blk) {
			blk = container_of(t->datablock_head->next, typeof(struct datablock), head);
			off = 0;
		} else if (off + rs > DATABLOCK_PAGE_SIZE) {
			blk = container_of(blk->head.next, typeof(struct datablock), head);
			off = 0;
			if (&blk->head == t->datablock_head)
				break;
		} else {
			off += rs;
		}
		if (off + rs > DATABLOCK_PAGE_SIZE)
			continue;	/* (the reference reads past the page here - SURVEY 8a D4; the walk skips to the next block instead) */
		row = (struct row *)&blk->data[off];
		if (row->flags.empty)
			break;
		if (n < cap) {
			size_t o = 0;
			for (int i = 0; i < col; i++)
				o += table_calc_column_space(&t->columns[i]);
			out[n] = *(int64_t *)(row->data + o);
		}
		n++;
	}
	return n;
}

/* query_output.n_rows_aff of the last executed statement (INSERT / DELETE / UPDATE). */
int64_t ref_rows_affected(void *h)
{
	return ((struct ref_ctx *)h)->rows_affected;
}

/*
 * Live rows of a BASE table in scan order (what proc_from_clause_table() would see), 8-byte columns only:
 * row-major raw cell bytes and the row header's NULL bits (reliable in base tables).  vals == NULL: count.
 */
int64_t ref_table_fetch(void *h, const char *table_name, int64_t *vals, uint8_t *nulls, int64_t cap_rows)
{
	struct ref_ctx *c = h;
	struct table *t = database_table_get(&c->db, (char *)table_name);
	struct list_head *pos;
	size_t rs;
	int64_t n = 0;

	if (!t)
		return -1;
	rs = table_calc_row_size(t);
	list_for_each(pos, t->datablock_head) {
		struct datablock *b = list_entry(pos, struct datablock, head);
		for (size_t i = 0; i < DATABLOCK_PAGE_SIZE / rs; i++) {
			struct row *r = (struct row *)&b->data[rs * i];
			size_t off = 0;
			if (r->flags.empty)
				break;
			if (r->flags.deleted)
				continue;
			if (vals && n < cap_rows) {
				for (int k = 0; k < t->column_count; k++) {
					size_t sp = table_calc_column_space(&t->columns[k]);
					int64_t v = 0;
					if (sp == 8)
						memcpy(&v, r->data + off, 8);
					vals[n * t->column_count + k] = v;
					nulls[n * t->column_count + k] = bit_test(r->null_bitmap, k, sizeof(r->null_bitmap));
					off += sp;
				}
			}
			n++;
		}
	}
	return n;
}

int ref_table_ncols(void *h, const char *table_name)
{
	struct ref_ctx *c = h;
	struct table *t = database_table_get(&c->db, (char *)table_name);
	return t ? t->column_count : -1;
}

/* Number of live rows of a base table (sanity helper for the tests). */
int64_t ref_table_rows(void *h, const char *table_name)
{
	struct ref_ctx *c = h;
	struct table *t = database_table_get(&c->db, (char *)table_name);
	struct list_head *pos;
	size_t rs;
	int64_t n = 0;

	if (!t)
		return -1;
	rs = table_calc_row_size(t);
	list_for_each(pos, t->datablock_head) {
		struct datablock *b = list_entry(pos, struct datablock, head);
		for (size_t i = 0; i < DATABLOCK_PAGE_SIZE / rs; i++) {
			struct row *r = (struct row *)&b->data[rs * i];
			if (r->flags.empty)
				break;
			if (!r->flags.deleted)
				n++;
		}
	}
	return n;
}
