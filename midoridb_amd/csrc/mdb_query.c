/*
 * mdb_query.c - the public API (include/mdb_query.h): same entry points, statuses, error
 * convention and ownership rules as the reference's src/engine/query.c and database.c, with the
 * MI355X executor behind them.
 *
 *   query_execute()  = reference query.c:35-106: parse -> plan -> (semantic + optimiser
 *                      normalisation) -> execute; never returns an error code, the returned
 *                      object carries status and a '\n'-terminated message (query.c:95-105).
 *   query_cur_step() = reference query.c:108-146, but correct for any result size (D4).
 *   query_column_int64() = reference query.c:148-167: the raw 8 bytes of the cell.
 */
#include "mdb_host.h"

int database_open(struct database *db)
{
	struct mdb_catalog *cat;

	if (!db)
		return -MIDORIDB_ERROR;
	cat = calloc(1, sizeof(*cat));
	if (!cat)
		return -MIDORIDB_NOMEM;
	db->tables = cat;
	if (pthread_mutex_init(&db->mutex, NULL)) {
		free(cat);
		db->tables = NULL;
		return -MIDORIDB_INTERNAL;
	}
	return MIDORIDB_OK;
}

void database_close(struct database *db)
{
	if (!db || !db->tables)
		return;
	mdb_catalog_free(db->tables);
	free(db->tables);
	db->tables = NULL;
	pthread_mutex_destroy(&db->mutex);
}

static struct query_output *run_rpn(struct database *db, struct mdb_rpn *rpn, struct query_output *output)
{
	struct mdb_catalog *cat = db->tables;
	struct mdb_stmt st;
	char *msg = output->error.message;
	const size_t msglen = sizeof(output->error.message) - 1;
	int rc;

	rc = mdb_plan_build(rpn, &st, msg, msglen);
	if (rc) {
		if (!msg[0])
			snprintf(msg, msglen, "error while running syntax analysis on query\n");
		output->status = ST_ERROR;
		return output;
	}
	switch (st.kind) {
	case MDB_ST_CREATE:
		pthread_mutex_lock(&db->mutex);		/* the reference locks the database for CREATE (executor_create.c:73) */
		rc = mdb_exec_create(cat, &st.crt, msg, msglen);
		pthread_mutex_unlock(&db->mutex);
		output->status = rc ? ST_ERROR : ST_OK_EXECUTED;
		break;
	case MDB_ST_INSERT:
		rc = mdb_exec_insert(cat, &st.ins, &output->n_rows_aff, msg, msglen);
		output->status = rc ? ST_ERROR : ST_OK_EXECUTED;
		break;
	case MDB_ST_DELETE:
		rc = mdb_exec_delete(cat, &st.dml, &output->n_rows_aff, msg, msglen);
		output->status = rc ? ST_ERROR : ST_OK_EXECUTED;
		break;
	case MDB_ST_UPDATE:
		rc = mdb_exec_update(cat, &st.dml, &output->n_rows_aff, msg, msglen);
		output->status = rc ? ST_ERROR : ST_OK_EXECUTED;
		break;
	case MDB_ST_SELECT: {
		struct mdb_result *res = NULL;
		rc = mdb_exec_select(cat, &st.sel, &res, msg, msglen);
		if (rc) {
			output->status = ST_ERROR;
		} else {
			output->status = ST_OK_WITH_RESULTS;
			output->results.table = res;
			output->results.cursor_blk = NULL;
			output->results.cursor_offset = 0;
		}
		break;
	}
	default:
		snprintf(msg, msglen, "statement not supported\n");
		output->status = ST_ERROR;
	}
	if (output->status == ST_ERROR && !msg[0])
		snprintf(msg, msglen, "execution phase: internal error\n");
	mdb_stmt_free(&st);
	return output;
}

struct query_output *query_execute(struct database *db, char *query)
{
	struct query_output *output;
	struct mdb_rpn rpn = {0};

	if (!db || !db->tables || !query)
		return NULL;
	output = calloc(1, sizeof(*output));
	if (!output)
		return NULL;
	if (mdb_sql_parse(query, &rpn, output->error.message, sizeof(output->error.message) - 2)) {
		size_t l = strlen(output->error.message);
		if (l && output->error.message[l - 1] != '\n')
			strcat(output->error.message, "\n");
		output->status = ST_ERROR;
		return output;
	}
	{
		/* MDB_PROF_DUMP=1: per-kernel device time of every statement on stderr (diagnostics; serialises nothing by itself,
		 * the events ride on the context's stream) */
		struct mdb_catalog *cat = (struct mdb_catalog *)db->tables;
		const int dump = mdb_knob("MDB_PROF_DUMP") != NULL && cat && cat->dev;
		if (dump) {
			mdb_dev_prof_enable(cat->dev, 1);
			mdb_dev_prof_reset(cat->dev);
		}
		run_rpn(db, &rpn, output);
		if (dump) {
			struct mdb_dev_prof_entry e[64];
			int ne = 0;
			if (mdb_dev_prof_read(cat->dev, e, 64, &ne) == 0) {
				fprintf(stderr, "[mdb prof] %.60s\n", query);
				for (int i = 0; i < ne; i++)
					fprintf(stderr, "[mdb prof]   %-28s x%-4llu %9.3f ms\n", e[i].name, (unsigned long long)e[i].launches, e[i].total_ms);
			}
			mdb_dev_prof_enable(cat->dev, 0);
		}
	}
	mdb_rpn_free(&rpn);
	return output;
}

struct query_output *mdb_query_execute_rpn(struct database *db, const char *rpn_lines)
{
	struct query_output *output;
	struct mdb_rpn rpn = {0};

	if (!db || !db->tables || !rpn_lines)
		return NULL;
	output = calloc(1, sizeof(*output));
	if (!output)
		return NULL;
	if (mdb_rpn_from_text(rpn_lines, &rpn)) {
		snprintf(output->error.message, sizeof(output->error.message) - 1, "error while initialising query\n");
		output->status = ST_ERROR;
		return output;
	}
	run_rpn(db, &rpn, output);
	mdb_rpn_free(&rpn);
	return output;
}

/* cursor: cursor_blk == NULL means "before the first row"; cursor_offset is the current row index */
int query_cur_step(struct result_set *res)
{
	struct mdb_result *r;

	if (!res || !res->table)
		return MIDORIDB_OK;
	r = res->table;
	if (!r->fetched && mdb_result_fetch(r))
		return -MIDORIDB_INTERNAL;	/* the device-to-host copy of a result kept on the device failed: no row is current, and the
						 * caller can tell this from the end of an empty result (MIDORIDB_OK); a loop over
						 * `== MIDORIDB_ROW` ends either way */
	if (!res->cursor_blk) {
		res->cursor_blk = r;
		res->cursor_offset = 0;
	} else {
		res->cursor_offset++;
	}
	if (res->cursor_offset >= r->nrows) {
		res->cursor_offset = r->nrows;
		return MIDORIDB_OK;
	}
	return MIDORIDB_ROW;
}

static struct mdb_result *cur_row(struct result_set *res, int col_idx, uint64_t *row)
{
	struct mdb_result *r;
	if (!res || !res->table || !res->cursor_blk)
		return NULL;
	r = res->table;
	if (col_idx < 0 || col_idx >= r->ncols || res->cursor_offset >= r->nrows)
		return NULL;
	*row = res->cursor_offset;
	return r;
}

int64_t query_column_int64(struct result_set *res, int col_idx)
{
	uint64_t row;
	struct mdb_result *r = cur_row(res, col_idx, &row);
	/* the reference BUG_ON()s (exit) on a bad cursor/column (query.c:155-160); a library must not
	 * kill its host process, so an invalid access reads as 0 */
	return r ? r->data[col_idx][row] : 0;
}

double query_column_double(struct result_set *res, int col_idx)
{
	int64_t bits = query_column_int64(res, col_idx);
	double d;
	memcpy(&d, &bits, 8);
	return d;
}

const char *query_column_text(struct result_set *res, int col_idx)
{
	uint64_t row;
	struct mdb_result *r = cur_row(res, col_idx, &row);
	if (!r || r->coltype[col_idx] != MDB_CT_VARCHAR || !r->dict)
		return NULL;
	if (r->nullbits[col_idx] && ((r->nullbits[col_idx][row >> 6] >> (row & 63)) & 1))
		return "";	/* upstream a NULL VARCHAR cell points at an empty string; query_column_is_null() tells them apart */
	return mdb_dict_str(r->dict, r->data[col_idx][row]);
}

const char *mdb_result_text_at(struct result_set *res, int col_idx, uint64_t row)
{
	struct mdb_result *r = res ? res->table : NULL;
	if (!r || col_idx < 0 || col_idx >= r->ncols || row >= r->nrows || r->coltype[col_idx] != MDB_CT_VARCHAR || !r->dict)
		return NULL;
	if (!r->fetched && mdb_result_fetch(r))
		return NULL;
	if (r->nullbits[col_idx] && ((r->nullbits[col_idx][row >> 6] >> (row & 63)) & 1))
		return "";
	return mdb_dict_str(r->dict, r->data[col_idx][row]);
}

bool query_column_is_null(struct result_set *res, int col_idx)
{
	uint64_t row;
	struct mdb_result *r = cur_row(res, col_idx, &row);
	if (!r || !r->nullbits[col_idx])
		return false;
	return (r->nullbits[col_idx][row >> 6] >> (row & 63)) & 1;
}

int query_column_count(struct result_set *res)
{
	return res && res->table ? ((struct mdb_result *)res->table)->ncols : 0;
}

const char *query_column_name(struct result_set *res, int col_idx)
{
	struct mdb_result *r = res ? res->table : NULL;
	return r && col_idx >= 0 && col_idx < r->ncols ? r->colname[col_idx] : NULL;
}

int query_column_type(struct result_set *res, int col_idx)
{
	struct mdb_result *r = res ? res->table : NULL;
	return r && col_idx >= 0 && col_idx < r->ncols ? r->coltype[col_idx] : -1;
}

uint64_t query_row_count(struct result_set *res)
{
	return res && res->table ? ((struct mdb_result *)res->table)->nrows : 0;
}

const int64_t *query_column_data(struct result_set *res, int col_idx)
{
	struct mdb_result *r = res ? res->table : NULL;
	if (r && !r->fetched && mdb_result_fetch(r))
		return NULL;
	return r && col_idx >= 0 && col_idx < r->ncols ? r->data[col_idx] : NULL;
}

const void *query_column_data_device(struct result_set *res, int col_idx)
{
	struct mdb_result *r = res ? res->table : NULL;
	return r && r->d_data && col_idx >= 0 && col_idx < r->ncols ? r->d_data[col_idx] : NULL;
}

const uint64_t *query_column_nulls_device(struct result_set *res, int col_idx)
{
	struct mdb_result *r = res ? res->table : NULL;
	return r && r->d_data && r->d_nullbits && col_idx >= 0 && col_idx < r->ncols ? r->d_nullbits[col_idx] : NULL;
}

int mdb_database_results_on_device(struct database *db, int on)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	if (!cat)
		return -MIDORIDB_ERROR;
	cat->results_on_device = on != 0;
	return MIDORIDB_OK;
}

int mdb_database_groups_any_order(struct database *db, int on)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	if (!cat)
		return -MIDORIDB_ERROR;
	cat->groups_any_order = on != 0;
	return MIDORIDB_OK;
}

unsigned long long mdb_database_joins_eliminated(struct database *db)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	return cat ? cat->joins_eliminated : 0ull;
}

double query_exec_ms(struct result_set *res)
{
	return res && res->table ? ((struct mdb_result *)res->table)->exec_ms : 0.0;
}

uint64_t query_joined_rows(struct result_set *res)
{
	return res && res->table ? ((struct mdb_result *)res->table)->joined_rows : 0;
}

void query_free(struct query_output *output)
{
	if (!output)
		return;
	if (output->status == ST_OK_WITH_RESULTS)
		mdb_result_free(output->results.table);
	free(output);
}

/* ------------------------------------------------------------------ sharded mode handed in by the host */

struct mdb_dev_ctx *mdb_database_device(struct database *db)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	char err[256];
	if (!cat || mdb_catalog_device(cat, err, sizeof(err)))
		return NULL;
	return cat->dev;
}

int mdb_database_set_dist(struct database *db, struct mdb_dist *dist)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	if (!cat || !cat->dev || !dist)
		return -MIDORIDB_ERROR;
	if (cat->dist)
		mdb_dist_destroy(cat->dist);
	cat->dist = dist;
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ bulk ingest */

int mdb_table_append_columns(struct database *db, const char *table, int ncols, uint64_t n, const int64_t *const *cols,
			     const uint8_t *const *nulls)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	struct mdb_table *t = cat ? mdb_catalog_find(cat, table) : NULL;
	int rc;

	if (!t || t->ncols != ncols || t->device_only)
		return -MIDORIDB_ERROR;
	rc = mdb_table_reserve(t, t->nrows + n);
	if (rc)
		return rc;
	/* everything is validated before any column is touched: a missing column array, NULLs in NOT NULL columns (reference
	 * semantic_insert.c:440-495) - for a VARCHAR column also a NULL string pointer -, a VARCHAR without room for a character */
	uint64_t add_nulls[MDB_MAX_COLS] = { 0 };
	for (int c = 0; c < ncols; c++) {
		const struct mdb_column *col = &t->cols[c];
		if (col->type == MDB_CT_VARCHAR) {
			const char *const *strs = (const char *const *)cols[c];
			if (col->precision < 1)
				return -MIDORIDB_ERROR;
			if (col->not_null)
				for (uint64_t i = 0; i < n; i++)
					if (!strs || !strs[i] || (nulls && nulls[c] && nulls[c][i]))
						return -MIDORIDB_ERROR;
			continue;
		}
		if (!cols[c])
			return -MIDORIDB_ERROR;
		if (col->not_null && nulls && nulls[c])
			for (uint64_t i = 0; i < n; i++)
				if (nulls[c][i])
					return -MIDORIDB_ERROR;
	}
	/* large appends of plain 8-byte columns without NULL flags: copied by several threads, chunk by chunk, the device mirror following
	 * behind (mdb_table_bulk_copy, mdb_store.c) - the first SELECT then uploads nothing */
	bool plain = n >= ((uint64_t)1 << 20) && !(mdb_knob("MDB_INGEST_BULK") && mdb_knob("MDB_INGEST_BULK")[0] == '0');
	for (int c = 0; c < ncols && plain; c++)
		plain = t->cols[c].type != MDB_CT_VARCHAR && !(nulls && nulls[c]);
	if (plain) {
		const uint64_t r0 = t->nrows, r1 = t->nrows + n, old_gen = t->generation;
		for (int c = 0; c < ncols; c++) {	/* the rows' NULL bits, cleared a word at a time (the first and last word may be shared) */
			struct mdb_column *col = &t->cols[c];
			uint64_t i = r0;
			for (; i < r1 && (i & 63); i++)
				col->nullbits[i >> 6] &= ~(1ull << (i & 63));
			if (i < r1) {
				const uint64_t full = (r1 - i) / 64;
				memset(col->nullbits + (i >> 6), 0, full * 8);
				i += full * 64;
			}
			for (; i < r1; i++)
				col->nullbits[i >> 6] &= ~(1ull << (i & 63));
		}
		bool mirrored = false;
		rc = mdb_table_bulk_copy(cat, t, ncols, n, cols, &mirrored);
		if (rc)
			return rc;
		t->nrows += n;
		t->generation++;
		if (mirrored)
			mdb_table_bulk_mirrored(cat, t, r0, old_gen);
		return MIDORIDB_OK;
	}
	for (int c = 0; c < ncols; c++) {
		struct mdb_column *col = &t->cols[c];
		if (col->type == MDB_CT_VARCHAR) {
			/* cells are `const char *` (NULL pointer, NULL flag or cols[c] == NULL: SQL NULL); the strings are interned */
			const char *const *strs = (const char *const *)cols[c];
			for (uint64_t i = 0; i < n; i++) {
				const uint64_t row = t->nrows + i;
				const bool isnull = !strs || !strs[i] || (nulls && nulls[c] && nulls[c][i]);
				col->data[row] = 0;
				if (isnull) {
					col->nullbits[row >> 6] |= 1ull << (row & 63);
					add_nulls[c]++;
				} else {
					size_t len = strlen(strs[i]);
					if (len + 1 > (size_t)col->precision)
						len = (size_t)col->precision - 1;
					col->data[row] = mdb_dict_intern(&cat->dict, strs[i], len);
					if (!col->data[row])
						return -MIDORIDB_NOMEM;
					col->nullbits[row >> 6] &= ~(1ull << (row & 63));
				}
			}
			continue;
		}
		if (!cols[c])
			return -MIDORIDB_ERROR;
		memcpy(col->data + t->nrows, cols[c], n * 8);
		if (!nulls || !nulls[c]) {
			/* no NULL flags: the rows' bits are cleared a word at a time (the first and last word may be shared with other rows) */
			const uint64_t r0 = t->nrows, r1 = t->nrows + n;
			uint64_t i = r0;
			for (; i < r1 && (i & 63); i++)
				col->nullbits[i >> 6] &= ~(1ull << (i & 63));
			if (i < r1) {
				const uint64_t full = (r1 - i) / 64;
				memset(col->nullbits + (i >> 6), 0, full * 8);
				i += full * 64;
			}
			for (; i < r1; i++)
				col->nullbits[i >> 6] &= ~(1ull << (i & 63));
			continue;
		}
		for (uint64_t i = 0; i < n; i++) {
			const uint64_t row = t->nrows + i;
			if (nulls[c][i]) {
				col->nullbits[row >> 6] |= 1ull << (row & 63);
				col->data[row] = 0;
				add_nulls[c]++;
			} else {
				col->nullbits[row >> 6] &= ~(1ull << (row & 63));
			}
		}
	}
	for (int c = 0; c < ncols; c++)
		t->cols[c].null_count += add_nulls[c];
	t->nrows += n;
	t->generation++;
	return MIDORIDB_OK;
}

int mdb_table_generate_shard(struct database *db, const char *table, uint64_t n, uint64_t first_index, uint64_t domain, uint64_t seed,
			     const uint64_t *modulus)
{
	struct mdb_catalog *cat = db ? db->tables : NULL;
	struct mdb_table *t = cat ? mdb_catalog_find(cat, table) : NULL;
	char err[256];
	int rc;

	if (!t || t->nrows)
		return -MIDORIDB_ERROR;
	rc = mdb_catalog_device(cat, err, sizeof(err));
	if (rc)
		return rc;
	for (int c = 0; c < t->ncols; c++)
		if (t->cols[c].type != MDB_CT_INTEGER && t->cols[c].type != MDB_CT_DOUBLE)
			return -MIDORIDB_ERROR;
	for (int c = 0; c < t->ncols; c++) {
		struct mdb_column *col = &t->cols[c];
		rc = mdb_dev_alloc(cat->dev, (n ? n : 1) * 8, &col->d_data);
		if (!rc && col->type == MDB_CT_DOUBLE)
			rc = mdb_dev_gen_payload(cat->dev, col->d_data, n, first_index, seed + (uint64_t)c, 1);
		else if (!rc)
			rc = mdb_dev_gen_keys(cat->dev, col->d_data, n, first_index, domain, seed + (uint64_t)c, modulus ? modulus[c] : 0);
		if (rc)
			return rc;
	}
	rc = mdb_dev_sync(cat->dev);
	if (rc)
		return rc;
	t->nrows = n;
	t->device_only = true;
	t->dev_rows = n;
	t->dev_cap = n ? n : 1;
	t->dev_generation = t->generation;
	for (int c = 0; c < t->ncols; c++) {	/* catalog statistics, as an ingest would leave them (mdb_table_sync_device) */
		int64_t lo, hi;
		(void)mdb_col_range(cat, t, &t->cols[c], &lo, &hi);
	}
	return MIDORIDB_OK;
}

int mdb_table_generate(struct database *db, const char *table, uint64_t n, uint64_t seed, const uint64_t *modulus)
{
	return mdb_table_generate_shard(db, table, n, 0, n, seed, modulus);
}
