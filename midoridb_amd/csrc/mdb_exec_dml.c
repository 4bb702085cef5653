/*
 * mdb_exec_dml.c - CREATE TABLE / INSERT on the host store and DELETE / UPDATE on the device mirror (reference:
 * src/engine/executor_insert.c:194-249, executor_delete.c:412-440, executor_update.c:460-484).  Split off mdb_exec.c in round 4.
 */
#include "mdb_exec_internal.h"

/* ------------------------------------------------------------------ CREATE / INSERT (host storage) */

int mdb_exec_create(struct mdb_catalog *cat, struct mdb_create *c, char *err, size_t errlen)
{
	struct mdb_table *t;

	if (mdb_catalog_find(cat, c->name)) {
		if (c->if_not_exists)
			return MIDORIDB_OK;
		ERR("table '%s' already exists\n", c->name);
		return -MIDORIDB_ERROR;
	}
	for (int i = 0; i < c->ncols; i++) {
		/* every reference column type can be declared (include/primitive/column.h:17-25).  INTEGER, DOUBLE, DATE,
		 * DATETIME and TINYINT cells are 8-byte values that live on the device; VARCHAR cells stay on the host and a
		 * statement is rejected only when it REFERENCES such a column */
		for (int k = 0; k < i; k++)
			if (strcmp(c->colname[i], c->colname[k]) == 0) {
				ERR("duplicate column name: '%s'\n", c->colname[i]);
				return -MIDORIDB_ERROR;
			}
	}
	t = mdb_table_new(c->name);
	if (!t)
		return -MIDORIDB_NOMEM;
	for (int i = 0; i < c->ncols; i++) {
		mdb_table_add_column(t, c->colname[i], c->coltype[i]);
		t->cols[i].precision = c->colprec[i];
		t->cols[i].not_null = c->notnull[i];
		t->cols[i].declared_unique = c->unique[i];
	}
	return mdb_catalog_add(cat, t);
}

int mdb_exec_insert(struct mdb_catalog *cat, struct mdb_insert *ins, size_t *n_rows_aff, char *err, size_t errlen)
{
	struct mdb_table *t = mdb_catalog_find(cat, ins->name);
	int map[MDB_MAX_COLS];
	int rc;

	if (!t) {
		ERR("table '%s' doesn't exist\n", ins->name);
		return -MIDORIDB_ERROR;
	}
	if (t->device_only) {
		ERR("table '%s' was generated on the device and is read-only\n", ins->name);
		return -MIDORIDB_ERROR;
	}
	for (int c = 0; c < t->ncols; c++)
		map[c] = -1;
	if (ins->ncolnames) {
		if (ins->ncolnames != ins->nvals) {
			ERR("column count doesn't match value count\n");
			return -MIDORIDB_ERROR;
		}
		for (int k = 0; k < ins->ncolnames; k++) {
			int found = -1;
			for (int c = 0; c < t->ncols; c++)
				if (strcmp(t->cols[c].name, ins->colname[k]) == 0)
					found = c;
			if (found < 0) {
				ERR("no such column: '%.128s'\n", ins->colname[k]);
				return -MIDORIDB_ERROR;
			}
			map[found] = k;
		}
	} else {
		if (ins->nvals != t->ncols) {
			ERR("column count doesn't match value count\n");
			return -MIDORIDB_ERROR;
		}
		for (int c = 0; c < t->ncols; c++)
			map[c] = c;
	}
	rc = mdb_table_reserve(t, t->nrows + (uint64_t)ins->ntuples);
	if (rc)
		return rc;
	/* validate everything before touching the table: NOT NULL (semantic_insert.c:440-495), then the value / column type
	 * rules of check_value_for_column (semantic_insert.c:283-330), with the reference's texts */
	for (int c = 0; c < t->ncols; c++)
		if (map[c] < 0 && t->cols[c].not_null) {
			ERR("NOT NULL constraint failed: %s.%s\n", t->name, t->cols[c].name);
			return -MIDORIDB_ERROR;
		}
	for (int r = 0; r < ins->ntuples; r++)
		for (int c = 0; c < t->ncols; c++) {
			struct mdb_expr *v = map[c] >= 0 ? ins->vals[r][map[c]] : NULL;
			const struct mdb_column *col = &t->cols[c];
			int64_t tv;
			if (!v)
				continue;
			if (v->kind == MDB_EX_NULL) {
				if (col->not_null) {
					ERR("NOT NULL constraint failed: %s.%s\n", t->name, col->name);
					return -MIDORIDB_ERROR;
				}
				continue;
			}
			if (v->kind == MDB_EX_STRING) {
				if (col->type == MDB_CT_DATE || col->type == MDB_CT_DATETIME) {
					if (!mdb_parse_time(v->sval, col->type, &tv)) {
						ERR("val: '%.256s' can't be parsed for DATE | DATETIME column\n", v->sval);
						return -MIDORIDB_ERROR;
					}
				} else if (col->type == MDB_CT_VARCHAR) {
					const size_t len = strlen(v->sval) - 2 + 1;	/* without the quotes, with the NUL */
					if (len > (size_t)col->precision) {
						ERR("column: '%s' supports up to %d ASCII chars, value contains %lu\n", col->name, col->precision,
						    (unsigned long)len);
						return -MIDORIDB_ERROR;
					}
				} else {
					ERR("val: '%.256s' requires an VARCHAR() column\n", v->sval);
					return -MIDORIDB_ERROR;
				}
			} else if (v->kind == MDB_EX_INT && col->type != MDB_CT_INTEGER) {
				ERR("val: '%ld' requires an INTEGER column\n", (long)v->ival);
				return -MIDORIDB_ERROR;
			} else if (v->kind == MDB_EX_FLOAT && col->type != MDB_CT_DOUBLE) {
				ERR("val: '%f' requires a DOUBLE column\n", v->dval);
				return -MIDORIDB_ERROR;
			} else if (v->kind == MDB_EX_BOOL && col->type != MDB_CT_TINYINT) {
				ERR("val: '%d' requires a TINYINT column\n", (int)v->ival);
				return -MIDORIDB_ERROR;
			} else if (v->kind != MDB_EX_INT && v->kind != MDB_EX_FLOAT && v->kind != MDB_EX_BOOL) {
				ERR("only literal values can be inserted on the MI355X path\n");
				return -MIDORIDB_ERROR;
			}
		}
	for (int r = 0; r < ins->ntuples; r++) {
		const uint64_t row = t->nrows;
		for (int c = 0; c < t->ncols; c++) {
			struct mdb_expr *v = map[c] >= 0 ? ins->vals[r][map[c]] : NULL;
			struct mdb_column *col = &t->cols[c];
			col->data[row] = 0;
			if (!v || v->kind == MDB_EX_NULL) {
				col->nullbits[row >> 6] |= 1ull << (row & 63);
				col->null_count++;
				continue;
			}
			col->nullbits[row >> 6] &= ~(1ull << (row & 63));
			if (v->kind == MDB_EX_FLOAT) {
				memcpy(&col->data[row], &v->dval, 8);
			} else if (v->kind == MDB_EX_STRING && col->type == MDB_CT_VARCHAR) {
				/* the cell is the string's id in the database's dictionary (struct mdb_strdict) */
				const int64_t id = mdb_dict_intern(&cat->dict, v->sval + 1, strlen(v->sval) - 2);
				if (!id)
					return -MIDORIDB_NOMEM;	/* (rows already appended stay: the statement reports the failure) */
				col->data[row] = id;
			} else if (v->kind == MDB_EX_STRING) {
				(void)mdb_parse_time(v->sval, col->type, &col->data[row]);	/* validated above */
			} else {
				col->data[row] = v->ival;	/* INT, BOOL (0 | 1) */
			}
		}
		t->nrows++;
	}
	t->generation++;
	*n_rows_aff = (size_t)ins->ntuples;
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ DELETE / UPDATE
 *
 * MI355X replacements of scan_delete() (reference src/engine/executor_delete.c:412-440) and scan_update()
 * (src/engine/executor_update.c:460-484), the two callers SURVEY.md 8f row 1 names beside INSERT: the
 * WHERE clause runs through the same device predicate program as SELECT's (mdb_dev_filter), the rows are
 * removed (order-preserving compaction = what a later scan of the reference's flagged rows sees) or
 * rewritten ON the device mirror, and the host copy - when the table has one - follows, so the mirror is
 * never re-uploaded because of a DELETE or an UPDATE.
 *
 * Semantics kept (and where the reference's defects bound the domain):
 *   - a comparison with a NULL operand is false; IS [NOT] NULL reads the bitmap (executor_delete.c:170-195, 300-316)
 *   - literal types must equal the column type (semantic_delete.c:226-262, semantic_update.c:229-265)
 *   - UPDATE evaluates WHERE on the row's old values, then applies every assignment (executor_update.c:474-476)
 *   - SET col = NULL sets the NULL bit and leaves the cell's bytes; a value clears it (:411-415)
 *   - `value <op> column` is rejected: the reference evaluates it as `column <op> value` (executor_delete.c:281-283)
 *   - INT comparisons upstream go through `int` parameters (executor_delete.c:52): identical for values in
 *     [-2^31, 2^31), which is SURVEY's agreement domain D5
 *   - x NOT IN (a, b, ...) with more than one value is true upstream when x differs from ANY value
 *     (executor_delete.c:318-352); here it has SQL semantics, as in SELECT (defect D3)
 */
int dml_value_on_left(const struct mdb_expr *e)
{
	if (!e)
		return 0;
	if (e->kind == MDB_EX_CMP && e->kids[0]->kind != MDB_EX_FIELD && e->kids[1]->kind == MDB_EX_FIELD)
		return 1;
	for (int i = 0; i < e->nkids; i++)
		if (dml_value_on_left(e->kids[i]))
			return 1;
	return 0;
}

/* value-to-value comparisons: both literals of one kind (semantic_delete.c:273-325, semantic_update.c:276-328) */
int dml_check_values(const struct mdb_expr *e, char *err, size_t errlen)
{
	int rc;
	if (!e)
		return MIDORIDB_OK;
	if (e->kind == MDB_EX_CMP && e->kids[0]->kind != MDB_EX_FIELD && e->kids[1]->kind != MDB_EX_FIELD) {
		if (e->kids[0]->kind != e->kids[1]->kind) {
			ERR("value-to-value comparison don't have the same type\n");
			return -MIDORIDB_ERROR;
		}
		if (e->kids[0]->kind == MDB_EX_NULL && e->op != MDB_CMP_EQ && e->op != MDB_CMP_NE) {
			ERR("value-to-value NULL comparisons can only use '=' or '<>'\n");
			return -MIDORIDB_ERROR;
		}
	}
	for (int i = 0; i < e->nkids; i++)
		if ((rc = dml_check_values(e->kids[i], err, errlen)))
			return rc;
	return MIDORIDB_OK;
}

/* common front half: table lookup, WHERE resolution and checks, device mirror, selection.
 * *sel = device vector of the selected row positions (ascending), NULL when every row is selected
 * (no WHERE); with `complement` the rows NOT matching the predicate are selected instead. */
int dml_select_rows(struct mdb_catalog *cat, struct mdb_dml *d, struct exec *x, struct mdb_select *s, struct mdb_from_tab *tab,
			   bool complement, struct mdb_table **out_t, const uint32_t **sel, uint64_t *m, char *err, size_t errlen)
{
	struct mdb_table *t = mdb_catalog_find(cat, d->name);
	int rc;

	*sel = NULL;
	*m = 0;
	if (!t) {
		ERR("table '%s' doesn't exist\n", d->name);
		return -MIDORIDB_ERROR;
	}
	*out_t = t;
	memset(s, 0, sizeof(*s));
	memset(tab, 0, sizeof(*tab));
	mdb_copy_name(tab->name, t->name);
	tab->t = t;
	s->tabs = tab;
	s->ntabs = 1;
	if (d->where) {
		if ((rc = resolve_expr(s, d->where, err, errlen)) || (rc = check_predicate_x(d->where, "where", true, err, errlen)) ||
		    (rc = dml_check_values(d->where, err, errlen)))
			return rc;
		if (dml_value_on_left(d->where)) {
			ERR("comparisons in DELETE/UPDATE must have the column on the left (the reference evaluates 'value <op> column' "
			    "as 'column <op> value', executor_delete.c:281-283)\n");
			return -MIDORIDB_ERROR;
		}
	}
	if ((rc = mdb_catalog_device(cat, err, errlen)) || (rc = mdb_table_sync_device(cat, t, err, errlen)))
		return rc;
	memset(x, 0, sizeof(*x));
	for (int i = 0; i < MDB_MAX_TABS; i++)
		x->same_col[i] = -1;
	x->cat = cat;
	x->dev = cat->dev;
	x->s = s;
	x->err = err;
	x->errlen = errlen;
	x->n = t->nrows;
	*m = t->nrows;
	if (d->where && t->nrows) {
		struct pred_prog p;
		uint32_t *v;
		memset(&p, 0, sizeof(p));
		if (pred_compile(x, &p, d->where) ||
		    (complement && (pred_emit(&p, MDB_P_CONST, 0, 0, 0, 0, 1) || pred_emit(&p, MDB_P_XOR, 0, 0, 0, 0, 0)))) {
			ERR("execution phase: predicate too large for the device program (max %d steps, %d columns)\n", MDB_PRED_MAX_INSNS,
			    MDB_PRED_MAX_SLOTS);
			return -MIDORIDB_ERROR;
		}
		v = dalloc(x, t->nrows * 4);
		if (!v)
			return dev_fail(x, "allocating the selection vector");
		if (mdb_dev_filter(x->dev, p.insn, p.n, p.cols, p.ncols, t->nrows, v, m))
			return dev_fail(x, "filter");
		*sel = v;
	} else if (complement) {
		*m = 0;		/* no WHERE: nothing is kept */
	}
	return MIDORIDB_OK;
}

int mdb_exec_delete(struct mdb_catalog *cat, struct mdb_dml *d, size_t *n_rows_aff, char *err, size_t errlen)
{
	stmt_dict = &cat->dict;
	struct exec x;
	struct mdb_select s;
	struct mdb_from_tab tab;
	struct mdb_table *t = NULL;
	const uint32_t *keep = NULL;
	uint32_t *h_keep = NULL;
	uint64_t n_keep = 0, n_old;
	int rc;

	memset(&x, 0, sizeof(x));
	for (int t = 0; t < MDB_MAX_TABS; t++)
		x.same_col[t] = -1;
	*n_rows_aff = 0;
	rc = dml_select_rows(cat, d, &x, &s, &tab, true, &t, &keep, &n_keep, err, errlen);
	if (rc)
		goto out;
	n_old = t->nrows;
	if (n_keep == n_old)
		goto out;	/* nothing matched */
	if (t->dev_cap < n_old)
		t->dev_cap = n_old;
	/* ---- device mirror: order-preserving compaction of every column */
	for (int c = 0; c < t->ncols; c++) {
		struct mdb_column *col = &t->cols[c];
		void *nd = NULL;
		uint64_t *nb = NULL;
		if (!col->d_data)
			continue;
		if (n_keep) {
			const uint64_t words = (t->dev_cap + 63) / 64;
			if (mdb_dev_alloc(x.dev, t->dev_cap * 8, &nd) ||
			    (col->d_nullbits && (mdb_dev_alloc(x.dev, words * 8, (void **)&nb) || mdb_dev_memset(x.dev, nb, 0, words * 8))) ||
			    mdb_dev_gather64(x.dev, col->d_data, col->d_nullbits, keep, n_keep, nd, nb)) {
				if (nd)
					mdb_dev_free(x.dev, nd);
				if (nb)
					mdb_dev_free(x.dev, nb);
				rc = dev_fail(&x, "compacting a column");
				/* columns already swapped are shorter than the rest: drop the mirror, the host copy is intact */
				t->dev_generation = 0;
				goto out;
			}
		}
		if (mdb_dev_sync(x.dev)) {
			rc = dev_fail(&x, "compacting a column");
			t->dev_generation = 0;
			goto out;
		}
		mdb_dev_free(x.dev, col->d_data);
		if (col->d_nullbits)
			mdb_dev_free(x.dev, col->d_nullbits);
		col->d_data = nd;
		col->d_nullbits = nb;
	}
	/* ---- host copy */
	if (!t->device_only) {
		if (n_keep) {
			h_keep = malloc(n_keep * 4);
			if (!h_keep) {
				rc = -MIDORIDB_NOMEM;
				t->dev_generation = 0;
				goto out;
			}
			if (mdb_dev_d2h(x.dev, h_keep, keep, n_keep * 4)) {
				rc = dev_fail(&x, "reading the surviving row ids");
				t->dev_generation = 0;
				goto out;
			}
		}
		for (int c = 0; c < t->ncols; c++) {
			struct mdb_column *col = &t->cols[c];
			uint64_t nulls = 0;
			for (uint64_t k = 0; k < n_keep; k++) {		/* ascending ids: in place */
				const uint64_t r = h_keep[k];
				const bool isnull = (col->nullbits[r >> 6] >> (r & 63)) & 1;
				col->data[k] = col->data[r];
				if (isnull)
					col->nullbits[k >> 6] |= 1ull << (k & 63);
				else
					col->nullbits[k >> 6] &= ~(1ull << (k & 63));
				nulls += isnull;
			}
			for (uint64_t k = n_keep; k < n_old; k++)		/* vacated tail: clean bits for later appends */
				col->nullbits[k >> 6] &= ~(1ull << (k & 63));
			col->null_count = nulls;
		}
	}
	t->nrows = n_keep;
	t->generation++;
	if (t->dev_generation) {
		if (n_keep == 0) {
			t->dev_generation = 0;	/* empty mirror: rebuilt by the next upload */
			t->dev_rows = 0;
			t->dev_cap = 0;
		} else {
			t->dev_generation = t->generation;
			t->dev_rows = n_keep;
		}
	}
	*n_rows_aff = (size_t)(n_old - n_keep);
out:
	free(h_keep);
	free_all(&x);
	return rc;
}

int mdb_exec_update(struct mdb_catalog *cat, struct mdb_dml *d, size_t *n_rows_aff, char *err, size_t errlen)
{
	stmt_dict = &cat->dict;
	struct exec x;
	struct mdb_select s;
	struct mdb_from_tab tab;
	struct mdb_table *t = mdb_catalog_find(cat, d->name);
	const uint32_t *sel = NULL;
	uint32_t *h_sel = NULL;
	uint64_t m = 0;
	int acol[MDB_MAX_COLS];
	int rc = MIDORIDB_OK;

	memset(&x, 0, sizeof(x));
	for (int t = 0; t < MDB_MAX_TABS; t++)
		x.same_col[t] = -1;
	*n_rows_aff = 0;
	if (!t) {
		ERR("table '%s' doesn't exist\n", d->name);
		return -MIDORIDB_ERROR;
	}
	if (d->nassign > MDB_MAX_COLS) {
		ERR("too many assignments\n");
		return -MIDORIDB_ERROR;
	}
	/* assignment checks (semantic_update.c:418-460: the value must have the column's type, NULL always fits) */
	for (int a = 0; a < d->nassign; a++) {
		const struct mdb_expr *v = d->assign[a].val;
		acol[a] = -1;
		for (int c = 0; c < t->ncols; c++)
			if (strcmp(t->cols[c].name, d->assign[a].col) == 0)
				acol[a] = c;
		if (acol[a] < 0) {
			ERR("no such column: '%.128s'\n", d->assign[a].col);
			return -MIDORIDB_ERROR;
		}
		if (v->kind != MDB_EX_INT && v->kind != MDB_EX_FLOAT && v->kind != MDB_EX_NULL && v->kind != MDB_EX_BOOL && v->kind != MDB_EX_STRING) {
			ERR("only literal values can be assigned on the MI355X path\n");
			return -MIDORIDB_ERROR;
		}
		if (v->kind == MDB_EX_INT && t->cols[acol[a]].type != MDB_CT_INTEGER) {
			ERR("val: '%ld' requires an INTEGER column\n", (long)v->ival);
			return -MIDORIDB_ERROR;
		}
		if (v->kind == MDB_EX_FLOAT && t->cols[acol[a]].type != MDB_CT_DOUBLE) {
			ERR("val: '%f' requires a DOUBLE column\n", v->dval);
			return -MIDORIDB_ERROR;
		}
		if (v->kind == MDB_EX_BOOL && t->cols[acol[a]].type != MDB_CT_TINYINT) {
			ERR("val: '%d' requires a TINYINT column\n", (int)v->ival);
			return -MIDORIDB_ERROR;
		}
		if (v->kind == MDB_EX_STRING) {
			const struct mdb_column *col = &t->cols[acol[a]];
			int64_t tv;
			if (col->type == MDB_CT_DATE || col->type == MDB_CT_DATETIME) {
				if (!mdb_parse_time(v->sval, col->type, &tv)) {
					ERR("val: '%.256s' can't be parsed for DATE | DATETIME column\n", v->sval);
					return -MIDORIDB_ERROR;
				}
			} else if (col->type == MDB_CT_VARCHAR) {
				/* UPDATE does not check the length (INSERT does): the reference copies the first precision - 1 characters
				 * (strncpy, executor_update.c:425-426); the literal is cut here so that everything below sees that string */
				const size_t len = strlen(v->sval) - 2, keep = col->precision > 0 ? (size_t)col->precision - 1 : 0;
				if (len > keep) {
					v->sval[1 + keep] = v->sval[0];
					v->sval[2 + keep] = 0;
				}
				if (!mdb_dict_intern(&cat->dict, v->sval + 1, strlen(v->sval) - 2))	/* lit_bits_for() finds the id below */
					return -MIDORIDB_NOMEM;
			} else {
				ERR("val: '%.256s' requires an VARCHAR() column\n", v->sval);
				return -MIDORIDB_ERROR;
			}
		}
		if (v->kind == MDB_EX_NULL && t->cols[acol[a]].not_null) {
			ERR("NOT NULL constraint failed: %s.%s\n", t->name, t->cols[acol[a]].name);
			return -MIDORIDB_ERROR;
		}
	}
	rc = dml_select_rows(cat, d, &x, &s, &tab, false, &t, &sel, &m, err, errlen);
	if (rc || m == 0)
		goto out;
	/* ---- device mirror */
	for (int a = 0; a < d->nassign; a++) {
		struct mdb_column *col = &t->cols[acol[a]];
		const struct mdb_expr *v = d->assign[a].val;
		const bool set_null = v->kind == MDB_EX_NULL;
		if (set_null && !col->d_nullbits) {
			const uint64_t words = (t->dev_cap + 63) / 64;
			if (mdb_dev_alloc(x.dev, words * 8, (void **)&col->d_nullbits) || mdb_dev_memset(x.dev, col->d_nullbits, 0, words * 8)) {
				rc = dev_fail(&x, "allocating a NULL bitmap");
				t->dev_generation = 0;
				goto out;
			}
		}
		/* a device-resident result may hold the column as well (mdb_dev_retain, mdb_exec.c): it keeps the old cells, the table moves on */
		for (int pass = 0; pass < 2; pass++) {
			void **slot = pass ? (void **)&col->d_nullbits : &col->d_data;
			if (!*slot || !mdb_dev_holders(x.dev, *slot))
				continue;
			const size_t bytes = pass ? (size_t)((t->dev_cap + 63) / 64) * 8 : (size_t)t->dev_cap * 8;
			void *mine = NULL;
			if (mdb_dev_alloc(x.dev, bytes, &mine) || mdb_dev_gather64(x.dev, *slot, NULL, NULL, bytes / 8, mine, NULL)) {
				rc = dev_fail(&x, "copying a column that a result still reads");
				t->dev_generation = 0;
				goto out;
			}
			mdb_dev_free(x.dev, *slot);	/* (one holder less: the results keep it) */
			*slot = mine;
		}
		if (mdb_dev_scatter_set64(x.dev, col->d_data, col->d_nullbits, sel, m, set_null ? 0 : lit_bits_for(v, col->type), set_null)) {
			rc = dev_fail(&x, "updating a column");
			t->dev_generation = 0;
			goto out;
		}
	}
	if (mdb_dev_sync(x.dev)) {
		rc = dev_fail(&x, "updating a column");
		t->dev_generation = 0;
		goto out;
	}
	/* ---- host copy */
	if (!t->device_only) {
		if (sel) {
			h_sel = malloc(m * 4);
			if (!h_sel) {
				rc = -MIDORIDB_NOMEM;
				t->dev_generation = 0;
				goto out;
			}
			if (mdb_dev_d2h(x.dev, h_sel, sel, m * 4)) {
				rc = dev_fail(&x, "reading the selected row ids");
				t->dev_generation = 0;
				goto out;
			}
		}
		for (int a = 0; a < d->nassign; a++) {
			struct mdb_column *col = &t->cols[acol[a]];
			const struct mdb_expr *v = d->assign[a].val;
			const int64_t bits = v->kind == MDB_EX_NULL ? 0 : lit_bits_for(v, col->type);
			for (uint64_t k = 0; k < m; k++) {
				const uint64_t r = h_sel ? h_sel[k] : k;
				const bool was_null = (col->nullbits[r >> 6] >> (r & 63)) & 1;
				if (v->kind == MDB_EX_NULL) {
					col->nullbits[r >> 6] |= 1ull << (r & 63);
					col->null_count += !was_null;
				} else {
					col->data[r] = bits;
					col->nullbits[r >> 6] &= ~(1ull << (r & 63));
					col->null_count -= was_null;
				}
			}
		}
	}
	t->generation++;
	if (t->dev_generation)
		t->dev_generation = t->generation;
	*n_rows_aff = (size_t)m;
out:
	free(h_sel);
	free_all(&x);
	return rc;
}
