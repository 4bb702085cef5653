/*
 * mdb_exec.c - the SELECT executor: lowers a statement plan onto the device C-ABI (mdb_dev.h).
 *
 * This file is the MI355X replacement of executor_run_select_stmt() (reference
 * src/engine/executor_select.c:1655-1744) and keeps its phase order:
 *
 *   reference phase (file:line)                        here
 *   -------------------------------------------------  --------------------------------------------
 *   build_cols_hashtable/build_table_scafold :267-322  result column set + order (R3), host only
 *   proc_from_clause_table :1282-1343 (scan)           tuple stream = identity row ids (no copy)
 *   _join_nested_loop_tbl2tbl :1076-1149               mdb_dev_join_pairs (+ residual ON filter)
 *   _join_nested_loop_tbl2mat :1151-1232 (3-way)       the same operator applied to the joined
 *                                                      stream (true (A x B) x C; the reference's
 *                                                      own tbl2mat is defective)
 *   proc_where_clause :1435-1463                       mdb_dev_filter: conjuncts that read one table filter
 *                                                      it BEFORE the join (same rows for inner joins),
 *                                                      the rest run over the joined stream
 *   proc_groupby_clause :1526-1588                     mdb_dev_group_count over the gathered key
 *                                                      (several fields: mdb_dev_group_count_multi)
 *   proc_select_clause :1369-1433 (projection)         mdb_dev_gather64 of the selected columns only
 *   handle_countonly_case :1590-1653                   COUNT(*) = stream length
 *   table_vacuum :1726                                 nothing to do (streams are always compact)
 *
 * plus one fused plan for the north-star shape (JOIN ... ON l = r GROUP BY that key, COUNT(*)),
 * which never materialises the joined rows (mdb_dev_join_group_count); it is chained over further
 * tables joined on the same key and also answers SELECT COUNT(*) over such joins.  After the
 * reference's phases come the clauses it parses but never executes - HAVING, DISTINCT, ORDER BY,
 * LIMIT (select_tail) - and, at the end of the file, DELETE and UPDATE on the device mirror.
 *
 * Early materialisation is replaced by late materialisation: a tuple stream is a set of uint32
 * row-id vectors (one per FROM table; NULL = identity), so joins / filters / grouping move 4-byte
 * ids and only the selected columns are gathered once at the end.
 *
 * The plan normalisation the reference does in its optimiser (optimiser_select.c:114-238: NAME ->
 * fully qualified FIELDNAME, alias -> table name, SELECT * expansion) and the checks of its
 * semantic phase that guard the executor (semantic_select.c: unknown table/column, duplicate
 * column names across FROM tables S1, operand types S2, GROUP BY rule S4) happen in
 * resolve_select() below.
 */
#include "mdb_host.h"
#include <time.h>

#define ERR(...) snprintf(err, errlen, __VA_ARGS__)

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

/* ------------------------------------------------------------------ plan resolution */

static bool field_eq(const struct mdb_expr *a, const struct mdb_expr *b)
{
	return a->kind == MDB_EX_FIELD && b->kind == MDB_EX_FIELD && a->tbl_idx == b->tbl_idx && a->col_idx == b->col_idx;
}

static int resolve_expr(struct mdb_select *s, struct mdb_expr *e, char *err, size_t errlen)
{
	int rc;

	if (!e)
		return MIDORIDB_OK;
	if (e->kind == MDB_EX_NAME) {
		int ft = -1, fc = -1, hits = 0;
		for (int t = 0; t < s->ntabs; t++)
			for (int c = 0; c < s->tabs[t].t->ncols; c++)
				if (strcmp(s->tabs[t].t->cols[c].name, e->col) == 0) {
					ft = t;
					fc = c;
					hits++;
				}
		if (hits == 0) {
			ERR("no such column: '%.128s'\n", e->col);
			return -MIDORIDB_ERROR;
		}
		if (hits > 1) {
			ERR("ambiguous column name: '%.128s'\n", e->col);
			return -MIDORIDB_ERROR;
		}
		e->kind = MDB_EX_FIELD;
		e->tbl_idx = ft;
		e->col_idx = fc;
		mdb_copy_name(e->tbl, s->tabs[ft].t->name);
	} else if (e->kind == MDB_EX_FIELD) {
		int ft = -1;
		for (int t = 0; t < s->ntabs; t++)
			if (strcmp(s->tabs[t].alias, e->tbl) == 0 || (!s->tabs[t].alias[0] && strcmp(s->tabs[t].name, e->tbl) == 0) ||
			    strcmp(s->tabs[t].name, e->tbl) == 0)
				ft = t;
		if (ft < 0) {
			ERR("table is not part of from clause: '%.128s'\n", e->tbl);
			return -MIDORIDB_ERROR;
		}
		e->tbl_idx = ft;
		e->col_idx = -1;
		for (int c = 0; c < s->tabs[ft].t->ncols; c++)
			if (strcmp(s->tabs[ft].t->cols[c].name, e->col) == 0)
				e->col_idx = c;
		if (e->col_idx < 0) {
			ERR("no such column: '%.128s'.'%.128s'\n", e->tbl, e->col);
			return -MIDORIDB_ERROR;
		}
		mdb_copy_name(e->tbl, s->tabs[ft].t->name);	/* alias -> real table name */
	}
	if (e->kind == MDB_EX_FIELD) {
		e->type = s->tabs[e->tbl_idx].t->cols[e->col_idx].type;
	}
	for (int i = 0; i < e->nkids; i++)
		if ((rc = resolve_expr(s, e->kids[i], err, errlen)))
			return rc;
	return MIDORIDB_OK;
}

/* predicate shape check (what the device predicate compiler accepts); in the HAVING clause COUNT(*) is an
 * INTEGER operand of comparisons (semantic_select.c:1983-1985 lets it through) */
static bool is_having_clause(const char *clause)
{
	return strcmp(clause, "having") == 0;
}

/* type of a comparison operand as the reference's semantic phase sees it (check_value_types_cmp, semantic_select.c:2135-2186):
 * a raw string is a VARCHAR - SELECT does not box it into a DATE ("raw values are not auto-boxed", executor_select.c:193) -
 * while DELETE / UPDATE parse it against a DATE / DATETIME column (semantic_delete.c:160-200): `dml` */
static int operand_type(const struct mdb_expr *o, const struct mdb_expr *other, bool dml)
{
	switch (o->kind) {
	case MDB_EX_FIELD: return o->type;
	case MDB_EX_INT: case MDB_EX_COUNT: return MDB_CT_INTEGER;
	case MDB_EX_FLOAT: return MDB_CT_DOUBLE;
	case MDB_EX_BOOL: return MDB_CT_TINYINT;
	case MDB_EX_STRING:
		if (dml && other->kind == MDB_EX_FIELD && (other->type == MDB_CT_DATE || other->type == MDB_CT_DATETIME))
			return other->type;
		return MDB_CT_VARCHAR;
	default: return -1;
	}
}

static int check_predicate_x(const struct mdb_expr *e, const char *clause, bool dml, char *err, size_t errlen)
{
	int rc;
	switch (e->kind) {
	case MDB_EX_LOGOP:
		if ((rc = check_predicate_x(e->kids[0], clause, dml, err, errlen)) || (rc = check_predicate_x(e->kids[1], clause, dml, err, errlen)))
			return rc;
		return MIDORIDB_OK;
	case MDB_EX_CMP: {
		const struct mdb_expr *l = e->kids[0], *r = e->kids[1];
		for (int i = 0; i < 2; i++) {
			const struct mdb_expr *o = e->kids[i];
			if (o->kind == MDB_EX_COUNT && is_having_clause(clause))
				continue;
			if (o->kind != MDB_EX_FIELD && o->kind != MDB_EX_INT && o->kind != MDB_EX_FLOAT && o->kind != MDB_EX_NULL &&
			    o->kind != MDB_EX_BOOL && o->kind != MDB_EX_STRING) {
				ERR("expressions in %s clause must compare columns with literal values\n", clause);
				return -MIDORIDB_ERROR;
			}
		}
		/* operand types must match exactly (reference check_value_types_cmp, semantic_select.c:2135-2186) */
		{
			const int tl = operand_type(l, r, dml), tr = operand_type(r, l, dml);
			int64_t tv;
			if (tl >= 0 && tr >= 0 && tl != tr) {
				ERR("comparison operands must have the same type\n");
				return -MIDORIDB_ERROR;
			}
			/* VARCHAR cells are dictionary ids: equal strings, equal ids - and nothing else (semantic_select.c:2171-2176,
			 * semantic_delete.c:211-216) */
			if ((tl == MDB_CT_VARCHAR || tr == MDB_CT_VARCHAR) && e->op != MDB_CMP_EQ && e->op != MDB_CMP_NE) {
				if (dml)
					ERR("VARCHAR fields can only use '=' or '<>' ops\n");
				else
					ERR("VARCHAR values can only use '=' or '<>' ops\n");
				return -MIDORIDB_ERROR;
			}
			for (int i = 0; i < 2; i++)
				if (e->kids[i]->kind == MDB_EX_STRING && (i ? tl : tr) != MDB_CT_VARCHAR &&
				    !mdb_parse_time(e->kids[i]->sval, i ? tl : tr, &tv)) {
					ERR("val: '%.256s' can't be parsed for DATE | DATETIME column\n", e->kids[i]->sval);
					return -MIDORIDB_ERROR;
				}
			if ((l->kind == MDB_EX_NULL || r->kind == MDB_EX_NULL) && e->op != MDB_CMP_EQ && e->op != MDB_CMP_NE) {
				ERR("NULL values can only use '=' or '<>' ops\n");
				return -MIDORIDB_ERROR;
			}
		}
		return MIDORIDB_OK;
	}
	case MDB_EX_ISNULL:
		if (e->kids[0]->kind != MDB_EX_FIELD) {
			ERR("only fields are allowed in IS NULL|IS NOT NULL\n");
			return -MIDORIDB_ERROR;
		}
		return MIDORIDB_OK;
	case MDB_EX_ISIN:
		if (e->kids[0]->kind != MDB_EX_FIELD) {
			ERR("Fields aren't allowed on IN-clauses\n");
			return -MIDORIDB_ERROR;
		}
		for (int i = 1; i < e->nkids; i++) {
			const struct mdb_expr *v = e->kids[i];
			if (v->kind != MDB_EX_INT && v->kind != MDB_EX_FLOAT && v->kind != MDB_EX_NULL && v->kind != MDB_EX_BOOL &&
			    v->kind != MDB_EX_STRING) {
				ERR("IN-clause can only contain raw values\n");
				return -MIDORIDB_ERROR;
			}
			if (v->kind == MDB_EX_STRING && e->kids[0]->type != MDB_CT_VARCHAR) {	/* (semantic_select.c:2308-2326) */
				ERR("val: '%.256s' requires an VARCHAR() column\n", v->sval);
				return -MIDORIDB_ERROR;
			}
			if ((v->kind == MDB_EX_INT && e->kids[0]->type != MDB_CT_INTEGER) ||
			    (v->kind == MDB_EX_BOOL && e->kids[0]->type != MDB_CT_TINYINT) ||
			    (v->kind == MDB_EX_FLOAT && e->kids[0]->type != MDB_CT_DOUBLE)) {
				ERR("comparison operands must have the same type\n");
				return -MIDORIDB_ERROR;
			}
		}
		return MIDORIDB_OK;
	case MDB_EX_COUNT:
		ERR("COUNT function can't be used in the %s-clause\n", clause);
		return -MIDORIDB_ERROR;
	default:
		ERR("expressions in %s clause must be a type of comparison\n", clause);
		return -MIDORIDB_ERROR;
	}
}

static int check_predicate(const struct mdb_expr *e, const char *clause, char *err, size_t errlen)
{
	return check_predicate_x(e, clause, false, err, errlen);
}

static bool expr_has_count(const struct mdb_expr *e)
{
	if (e->kind == MDB_EX_COUNT)
		return true;
	for (int i = 0; i < e->nkids; i++)
		if (expr_has_count(e->kids[i]))
			return true;
	return false;
}

/* every field under e appears in the select list ("SELECT list is not in <clause> clause", the reference's
 * wording, semantic_select.c:1836-1852, 1965-1985) */
static int fields_in_select_list(const struct mdb_select *s, const struct mdb_expr *e, const char *clause, char *err, size_t errlen)
{
	int rc;
	if (e->kind == MDB_EX_FIELD && !s->select_all) {
		bool ok = false;
		for (int i = 0; i < s->nsel; i++)
			ok |= field_eq(s->sel[i], e);
		if (!ok) {
			ERR("SELECT list is not in %s clause: '%.128s'.'%.128s'\n", clause, e->tbl, e->col);
			return -MIDORIDB_ERROR;
		}
	}
	if (e->kind == MDB_EX_COUNT)
		return MIDORIDB_OK;
	for (int i = 0; i < e->nkids; i++)
		if ((rc = fields_in_select_list(s, e->kids[i], clause, err, errlen)))
			return rc;
	return MIDORIDB_OK;
}

static int resolve_select(struct mdb_catalog *cat, struct mdb_select *s, char *err, size_t errlen)
{
	int rc;

	if (s->ntabs == 0) {
		ERR("SELECT without FROM is not supported by the MI355X path\n");
		return -MIDORIDB_ERROR;
	}
	if (s->ntabs > MDB_MAX_TABS) {
		ERR("more than %d tables in the FROM clause are not supported\n", MDB_MAX_TABS);
		return -MIDORIDB_ERROR;
	}
	for (int t = 0; t < s->ntabs; t++) {
		s->tabs[t].t = mdb_catalog_find(cat, s->tabs[t].name);
		if (!s->tabs[t].t) {
			ERR("table doesn't exist: '%.128s'\n", s->tabs[t].name);
			return -MIDORIDB_ERROR;
		}
		for (int u = 0; u < t; u++) {
			const char *a = s->tabs[t].alias[0] ? s->tabs[t].alias : s->tabs[t].name;
			const char *b = s->tabs[u].alias[0] ? s->tabs[u].alias : s->tabs[u].name;
			if (strcmp(a, b) == 0) {
				ERR("Not unique table/alias: '%.128s'\n", a);
				return -MIDORIDB_ERROR;
			}
			/* S1: bare column names must be unique across all FROM tables (semantic_select.c:2470-2478) */
			for (int c = 0; c < s->tabs[t].t->ncols; c++)
				for (int d = 0; d < s->tabs[u].t->ncols; d++)
					if (strcmp(s->tabs[t].t->cols[c].name, s->tabs[u].t->cols[d].name) == 0) {
						ERR("duplicate column name: '%s'\n", s->tabs[t].t->cols[c].name);
						return -MIDORIDB_ERROR;
					}
		}
		if (s->join_type[t] != 1) {
			ERR("only INNER JOIN is executed (the reference aborts on other join types, executor_select.c:1094)\n");
			return -MIDORIDB_ERROR;
		}
	}
	for (int i = 0; i < s->nsel; i++) {
		struct mdb_expr *e = s->sel[i];
		if (e->kind == MDB_EX_COUNT) {
			for (int k = 0; k < e->nkids; k++)
				if ((rc = resolve_expr(s, e->kids[k], err, errlen)))
					return rc;
			continue;
		}
		if (e->kind != MDB_EX_NAME && e->kind != MDB_EX_FIELD) {
			ERR("only columns and COUNT(*) are supported in the select list (aliases and expressions are not executed by the reference)\n");
			return -MIDORIDB_ERROR;
		}
		if ((rc = resolve_expr(s, e, err, errlen)))
			return rc;
	}
	for (int t = 1; t < s->ntabs; t++)
		if (s->on[t]) {
			if ((rc = resolve_expr(s, s->on[t], err, errlen)) || (rc = check_predicate(s->on[t], "JOIN ON", err, errlen)))
				return rc;
		}
	if (s->where && ((rc = resolve_expr(s, s->where, err, errlen)) || (rc = check_predicate(s->where, "where", err, errlen))))
		return rc;
	if (s->ngroup > MDB_SORT_MAX_KEYS) {
		ERR("GROUP BY over more than %d fields is not supported\n", MDB_SORT_MAX_KEYS);
		return -MIDORIDB_ERROR;
	}
	for (int i = 0; i < s->ngroup; i++) {
		if (s->group[i]->kind != MDB_EX_NAME && s->group[i]->kind != MDB_EX_FIELD) {
			ERR("group-by clauses support only fields and aliases\n");
			return -MIDORIDB_ERROR;
		}
		if ((rc = resolve_expr(s, s->group[i], err, errlen)))
			return rc;
	}
	/* S4: with GROUP BY or COUNT, every plain select field must be a GROUP BY field */
	{
		int ncount = 0, nfield = 0;
		for (int i = 0; i < s->nsel; i++) {
			if (s->sel[i]->kind == MDB_EX_COUNT) {
				ncount++;
				continue;
			}
			nfield++;
			if (s->ngroup) {
				bool ok = false;
				for (int g = 0; g < s->ngroup; g++)
					ok |= field_eq(s->sel[i], s->group[g]);
				if (!ok) {
					ERR("SELECT list is not in GROUP BY clause: '%.128s'.'%.128s'\n", s->sel[i]->tbl, s->sel[i]->col);
					return -MIDORIDB_ERROR;
				}
			}
		}
		if (s->select_all && (s->ngroup || ncount)) {
			ERR("SELECT * can't be combined with GROUP BY / COUNT\n");
			return -MIDORIDB_ERROR;
		}
		if (ncount && nfield && !s->ngroup) {
			ERR("mixing fields and COUNT in the select list requires a GROUP BY clause\n");
			return -MIDORIDB_ERROR;
		}
		/* ---- DISTINCT / HAVING / ORDER BY / LIMIT: parsed and checked but never executed upstream (SURVEY 8a
		 *      D7); executed here with SQL semantics (8f row 4), under the reference's own semantic rules */
		if (s->distinct && (s->ngroup || ncount)) {
			ERR("DISTINCT can't be combined with GROUP BY / COUNT on the MI355X path\n");
			return -MIDORIDB_ERROR;
		}
		if (s->having) {
			if ((rc = resolve_expr(s, s->having, err, errlen)) || (rc = check_predicate(s->having, "having", err, errlen)))
				return rc;
			/* fields must come from the SELECT list (check_having_clause_inselect, semantic_select.c:1953-2001) */
			if ((rc = fields_in_select_list(s, s->having, "HAVING", err, errlen)))
				return rc;
			if (expr_has_count(s->having) && !s->ngroup) {
				ERR("COUNT in the having-clause requires a GROUP BY clause on the MI355X path\n");
				return -MIDORIDB_ERROR;
			}
			if (ncount && !s->ngroup) {
				ERR("HAVING over an ungrouped COUNT is not supported on the MI355X path\n");
				return -MIDORIDB_ERROR;
			}
		}
		for (int i = 0; i < s->norder; i++) {
			struct mdb_expr *o = s->order[i];
			if (o->kind == MDB_EX_COUNT) {		/* check_orderby_clause_count, semantic_select.c:1755-1795 */
				ERR("COUNT function can't be used in the orderby-clause\n");
				return -MIDORIDB_ERROR;
			}
			if (o->kind != MDB_EX_NAME && o->kind != MDB_EX_FIELD) {	/* check_orderby_clause_expr :1718-1753 */
				ERR("order-by clauses support only fields and aliases\n");
				return -MIDORIDB_ERROR;
			}
			if ((rc = resolve_expr(s, o, err, errlen)) || (rc = fields_in_select_list(s, o, "ORDER BY", err, errlen)))
				return rc;
			if (o->type == MDB_CT_VARCHAR) {	/* cells are dictionary ids: equality only, no collation order */
				ERR("ORDER BY over the VARCHAR column '%s.%s' is not supported on the MI355X path\n", o->tbl, o->col);
				return -MIDORIDB_ERROR;
			}
		}
		if (s->norder > MDB_SORT_MAX_KEYS) {
			ERR("too many ORDER BY items (max %d)\n", MDB_SORT_MAX_KEYS);
			return -MIDORIDB_ERROR;
		}
	}
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ device-side execution state */

struct dbuf_list {
	void **p;
	int n, cap;
};

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
	bool need[MDB_MAX_TABS][MDB_MAX_COLS];
};

static int dev_fail(struct exec *x, const char *what)
{
	snprintf(x->err, x->errlen, "execution phase: %s: %s\n", what, mdb_dev_last_error(x->dev));
	return -MIDORIDB_INTERNAL;
}

static int track(struct exec *x, void *p)
{
	if (x->bufs.n == x->bufs.cap) {
		int nc = x->bufs.cap ? x->bufs.cap * 2 : 32;
		void **np = realloc(x->bufs.p, sizeof(void *) * (size_t)nc);
		if (!np)
			return -MIDORIDB_NOMEM;
		x->bufs.p = np;
		x->bufs.cap = nc;
	}
	x->bufs.p[x->bufs.n++] = p;
	return 0;
}

static void *dalloc(struct exec *x, size_t bytes)
{
	void *p = NULL;
	if (mdb_dev_alloc(x->dev, bytes ? bytes : 8, &p))
		return NULL;
	if (track(x, p)) {
		mdb_dev_free(x->dev, p);
		return NULL;
	}
	return p;
}

static void free_all(struct exec *x)
{
	for (int i = 0; i < x->bufs.n; i++)
		mdb_dev_free(x->dev, x->bufs.p[i]);
	free(x->bufs.p);
	x->bufs.p = NULL;
	x->bufs.n = x->bufs.cap = 0;
}

/* Re-map every row-id vector of the stream through `sel` (n_new positions into the old stream). */
static int stream_select(struct exec *x, int ntabs_in_stream, const uint32_t *sel, uint64_t n_new)
{
	for (int t = 0; t < ntabs_in_stream; t++) {
		if (x->rid[t]) {
			uint32_t *nr = dalloc(x, n_new * 4);
			if (!nr)
				return dev_fail(x, "allocating row ids");
			if (n_new && mdb_dev_gather32(x->dev, x->rid[t], sel, n_new, nr))
				return dev_fail(x, "re-mapping row ids");
			x->rid[t] = nr;
		} else {
			x->rid[t] = (uint32_t *)sel;	/* identity composed with sel */
		}
	}
	x->n = n_new;
	return MIDORIDB_OK;
}

/* keep the rows sel[0..n_new) of the current stream: row-id vectors (or the fused key column) and COUNT(*) */
static int stream_apply_sel(struct exec *x, int ntabs_in_stream, const uint32_t *sel, uint64_t n_new)
{
	if (x->d_count) {
		int64_t *nc = dalloc(x, (n_new ? n_new : 1) * 8);
		if (!nc || (n_new && mdb_dev_gather64(x->dev, x->d_count, NULL, sel, n_new, nc, NULL)))
			return dev_fail(x, "re-mapping COUNT(*)");
		x->d_count = nc;
	}
	if (x->fused) {
		int64_t *nk = dalloc(x, (n_new ? n_new : 1) * 8);
		if (!nk || (n_new && mdb_dev_gather64(x->dev, x->d_fused_key, NULL, sel, n_new, nk, NULL)))
			return dev_fail(x, "re-mapping the group key");
		x->d_fused_key = nk;
		x->n = n_new;
		return MIDORIDB_OK;
	}
	return stream_select(x, ntabs_in_stream, sel, n_new);
}

/* device pointer to a column's key/value vector for the current stream (gathered when needed) */
static int stream_column(struct exec *x, const struct mdb_expr *f, const int64_t **vals, const uint64_t **nulls)
{
	struct mdb_column *col = &x->s->tabs[f->tbl_idx].t->cols[f->col_idx];
	*vals = col->d_data;
	*nulls = col->d_nullbits;
	if (x->rid[f->tbl_idx] && x->n) {
		int64_t *v = dalloc(x, x->n * 8);
		uint64_t *nb = col->d_nullbits ? dalloc(x, ((x->n + 63) / 64) * 8) : NULL;
		if (!v || (col->d_nullbits && !nb))
			return dev_fail(x, "allocating a key column");
		if (mdb_dev_gather64(x->dev, col->d_data, col->d_nullbits, x->rid[f->tbl_idx], x->n, v, nb))
			return dev_fail(x, "gathering a key column");
		*vals = v;
		*nulls = nb;
	}
	return MIDORIDB_OK;
}

/* Key vector of an equi-join over table column `col`, rows rid[0..n) (rid == NULL: rows 0..n-1).  INTEGER keys are the
 * column itself; DOUBLE keys go through mdb_dev_double_join_keys so that the join's word comparison is the reference's
 * IEEE `==` (cmp_double_value_to_value, executor_select.c:440-460): -0.0 joins +0.0, NaN joins nothing. */
static int double_join_keys(struct exec *x, const struct mdb_column *col, const uint32_t *rid, uint64_t n, const void **vals,
			    const uint64_t **nulls)
{
	int64_t *v = dalloc(x, (n ? n : 1) * 8);
	uint64_t *nb = dalloc(x, ((n + 63) / 64 + 1) * 8);
	if (!v || !nb)
		return dev_fail(x, "allocating a key column");
	if (mdb_dev_double_join_keys(x->dev, col->d_data, col->d_nullbits, rid, n, v, nb))
		return dev_fail(x, "preparing DOUBLE join keys");
	*vals = v;
	*nulls = nb;
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ sharded mode: row exchange
 *
 * One process per GPU, every process holds ITS rows of every table (include/mdb_dist.h).  The general plan stays what it
 * is - the reference's phases, executor_select.c:1655-1744 - and gains ONE step: before an operator that must see all the
 * rows of a key together (equi-join, GROUP BY, DISTINCT), the tuple stream is re-distributed so that every row lands on
 * the rank its key hashes to (mdb_dist_shuffle_rows: key + the columns the statement still reads), unless it already is
 * (x->part: the join keys tied together so far).  What arrives replaces the table for the rest of the statement. */

static void mark_needed(struct exec *x, const struct mdb_expr *e)
{
	if (!e)
		return;
	if (e->kind == MDB_EX_FIELD && e->tbl_idx >= 0 && e->tbl_idx < MDB_MAX_TABS && e->col_idx >= 0 && e->col_idx < MDB_MAX_COLS)
		x->need[e->tbl_idx][e->col_idx] = true;
	for (int i = 0; i < e->nkids; i++)
		mark_needed(x, e->kids[i]);
}

static void mark_needed_all(struct exec *x)
{
	const struct mdb_select *s = x->s;
	for (int i = 0; i < s->nsel; i++)
		mark_needed(x, s->sel[i]);
	if (s->select_all)
		for (int t = 0; t < s->ntabs; t++)
			for (int c = 0; c < s->tabs[t].t->ncols; c++)
				x->need[t][c] = true;
	for (int t = 1; t < s->ntabs; t++)
		mark_needed(x, s->on[t]);
	mark_needed(x, s->where);
	for (int g = 0; g < s->ngroup; g++)
		mark_needed(x, s->group[g]);
	mark_needed(x, s->having);
	for (int o = 0; o < s->norder; o++)
		mark_needed(x, s->order[o]);
}

static bool in_part(const struct exec *x, const struct mdb_expr *f)
{
	for (int i = 0; i < x->npart; i++)
		if (x->part[i]->tbl_idx == f->tbl_idx && x->part[i]->col_idx == f->col_idx)
			return true;
	return false;
}

/* Tables tabs[0..nt) share one tuple stream of n tuples, table tabs[i] read through rid_of[i] (NULL = identity); kv / kn is
 * the stream's partitioning key.  Afterwards each of those tables is a shadow over the rows this rank received, *n_out of
 * them, all read by identity.  Collective: every rank calls it for the same statement at the same point. */
static int shard_rows(struct exec *x, const int *tabs, int nt, uint32_t *const *rid_of, uint64_t n, const int64_t *kv, const uint64_t *kn,
		      uint32_t flags, uint64_t *n_out)
{
	struct mdb_select *s = x->s;
	struct mdb_dist_col cols[MDB_DIST_SHUFFLE_MAX_COLS] = { { NULL, NULL, NULL } };
	int col_i[MDB_DIST_SHUFFLE_MAX_COLS], col_c[MDB_DIST_SHUFFLE_MAX_COLS], nc = 0;
	void *ov[MDB_DIST_SHUFFLE_MAX_COLS];
	uint64_t *on[MDB_DIST_SHUFFLE_MAX_COLS];
	uint64_t got = 0;

	for (int i = 0; i < nt; i++) {
		const struct mdb_table *tb = s->tabs[tabs[i]].t;
		for (int c = 0; c < tb->ncols; c++) {
			if (!x->need[tabs[i]][c])
				continue;
			if (tb->cols[c].type == MDB_CT_VARCHAR) {
				snprintf(x->err, x->errlen, "execution phase: sharded mode: VARCHAR column %s.%s cannot travel between the ranks (its cells are "
							    "ids of this process's string dictionary)\n", tb->name, tb->cols[c].name);
				return -MIDORIDB_ERROR;
			}
			if (nc == MDB_DIST_SHUFFLE_MAX_COLS) {
				snprintf(x->err, x->errlen, "execution phase: sharded mode: more than %d columns in one exchange\n", MDB_DIST_SHUFFLE_MAX_COLS);
				return -MIDORIDB_ERROR;
			}
			cols[nc].values = tb->cols[c].d_data;
			cols[nc].nullbits = tb->cols[c].d_nullbits;
			cols[nc].rid = rid_of[i];
			col_i[nc] = i;
			col_c[nc] = c;
			nc++;
		}
	}
	if (mdb_dist_shuffle_rows(x->cat->dist, kv, kn, n, flags, cols, nc, ov, on, &got)) {
		snprintf(x->err, x->errlen, "execution phase: sharded exchange: %s\n", mdb_dist_last_error(x->cat->dist));
		return -MIDORIDB_INTERNAL;
	}
	for (int k = 0; k < nc; k++)
		if (track(x, ov[k]) || (on[k] && track(x, on[k])))
			return -MIDORIDB_NOMEM;
	for (int i = 0; i < nt; i++) {
		const int t = tabs[i];
		const struct mdb_table *tb = s->tabs[t].t;
		struct mdb_table *sh = calloc(1, sizeof(*sh));
		if (!sh)
			return -MIDORIDB_NOMEM;
		memcpy(sh->name, tb->name, sizeof(sh->name));
		sh->ncols = tb->ncols;
		for (int c = 0; c < tb->ncols; c++) {
			memcpy(sh->cols[c].name, tb->cols[c].name, sizeof(sh->cols[c].name));
			sh->cols[c].type = tb->cols[c].type;
			sh->cols[c].precision = tb->cols[c].precision;
			sh->cols[c].not_null = tb->cols[c].not_null;
		}
		for (int k = 0; k < nc; k++)
			if (col_i[k] == i) {
				sh->cols[col_c[k]].d_data = ov[k];
				sh->cols[col_c[k]].d_nullbits = on[k];
			}
		sh->nrows = sh->dev_rows = got;
		sh->dev_cap = got ? got : 1;
		sh->device_only = true;
		if (!x->orig_tab[t])
			x->orig_tab[t] = s->tabs[t].t;
		free(x->shadow[t]);
		x->shadow[t] = sh;
		s->tabs[t].t = sh;
	}
	*n_out = got;
	return MIDORIDB_OK;
}

/* the current stream (tables 0..nt-1) partitioned by field f: afterwards every rank holds the tuples whose f hashes to it */
static int shard_stream(struct exec *x, int nt, const struct mdb_expr *f, uint32_t flags)
{
	int tabs[MDB_MAX_TABS] = { 0 };
	uint32_t *rids[MDB_MAX_TABS] = { NULL };
	const int64_t *kv;
	const uint64_t *kn;
	const void *dv;
	uint64_t got = 0;
	int rc;
	if (f->type == MDB_CT_DOUBLE && !(flags & MDB_DIST_KEEP_NULL_KEYS)) {	/* a join key: -0.0 meets +0.0, NaN meets nothing */
		if ((rc = double_join_keys(x, &x->s->tabs[f->tbl_idx].t->cols[f->col_idx], x->rid[f->tbl_idx], x->n, &dv, &kn)))
			return rc;
		kv = dv;
	} else if ((rc = stream_column(x, f, &kv, &kn))) {
		return rc;
	}
	for (int t = 0; t < nt; t++) {
		tabs[t] = t;
		rids[t] = x->rid[t];
	}
	if ((rc = shard_rows(x, tabs, nt, rids, x->n, kv, kn, flags, &got)))
		return rc;
	for (int t = 0; t < nt; t++)
		x->rid[t] = NULL;
	x->n = got;
	x->npart = 0;
	x->part[x->npart++] = f;
	return MIDORIDB_OK;
}

/* Sharded joins on key columns of BASE tables: the exchange is told the two tables' GLOBAL key ranges from catalog statistics -
 * the smallest / largest key of each rank's mirror, computed once per table generation (one pass) and agreed on with one tiny
 * all-gather per statement - instead of measuring both columns on every call (MDB_WIRE_AUTO: two passes over the columns per
 * query).  With the ranges known the operator ships first-level partition regions (mdb_dev_shard.hip).  A filtered table's keys
 * lie inside its column's range: a superset is fine. */
static int shard_promise_ranges(struct exec *x, const struct mdb_expr *fl, const struct mdb_expr *fr)
{
	struct mdb_column *cols[2] = { &x->s->tabs[fl->tbl_idx].t->cols[fl->col_idx], &x->s->tabs[fr->tbl_idx].t->cols[fr->col_idx] };
	struct mdb_table *tabs[2] = { x->s->tabs[fl->tbl_idx].t, x->s->tabs[fr->tbl_idx].t };
	uint64_t mine[4], all[4 * 512];
	const int W = mdb_dist_world(x->cat->dist);
	/* a promise an earlier step of this statement made is about OTHER columns: forgotten before anything else, so that a return
	 * without a new promise (below) leaves the handle measuring by itself instead of holding ranges that are not these columns' */
	if (x->promised) {
		(void)mdb_dist_set_key_ranges(x->cat->dist, NULL, NULL);
		(void)mdb_dist_set_wire(x->cat->dist, MDB_WIRE_AUTO);
		x->promised = false;
	}
	if (W > 512)
		return MIDORIDB_OK;
	for (int i = 0; i < 2; i++) {
		if (cols[i]->st_generation != tabs[i]->generation + 1) {
			int64_t lo = 0, hi = -1;
			if (tabs[i]->nrows && mdb_dev_key_range(x->dev, cols[i]->d_data, cols[i]->d_nullbits, tabs[i]->nrows, &lo, &hi))
				return dev_fail(x, "column statistics");
			cols[i]->st_lo = lo;
			cols[i]->st_hi = hi;
			cols[i]->st_generation = tabs[i]->generation + 1;
		}
		const bool none = cols[i]->st_lo > cols[i]->st_hi;
		/* (as offsets from the smallest int64: every rank's minimum of the unsigned images is the global minimum) */
		mine[2 * i] = none ? ~0ull : (uint64_t)cols[i]->st_lo ^ 0x8000000000000000ull;
		mine[2 * i + 1] = none ? 0ull : (uint64_t)cols[i]->st_hi ^ 0x8000000000000000ull;
	}
	if (mdb_dist_allgather_u64(x->cat->dist, mine, 4, all)) {
		snprintf(x->err, x->errlen, "execution phase: %s\n", mdb_dist_last_error(x->cat->dist));
		return -MIDORIDB_INTERNAL;
	}
	int64_t g[2][2];
	bool fits32 = true;
	for (int i = 0; i < 2; i++) {
		uint64_t lo = ~0ull, hi = 0;
		for (int p = 0; p < W; p++) {
			lo = all[4 * p + 2 * i] < lo ? all[4 * p + 2 * i] : lo;
			hi = all[4 * p + 2 * i + 1] > hi ? all[4 * p + 2 * i + 1] : hi;
		}
		g[i][0] = (int64_t)(lo ^ 0x8000000000000000ull);
		g[i][1] = (int64_t)(hi ^ 0x8000000000000000ull);
		if (lo > hi) {		/* no key on any rank: an empty range (lo > hi) */
			g[i][0] = 0;
			g[i][1] = -1;
		} else if (g[i][0] < -(1ll << 31) || g[i][1] >= (1ll << 31)) {
			fits32 = false;
		}
	}
	if (g[0][0] > g[0][1] || g[1][0] > g[1][1])
		return MIDORIDB_OK;	/* (a table without keys: the measuring path answers "no groups") */
	if (mdb_dist_set_key_ranges(x->cat->dist, g[0], g[1]) || mdb_dist_set_wire(x->cat->dist, fits32 ? MDB_WIRE_32 : MDB_WIRE_64))
		return -MIDORIDB_INTERNAL;
	x->promised = true;
	return MIDORIDB_OK;
}

static void shard_cleanup(struct exec *x)
{
	if (x->promised) {	/* back to per-call measurement for whoever uses the handle next */
		(void)mdb_dist_set_key_ranges(x->cat->dist, NULL, NULL);
		(void)mdb_dist_set_wire(x->cat->dist, MDB_WIRE_AUTO);
		x->promised = false;
	}
	for (int t = 0; t < MDB_MAX_TABS; t++) {
		if (x->orig_tab[t])
			x->s->tabs[t].t = x->orig_tab[t];
		free(x->shadow[t]);
		x->shadow[t] = NULL;
		x->orig_tab[t] = NULL;
	}
}

/* ------------------------------------------------------------------ predicate compiler */

struct pred_prog {
	struct mdb_pred_insn insn[MDB_PRED_MAX_INSNS];
	int n;
	struct mdb_col_binding cols[MDB_PRED_MAX_SLOTS];
	int slot_tbl[MDB_PRED_MAX_SLOTS], slot_col[MDB_PRED_MAX_SLOTS];
	int ncols;
};

/* device binding of a column-like operand for the current stream: a table column read through the table's
 * row-id vector, the COUNT(*) column (HAVING), or - in the fused north-star plan, whose stream carries no row
 * ids - the group key column (the only field S4 lets such a query name) */
static void bind_operand(struct exec *x, const struct mdb_expr *f, const void **values, const uint64_t **nullbits, const uint32_t **rid)
{
	if (f->kind == MDB_EX_COUNT) {
		*values = x->d_count;
		*nullbits = NULL;
		*rid = NULL;
	} else if (x->fused) {
		*values = x->d_fused_key;
		*nullbits = NULL;
		*rid = NULL;
	} else {
		struct mdb_column *col = &x->s->tabs[f->tbl_idx].t->cols[f->col_idx];
		*values = col->d_data;
		*nullbits = col->d_nullbits;
		*rid = x->rid[f->tbl_idx];
	}
}

static int pred_slot(struct exec *x, struct pred_prog *p, const struct mdb_expr *f)
{
	const int st = f->kind == MDB_EX_COUNT ? -2 : f->tbl_idx, sc = f->kind == MDB_EX_COUNT ? -2 : f->col_idx;
	for (int i = 0; i < p->ncols; i++)
		if (p->slot_tbl[i] == st && p->slot_col[i] == sc)
			return i;
	if (p->ncols == MDB_PRED_MAX_SLOTS)
		return -1;
	p->slot_tbl[p->ncols] = st;
	p->slot_col[p->ncols] = sc;
	bind_operand(x, f, &p->cols[p->ncols].values, &p->cols[p->ncols].nullbits, &p->cols[p->ncols].rid);
	return p->ncols++;
}

static int pred_emit(struct pred_prog *p, int op, int cmp, int type, int a, int b, int64_t imm)
{
	struct mdb_pred_insn *in;
	if (p->n == MDB_PRED_MAX_INSNS)
		return -1;
	in = &p->insn[p->n++];
	memset(in, 0, sizeof(*in));
	in->op = op;
	in->cmp = cmp;
	in->type = type;
	in->a = a;
	in->b = b;
	in->imm = imm;
	return 0;
}

/* the 8 bytes a literal stands for in a column of type coltype (a DATE / DATETIME string: its time_t, validated by
 * check_predicate_x / the UPDATE checks) */
/* the string dictionary of the database the running statement belongs to (set by the statement entry points) */
static __thread const struct mdb_strdict *stmt_dict;

static int64_t lit_bits_for(const struct mdb_expr *v, int coltype)
{
	int64_t bits = 0;
	if (v->kind == MDB_EX_FLOAT) {
		memcpy(&bits, &v->dval, 8);
		return bits;
	}
	if (v->kind == MDB_EX_STRING && coltype == MDB_CT_VARCHAR)	/* a string no cell holds has id -1: equal to nothing */
		return stmt_dict ? mdb_dict_find(stmt_dict, v->sval + 1, strlen(v->sval) - 2) : -1;
	if (v->kind == MDB_EX_STRING) {
		(void)mdb_parse_time(v->sval, coltype, &bits);
		return bits;
	}
	return v->ival;
}

static bool const_cmp(int op, const struct mdb_expr *l, const struct mdb_expr *r)
{
	if (l->kind == MDB_EX_NULL || r->kind == MDB_EX_NULL)
		return false;			/* executor_select.c:660-662 */
	if (l->kind == MDB_EX_STRING || r->kind == MDB_EX_STRING)
		return false;			/* (rejected by the type check; never evaluated) */
	if (l->kind == MDB_EX_FLOAT) {
		double a = l->dval, b = r->dval;
		return op == 1 ? a < b : op == 2 ? a > b : op == 3 ? a != b : op == 4 ? a == b : op == 5 ? a <= b : a >= b;
	} else {
		int64_t a = l->ival, b = r->ival;
		return op == 1 ? a < b : op == 2 ? a > b : op == 3 ? a != b : op == 4 ? a == b : op == 5 ? a <= b : a >= b;
	}
}

static int pred_compile(struct exec *x, struct pred_prog *p, const struct mdb_expr *e)
{
	int rc = 0, a, b;

	switch (e->kind) {
	case MDB_EX_LOGOP:
		if ((rc = pred_compile(x, p, e->kids[0])) || (rc = pred_compile(x, p, e->kids[1])))
			return rc;
		return pred_emit(p, e->op == 0 ? MDB_P_AND : (e->op == 1 ? MDB_P_OR : MDB_P_XOR), 0, 0, 0, 0, 0);
	case MDB_EX_CMP: {
		const struct mdb_expr *l = e->kids[0], *r = e->kids[1];
		const bool lcol = l->kind == MDB_EX_FIELD || l->kind == MDB_EX_COUNT, rcol = r->kind == MDB_EX_FIELD || r->kind == MDB_EX_COUNT;
		if (lcol && rcol) {
			if (l->kind == MDB_EX_FIELD && l->type == MDB_CT_TINYINT && e->op != MDB_CMP_EQ && e->op != MDB_CMP_NE)
				return pred_emit(p, MDB_P_CONST, 0, 0, 0, 0, 0);
			a = pred_slot(x, p, l);
			b = pred_slot(x, p, r);
			if (a < 0 || b < 0)
				return -1;
			return pred_emit(p, MDB_P_CMP_COL_COL, e->op, (l->kind == MDB_EX_FIELD && l->type == MDB_CT_DOUBLE) ? MDB_T_DOUBLE : MDB_T_INT64,
					 a, b, 0);
		}
		if (lcol || rcol) {
			const struct mdb_expr *f = lcol ? l : r, *v = lcol ? r : l;
			if (v->kind == MDB_EX_NULL)	/* NULL operand: never true (executor_select.c:793-795) */
				return pred_emit(p, MDB_P_CONST, 0, 0, 0, 0, 0);
			/* TINYINT (bool) operands only know = and <> upstream (cmp_bool_value_to_value, executor_select.c:484-494): false */
			if (f->kind == MDB_EX_FIELD && f->type == MDB_CT_TINYINT && e->op != MDB_CMP_EQ && e->op != MDB_CMP_NE)
				return pred_emit(p, MDB_P_CONST, 0, 0, 0, 0, 0);
			a = pred_slot(x, p, f);
			if (a < 0)
				return -1;
			return pred_emit(p, lcol ? MDB_P_CMP_COL_CONST : MDB_P_CMP_CONST_COL, e->op,
					 (f->kind == MDB_EX_FIELD && f->type == MDB_CT_DOUBLE) ? MDB_T_DOUBLE : MDB_T_INT64, a, 0,
					 lit_bits_for(v, f->kind == MDB_EX_FIELD ? f->type : MDB_CT_INTEGER));
		}
		return pred_emit(p, MDB_P_CONST, 0, 0, 0, 0, const_cmp(e->op, l, r));
	}
	case MDB_EX_ISNULL:
		a = pred_slot(x, p, e->kids[0]);
		if (a < 0)
			return -1;
		return pred_emit(p, MDB_P_ISNULL, e->op ? 1 : 0, 0, a, 0, 0);
	case MDB_EX_ISIN: {
		/* x IN (v1..vk)  = (x = v1) OR ... OR (x = vk)   - SQL semantics; the reference's
		 *                  conjunction (eval_isxin :1013-1021) is defect D3, identical for k = 1
		 * x NOT IN (...) = (x <> v1) AND ... AND (x <> vk) - same as the reference */
		const struct mdb_expr *f = e->kids[0];
		a = pred_slot(x, p, f);
		if (a < 0)
			return -1;
		for (int i = 1; i < e->nkids; i++) {
			const struct mdb_expr *v = e->kids[i];
			if (v->kind == MDB_EX_NULL)
				rc = pred_emit(p, MDB_P_CONST, 0, 0, 0, 0, 0);
			else
				rc = pred_emit(p, MDB_P_CMP_COL_CONST, e->op ? MDB_CMP_NE : MDB_CMP_EQ,
					       f->type == MDB_CT_DOUBLE ? MDB_T_DOUBLE : MDB_T_INT64, a, 0, lit_bits_for(v, f->type));
			if (rc)
				return rc;
			if (i > 1 && (rc = pred_emit(p, e->op ? MDB_P_AND : MDB_P_OR, 0, 0, 0, 0, 0)))
				return rc;
		}
		return 0;
	}
	default:
		return -1;
	}
}

/* filter the current stream (tables 0..ntabs-1) by predicate e */
static int stream_filter(struct exec *x, int ntabs_in_stream, const struct mdb_expr *e)
{
	struct pred_prog p;
	uint32_t *sel;
	uint64_t m = 0;

	if (x->n == 0)
		return MIDORIDB_OK;
	memset(&p, 0, sizeof(p));
	if (pred_compile(x, &p, e)) {
		snprintf(x->err, x->errlen, "execution phase: predicate too large for the device program (max %d steps, %d columns)\n",
			 MDB_PRED_MAX_INSNS, MDB_PRED_MAX_SLOTS);
		return -MIDORIDB_ERROR;
	}
	sel = dalloc(x, x->n * 4);
	if (!sel)
		return dev_fail(x, "allocating the selection vector");
	if (mdb_dev_filter(x->dev, p.insn, p.n, p.cols, p.ncols, x->n, sel, &m))
		return dev_fail(x, "filter");
	return stream_apply_sel(x, ntabs_in_stream, sel, m);
}

/* ------------------------------------------------------------------ FROM clause */

/* split an ON expression into conjuncts; pick the first "left-stream field = field of table t" as the hash key */
static void collect_conjuncts(struct mdb_expr *e, struct mdb_expr **out, int *n, int cap)
{
	if (e->kind == MDB_EX_LOGOP && e->op == 0) {
		collect_conjuncts(e->kids[0], out, n, cap);
		collect_conjuncts(e->kids[1], out, n, cap);
	} else if (*n < cap) {
		out[(*n)++] = e;
	} else {
		*n = cap + 1;	/* overflow marker */
	}
}

/* which FROM tables an expression reads: bit t set for table t */
static uint64_t expr_tables(const struct mdb_expr *e)
{
	uint64_t m = 0;
	if (!e)
		return 0;
	if (e->kind == MDB_EX_FIELD && e->tbl_idx >= 0 && e->tbl_idx < 64)
		m |= 1ull << e->tbl_idx;
	for (int i = 0; i < e->nkids; i++)
		m |= expr_tables(e->kids[i]);
	return m;
}

/* WHERE push-down: the reference filters AFTER the joins (proc_where_clause :1435-1463), which for inner joins gives
 * the same rows as filtering a table first whenever a conjunct reads only that table.  Splits the top-level
 * AND-conjuncts of the WHERE clause: push[t][..] = table t's own conjuncts (constants go with table 0),
 * residual[..] = conjuncts that read several tables and stay above the joins.  false = too many to split. */
#define PUSH_MAX 16
#define PUSH_TABS MDB_MAX_TABS
struct where_split {
	const struct mdb_expr *push[PUSH_TABS][PUSH_MAX];
	int npush[PUSH_TABS];
	const struct mdb_expr *residual[64];
	int nresidual;
};

static bool where_split(const struct mdb_select *s, struct where_split *w)
{
	struct mdb_expr *all[64];
	int n = 0;
	memset(w, 0, sizeof(*w));
	if (!s->where)
		return true;
	if (s->ntabs > PUSH_TABS)
		return false;
	collect_conjuncts(s->where, all, &n, 64);
	if (n > 64)
		return false;
	for (int i = 0; i < n; i++) {
		const uint64_t m = expr_tables(all[i]);
		int t = 0;
		if (m & (m - 1)) {
			w->residual[w->nresidual++] = all[i];
			continue;
		}
		while (m && !((m >> t) & 1))
			t++;
		if (w->npush[t] == PUSH_MAX)
			return false;
		w->push[t][w->npush[t]++] = all[i];
	}
	return true;
}

/* rows of base table t that pass its pushed-down conjuncts: *sel = ascending row ids (NULL = every row), *m = how many */
static int table_filter(struct exec *x, int t, const struct mdb_expr *const *conj, int nconj, const uint32_t **sel, uint64_t *m)
{
	struct mdb_table *tb = x->s->tabs[t].t;
	struct pred_prog p;
	uint32_t *v;
	*sel = NULL;
	*m = tb->nrows;
	if (!nconj || !tb->nrows)
		return MIDORIDB_OK;
	memset(&p, 0, sizeof(p));
	{
		uint32_t *saved = x->rid[t];	/* the program reads the BASE table, whatever the stream holds for t */
		x->rid[t] = NULL;
		for (int i = 0; i < nconj; i++)
			if (pred_compile(x, &p, conj[i]) || (i && pred_emit(&p, MDB_P_AND, 0, 0, 0, 0, 0))) {
				x->rid[t] = saved;
				snprintf(x->err, x->errlen, "execution phase: predicate too large for the device program (max %d steps, %d columns)\n",
					 MDB_PRED_MAX_INSNS, MDB_PRED_MAX_SLOTS);
				return -MIDORIDB_ERROR;
			}
		x->rid[t] = saved;
	}
	v = dalloc(x, tb->nrows * 4);
	if (!v)
		return dev_fail(x, "allocating the selection vector");
	if (mdb_dev_filter(x->dev, p.insn, p.n, p.cols, p.ncols, tb->nrows, v, m))
		return dev_fail(x, "filter");
	*sel = v;
	return MIDORIDB_OK;
}

/* column `key` of base table t restricted to the rows in sel (NULL = all): device pointers for an operator */
static int table_column(struct exec *x, int t, const struct mdb_expr *key, const uint32_t *sel, uint64_t m, const void **vals,
			const uint64_t **nulls)
{
	struct mdb_column *col = &x->s->tabs[t].t->cols[key->col_idx];
	*vals = col->d_data;
	*nulls = col->d_nullbits;
	if (sel) {
		int64_t *v = dalloc(x, (m ? m : 1) * 8);
		uint64_t *nb = col->d_nullbits ? dalloc(x, ((m + 63) / 64 + 1) * 8) : NULL;
		if (!v || (col->d_nullbits && !nb))
			return dev_fail(x, "allocating a key column");
		if (m && mdb_dev_gather64(x->dev, col->d_data, col->d_nullbits, sel, m, v, nb))
			return dev_fail(x, "gathering a key column");
		*vals = v;
		*nulls = nb;
	}
	return MIDORIDB_OK;
}

/* key column of table t for the fused plan: the base column, or the keys of the rows that pass the pushed conjuncts */
static int fused_operand(struct exec *x, int t, const struct mdb_expr *key, const struct mdb_expr *const *conj, int nconj,
			 const void **vals, const uint64_t **nulls, uint64_t *n)
{
	const uint32_t *sel;
	int rc = table_filter(x, t, conj, nconj, &sel, n);
	if (rc)
		return rc;
	return table_column(x, t, key, sel, *n, vals, nulls);
}

/* The join of table t when EVERY row of the stream finds exactly one partner in it (a foreign key to a primary key) and the statement
 * reads at most two other columns of t, neither with NULLs: mdb_dev_join_payload carries those cells through t's one partition level
 * and delivers them in stream order - no partner row ids, no compaction, no random gather in the projection (BASELINE configs[1]).
 * Table t is then read through a SHADOW (like a table that arrived over the wire in sharded mode): its key column is the stream's own
 * key column, its payload columns are the carried cells, its row-id vector the identity.  0 = done, 1 = not such a join (nothing
 * changed: the pairs path answers), < 0 = error. */
static int join_with_payload(struct exec *x, int t, const struct mdb_expr *kr, const int64_t *vl, const uint64_t *nl, const void *vr,
			     const uint64_t *nr, uint64_t r_rows)
{
	struct mdb_select *s = x->s;
	const struct mdb_table *rt = s->tabs[t].t;
	int pc[2], np = 0;
	for (int c = 0; c < rt->ncols; c++) {
		if (!x->need[t][c] || c == kr->col_idx)
			continue;
		if (np == 2 || rt->cols[c].d_nullbits || !rt->cols[c].d_data)
			return 1;
		pc[np++] = c;
	}
	if (!np || !x->n || x->orig_tab[t])
		return 1;
	const void *pin[2] = { NULL, NULL };
	void *out[2] = { NULL, NULL };
	for (int i = 0; i < np; i++) {
		pin[i] = rt->cols[pc[i]].d_data;
		out[i] = dalloc(x, x->n * 8);
		if (!out[i])
			return dev_fail(x, "allocating carried columns");
	}
	const int rc = mdb_dev_join_payload(x->dev, vl, nl, x->n, (const int64_t *)vr, nr, r_rows, pin, np, out);
	if (rc == 1)
		return 1;
	if (rc)
		return dev_fail(x, "join with payload");
	struct mdb_table *sh = calloc(1, sizeof(*sh));
	if (!sh)
		return -MIDORIDB_NOMEM;
	memcpy(sh->name, rt->name, sizeof(sh->name));
	sh->ncols = rt->ncols;
	for (int c = 0; c < rt->ncols; c++) {
		memcpy(sh->cols[c].name, rt->cols[c].name, sizeof(sh->cols[c].name));
		sh->cols[c].type = rt->cols[c].type;
		sh->cols[c].precision = rt->cols[c].precision;
		sh->cols[c].not_null = rt->cols[c].not_null;
	}
	sh->cols[kr->col_idx].d_data = (void *)vl;	/* (in every joined tuple the key of t IS the stream's key; a NULL key joined nothing) */
	for (int i = 0; i < np; i++)
		sh->cols[pc[i]].d_data = out[i];
	sh->nrows = sh->dev_rows = x->n;
	sh->dev_cap = x->n;
	sh->device_only = true;
	x->orig_tab[t] = s->tabs[t].t;
	x->shadow[t] = sh;
	s->tabs[t].t = sh;
	x->rid[t] = NULL;
	x->joined_rows = x->n;
	return 0;
}

static int join_next_table(struct exec *x, int t, const struct mdb_expr *const *pconj, int npconj)
{
	struct mdb_select *s = x->s;
	struct mdb_table *rt = s->tabs[t].t;
	struct mdb_expr *conj[32];
	int nconj = 0, key = -1;
	const struct mdb_expr *kl = NULL, *kr = NULL;
	uint32_t *pl = NULL, *pr = NULL;
	uint64_t J = 0;
	int rc;

	if (s->on[t]) {
		collect_conjuncts(s->on[t], conj, &nconj, 32);
		if (nconj > 32)
			nconj = 0;	/* too many conjuncts: treat the whole ON as a residual predicate */
		for (int i = 0; i < nconj && key < 0; i++) {
			struct mdb_expr *c = conj[i];
			if (c->kind == MDB_EX_CMP && c->op == MDB_CMP_EQ && c->kids[0]->kind == MDB_EX_FIELD && c->kids[1]->kind == MDB_EX_FIELD) {
				struct mdb_expr *a = c->kids[0], *b = c->kids[1];
				if (a->tbl_idx < t && b->tbl_idx == t) {
					kl = a;
					kr = b;
					key = i;
				} else if (b->tbl_idx < t && a->tbl_idx == t) {
					kl = b;
					kr = a;
					key = i;
				}
			}
		}
	}
	/* the new table's own WHERE conjuncts filter it before it is joined */
	const uint32_t *rsel = NULL;
	uint64_t r_rows = rt->nrows;
	if ((rc = table_filter(x, t, pconj, npconj, &rsel, &r_rows)))
		return rc;
	if (x->cat->dist) {
		/* sharded mode: both sides go where their join key hashes to - the left stream unless it already is there (joined on
		 * this key before), the new table always (its rows that passed its own WHERE conjuncts at home) */
		if (key < 0) {
			snprintf(x->err, x->errlen, "execution phase: sharded mode (MIDORIDB_WORLD_SIZE): a join needs an equi-join key in its ON clause "
						    "(l.col = r.col) to be exchanged between the ranks; this one would only see local rows\n");
			return -MIDORIDB_ERROR;
		}
		if (!in_part(x, kl) && (rc = shard_stream(x, t, kl, 0)))
			return rc;
		{
			const void *kv;
			const uint64_t *kn;
			const int tabs1[1] = { t };
			uint32_t *rids1[1] = { (uint32_t *)rsel };
			if (kr->type == MDB_CT_DOUBLE)
				rc = double_join_keys(x, &rt->cols[kr->col_idx], rsel, r_rows, &kv, &kn);
			else
				rc = table_column(x, t, kr, rsel, r_rows, &kv, &kn);
			if (rc || (rc = shard_rows(x, tabs1, 1, rids1, r_rows, kv, kn, 0, &r_rows)))
				return rc;
			rsel = NULL;
			rt = s->tabs[t].t;
		}
		if (x->npart < 2 * MDB_MAX_TABS)
			x->part[x->npart++] = kr;
	}
	if (key >= 0) {
		const int64_t *vl;
		const uint64_t *nl;
		const void *vr;
		const uint64_t *nr;
		if (kl->type == MDB_CT_DOUBLE) {
			const void *dl;
			if ((rc = double_join_keys(x, &s->tabs[kl->tbl_idx].t->cols[kl->col_idx], x->rid[kl->tbl_idx], x->n, &dl, &nl)) ||
			    (rc = double_join_keys(x, &rt->cols[kr->col_idx], rsel, r_rows, &vr, &nr)))
				return rc;
			vl = dl;
		} else if ((rc = stream_column(x, kl, &vl, &nl)) || (rc = table_column(x, t, kr, rsel, r_rows, &vr, &nr))) {
			return rc;
		}
		if (kl->type != MDB_CT_DOUBLE && kr->type != MDB_CT_DOUBLE) {
			x->same_col[t] = kr->col_idx;
			x->same_as_tbl[t] = kl->tbl_idx;
			x->same_as_col[t] = kl->col_idx;
		}
		if (x->n && r_rows && !x->cat->dist && !rsel && kl->type != MDB_CT_DOUBLE && kr->type != MDB_CT_DOUBLE) {
			const int prc = join_with_payload(x, t, kr, vl, nl, vr, nr, r_rows);
			if (prc < 0)
				return prc;
			if (prc == 0) {
				for (int i = 0; i < nconj; i++)		/* residual ON conjuncts, on the merged tuples */
					if (i != key && (rc = stream_filter(x, t + 1, conj[i])))
						return rc;
				if (nconj > 1)
					x->joined_rows = x->n;
				return MIDORIDB_OK;
			}
		}
		if (x->n && r_rows) {
			if (mdb_dev_join_pairs(x->dev, vl, nl, x->n, vr, nr, r_rows, &pl, &pr, &J))
				return dev_fail(x, "hash join");
			if (pl && track(x, pl))
				return -MIDORIDB_NOMEM;
			if (pr && track(x, pr))
				return -MIDORIDB_NOMEM;
		}
	} else {
		/* no equi-join key: FROM A, B (ON 1=1) or a general ON -> all pairs, then the ON predicate */
		const uint64_t total = x->n * r_rows;
		if (total > (1ull << 28)) {
			snprintf(x->err, x->errlen, "execution phase: cross join of %llu x %llu rows is too large (no equi-join key in the ON clause)\n",
				 (unsigned long long)x->n, (unsigned long long)r_rows);
			return -MIDORIDB_ERROR;
		}
		J = total;
		if (J) {
			pl = dalloc(x, J * 4);
			pr = dalloc(x, J * 4);
			if (!pl || !pr)
				return dev_fail(x, "allocating join pairs");
			if (mdb_dev_cross_pairs(x->dev, x->n, r_rows, pl, pr))
				return dev_fail(x, "cross join");
		}
	}
	if (rsel && J) {	/* pairs index the filtered right rows: back to row ids of the base table */
		uint32_t *mapped = dalloc(x, J * 4);
		if (!mapped || mdb_dev_gather32(x->dev, rsel, pr, J, mapped))
			return dev_fail(x, "re-mapping row ids");
		pr = mapped;
	}
	/* compose the stream: earlier tables through pl, the new table = pr - unless pl is 0, 1, 2 ... (every row of the stream joined
	 * exactly one right row: a primary-key join): the earlier tables' row ids then stand as they are, nothing is gathered */
	if (key >= 0 && J == x->n && pl && mdb_dev_last_pairs_identity(x->dev))
		x->n = J;
	else if ((rc = stream_select(x, t, pl, J)))
		return rc;
	x->rid[t] = pr;
	x->joined_rows = J;
	/* residual ON conjuncts (everything except the hash key), evaluated on the merged tuples */
	if (s->on[t]) {
		if (key < 0) {
			if ((rc = stream_filter(x, t + 1, s->on[t])))
				return rc;
		} else {
			for (int i = 0; i < nconj; i++)
				if (i != key && (rc = stream_filter(x, t + 1, conj[i])))
					return rc;
		}
		x->joined_rows = x->n;
	}
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ result assembly */

void mdb_result_free(struct mdb_result *r)
{
	if (!r)
		return;
	for (int c = 0; c < r->ncols; c++) {
		if (r->data)
			mdb_dev_host_free(r->data[c]);
		if (r->nullbits)
			free(r->nullbits[c]);
		if (r->d_data && r->d_data[c])
			mdb_dev_free(r->dev, r->d_data[c]);
		if (r->d_nullbits && r->d_nullbits[c])
			mdb_dev_free(r->dev, r->d_nullbits[c]);
	}
	free(r->data);
	free(r->nullbits);
	free(r->d_data);
	free(r->d_nullbits);
	free(r->colname);
	free(r->coltype);
	free(r);
}

/* one result column device -> host; a NULL cell reads as 0 through query_column_int64(), like the reference
 * (cpy_cols skips the copy into the zeroed row, executor_select.c:384-387) */
static int result_column_to_host(mdb_dev_ctx *dev, struct mdb_result *res, int c, const void *d_vals, const uint64_t *d_nulls)
{
	const uint64_t rows = res->nrows;
	if (!d_vals || !rows)
		return MIDORIDB_OK;
	if (mdb_dev_d2h(dev, res->data[c], d_vals, rows * 8))
		return -MIDORIDB_INTERNAL;
	if (d_nulls) {
		const uint64_t words = (rows + 63) / 64;
		if (!res->nullbits[c])		/* (a fetch that is tried again after a failure finds the bitmap of its first attempt) */
			res->nullbits[c] = calloc((size_t)words, 8);
		if (!res->nullbits[c] || mdb_dev_d2h(dev, res->nullbits[c], d_nulls, words * 8))
			return -MIDORIDB_INTERNAL;
		for (uint64_t i = 0; i < rows; i++)
			if ((res->nullbits[c][i >> 6] >> (i & 63)) & 1)
				res->data[c][i] = 0;
	}
	return MIDORIDB_OK;
}

int mdb_result_fetch(struct mdb_result *r)
{
	if (!r || r->fetched)
		return MIDORIDB_OK;
	for (int c = 0; c < r->ncols; c++) {
		if (!r->d_data || !r->d_data[c])
			continue;
		if (!r->data[c]) {
			r->data[c] = mdb_dev_host_alloc((size_t)(r->nrows ? r->nrows : 1) * 8);
			if (!r->data[c])
				return -MIDORIDB_NOMEM;
		}
		if (result_column_to_host(r->dev, r, c, r->d_data[c], r->d_nullbits ? r->d_nullbits[c] : NULL))
			return -MIDORIDB_INTERNAL;
	}
	r->fetched = true;
	return MIDORIDB_OK;
}

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

/* is the plan the fused north-star shape?  returns the group field's side (0 = left key, 1 = right key) or -1 */
/* The fused plan applies to  T0 JOIN T1 ON k0 = k1 [JOIN T2 ON (k0 | k1) = k2 ...] GROUP BY one of those keys, COUNT(*):
 * every join is an equi-join on the SAME key (each ON clause ties the new table's column to a key column already
 * in the chain); a WHERE clause must be pushable below the joins (where_pushable).  keys[t] = the key field of
 * table t.  count_only: the query is SELECT COUNT(*) over the join without GROUP BY - the same operator, of which only
 * the joined-row total is used.  Returns 0 when it applies, -1 otherwise. */
static int fused_chain(struct mdb_select *s, const struct mdb_expr **keys, bool count_only)
{
	if (s->ntabs < 2 || (count_only ? s->ngroup != 0 : s->ngroup != 1))
		return -1;
	for (int t = 0; t < s->ntabs; t++)
		keys[t] = NULL;
	for (int t = 1; t < s->ntabs; t++) {
		const struct mdb_expr *on = s->on[t], *mine, *other;
		if (!on || on->kind != MDB_EX_CMP || on->op != MDB_CMP_EQ || on->kids[0]->kind != MDB_EX_FIELD || on->kids[1]->kind != MDB_EX_FIELD)
			return -1;
		if (on->kids[0]->tbl_idx == t && on->kids[1]->tbl_idx < t) {
			mine = on->kids[0];
			other = on->kids[1];
		} else if (on->kids[1]->tbl_idx == t && on->kids[0]->tbl_idx < t) {
			mine = on->kids[1];
			other = on->kids[0];
		} else {
			return -1;
		}
		if (t == 1)
			keys[0] = other;
		else if (!keys[other->tbl_idx] || !field_eq(other, keys[other->tbl_idx]))
			return -1;
		if (mine->type != other->type)
			return -1;
		if (mine->type == MDB_CT_DOUBLE)
			return -1;	/* DOUBLE keys join through their IEEE-canonical words (join_next_table), whose values are not
					 * the column's: the general plan gathers the group keys from the table itself */
		keys[t] = mine;
	}
	if (count_only)
		return 0;	/* SELECT COUNT(*) FROM the join: only the number of joined rows is wanted */
	for (int t = 0; t < s->ntabs; t++)
		if (field_eq(s->group[0], keys[t]))
			return 0;
	return -1;
}

/* a two-table INNER JOIN ON l = r whose result columns are all one of the two key columns, nothing else asked of it, order left
 * open by the host (mdb_database_groups_any_order): kj[0..1] = the key fields (left table's first) */
static bool keys_only_join(const struct mdb_select *s, const struct mdb_catalog *cat, int has_count, const int *key_tbl, const int *key_col,
			   const int *src, int ncols, const struct mdb_expr **kj)
{
	if (cat->dist || !cat->groups_any_order || s->ntabs != 2 || s->where || s->ngroup || has_count || s->distinct || s->norder || s->having ||
	    s->has_limit || !ncols)
		return false;
	const struct mdb_expr *on = s->on[1];
	if (!on || on->kind != MDB_EX_CMP || on->op != MDB_CMP_EQ || on->kids[0]->kind != MDB_EX_FIELD || on->kids[1]->kind != MDB_EX_FIELD)
		return false;
	const struct mdb_expr *a = on->kids[0], *b = on->kids[1];
	if (a->tbl_idx == b->tbl_idx || a->type != b->type || a->type == MDB_CT_DOUBLE)
		return false;	/* (DOUBLE keys join through their IEEE-canonical words, whose values are not the column's) */
	kj[0] = a->tbl_idx == 0 ? a : b;
	kj[1] = a->tbl_idx == 0 ? b : a;
	if (!s->tabs[0].t->nrows || !s->tabs[1].t->nrows)
		return false;	/* (an empty side: the general plan answers "no rows") */
	for (int c = 0; c < ncols; c++) {
		const int t = key_tbl[src[c]], col = key_col[src[c]];
		if (t < 0 || !((t == kj[0]->tbl_idx && col == kj[0]->col_idx) || (t == kj[1]->tbl_idx && col == kj[1]->col_idx)))
			return false;
	}
	return true;
}

/* HAVING, DISTINCT, ORDER BY, LIMIT over the finished stream (after FROM / WHERE / GROUP BY) */
static int select_tail(struct exec *x, int has_count)
{
	struct mdb_select *s = x->s;
	const bool count_only = has_count && !s->ngroup;
	int rc;

	if (count_only)
		return MIDORIDB_OK;	/* one row: HAVING is rejected at plan time, ORDER BY has nothing to order, LIMIT is applied by the caller */
	if (s->having && (rc = stream_filter(x, s->ntabs, s->having)))
		return rc;
	if (s->distinct && x->n > 1) {
		struct mdb_sort_key keys[MDB_SORT_MAX_KEYS];
		int nk = 0;
		uint32_t *sel;
		uint64_t m = 0;
		for (int t = 0; t < s->ntabs; t++)
			for (int c = 0; c < s->tabs[t].t->ncols; c++) {
				struct mdb_expr f;
				bool want = s->select_all;
				for (int i = 0; i < s->nsel && !want; i++)
					want = s->sel[i]->kind == MDB_EX_FIELD && s->sel[i]->tbl_idx == t && s->sel[i]->col_idx == c;
				if (!want)
					continue;
				if (nk == MDB_SORT_MAX_KEYS) {
					snprintf(x->err, x->errlen, "DISTINCT over more than %d columns is not supported\n", MDB_SORT_MAX_KEYS);
					return -MIDORIDB_ERROR;
				}
				memset(&f, 0, sizeof(f));
				f.kind = MDB_EX_FIELD;
				f.tbl_idx = t;
				f.col_idx = c;
				bind_operand(x, &f, &keys[nk].values, &keys[nk].nullbits, &keys[nk].rid);
				keys[nk].type = s->tabs[t].t->cols[c].type == MDB_CT_DOUBLE ? MDB_T_DOUBLE : MDB_T_INT64;
				keys[nk].desc = 0;
				nk++;
			}
		sel = dalloc(x, x->n * 4);
		if (!sel)
			return dev_fail(x, "allocating the DISTINCT selection");
		if (nk == 1) {
			/* one column: the hash GROUP BY operator already returns first occurrences in order (2-3x faster
			 * than sorting at 10^8 rows); its COUNT(*) output is not needed */
			const void *kv = keys[0].values;
			const uint64_t *kn = keys[0].nullbits;
			int64_t *cnt = dalloc(x, x->n * 8);
			if (!cnt)
				return dev_fail(x, "allocating the DISTINCT selection");
			if (keys[0].rid) {
				int64_t *v = dalloc(x, x->n * 8);
				uint64_t *nb = kn ? dalloc(x, ((x->n + 63) / 64 + 1) * 8) : NULL;
				if (!v || (kn && !nb) || mdb_dev_gather64(x->dev, kv, kn, keys[0].rid, x->n, v, nb))
					return dev_fail(x, "gathering the DISTINCT column");
				kv = v;
				kn = nb;
			}
			if (mdb_dev_group_count(x->dev, kv, kn, x->n, MDB_ORDER_FIRST, sel, cnt, x->n, &m))
				return dev_fail(x, "DISTINCT");
		} else if (mdb_dev_distinct_sel(x->dev, keys, nk, x->n, sel, &m)) {
			return dev_fail(x, "DISTINCT");
		}
		if ((rc = stream_apply_sel(x, s->ntabs, sel, m)))
			return rc;
	}
	if (s->norder && x->n > 1) {
		struct mdb_sort_key keys[MDB_SORT_MAX_KEYS];
		uint32_t *perm;
		for (int i = 0; i < s->norder; i++) {
			bind_operand(x, s->order[i], &keys[i].values, &keys[i].nullbits, &keys[i].rid);
			keys[i].type = s->order[i]->type == MDB_CT_DOUBLE ? MDB_T_DOUBLE : MDB_T_INT64;
			keys[i].desc = s->order_desc[i];
		}
		/* ORDER BY ... LIMIT: only the first offset + count rows of the order are ever looked at - top-k selection instead
		 * of a sort of the whole stream (mdb_dev_topk_perm falls back to the sort by itself when that does not pay) */
		uint64_t want = x->n;
		if (s->has_limit && s->limit_off >= 0 && s->limit_cnt >= 0 && (uint64_t)s->limit_off + (uint64_t)s->limit_cnt < x->n)
			want = (uint64_t)s->limit_off + (uint64_t)s->limit_cnt;
		perm = dalloc(x, (want ? want : 1) * 4);
		if (!perm)
			return dev_fail(x, "allocating the ORDER BY permutation");
		if (want < x->n) {
			if (want && mdb_dev_topk_perm(x->dev, keys, s->norder, x->n, want, perm, NULL))
				return dev_fail(x, "ORDER BY ... LIMIT");
		} else if (mdb_dev_sort_perm(x->dev, keys, s->norder, x->n, perm))
			return dev_fail(x, "ORDER BY");
		if ((rc = stream_apply_sel(x, s->ntabs, perm, want)))
			return rc;
	}
	if (s->has_limit) {
		const uint64_t off = (uint64_t)s->limit_off < x->n ? (uint64_t)s->limit_off : x->n;
		const uint64_t cnt = (uint64_t)s->limit_cnt < x->n - off ? (uint64_t)s->limit_cnt : x->n - off;
		if (off || cnt < x->n) {
			uint32_t *idx = dalloc(x, (off + cnt ? off + cnt : 1) * 4);
			if (!idx || mdb_dev_iota32(x->dev, idx, off + cnt))
				return dev_fail(x, "LIMIT");
			if ((rc = stream_apply_sel(x, s->ntabs, idx + off, cnt)))
				return rc;
		}
	}
	return MIDORIDB_OK;
}

int mdb_exec_select(struct mdb_catalog *cat, struct mdb_select *s, struct mdb_result **out, char *err, size_t errlen)
{
	stmt_dict = &cat->dict;
	struct exec x;
	struct mdb_result *res = NULL;
	char (*keys)[MDB_NAME_LEN] = NULL;
	int *order = NULL, *key_tbl = NULL, *key_col = NULL, *src = NULL;
	int nkeys = 0, ncols = 0, rc, has_count = 0;
	void **direct_vals = NULL;		/* scan + WHERE + projection plan: the result columns on the device, final */
	uint64_t **direct_nulls = NULL;
	double t0;
	const struct mdb_expr *fkeys[MDB_MAX_TABS];
	const struct mdb_expr *kj[2] = { NULL, NULL };
	int fused;

	*out = NULL;
	memset(&x, 0, sizeof(x));
	for (int t = 0; t < MDB_MAX_TABS; t++)
		x.same_col[t] = -1;
	if ((rc = resolve_select(cat, s, err, errlen)))
		return rc;
	if ((rc = mdb_catalog_device(cat, err, errlen)))
		return rc;
	for (int t = 0; t < s->ntabs; t++)
		if ((rc = mdb_table_sync_device(cat, s->tabs[t].t, err, errlen)))
			return rc;
	x.cat = cat;
	x.dev = cat->dev;
	x.s = s;
	x.err = err;
	x.errlen = errlen;
	mark_needed_all(&x);	/* (which columns the statement reads: the sharded exchange and the join that carries payload cells ask) */

	/* ---- result column set in the reference's order (R3): COUNT(*) first if selected, then every column
	 *      of every FROM table left to right; projected afterwards to the select list */
	for (int i = 0; i < s->nsel; i++)
		has_count |= s->sel[i]->kind == MDB_EX_COUNT;
	{
		int total = has_count;
		for (int t = 0; t < s->ntabs; t++)
			total += s->tabs[t].t->ncols;
		keys = calloc((size_t)total, sizeof(*keys));
		order = calloc((size_t)total, sizeof(int));
		key_tbl = calloc((size_t)total, sizeof(int));
		key_col = calloc((size_t)total, sizeof(int));
		if (!keys || !order || !key_tbl || !key_col) {
			rc = -MIDORIDB_NOMEM;
			goto out;
		}
		if (has_count) {
			strcpy(keys[nkeys], "COUNT(*)");
			key_tbl[nkeys] = -1;
			nkeys++;
		}
		for (int t = 0; t < s->ntabs; t++)
			for (int c = 0; c < s->tabs[t].t->ncols; c++) {
				snprintf(keys[nkeys], MDB_NAME_LEN, "%.60s.%.60s", s->tabs[t].t->name, s->tabs[t].t->cols[c].name);
				key_tbl[nkeys] = t;
				key_col[nkeys] = c;
				nkeys++;
			}
		if ((rc = mdb_reference_column_order((const char (*)[MDB_NAME_LEN])keys, nkeys, order)))
			goto out;
	}

	/* the result columns that are wanted, in the reference's order */
	src = calloc((size_t)(nkeys ? nkeys : 1), sizeof(int));
	if (!src) {
		rc = -MIDORIDB_NOMEM;
		goto out;
	}
	for (int k = 0; k < nkeys; k++) {
		int key = order[k];
		bool want = false;
		if (key_tbl[key] < 0) {
			want = true;
		} else if (s->select_all) {
			want = true;
		} else {
			for (int i = 0; i < s->nsel; i++)
				if (s->sel[i]->kind == MDB_EX_FIELD && s->sel[i]->tbl_idx == key_tbl[key] && s->sel[i]->col_idx == key_col[key])
					want = true;
		}
		if (want)
			src[ncols++] = key;
	}

	t0 = now_ms();
	struct where_split ws;
	const bool split_ok = where_split(s, &ws);
	bool only_count = has_count && !s->ngroup && !s->select_all;
	for (int i = 0; i < s->nsel; i++)
		only_count = only_count && s->sel[i]->kind == MDB_EX_COUNT;
	fused = s->ntabs <= PUSH_TABS ? fused_chain(s, fkeys, only_count) : -1;
	if (fused >= 0 && (!split_ok || ws.nresidual))
		fused = -1;	/* a conjunct reads several tables: it has to see the joined rows */
	if (fused >= 0) {
		/* ---- north-star plan: join + GROUP BY key + COUNT(*) without materialising the join.  More than two tables
		 *      on the same key chain the operator: the group keys of (T0, T1) are joined with T2, and so on; a group's
		 *      COUNT(*) is the product of the per-table multiplicities (mdb_dev_combine_counts).  WHERE conjuncts that
		 *      read one table filter that table before it enters the join. */
		const void *lv, *rv;
		const uint64_t *ln, *rn;
		uint64_t nl_rows, nr_rows;
		if ((rc = fused_operand(&x, 0, fkeys[0], ws.push[0], ws.npush[0], &lv, &ln, &nl_rows)) ||
		    (rc = fused_operand(&x, 1, fkeys[1], ws.push[1], ws.npush[1], &rv, &rn, &nr_rows)))
			goto out;
		uint64_t cap = nl_rows ? nl_rows : 1, G = 0, J = 0;
		bool multi_done = false;
		if (cat->dist && fkeys[0]->type == MDB_CT_VARCHAR) {
			ERR("sharded mode: VARCHAR join keys are ids of this process's string dictionary and mean nothing to the other ranks\n");
			rc = -MIDORIDB_ERROR;
			goto out;
		}
		if (cat->dist && fkeys[0]->type != MDB_CT_DOUBLE && (rc = shard_promise_ranges(&x, fkeys[0], fkeys[1])))
			goto out;
		if (cat->dist && s->ntabs > 2 && s->ntabs <= 4) {
			/* sharded mode, three or four tables on one key: ONE exchange - every table partitioned once with the same hash, the
			 * right tables' counts multiplied where the regions meet (mdb_dist_join_group_count_multi_alloc); when that form is
			 * not served (every rank learns so together) the chain of two-table calls below runs */
			const int64_t *rk[3];
			const uint64_t *rnb[3];
			uint64_t rrows[3];
			rk[0] = rv;
			rnb[0] = rn;
			rrows[0] = nr_rows;
			for (int t = 2; t < s->ntabs; t++) {
				const void *cv;
				if ((rc = fused_operand(&x, t, fkeys[t], ws.push[t], ws.npush[t], &cv, &rnb[t - 1], &rrows[t - 1])))
					goto out;
				rk[t - 1] = cv;
			}
			const int mrc = mdb_dist_join_group_count_multi_alloc(cat->dist, lv, ln, nl_rows, s->ntabs - 1, rk, rnb, rrows, &x.d_fused_key, &x.d_count,
									      &G, &J);
			if (mrc < 0) {
				snprintf(err, errlen, "execution phase: sharded join + group count: %s\n", mdb_dist_last_error(cat->dist));
				rc = -MIDORIDB_INTERNAL;
				goto out;
			}
			if (mrc == 0) {
				if (track(&x, x.d_fused_key) || track(&x, x.d_count)) {
					rc = -MIDORIDB_NOMEM;
					goto out;
				}
				multi_done = true;
			}
		}
		if (cat->dist && !multi_done) {
			/* sharded mode: the tables hold this rank's rows; both key columns are exchanged (RCCL all-to-all per table,
			 * include/mdb_dist.h) and this rank keeps the groups whose key hashes to it.  Collective: every rank runs the
			 * same statement. */
			/* (more tables follow on the same key: the groups must lie where their key hashes to, for the tables sent after them) */
			if (mdb_dist_join_group_count_alloc(cat->dist, lv, ln, nl_rows, rv, rn, nr_rows, s->ntabs > 2 ? MDB_DIST_PLACE_BY_KEY_HASH : 0u,
							    &x.d_fused_key, &x.d_count, NULL, &G, &J)) {
				snprintf(err, errlen, "execution phase: sharded join + group count: %s\n", mdb_dist_last_error(cat->dist));
				rc = -MIDORIDB_INTERNAL;
				goto out;
			}
			if (track(&x, x.d_fused_key) || track(&x, x.d_count)) {
				rc = -MIDORIDB_NOMEM;
				goto out;
			}
		} else if (!cat->dist) {
			x.d_fused_key = dalloc(&x, cap * 8);
			x.d_count = dalloc(&x, cap * 8);
			if (!x.d_fused_key || !x.d_count) {
				rc = dev_fail(&x, "allocating group outputs");
				goto out;
			}
			if (s->ntabs > 2 && s->ntabs <= 4) {
				/* A JOIN B ON a = b JOIN C ON a = c [JOIN D ...]: every table partitioned once, the right tables' counts multiplied
				 * in the leaf kernel, the groups ordered once (mdb_dev_join_group_count_multi; it chains by itself when the keys
				 * do not take the compact form) */
				const int64_t *rk[3];
				const uint64_t *rnb[3];
				uint64_t rrows[3];
				rk[0] = rv;
				rnb[0] = rn;
				rrows[0] = nr_rows;
				for (int t = 2; t < s->ntabs; t++) {
					const void *cv;
					if ((rc = fused_operand(&x, t, fkeys[t], ws.push[t], ws.npush[t], &cv, &rnb[t - 1], &rrows[t - 1])))
						goto out;
					rk[t - 1] = cv;
				}
				if (mdb_dev_join_group_count_multi(x.dev, lv, ln, nl_rows, s->ntabs - 1, rk, rnb, rrows,
								   (only_count || cat->groups_any_order) ? 0u : MDB_ORDER_FIRST, x.d_fused_key, x.d_count, NULL,
								   cap, &G, &J)) {
					rc = dev_fail(&x, "join + group count over several tables");
					goto out;
				}
				multi_done = true;
			} else if (mdb_dev_join_group_count(x.dev, lv, ln, nl_rows, rv, rn, nr_rows,
							    /* (a bare COUNT(*) has no group order to keep; nor has a GROUP BY when the database says so) */
							    (only_count || cat->groups_any_order) ? 0u : MDB_ORDER_FIRST, x.d_fused_key,
							    x.d_count, NULL, cap, &G, &J)) {
				rc = dev_fail(&x, "join + group count");
				goto out;
			}
		}
		for (int t = 2; t < s->ntabs && (G || cat->dist) && !multi_done; t++) {
			const void *cv;
			const uint64_t *cn;
			uint64_t nc_rows;
			if ((rc = fused_operand(&x, t, fkeys[t], ws.push[t], ws.npush[t], &cv, &cn, &nc_rows)))
				goto out;
			int64_t *key2 = NULL, *cnt2 = NULL, *cnt3 = dalloc(&x, (G ? G : 1) * 8);
			uint32_t *first2 = NULL;
			uint64_t G2 = 0, J2 = 0;
			if (cat->dist) {
				/* the groups so far already live on the rank their key hashes to: only the new table travels */
				if (mdb_dist_join_group_count_alloc(cat->dist, x.d_fused_key, NULL, G, cv, cn, nc_rows, MDB_DIST_LEFT_IN_PLACE, &key2, &cnt2,
								    &first2, &G2, &J2)) {
					snprintf(err, errlen, "execution phase: sharded join + group count: %s\n", mdb_dist_last_error(cat->dist));
					rc = -MIDORIDB_INTERNAL;
					goto out;
				}
				if (track(&x, key2) || track(&x, cnt2) || track(&x, first2)) {
					rc = -MIDORIDB_NOMEM;
					goto out;
				}
			} else {
				key2 = dalloc(&x, G * 8);
				cnt2 = dalloc(&x, G * 8);
				first2 = dalloc(&x, G * 4);
			}
			if (!key2 || !cnt2 || !cnt3 || !first2) {
				rc = dev_fail(&x, "allocating group outputs");
				goto out;
			}
			if ((!cat->dist && mdb_dev_join_group_count(x.dev, x.d_fused_key, NULL, G, cv, cn, nc_rows, MDB_ORDER_FIRST, key2, cnt2, first2, G,
								    &G2, &J2)) ||
			    mdb_dev_combine_counts(x.dev, x.d_count, NULL, first2, cnt2, G2, cnt3, NULL, &J)) {
				rc = dev_fail(&x, "chained join + group count");
				goto out;
			}
			x.d_fused_key = key2;
			x.d_count = cnt3;
			G = G2;
		}
		if (!G)
			J = 0;
		if (cat->dist && only_count) {
			/* SELECT COUNT(*) over the sharded join: every rank reports the global number of joined rows */
			uint64_t tot = J;
			if (mdb_dist_allreduce_sum_u64(cat->dist, &tot, 1)) {
				snprintf(err, errlen, "execution phase: %s\n", mdb_dist_last_error(cat->dist));
				rc = -MIDORIDB_INTERNAL;
				goto out;
			}
			J = tot;
		}
		x.fused = true;
		x.n = only_count ? J : G;	/* COUNT(*) without GROUP BY = the stream length = the joined rows */
		x.joined_rows = J;
	} else if (keys_only_join(s, cat, has_count, key_tbl, key_col, src, ncols, kj)) {
		/* ---- a two-table equi-join whose select list names nothing but the two key columns (BASELINE configs[3]: SELECT * over
		 *      two key columns), any order allowed (mdb_database_groups_any_order): both sides hold the same value in every
		 *      joined row, so no row has to be identified - the any-order join + GROUP BY pipeline counts every key's partners and
		 *      the key is written COUNT times (mdb_dev_join_keys); the stream is the fused plan's: one key column, no row ids */
		const struct mdb_column *cl = &s->tabs[kj[0]->tbl_idx].t->cols[kj[0]->col_idx], *cr = &s->tabs[kj[1]->tbl_idx].t->cols[kj[1]->col_idx];
		int64_t *jk = NULL;
		uint64_t J = 0;
		if (mdb_dev_join_keys(x.dev, cl->d_data, cl->d_nullbits, s->tabs[kj[0]->tbl_idx].t->nrows, cr->d_data, cr->d_nullbits,
				      s->tabs[kj[1]->tbl_idx].t->nrows, &jk, &J)) {
			rc = dev_fail(&x, "join of two key columns");
			goto out;
		}
		if (jk && track(&x, jk)) {
			rc = -MIDORIDB_NOMEM;
			goto out;
		}
		x.d_fused_key = jk;
		x.fused = true;
		x.n = J;
		x.joined_rows = J;
	} else if (s->ntabs == 1 && s->where && split_ok && !ws.nresidual && ws.npush[0] && !s->ngroup && !has_count && !s->distinct &&
		   !s->norder && !s->having && !s->has_limit && s->tabs[0].t->nrows && ncols && ncols <= MDB_GATHER_MAX_COLS) {
		/* ---- scan + WHERE + projection of one table (BASELINE configs[0] shape): the predicate bitmap is turned
		 *      straight into the compacted result columns (mdb_dev_filter_project) - no selection vector, no gathers */
		struct mdb_table *tb = s->tabs[0].t;
		struct pred_prog p;
		struct mdb_project_col pc[MDB_GATHER_MAX_COLS];
		uint64_t m = 0;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < ws.npush[0]; i++)
			if (pred_compile(&x, &p, ws.push[0][i]) || (i && pred_emit(&p, MDB_P_AND, 0, 0, 0, 0, 0))) {
				ERR("execution phase: predicate too large for the device program (max %d steps, %d columns)\n", MDB_PRED_MAX_INSNS,
				    MDB_PRED_MAX_SLOTS);
				rc = -MIDORIDB_ERROR;
				goto out;
			}
		direct_vals = calloc((size_t)ncols, sizeof(void *));
		direct_nulls = calloc((size_t)ncols, sizeof(uint64_t *));
		if (!direct_vals || !direct_nulls) {
			rc = -MIDORIDB_NOMEM;
			goto out;
		}
		for (int c = 0; c < ncols; c++) {
			struct mdb_column *col = &tb->cols[key_col[src[c]]];
			pc[c].values = col->d_data;
			pc[c].nullbits = col->d_nullbits;
			pc[c].out_values = &direct_vals[c];
			pc[c].out_nullbits = &direct_nulls[c];
		}
		if (mdb_dev_filter_project(x.dev, p.insn, p.n, p.cols, p.ncols, tb->nrows, pc, ncols, &m)) {
			rc = dev_fail(&x, "scan + filter + projection");
			goto out;
		}
		for (int c = 0; c < ncols; c++)
			if ((direct_vals[c] && track(&x, direct_vals[c])) || (direct_nulls[c] && track(&x, direct_nulls[c]))) {
				rc = -MIDORIDB_NOMEM;
				goto out;
			}
		x.n = m;
	} else {
		/* ---- general plan */
		if (cat->dist && s->has_limit && s->limit_off > 0) {
			ERR("execution phase: sharded mode: LIMIT with an offset cannot be answered from one rank's rows\n");
			rc = -MIDORIDB_ERROR;
			goto out;
		}
		x.n = s->tabs[0].t->nrows;	/* scan: identity stream over the first table */
		if (split_ok) {
			/* WHERE conjuncts that read one table filter that table before it is joined; the others after the joins */
			const uint32_t *sel0;
			uint64_t m0;
			if ((rc = table_filter(&x, 0, ws.push[0], ws.npush[0], &sel0, &m0)))
				goto out;
			if (sel0) {
				x.rid[0] = (uint32_t *)sel0;
				x.n = m0;
			}
			for (int t = 1; t < s->ntabs; t++)
				if ((rc = join_next_table(&x, t, ws.push[t], ws.npush[t])))
					goto out;
			for (int i = 0; i < ws.nresidual; i++)
				if ((rc = stream_filter(&x, s->ntabs, ws.residual[i])))
					goto out;
		} else {
			for (int t = 1; t < s->ntabs; t++)
				if ((rc = join_next_table(&x, t, NULL, 0)))
					goto out;
			if (s->where && (rc = stream_filter(&x, s->ntabs, s->where)))
				goto out;
		}
		if (cat->dist && s->ngroup) {
			/* sharded mode: a group's rows must meet on one rank - they do when the stream is partitioned by one of the group
			 * fields (a join key); otherwise it is exchanged by the first one, NULL keys included (one group, :1477-1482) */
			bool placed = false;
			for (int g = 0; g < s->ngroup; g++)
				placed = placed || in_part(&x, s->group[g]);
			if (!placed && cat->groups_any_order && s->ngroup == 1 && !x.fused && !s->select_all && !s->distinct &&
			    s->group[0]->kind == MDB_EX_FIELD && s->group[0]->type != MDB_CT_DOUBLE && s->group[0]->type != MDB_CT_VARCHAR) {
				/* any order allowed and only the group key and COUNT(*) can be named (S4): no rows need to travel - every rank's ONE
				 * partition pass over its key column, the first-level regions exchanged, counted where they land
				 * (mdb_dist_group_count_keys_alloc).  Every rank must take the same way: they agree on "no NULL keys anywhere" first.
				 * The decision is taken from the STREAM's key column (its NULL bitmap as it is after joins and exchanges: a shadow table
				 * that arrived over the wire carries bitmaps but no NULL counts), never from catalog counters. */
				const int64_t *kv;
				const uint64_t *kn;
				if ((rc = stream_column(&x, s->group[0], &kv, &kn)))
					goto out;
				uint64_t ok = kn == NULL ? 1u : 0u;
				if (mdb_dist_allreduce_sum_u64(cat->dist, &ok, 1)) {
					snprintf(err, errlen, "execution phase: %s\n", mdb_dist_last_error(cat->dist));
					rc = -MIDORIDB_INTERNAL;
					goto out;
				}
				if (ok == (uint64_t)mdb_dist_world(cat->dist)) {
					int64_t *gk = NULL, *gc = NULL;
					uint64_t Gk = 0;
					if ((rc = shard_promise_ranges(&x, s->group[0], s->group[0])))
						goto out;
					if (!x.promised)
						goto exchange_rows;	/* (no range to promise - an empty column somewhere: the row exchange answers) */
					const int krc = mdb_dist_group_count_keys_alloc(cat->dist, kv, NULL, x.n, &gk, &gc, &Gk);
					if (krc < 0) {
						snprintf(err, errlen, "execution phase: sharded group count: %s\n", mdb_dist_last_error(cat->dist));
						rc = -MIDORIDB_INTERNAL;
						goto out;
					}
					if (krc == 0) {
						if (track(&x, gk) || track(&x, gc)) {
							rc = -MIDORIDB_NOMEM;
							goto out;
						}
						x.d_fused_key = gk;
						x.d_count = gc;
						x.fused = true;
						x.n = Gk;
						goto grouped;
					}
				}
			}
exchange_rows:
			if (!placed && (rc = shard_stream(&x, s->ntabs, s->group[0], MDB_DIST_KEEP_NULL_KEYS)))
				goto out;
		}
		if (cat->dist && s->distinct) {
			/* ... and so must equal rows under DISTINCT */
			bool placed = false;
			const struct mdb_expr *by = NULL;
			static __thread struct mdb_expr first_col;
			for (int i = 0; i < s->nsel; i++)
				if (s->sel[i]->kind == MDB_EX_FIELD) {
					placed = placed || in_part(&x, s->sel[i]);
					by = by ? by : s->sel[i];
				}
			if (s->select_all) {
				placed = placed || x.npart > 0;
				if (!by && s->tabs[0].t->ncols) {
					memset(&first_col, 0, sizeof(first_col));
					first_col.kind = MDB_EX_FIELD;
					first_col.type = s->tabs[0].t->cols[0].type;
					by = &first_col;
				}
			}
			if (!placed && (s->ngroup || !by)) {
				ERR("execution phase: sharded mode: DISTINCT over an aggregate alone cannot be answered from one rank's groups\n");
				rc = -MIDORIDB_ERROR;
				goto out;
			}
			if (!placed && (rc = shard_stream(&x, s->ntabs, by, MDB_DIST_KEEP_NULL_KEYS)))
				goto out;
		}
		if (s->ngroup == 1) {
			const int64_t *kv;
			const uint64_t *kn;
			uint32_t *first;
			uint64_t G = 0;
			if ((rc = stream_column(&x, s->group[0], &kv, &kn)))
				goto out;
			/* any order allowed (mdb_database_groups_any_order), an INTEGER-like key without NULLs: (key, COUNT) pairs without row
			 * ids or an ordering sort - the stream becomes what the fused plan's is (S4: only the group key and COUNT(*) can be
			 * named from here on) */
			if (cat->groups_any_order && !cat->dist && !x.fused && x.n && s->group[0]->kind == MDB_EX_FIELD && s->group[0]->type != MDB_CT_DOUBLE &&
			    !s->select_all && !s->distinct &&
			    s->tabs[s->group[0]->tbl_idx].t->cols[s->group[0]->col_idx].null_count == 0) {
				int64_t *gk = dalloc(&x, x.n * 8), *gc = dalloc(&x, x.n * 8);
				uint64_t Gk = 0;
				if (!gk || !gc) {
					rc = dev_fail(&x, "allocating group outputs");
					goto out;
				}
				const int krc = mdb_dev_group_count_keys(x.dev, kv, NULL, x.n, gk, gc, x.n, &Gk);
				if (krc < 0) {
					rc = dev_fail(&x, "group count (any order)");
					goto out;
				}
				if (krc == 0) {
					x.d_fused_key = gk;
					x.d_count = gc;
					x.fused = true;
					x.n = Gk;
					goto grouped;
				}
			}
			first = dalloc(&x, (x.n ? x.n : 1) * 4);
			x.d_count = dalloc(&x, (x.n ? x.n : 1) * 8);
			if (!first || !x.d_count) {
				rc = dev_fail(&x, "allocating group outputs");
				goto out;
			}
			if (x.n && mdb_dev_group_count(x.dev, kv, kn, x.n, MDB_ORDER_FIRST, first, x.d_count, x.n, &G)) {
				rc = dev_fail(&x, "group count");
				goto out;
			}
			if ((rc = stream_select(&x, s->ntabs, first, G)))
				goto out;
		} else if (s->ngroup > 1) {
			/* several fields: groups = distinct combinations (the reference applies its single-field loop once
			 * per field, executor_select.c:1537-1541, which is not a grouping by the combination: DESIGN.md 2) */
			struct mdb_sort_key gk[MDB_SORT_MAX_KEYS];
			uint32_t *first;
			uint64_t G = 0;
			for (int g = 0; g < s->ngroup; g++) {
				bind_operand(&x, s->group[g], &gk[g].values, &gk[g].nullbits, &gk[g].rid);
				gk[g].type = s->group[g]->type == MDB_CT_DOUBLE ? MDB_T_DOUBLE : MDB_T_INT64;
				gk[g].desc = 0;
			}
			first = dalloc(&x, (x.n ? x.n : 1) * 4);
			x.d_count = dalloc(&x, (x.n ? x.n : 1) * 8);
			if (!first || !x.d_count) {
				rc = dev_fail(&x, "allocating group outputs");
				goto out;
			}
			if (x.n && mdb_dev_group_count_multi(x.dev, gk, s->ngroup, x.n, first, x.d_count, x.n, &G)) {
				rc = dev_fail(&x, "group count");
				goto out;
			}
			{
				int64_t *cnt = x.d_count;	/* stream_select must not re-map the fresh counts */
				x.d_count = NULL;
				rc = stream_select(&x, s->ntabs, first, G);
				x.d_count = cnt;
				if (rc)
					goto out;
			}
		}
	}

grouped:
	if (cat->dist && fused < 0 && has_count && !s->ngroup) {
		/* SELECT COUNT(*) [WHERE ...] in sharded mode: every rank reports the global count */
		uint64_t tot = x.n;
		if (mdb_dist_allreduce_sum_u64(cat->dist, &tot, 1)) {
			snprintf(err, errlen, "execution phase: %s\n", mdb_dist_last_error(cat->dist));
			rc = -MIDORIDB_INTERNAL;
			goto out;
		}
		x.n = tot;
	}
	if (cat->dist && fused >= 0 && s->distinct) {
		bool key_selected = false;
		for (int i = 0; i < s->nsel; i++)
			key_selected = key_selected || s->sel[i]->kind == MDB_EX_FIELD;
		if (!key_selected) {
			ERR("execution phase: sharded mode: DISTINCT over an aggregate alone cannot be answered from one rank's groups\n");
			rc = -MIDORIDB_ERROR;
			goto out;
		}
	}
	/* ---- HAVING -> DISTINCT -> ORDER BY -> LIMIT (SQL order of evaluation; extension) */
	if ((rc = select_tail(&x, has_count)))
		goto out;

	/* ---- projection + COUNT-only handling */
	res = calloc(1, sizeof(*res));
	if (!res) {
		rc = -MIDORIDB_NOMEM;
		goto out;
	}
	{
		bool count_only = has_count && !s->ngroup;	/* SELECT COUNT(*) FROM ... [WHERE ...] */
		uint64_t out_rows = count_only ? (x.n ? 1 : 0) : x.n;	/* the reference returns no row for an empty input */
		if (count_only && s->has_limit && (s->limit_off > 0 || s->limit_cnt == 0))
			out_rows = 0;
		const void **d_vals = calloc((size_t)(ncols ? ncols : 1), sizeof(void *));
		const uint64_t **d_nulls = calloc((size_t)(ncols ? ncols : 1), sizeof(uint64_t *));
		res->ncols = ncols;
		res->nrows = out_rows;
		res->colname = calloc((size_t)(ncols ? ncols : 1), sizeof(*res->colname));
		res->coltype = calloc((size_t)(ncols ? ncols : 1), sizeof(int));
		res->data = calloc((size_t)(ncols ? ncols : 1), sizeof(int64_t *));
		res->nullbits = calloc((size_t)(ncols ? ncols : 1), sizeof(uint64_t *));
		if (!d_vals || !d_nulls || !res->colname || !res->coltype || !res->data || !res->nullbits) {
			free(d_vals);
			free(d_nulls);
			rc = -MIDORIDB_NOMEM;
			goto out;
		}
		const bool keep = cat->results_on_device && out_rows > 1;
		/* pass 1: where every result column lives on the device.  Columns read through a row-id vector are gathered -
		 * all of them in one launch per MDB_GATHER_MAX_COLS columns (mdb_dev_gather_cols), not one launch each */
		struct mdb_gather_col gl[MDB_GATHER_MAX_COLS];
		int ngl = 0, nrid = 0;
		const uint32_t *seen_rid[MDB_GATHER_MAX_RIDS];
		for (int c = 0; c < ncols && rc == MIDORIDB_OK; c++) {
			int key = src[c];
			memcpy(res->colname[c], keys[key], MDB_NAME_LEN);
			if (!keep) {	/* (a result kept on the device gets its host columns on first use, mdb_result_fetch) */
				res->data[c] = mdb_dev_host_alloc((size_t)(out_rows ? out_rows : 1) * 8);	/* pinned when large */
				if (!res->data[c]) {
					rc = -MIDORIDB_NOMEM;
					break;
				}
				if (out_rows <= 1)
					res->data[c][0] = 0;
			}
			if (key_tbl[key] < 0) {
				res->coltype[c] = MDB_CT_INTEGER;
				if (count_only) {
					if (out_rows)
						res->data[c][0] = (int64_t)x.n;
				} else if (out_rows) {
					if (!x.d_count) {	/* COUNT without aggregation cannot reach here (S4) */
						ERR("execution phase: internal error\n");
						rc = -MIDORIDB_INTERNAL;
						break;
					}
					d_vals[c] = x.d_count;
				}
				continue;
			}
			struct mdb_column *col = &s->tabs[key_tbl[key]].t->cols[key_col[key]];
			res->coltype[c] = col->type;
			if (!out_rows || count_only)
				continue;
			if (fused >= 0 || x.fused) {
				d_vals[c] = x.d_fused_key;	/* only the group key can be selected (S4); both sides hold the same value */
				continue;
			}
			if (direct_vals) {			/* scan + WHERE + projection plan: the columns are final already */
				d_vals[c] = direct_vals[c];
				d_nulls[c] = direct_nulls[c];
				continue;
			}
			int from_tbl = key_tbl[key];
			const struct mdb_column *from = col;
			while (x.same_col[from_tbl] == (int)(from - s->tabs[from_tbl].t->cols)) {	/* the join key of a joined table: read the earlier table's column */
				const int t2 = x.same_as_tbl[from_tbl];
				from = &s->tabs[t2].t->cols[x.same_as_col[from_tbl]];
				from_tbl = t2;
			}
			const bool aliased = from != col;
			const uint64_t *src_nb = aliased ? NULL : from->d_nullbits;
			const uint32_t *rid = x.rid[from_tbl];
			if (!rid) {
				d_vals[c] = from->d_data;
				d_nulls[c] = aliased ? NULL : from->d_nullbits;
				continue;
			}
			{
				/* the same column through the same row ids twice (SELECT * after an equi-join: both key columns): gathered once */
				int g = 0;
				while (g < ngl && !(gl[g].src == from->d_data && gl[g].rid == rid && gl[g].src_nullbits == src_nb))
					g++;
				if (g < ngl) {
					d_vals[c] = gl[g].dst;
					d_nulls[c] = gl[g].dst_nullbits;
					continue;
				}
			}
			{
				int t = 0;
				while (t < nrid && seen_rid[t] != rid)
					t++;
				if (ngl == MDB_GATHER_MAX_COLS || (t == nrid && nrid == MDB_GATHER_MAX_RIDS)) {
					if (mdb_dev_gather_cols(x.dev, gl, ngl, out_rows)) {
						rc = dev_fail(&x, "projection gather");
						break;
					}
					ngl = nrid = 0;
					t = 0;
				}
				if (t == nrid)
					seen_rid[nrid++] = rid;
			}
			int64_t *v = dalloc(&x, out_rows * 8);
			uint64_t *nb = src_nb ? dalloc(&x, ((out_rows + 63) / 64) * 8) : NULL;
			if (!v || (src_nb && !nb)) {
				rc = dev_fail(&x, "projection gather");
				break;
			}
			gl[ngl].src = from->d_data;
			gl[ngl].src_nullbits = src_nb;
			gl[ngl].rid = rid;
			gl[ngl].dst = v;
			gl[ngl].dst_nullbits = nb;
			ngl++;
			d_vals[c] = v;
			d_nulls[c] = nb;
		}
		if (rc == MIDORIDB_OK && ngl && mdb_dev_gather_cols(x.dev, gl, ngl, out_rows))
			rc = dev_fail(&x, "projection gather");
		/* pass 2: device -> host - or, with results kept on the device (mdb_database_results_on_device), the device columns
		 * become the result's own: a buffer of this statement changes hands, a base-table column is copied on the device */
		res->fetched = !keep;
		if (keep) {
			res->dev = x.dev;
			res->d_data = calloc((size_t)(ncols ? ncols : 1), sizeof(void *));
			res->d_nullbits = calloc((size_t)(ncols ? ncols : 1), sizeof(uint64_t *));
			if (!res->d_data || !res->d_nullbits)
				rc = -MIDORIDB_NOMEM;
		}
		for (int c = 0; c < ncols && rc == MIDORIDB_OK; c++) {
			if (!d_vals[c] || !out_rows)
				continue;
			if (!keep) {
				if (result_column_to_host(x.dev, res, c, d_vals[c], d_nulls[c]))
					rc = dev_fail(&x, "reading a result column");
				continue;
			}
			for (int pass = 0; pass < 2 && rc == MIDORIDB_OK; pass++) {
				const void *src = pass ? (const void *)d_nulls[c] : d_vals[c];
				const size_t bytes = pass ? (size_t)((out_rows + 63) / 64) * 8 : (size_t)out_rows * 8;
				void *own = NULL;
				bool shared = false;
				if (!src)
					continue;
				for (int k = 0; k < c; k++)	/* the same device column under two result columns (both key columns of SELECT *) */
					shared = shared || d_vals[k] == src || (const void *)d_nulls[k] == src;
				/* (a statement buffer sized for the worst case - every left row a group - must not pin hundreds of megabytes
				 * behind a small result until query_free: such a column is copied into a buffer of its own size instead) */
				for (int i = 0; i < x.bufs.n && !shared; i++)
					if (x.bufs.p[i] == src) {
						if (mdb_dev_alloc_size(x.dev, src) > 4 * bytes + ((size_t)1 << 20))
							break;
						own = x.bufs.p[i];
						x.bufs.p[i] = x.bufs.p[--x.bufs.n];
						break;
					}
				if (!own) {
					if (mdb_dev_alloc(x.dev, bytes, &own) || mdb_dev_gather64(x.dev, src, NULL, NULL, bytes / 8, own, NULL)) {
						rc = dev_fail(&x, "keeping a result column on the device");
						break;
					}
				}
				if (pass)
					res->d_nullbits[c] = own;
				else
					res->d_data[c] = own;
			}
		}
		if (keep && rc == MIDORIDB_OK && mdb_dev_sync(x.dev))
			rc = dev_fail(&x, "result columns");
		free(d_vals);
		free(d_nulls);
		if (rc)
			goto out;
	}
	res->dict = &cat->dict;
	res->exec_ms = now_ms() - t0;
	res->joined_rows = x.joined_rows;
	*out = res;
	res = NULL;
	rc = MIDORIDB_OK;
out:
	shard_cleanup(&x);
	free_all(&x);
	mdb_result_free(res);
	free(keys);
	free(order);
	free(key_tbl);
	free(key_col);
	free(src);
	free(direct_vals);
	free(direct_nulls);
	return rc;
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
static int dml_value_on_left(const struct mdb_expr *e)
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
static int dml_check_values(const struct mdb_expr *e, char *err, size_t errlen)
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
static int dml_select_rows(struct mdb_catalog *cat, struct mdb_dml *d, struct exec *x, struct mdb_select *s, struct mdb_from_tab *tab,
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
