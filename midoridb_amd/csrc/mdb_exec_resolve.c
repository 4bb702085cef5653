/*
 * mdb_exec_resolve.c - plan normalisation and the semantic checks that guard the executor: what the reference does in
 * its optimiser (src/engine/optimiser_select.c:114-238: NAME -> fully qualified FIELDNAME, alias -> table name, SELECT * expansion)
 * and in its semantic phase (src/parser/semantic_select.c:1575-1718, 2135-2186, 2470-2478: unknown table / column, duplicate column
 * names across FROM tables, operand types, the GROUP BY rule), with the reference's error texts.  Split off mdb_exec.c in round 4.
 */
#include "mdb_exec_internal.h"

/* ------------------------------------------------------------------ plan resolution */

bool field_eq(const struct mdb_expr *a, const struct mdb_expr *b)
{
	return a->kind == MDB_EX_FIELD && b->kind == MDB_EX_FIELD && a->tbl_idx == b->tbl_idx && a->col_idx == b->col_idx;
}

/* a bare name that is a select-list alias stands for the aliased item: a column (any clause) or COUNT(*) (HAVING); names of real columns win */
static void subst_alias(const struct mdb_select *s, struct mdb_expr *e, bool count_ok)
{
	if (!e)
		return;
	if (e->kind == MDB_EX_NAME && s->sel_alias) {
		for (int t = 0; t < s->ntabs; t++)
			for (int c = 0; c < s->tabs[t].t->ncols; c++)
				if (strcmp(s->tabs[t].t->cols[c].name, e->col) == 0)
					return;
		for (int i = 0; i < s->nsel; i++) {
			if (!s->sel_alias[i][0] || strcmp(s->sel_alias[i], e->col) != 0)
				continue;
			const struct mdb_expr *to = s->sel[i];
			if (to->kind == MDB_EX_FIELD) {		/* (resolved: the select list is resolved first) */
				e->kind = MDB_EX_FIELD;
				mdb_copy_name(e->tbl, to->tbl);
				mdb_copy_name(e->col, to->col);
				e->tbl_idx = to->tbl_idx;
				e->col_idx = to->col_idx;
				e->type = to->type;
			} else if (to->kind == MDB_EX_COUNT && count_ok && to->nkids == 0) {
				e->kind = MDB_EX_COUNT;
				e->op = to->op;
				e->ival = to->ival;
			}
			return;
		}
	}
	for (int i = 0; i < e->nkids; i++)
		subst_alias(s, e->kids[i], count_ok);
}

int resolve_expr(struct mdb_select *s, struct mdb_expr *e, char *err, size_t errlen)
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
bool is_having_clause(const char *clause)
{
	return strcmp(clause, "having") == 0;
}

/* type of a comparison operand as the reference's semantic phase sees it (check_value_types_cmp, semantic_select.c:2135-2186):
 * a raw string is a VARCHAR - SELECT does not box it into a DATE ("raw values are not auto-boxed", executor_select.c:193) -
 * while DELETE / UPDATE parse it against a DATE / DATETIME column (semantic_delete.c:160-200): `dml` */
int operand_type(const struct mdb_expr *o, const struct mdb_expr *other, bool dml)
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

int check_predicate_x(const struct mdb_expr *e, const char *clause, bool dml, char *err, size_t errlen)
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
	case MDB_EX_LIKE:
		/* upstream parses LIKE, checks its operands (semantic_select.c) and then evaluates it - and NOT LIKE - to TRUE for every
		 * row, NULL cells included (executor_select.c:1027-1074: no case for it): a defect, not a semantics to keep.  Here it
		 * has SQL's meaning over a VARCHAR column and a string pattern ('%' any run of characters, '_' any one character; a
		 * NULL cell matches nothing, neither does it under NOT LIKE), SELECT only */
		if (dml || e->nkids != 2 || e->kids[0]->kind != MDB_EX_FIELD || e->kids[0]->type != MDB_CT_VARCHAR || e->kids[1]->kind != MDB_EX_STRING) {
			ERR("expressions in %s clause must be a type of comparison (LIKE takes a VARCHAR column and a string pattern)\n", clause);
			return -MIDORIDB_ERROR;
		}
		return MIDORIDB_OK;
	default:
		ERR("expressions in %s clause must be a type of comparison\n", clause);
		return -MIDORIDB_ERROR;
	}
}

int check_predicate(const struct mdb_expr *e, const char *clause, char *err, size_t errlen)
{
	return check_predicate_x(e, clause, false, err, errlen);
}

bool expr_has_count(const struct mdb_expr *e)
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
int fields_in_select_list(const struct mdb_select *s, const struct mdb_expr *e, const char *clause, char *err, size_t errlen)
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

int resolve_select(struct mdb_catalog *cat, struct mdb_select *s, char *err, size_t errlen)
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
	/* `column AS name` / `COUNT(*) AS name` (upstream parses and checks aliases, then fails at execution: "cannot build columns hashtable" -
	 * SURVEY.md 8a D7; here the alias names the result column and may stand for its item in GROUP BY, HAVING and ORDER BY) */
	for (int i = 0; i < s->nsel; i++) {
		struct mdb_expr *e = s->sel[i];
		if (e->kind != MDB_EX_ALIAS || e->nkids != 1)
			continue;
		struct mdb_expr *kid = e->kids[0];
		if (kid->kind != MDB_EX_NAME && kid->kind != MDB_EX_FIELD && kid->kind != MDB_EX_COUNT)
			continue;	/* (an aliased expression: rejected below like the expression itself) */
		if (!s->sel_alias && !(s->sel_alias = calloc((size_t)s->nsel, sizeof(*s->sel_alias)))) {
			ERR("out of memory\n");
			return -MIDORIDB_NOMEM;
		}
		for (int k = 0; k < i; k++)
			if (s->sel_alias[k][0] && strcmp(s->sel_alias[k], e->col) == 0) {
				ERR("duplicate alias: '%.128s'\n", e->col);
				return -MIDORIDB_ERROR;
			}
		mdb_copy_name(s->sel_alias[i], e->col);
		e->nkids = 0;
		mdb_expr_free(e);
		s->sel[i] = kid;
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
			ERR("only columns and COUNT(*) - with or without an alias - are supported in the select list (expressions are not executed by the reference)\n");
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
		subst_alias(s, s->group[i], false);
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
			subst_alias(s, s->having, true);
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
			subst_alias(s, o, false);
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
