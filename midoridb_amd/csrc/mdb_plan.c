/*
 * mdb_plan.c - RPN token queue -> statement plan.
 *
 * The same stack machine as the reference's AST builders (reference
 * src/parser/ast_select.c:1021-1140 for SELECT, ast_create.c / ast_insert.c for DDL/DML): every
 * token either pushes a leaf or pops its operands and pushes the combined node; "SELECT d n" pops
 * its n children (select items, table references, then WHERE / GROUP BY / HAVING / ORDER BY /
 * LIMIT wrappers) and "STMT" ends the statement.  The result is a flat struct mdb_select
 * (left-deep join list) instead of the reference's pointer-linked AST: that is the executor's
 * input contract (SURVEY.md 8b "internal seam 1"), normalised later by mdb_exec.c the way the
 * reference optimiser does (optimiser_select.c:114-238, 395-464).
 */
#include "mdb_host.h"

/* wrapper node kinds that only live on the builder stack */
enum {
	BX_TABLE = 100, BX_JOIN, BX_ONEXPR, BX_WHERE, BX_GROUPBY, BX_HAVING, BX_ORDERBYITEM, BX_ORDERBYLIST,
	BX_LIMIT, BX_SELECTALL, BX_ASSIGN,
};

static struct mdb_expr *ex_new(int kind)
{
	struct mdb_expr *e = calloc(1, sizeof(*e));
	if (e) {
		e->kind = kind;
		e->tbl_idx = e->col_idx = -1;
	}
	return e;
}

static int ex_add_kid(struct mdb_expr *e, struct mdb_expr *k)
{
	struct mdb_expr **nk = realloc(e->kids, sizeof(*nk) * (size_t)(e->nkids + 1));
	if (!nk)
		return -MIDORIDB_NOMEM;
	e->kids = nk;
	e->kids[e->nkids++] = k;
	return 0;
}

void mdb_expr_free(struct mdb_expr *e)
{
	if (!e)
		return;
	for (int i = 0; i < e->nkids; i++)
		mdb_expr_free(e->kids[i]);
	free(e->kids);
	free(e->sval);
	free(e);
}

struct bstack {
	struct mdb_expr **v;
	int n, cap;
};

static int push(struct bstack *s, struct mdb_expr *e)
{
	if (!e)
		return -MIDORIDB_NOMEM;
	if (s->n == s->cap) {
		int nc = s->cap ? s->cap * 2 : 32;
		struct mdb_expr **nv = realloc(s->v, sizeof(*nv) * (size_t)nc);
		if (!nv)
			return -MIDORIDB_NOMEM;
		s->v = nv;
		s->cap = nc;
	}
	s->v[s->n++] = e;
	return 0;
}

static struct mdb_expr *pop(struct bstack *s)
{
	return s->n ? s->v[--s->n] : NULL;
}

static bool starts(const char *s, const char *prefix)
{
	return strncmp(s, prefix, strlen(prefix)) == 0;
}

/* pops n nodes and appends them to `parent` in their original (push) order */
static int pop_n_into(struct bstack *st, int n, struct mdb_expr *parent)
{
	if (n < 0 || st->n < n)
		return -MIDORIDB_ERROR;
	for (int i = st->n - n; i < st->n; i++)
		if (ex_add_kid(parent, st->v[i]))
			return -MIDORIDB_NOMEM;
	st->n -= n;
	return 0;
}

static void copy_name(char *dst, const char *src)
{
	mdb_copy_name(dst, src);
	dst[MDB_NAME_LEN - 1] = 0;
}

/* flatten the FROM tree into the left-deep list; returns 0 or error */
static int flatten_from(struct mdb_expr *ref, struct mdb_select *s, char *err, size_t errlen, bool is_first_of_comma_item)
{
	if (ref->kind == BX_TABLE) {
		struct mdb_from_tab *nt = realloc(s->tabs, sizeof(*nt) * (size_t)(s->ntabs + 1));
		struct mdb_expr **non = realloc(s->on, sizeof(*non) * (size_t)(s->ntabs + 1));
		int *njt = realloc(s->join_type, sizeof(int) * (size_t)(s->ntabs + 1));
		if (nt)
			s->tabs = nt;
		if (non)
			s->on = non;
		if (njt)
			s->join_type = njt;
		if (!nt || !non || !njt)
			return -MIDORIDB_NOMEM;
		memset(&s->tabs[s->ntabs], 0, sizeof(s->tabs[0]));
		copy_name(s->tabs[s->ntabs].name, ref->tbl);
		copy_name(s->tabs[s->ntabs].alias, ref->col);
		s->on[s->ntabs] = NULL;
		s->join_type[s->ntabs] = 1;
		s->ntabs++;
		(void)is_first_of_comma_item;
		return 0;
	}
	if (ref->kind == BX_JOIN) {
		int rc;
		if (ref->nkids != 3 || ref->kids[1]->kind != BX_TABLE) {
			snprintf(err, errlen, "unsupported join shape\n");
			return -MIDORIDB_ERROR;
		}
		rc = flatten_from(ref->kids[0], s, err, errlen, false);
		if (rc)
			return rc;
		rc = flatten_from(ref->kids[1], s, err, errlen, false);
		if (rc)
			return rc;
		/* steal the ON expression */
		s->on[s->ntabs - 1] = ref->kids[2]->kids[0];
		ref->kids[2]->kids[0] = NULL;
		ref->kids[2]->nkids = 0;
		s->join_type[s->ntabs - 1] = ref->op;
		return 0;
	}
	snprintf(err, errlen, "unsupported FROM clause\n");
	return -MIDORIDB_ERROR;
}

static int finish_select(struct mdb_expr *root, struct mdb_stmt *out, char *err, size_t errlen)
{
	struct mdb_select *s = &out->sel;
	int rc = 0;

	out->kind = MDB_ST_SELECT;
	s->distinct = (root->op & 2) != 0;
	for (int i = 0; i < root->nkids && !rc; i++) {
		struct mdb_expr *k = root->kids[i];
		switch (k->kind) {
		case BX_SELECTALL:
			s->select_all = true;
			break;
		case BX_TABLE:
		case BX_JOIN:
			rc = flatten_from(k, s, err, errlen, true);
			break;
		case BX_WHERE:
			s->where = k->kids[0];
			k->kids[0] = NULL;
			k->nkids = 0;
			break;
		case BX_GROUPBY:
			s->group = k->kids;
			s->ngroup = k->nkids;
			k->kids = NULL;
			k->nkids = 0;
			break;
		case BX_HAVING:
			s->having = k->kids[0];
			k->kids[0] = NULL;
			k->nkids = 0;
			break;
		case BX_ORDERBYLIST:
			s->order = calloc((size_t)(k->nkids ? k->nkids : 1), sizeof(*s->order));
			s->order_desc = calloc((size_t)(k->nkids ? k->nkids : 1), sizeof(int));
			if (!s->order || !s->order_desc) {
				rc = -MIDORIDB_NOMEM;
				break;
			}
			for (int j = 0; j < k->nkids; j++) {
				struct mdb_expr *it = k->kids[j];
				if (it->kind != BX_ORDERBYITEM || it->nkids != 1) {
					snprintf(err, errlen, "error while running syntax analysis on query\n");
					rc = -MIDORIDB_ERROR;
					break;
				}
				s->order[s->norder] = it->kids[0];
				s->order_desc[s->norder] = it->op;
				s->norder++;
				it->kids[0] = NULL;
				it->nkids = 0;
			}
			break;
		case BX_LIMIT:
			for (int j = 0; j < k->nkids; j++)
				if (k->kids[j]->kind != MDB_EX_INT || k->kids[j]->ival < 0) {
					snprintf(err, errlen, "LIMIT takes non-negative integer literals\n");
					rc = -MIDORIDB_ERROR;
				}
			if (rc)
				break;
			s->has_limit = true;
			s->limit_off = k->nkids == 2 ? k->kids[0]->ival : 0;
			s->limit_cnt = k->kids[k->nkids - 1]->ival;
			break;
		default: {
			struct mdb_expr **ns = realloc(s->sel, sizeof(*ns) * (size_t)(s->nsel + 1));
			if (!ns) {
				rc = -MIDORIDB_NOMEM;
				break;
			}
			s->sel = ns;
			s->sel[s->nsel++] = k;
			root->kids[i] = NULL;
			break;
		}
		}
	}
	/* kids moved into the plan were NULLed; free the rest of the scaffold */
	for (int i = 0; i < root->nkids; i++)
		if (root->kids[i])
			mdb_expr_free(root->kids[i]);
	free(root->kids);
	free(root);
	return rc;
}

/* 32-bit folding of INSERT literal arithmetic, like the reference (optimiser_insert.c:21-110) */
static struct mdb_expr *fold_insert_arith(const char *op, struct mdb_expr *l, struct mdb_expr *r)
{
	struct mdb_expr *e;
	bool isf = l->kind == MDB_EX_FLOAT || (r && r->kind == MDB_EX_FLOAT);

	if ((l->kind != MDB_EX_INT && l->kind != MDB_EX_FLOAT) || (r && r->kind != MDB_EX_INT && r->kind != MDB_EX_FLOAT))
		return NULL;
	e = ex_new(isf ? MDB_EX_FLOAT : MDB_EX_INT);
	if (!e)
		return NULL;
	if (isf) {
		double a = l->kind == MDB_EX_FLOAT ? l->dval : (double)l->ival;
		double b = r ? (r->kind == MDB_EX_FLOAT ? r->dval : (double)r->ival) : 0;
		if (!strcmp(op, "ADD"))
			e->dval = a + b;
		else if (!strcmp(op, "SUB"))
			e->dval = a - b;
		else if (!strcmp(op, "MUL"))
			e->dval = a * b;
		else if (!strcmp(op, "DIV"))
			e->dval = a / b;
		else if (!strcmp(op, "NEG"))
			e->dval = -a;
		else {
			mdb_expr_free(e);
			return NULL;
		}
	} else {
		int a = (int)l->ival, b = r ? (int)r->ival : 0;
		if (!strcmp(op, "ADD"))
			e->ival = a + b;
		else if (!strcmp(op, "SUB"))
			e->ival = a - b;
		else if (!strcmp(op, "MUL"))
			e->ival = a * b;
		else if (!strcmp(op, "DIV") && b)
			e->ival = a / b;
		else if (!strcmp(op, "MOD") && b)
			e->ival = a % b;
		else if (!strcmp(op, "NEG"))
			e->ival = -a;
		else {
			mdb_expr_free(e);
			return NULL;
		}
	}
	return e;
}

int mdb_plan_build(const struct mdb_rpn *rpn, struct mdb_stmt *out, char *err, size_t errlen)
{
	struct bstack st = {0};
	int rc = -MIDORIDB_ERROR;
	bool done = false;
	bool is_insert = false;

	memset(out, 0, sizeof(*out));
	/* INSERT queues reuse the literal/arithmetic tokens with constant folding */
	for (int i = 0; i < rpn->n; i++)
		if (starts(rpn->tok[i], "INSERTVALS"))
			is_insert = true;

#define FAIL(...)                                                                                   \
	do {                                                                                        \
		snprintf(err, errlen, __VA_ARGS__);                                                 \
		goto out;                                                                           \
	} while (0)
#define NEED(k)                                                                                     \
	do {                                                                                        \
		if (st.n < (k))                                                                     \
			FAIL("error while running syntax analysis on query\n");                     \
	} while (0)

	for (int i = 0; i < rpn->n && !done; i++) {
		const char *t = rpn->tok[i];
		struct mdb_expr *e = NULL;
		char a[256], b[256];
		int x, y;

		if (starts(t, "NAME ")) {
			e = ex_new(MDB_EX_NAME);
			if (e)
				copy_name(e->col, t + 5);
		} else if (starts(t, "NUMBER ")) {
			e = ex_new(MDB_EX_INT);
			if (e)
				e->ival = atoi(t + 7);	/* 32-bit, as the reference (ast_select.c:75) */
		} else if (starts(t, "STRING ")) {
			/* the lexer hands a string on WITH its quotes (midorisql.l: '...' or "..."); every later site strips them by
			 * position, so a token without a matching pair - possible only through mdb_query_execute_rpn() - stops here */
			const size_t sl = strlen(t + 7);
			if (sl < 2 || (t[7] != '\'' && t[7] != '"') || t[7 + sl - 1] != t[7])
				FAIL("error while running syntax analysis on query\n");
			e = ex_new(MDB_EX_STRING);
			if (e && !(e->sval = strdup(t + 7))) {	/* the literal with its quotes, as the lexer hands it on */
				mdb_expr_free(e);
				e = NULL;
			}
		} else if (starts(t, "FLOAT ")) {
			e = ex_new(MDB_EX_FLOAT);
			if (e)
				e->dval = atof(t + 6);
		} else if (starts(t, "BOOL ")) {
			e = ex_new(MDB_EX_BOOL);
			if (e)
				e->ival = atoi(t + 5);
		} else if (!strcmp(t, "NULL")) {
			e = ex_new(MDB_EX_NULL);
		} else if (!strcmp(t, "ADD") || !strcmp(t, "SUB") || !strcmp(t, "MUL") || !strcmp(t, "DIV") || !strcmp(t, "MOD")) {
			struct mdb_expr *r, *l;
			NEED(2);
			r = pop(&st);
			l = pop(&st);
			if (is_insert) {
				e = fold_insert_arith(t, l, r);
				mdb_expr_free(l);
				mdb_expr_free(r);
				if (!e)
					FAIL("unsupported expression in INSERT values\n");
			} else {
				e = ex_new(MDB_EX_ARITH);
				if (e && (ex_add_kid(e, l) || ex_add_kid(e, r)))
					goto nomem;
			}
		} else if (!strcmp(t, "NEG")) {
			struct mdb_expr *l;
			NEED(1);
			l = pop(&st);
			if (is_insert) {
				e = fold_insert_arith(t, l, NULL);
				mdb_expr_free(l);
				if (!e)
					FAIL("unsupported expression in INSERT values\n");
			} else {
				e = ex_new(MDB_EX_ARITH);
				if (e && ex_add_kid(e, l))
					goto nomem;
			}
		} else if (starts(t, "ALIAS ")) {
			struct mdb_expr *l;
			NEED(1);
			l = pop(&st);
			if (l->kind == BX_TABLE) {
				copy_name(l->col, t + 6);
				e = l;
			} else {
				e = ex_new(MDB_EX_ALIAS);
				if (e) {
					copy_name(e->col, t + 6);
					if (ex_add_kid(e, l))
						goto nomem;
				}
			}
		} else if (starts(t, "FIELDNAME ")) {
			if (sscanf(t + 10, "%127[A-Za-z0-9_].%127[A-Za-z0-9_]", a, b) != 2)
				FAIL("error while running syntax analysis on query\n");
			e = ex_new(MDB_EX_FIELD);
			if (e) {
				copy_name(e->tbl, a);
				copy_name(e->col, b);
			}
		} else if (!strcmp(t, "SELECTALL")) {
			e = ex_new(BX_SELECTALL);
		} else if (starts(t, "TABLE ")) {
			e = ex_new(BX_TABLE);
			if (e)
				copy_name(e->tbl, t + 6);
		} else if (starts(t, "GROUPBYLIST ")) {
			e = ex_new(BX_GROUPBY);
			if (e && pop_n_into(&st, atoi(t + 12), e))
				FAIL("error while running syntax analysis on query\n");
		} else if (starts(t, "ORDERBYLIST ")) {
			e = ex_new(BX_ORDERBYLIST);
			if (e && pop_n_into(&st, atoi(t + 12), e))
				FAIL("error while running syntax analysis on query\n");
		} else if (starts(t, "ORDERBYITEM ")) {
			e = ex_new(BX_ORDERBYITEM);
			if (e) {
				e->op = atoi(t + 12) != 0;
				if (pop_n_into(&st, 1, e))
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (starts(t, "CMP ")) {
			e = ex_new(MDB_EX_CMP);
			if (e) {
				e->op = atoi(t + 4);
				if (e->op < 1 || e->op > 6 || pop_n_into(&st, 2, e))
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (!strcmp(t, "AND") || !strcmp(t, "OR") || !strcmp(t, "XOR")) {
			e = ex_new(MDB_EX_LOGOP);
			if (e) {
				e->op = !strcmp(t, "AND") ? 0 : (!strcmp(t, "OR") ? 1 : 2);
				if (pop_n_into(&st, 2, e))
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (!strcmp(t, "ISNULL") || !strcmp(t, "ISNOTNULL")) {
			e = ex_new(MDB_EX_ISNULL);
			if (e) {
				e->op = !strcmp(t, "ISNOTNULL");
				if (pop_n_into(&st, 1, e))
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (starts(t, "ISIN ") || starts(t, "ISNOTIN ")) {
			const bool neg = starts(t, "ISNOTIN ");
			e = ex_new(MDB_EX_ISIN);
			if (e) {
				e->op = neg;
				if (pop_n_into(&st, atoi(t + (neg ? 8 : 5)) + 1, e))
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (!strcmp(t, "COUNTALL")) {
			e = ex_new(MDB_EX_COUNT);
		} else if (!strcmp(t, "COUNTFIELD")) {
			e = ex_new(MDB_EX_COUNT);
			if (e && pop_n_into(&st, 1, e))
				FAIL("error while running syntax analysis on query\n");
		} else if (!strcmp(t, "LIKE") || !strcmp(t, "NOTLIKE")) {
			e = ex_new(MDB_EX_LIKE);
			if (e)
				e->op = !strcmp(t, "NOTLIKE");
			if (e && pop_n_into(&st, 2, e))
				FAIL("error while running syntax analysis on query\n");
		} else if (!strcmp(t, "ONEXPR")) {
			e = ex_new(BX_ONEXPR);
			if (e && pop_n_into(&st, 1, e))
				FAIL("error while running syntax analysis on query\n");
		} else if (starts(t, "JOIN ")) {
			e = ex_new(BX_JOIN);
			if (e) {
				e->op = atoi(t + 5);
				if (pop_n_into(&st, 3, e) || e->kids[2]->kind != BX_ONEXPR)
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (!strcmp(t, "WHERE")) {
			e = ex_new(BX_WHERE);
			if (e && pop_n_into(&st, 1, e))
				FAIL("error while running syntax analysis on query\n");
		} else if (!strcmp(t, "HAVING")) {
			e = ex_new(BX_HAVING);
			if (e && pop_n_into(&st, 1, e))
				FAIL("error while running syntax analysis on query\n");
		} else if (starts(t, "LIMIT ")) {
			e = ex_new(BX_LIMIT);
			if (e && pop_n_into(&st, atoi(t + 6), e))
				FAIL("error while running syntax analysis on query\n");
		} else if (starts(t, "SELECT ")) {
			if (sscanf(t + 7, "%d %d", &x, &y) != 2)
				FAIL("error while running syntax analysis on query\n");
			e = ex_new(MDB_EX_ALIAS + 1000);	/* scaffold root */
			if (e) {
				e->op = x;
				if (pop_n_into(&st, y, e))
					FAIL("error while running syntax analysis on query\n");
				rc = finish_select(e, out, err, errlen);
				if (rc)
					goto out;
				continue;
			}
		} else if (!strcmp(t, "STMT")) {
			done = true;
			continue;
		/* ---- DELETE / UPDATE ---- */
		} else if (starts(t, "ASSIGN ")) {
			e = ex_new(BX_ASSIGN);
			if (e) {
				copy_name(e->col, t + 7);
				if (pop_n_into(&st, 1, e))
					FAIL("error while running syntax analysis on query\n");
			}
		} else if (starts(t, "DELETEONE ")) {
			struct mdb_dml *d = &out->dml;
			copy_name(d->name, t + 10);
			if (st.n > 1 || (st.n == 1 && st.v[0]->kind != BX_WHERE))
				FAIL("error while running syntax analysis on query\n");
			if (st.n == 1) {
				struct mdb_expr *w = pop(&st);
				d->where = w->kids[0];
				w->kids[0] = NULL;
				w->nkids = 0;
				mdb_expr_free(w);
			}
			out->kind = MDB_ST_DELETE;
			continue;
		} else if (starts(t, "UPDATE ")) {
			struct mdb_dml *d = &out->dml;
			if (sscanf(t + 7, "%127[A-Za-z0-9_] %d %d", a, &x, &y) != 3 || x < 1 || y < 0 || y > 1 || st.n != x + y)
				FAIL("error while running syntax analysis on query\n");
			copy_name(d->name, a);
			if (y) {
				struct mdb_expr *w = pop(&st);
				if (w->kind != BX_WHERE) {
					mdb_expr_free(w);
					FAIL("error while running syntax analysis on query\n");
				}
				d->where = w->kids[0];
				w->kids[0] = NULL;
				w->nkids = 0;
				mdb_expr_free(w);
			}
			d->assign = calloc((size_t)x, sizeof(*d->assign));
			if (!d->assign)
				goto nomem;
			d->nassign = x;
			for (int k = x - 1; k >= 0; k--) {
				struct mdb_expr *as = pop(&st);
				if (as->kind != BX_ASSIGN) {
					mdb_expr_free(as);
					FAIL("error while running syntax analysis on query\n");
				}
				copy_name(d->assign[k].col, as->col);
				d->assign[k].val = as->kids[0];
				as->kids[0] = NULL;
				as->nkids = 0;
				mdb_expr_free(as);
			}
			out->kind = MDB_ST_UPDATE;
			continue;
		/* ---- CREATE TABLE ---- */
		} else if (!strcmp(t, "STARTCOL")) {
			out->crt.pending_notnull = false;
			out->crt.pending_unique = false;
			continue;
		} else if (starts(t, "ATTR ")) {
			/* NOT NULL and PRIMARY KEY make the column non-nullable (reference executor_create.c:37,53); AUTO_INCREMENT
			 * and UNIQUE are parsed and ignored upstream as well */
			if (!strcmp(t + 5, "NOTNULL") || !strcmp(t + 5, "PRIKEY"))
				out->crt.pending_notnull = true;
			/* (UNIQUE / PRIMARY KEY: not enforced here either - remembered as "worth measuring", mdb_col_distinct) */
			if (!strcmp(t + 5, "UNIQUEKEY") || !strcmp(t + 5, "PRIKEY"))
				out->crt.pending_unique = true;
			continue;
		} else if (starts(t, "COLUMNDEF ")) {
			struct mdb_create *c = &out->crt;
			int type;
			if (sscanf(t + 10, "%d %127s", &x, a) != 2 || c->ncols >= MDB_MAX_COLS)
				FAIL("error while running syntax analysis on query\n");
			/* code = type * 10000 (+ varchar length): midorisql.y:475-483, ast_create.c:13-48 */
			switch (x / 10000) {
			case 4: case 5: type = MDB_CT_INTEGER; break;
			case 6: type = MDB_CT_TINYINT; break;
			case 8: type = MDB_CT_DOUBLE; break;
			case 10: type = MDB_CT_DATE; break;
			case 11: type = MDB_CT_DATETIME; break;
			case 13: type = MDB_CT_VARCHAR; break;
			default: FAIL("error while running syntax analysis on query\n");
			}
			copy_name(c->colname[c->ncols], a);
			c->coltype[c->ncols] = type;
			c->colprec[c->ncols] = type == MDB_CT_VARCHAR ? x % 10000 : 8;
			c->notnull[c->ncols] = c->pending_notnull;
			c->unique[c->ncols] = c->pending_unique;
			c->pending_notnull = false;
			c->pending_unique = false;
			c->ncols++;
			continue;
		} else if (starts(t, "CREATE ")) {
			if (sscanf(t + 7, "%d %d %127s", &x, &y, a) != 3 || y != out->crt.ncols)
				FAIL("error while running syntax analysis on query\n");
			out->kind = MDB_ST_CREATE;
			out->crt.if_not_exists = x != 0;
			copy_name(out->crt.name, a);
			continue;
		/* ---- INSERT ... VALUES ---- */
		} else if (starts(t, "COLUMN ")) {
			struct mdb_insert *ins = &out->ins;
			if (ins->ncolnames >= MDB_MAX_COLS)
				FAIL("too many columns\n");
			copy_name(ins->colname[ins->ncolnames++], t + 7);
			continue;
		} else if (starts(t, "INSERTCOLS ")) {
			continue;
		} else if (starts(t, "VALUES ")) {
			struct mdb_insert *ins = &out->ins;
			struct mdb_expr ***nv;
			int n = atoi(t + 7);
			NEED(n);
			if (ins->ntuples && n != ins->nvals)
				FAIL("column count doesn't match value count\n");
			nv = realloc(ins->vals, sizeof(*nv) * (size_t)(ins->ntuples + 1));
			if (!nv)
				goto nomem;
			ins->vals = nv;
			ins->vals[ins->ntuples] = calloc((size_t)(n ? n : 1), sizeof(struct mdb_expr *));
			if (!ins->vals[ins->ntuples])
				goto nomem;
			for (int k = 0; k < n; k++)
				ins->vals[ins->ntuples][k] = st.v[st.n - n + k];
			st.n -= n;
			ins->nvals = n;
			ins->ntuples++;
			continue;
		} else if (starts(t, "INSERTVALS ")) {
			if (sscanf(t + 11, "%d %d %127s", &x, &y, a) != 3 || y != out->ins.ntuples)
				FAIL("error while running syntax analysis on query\n");
			out->kind = MDB_ST_INSERT;
			copy_name(out->ins.name, a);
			continue;
		} else {
			FAIL("statement not supported by the MI355X SELECT path: '%.64s'\n", t);
		}
		if (!e)
			goto nomem;
		if (push(&st, e)) {
			mdb_expr_free(e);
			goto nomem;
		}
	}
	if (!done || st.n != 0 || !out->kind)
		FAIL("error while running syntax analysis on query\n");
	rc = MIDORIDB_OK;
	goto out;
nomem:
	snprintf(err, errlen, "out of memory\n");
	rc = -MIDORIDB_NOMEM;
out:
	while (st.n)
		mdb_expr_free(pop(&st));
	free(st.v);
	if (rc)
		mdb_stmt_free(out);
	return rc;
#undef FAIL
#undef NEED
}

void mdb_stmt_free(struct mdb_stmt *s)
{
	for (int i = 0; i < s->sel.nsel; i++)
		mdb_expr_free(s->sel.sel[i]);
	free(s->sel.sel);
	free(s->sel.sel_alias);
	for (int i = 0; i < s->sel.ntabs; i++)
		mdb_expr_free(s->sel.on[i]);
	free(s->sel.on);
	free(s->sel.tabs);
	free(s->sel.join_type);
	mdb_expr_free(s->sel.where);
	for (int i = 0; i < s->sel.ngroup; i++)
		mdb_expr_free(s->sel.group[i]);
	free(s->sel.group);
	mdb_expr_free(s->sel.having);
	for (int i = 0; i < s->sel.norder; i++)
		mdb_expr_free(s->sel.order[i]);
	free(s->sel.order);
	free(s->sel.order_desc);
	for (int i = 0; i < s->ins.ntuples; i++) {
		for (int k = 0; k < s->ins.nvals; k++)
			mdb_expr_free(s->ins.vals[i][k]);
		free(s->ins.vals[i]);
	}
	free(s->ins.vals);
	mdb_expr_free(s->dml.where);
	for (int i = 0; i < s->dml.nassign; i++)
		mdb_expr_free(s->dml.assign[i].val);
	free(s->dml.assign);
	memset(s, 0, sizeof(*s));
}
