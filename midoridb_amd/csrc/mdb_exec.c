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
#include "mdb_exec_internal.h"

/* ------------------------------------------------------------------ device-side execution state */



int dev_fail(struct exec *x, const char *what)
{
	snprintf(x->err, x->errlen, "execution phase: %s: %s\n", what, mdb_dev_last_error(x->dev));
	return -MIDORIDB_INTERNAL;
}

int track(struct exec *x, void *p)
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

void *dalloc(struct exec *x, size_t bytes)
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

void free_all(struct exec *x)
{
	for (int i = 0; i < x->bufs.n; i++)
		mdb_dev_free(x->dev, x->bufs.p[i]);
	free(x->bufs.p);
	x->bufs.p = NULL;
	x->bufs.n = x->bufs.cap = 0;
}

/* Re-map every row-id vector of the stream through `sel` (n_new positions into the old stream). */
int stream_select(struct exec *x, int ntabs_in_stream, const uint32_t *sel, uint64_t n_new)
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
int stream_apply_sel(struct exec *x, int ntabs_in_stream, const uint32_t *sel, uint64_t n_new)
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
int stream_column(struct exec *x, const struct mdb_expr *f, const int64_t **vals, const uint64_t **nulls)
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
int double_join_keys(struct exec *x, const struct mdb_column *col, const uint32_t *rid, uint64_t n, const void **vals,
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

/* ------------------------------------------------------------------ FROM clause */

/* split an ON expression into conjuncts; pick the first "left-stream field = field of table t" as the hash key */
void collect_conjuncts(struct mdb_expr *e, struct mdb_expr **out, int *n, int cap)
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
uint64_t expr_tables(const struct mdb_expr *e)
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

bool where_split(const struct mdb_select *s, struct where_split *w)
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
int table_filter(struct exec *x, int t, const struct mdb_expr *const *conj, int nconj, const uint32_t **sel, uint64_t *m)
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
int table_column(struct exec *x, int t, const struct mdb_expr *key, const uint32_t *sel, uint64_t m, const void **vals,
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
int fused_operand(struct exec *x, int t, const struct mdb_expr *key, const struct mdb_expr *const *conj, int nconj,
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
/* The catalog's statistics of the key columns of the operator call that follows (mdb_dev_call_stats, include/mdb_dev.h): fl / fr = the
 * fields the key columns pl / pr come from - the columns themselves or filtered streams of them (a superset range is fine).  Nothing is
 * handed over for column types without a range (DOUBLE, VARCHAR), for shadow tables of the sharded mode, or when a range cannot be had:
 * the operator then looks at the data itself.  op_stats_end() after the call. */
void op_stats_begin(struct exec *x, const struct mdb_expr *fl, const void *pl, const struct mdb_expr *fr, const void *pr)
{
	struct mdb_dev_col_stats st[2];
	const struct mdb_expr *f[2] = { fl, fr };
	memset(st, 0, sizeof(st));
	if (x->cat->dist || !fl || !pl || (fr && !pr))
		return;
	for (int i = 0; i < (fr ? 2 : 1); i++) {
		if (f[i]->kind != MDB_EX_FIELD || f[i]->tbl_idx < 0 || x->orig_tab[f[i]->tbl_idx])
			return;
		struct mdb_table *tb = x->s->tabs[f[i]->tbl_idx].t;
		struct mdb_column *col = &tb->cols[f[i]->col_idx];
		if (mdb_col_range(x->cat, tb, col, &st[i].min, &st[i].max) != MIDORIDB_OK)
			return;
		st[i].rows = tb->device_only ? tb->dev_rows : tb->nrows;
		st[i].nulls = col->null_count;
		/* (measured at ingest, followed through appended rows; a filtered stream of a column without a value twice has none either) */
		if (mdb_col_distinct(x->cat, tb, col))
			st[i].flags |= MDB_COL_DISTINCT;
	}
	(void)mdb_dev_call_stats(x->dev, pl, &st[0], fr ? pr : NULL, fr ? &st[1] : NULL);
}

/* Most groups a GROUP BY over the key field f can have, by the catalog: a group needs a key value, and the column holds at most
 * max - min + 1 of them (a filtered stream of the column: fewer).  `rows` when nothing is known. */
uint64_t op_groups_bound(struct exec *x, const struct mdb_expr *f, uint64_t rows)
{
	int64_t lo, hi;
	if (x->cat->dist || !f || f->kind != MDB_EX_FIELD || f->tbl_idx < 0 || x->orig_tab[f->tbl_idx])
		return rows;
	struct mdb_table *tb = x->s->tabs[f->tbl_idx].t;
	if (mdb_col_range(x->cat, tb, &tb->cols[f->col_idx], &lo, &hi) != MIDORIDB_OK || lo > hi)
		return rows;
	const uint64_t span = (uint64_t)hi - (uint64_t)lo;	/* (max - min: + 1 below, unless that is 2^64) */
	return span + 1 && span + 1 < rows ? span + 1 : rows;
}

void op_stats_end(struct exec *x)
{
	(void)mdb_dev_call_stats(x->dev, NULL, NULL, NULL, NULL);
}

/* Table t after a join that carried its payload cells to the stream's rows: read through a SHADOW whose key column is the stream's own key
 * column, whose payload columns pc[0 .. np) are the carried cells out[], whose row-id vector is the identity */
static int payload_shadow(struct exec *x, int t, int key_col, const int64_t *vl, const int *pc, int np, void *const *out)
{
	struct mdb_select *s = x->s;
	const struct mdb_table *rt = s->tabs[t].t;
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
	sh->cols[key_col].d_data = (void *)vl;	/* (in every joined tuple the key of t IS the stream's key; a NULL key joined nothing) */
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

/* the payload columns of table t a join would have to carry: the columns the statement reads, the key column apart - at most two, none
 * with NULLs, all on the device; -1 when the table does not qualify */
static int payload_columns(struct exec *x, int t, int key_col, int *pc)
{
	const struct mdb_table *rt = x->s->tabs[t].t;
	int np = 0;
	for (int c = 0; c < rt->ncols; c++) {
		if (!x->need[t][c] || c == key_col)
			continue;
		if (np == 2 || rt->cols[c].d_nullbits || !rt->cols[c].d_data)
			return -1;
		pc[np++] = c;
	}
	return np;
}

/* Join elimination (round 6): table t joined on `stream key = its own column kr`, when the CATALOG says that every row of the
 * stream finds exactly one partner in it - kr holds no value twice (MDB_COL_DISTINCT: measured at ingest, followed through appends, looked
 * at again after UPDATE / DELETE), no NULL, and as many rows as its range has values: EVERY value of [min, max]; the stream's key column
 * has no NULL and its range lies inside - and the statement reads nothing of t but that key: the joined rows ARE the stream's rows, t's key
 * column in them IS the stream's key column.  The reference runs its nested loop over B for every row of A
 * (/root/reference/src/engine/executor_select.c:1076-1149) to find the same; here the table is not touched at all.  The statistics are the
 * store's own invariants (ranges are never narrower than the data, the distinct flag is never set on a column that holds a value twice):
 * a range that went stale wider only makes the rule not apply.  MDB_JOIN_ELIMINATION=0: never. */
static bool join_is_total_by_catalog(struct exec *x, const struct mdb_expr *kl, const struct mdb_expr *kr)
{
	const char *knob = mdb_knob("MDB_JOIN_ELIMINATION");
	if (x->cat->dist || (knob && knob[0] == '0') || !kl || !kr || kl->kind != MDB_EX_FIELD || kr->kind != MDB_EX_FIELD || kl->tbl_idx < 0 || kr->tbl_idx < 0 ||
	    kl->type == MDB_CT_DOUBLE || kr->type == MDB_CT_DOUBLE || kl->type != kr->type || x->orig_tab[kr->tbl_idx])
		return false;
	/* (the stream's key may be named through a table that was itself joined this way: its key column is an earlier table's) */
	int lt = kl->tbl_idx, lc = kl->col_idx;
	for (int hops = 0; x->orig_tab[lt] && hops < MDB_MAX_TABS; hops++) {
		if (x->same_col[lt] != lc)
			return false;
		const int nt = x->same_as_tbl[lt];
		lc = x->same_as_col[lt];
		lt = nt;
	}
	if (x->orig_tab[lt])
		return false;
	struct mdb_table *tl = x->s->tabs[lt].t, *tr = x->s->tabs[kr->tbl_idx].t;
	struct mdb_column *cl = &tl->cols[lc], *cr = &tr->cols[kr->col_idx];
	const uint64_t r_rows = tr->device_only ? tr->dev_rows : tr->nrows;
	int64_t llo, lhi, rlo, rhi;
	if (!r_rows || cl->null_count || cr->null_count || mdb_col_range(x->cat, tl, cl, &llo, &lhi) != MIDORIDB_OK ||
	    mdb_col_range(x->cat, tr, cr, &rlo, &rhi) != MIDORIDB_OK || rlo > rhi)
		return false;
	if ((uint64_t)rhi - (uint64_t)rlo + 1 != r_rows || (llo <= lhi && (llo < rlo || lhi > rhi)))
		return false;
	return mdb_col_distinct(x->cat, tr, cr);	/* (last: it may scan the column) */
}

/* ... and nothing but the key column of table t is read by the statement */
static bool only_key_needed(const struct exec *x, int t, int key_col)
{
	const struct mdb_table *rt = x->s->tabs[t].t;
	for (int c = 0; c < rt->ncols; c++)
		if (x->need[t][c] && c != key_col)
			return false;
	return true;
}

/* SEVERAL tables joined to the stream on ONE key, each a primary-key table that every row of the stream finds exactly one partner in
 * (SELECT * FROM A JOIN B ON A.k = B.k JOIN C ON A.k = C.k - BASELINE configs[4]'s join-only form; the reference joins B, then C against the
 * materialised A x B: executor_select.c:1076-1232): table t and the tables behind it whose whole ON clause is `earlier key = own column`,
 * that have no WHERE conjunct of their own and qualify like table t does, are handed to mdb_dev_join_payload_multi in one call - the
 * stream's key column is sorted once for all of them - with the catalog's key ranges as the window.  0 = done for table t and
 * x->joined_ahead[] tables (join_next_table returns at once for those), 1 = not such a statement or not such a join (nothing changed:
 * table t is joined on its own), < 0 = error. */
static int join_with_payload_multi(struct exec *x, int t, const struct mdb_expr *kl, const struct mdb_expr *kr, const int64_t *vl, const uint64_t *nl,
				   const void *vr, const uint64_t *nr, uint64_t r_rows)
{
	struct mdb_select *s = x->s;
	struct mdb_dev_payload_right right[4];
	const struct mdb_expr *keys[4];
	int tabs[4], pcs[4][2], nt = 0, streams = 0;
	int64_t lo, hi;
	const char *rowjoin = mdb_knob("MDB_ROWJOIN");	/* ("2": the row-order form for tables of any size - tests) */
	if ((x->n < ((uint64_t)1 << 24) && !(rowjoin && rowjoin[0] == '2')) || nl || nr || x->orig_tab[t] || kl->kind != MDB_EX_FIELD || kl->tbl_idx < 0 || x->orig_tab[kl->tbl_idx])
		return 1;
	{
		struct mdb_table *lt = s->tabs[kl->tbl_idx].t;
		if (mdb_col_range(x->cat, lt, &lt->cols[kl->col_idx], &lo, &hi) != MIDORIDB_OK || lo > hi)
			return 1;
	}
	memset(right, 0, sizeof(right));
	for (int t2 = t; t2 < s->ntabs && nt < 4; t2++) {
		const struct mdb_expr *k2 = kr;
		if (t2 > t) {
			const struct mdb_expr *on = s->on[t2], *mine = NULL, *other = NULL;
			if (!on || on->kind != MDB_EX_CMP || on->op != MDB_CMP_EQ || on->kids[0]->kind != MDB_EX_FIELD || on->kids[1]->kind != MDB_EX_FIELD)
				break;
			if (on->kids[0]->tbl_idx == t2 && on->kids[1]->tbl_idx < t2) {
				mine = on->kids[0];
				other = on->kids[1];
			} else if (on->kids[1]->tbl_idx == t2 && on->kids[0]->tbl_idx < t2) {
				mine = on->kids[1];
				other = on->kids[0];
			} else {
				break;
			}
			bool same_key = field_eq(other, kl);	/* (the stream's key, or the key column of a table this call joins on it) */
			for (int i = 0; i < nt && !same_key; i++)
				same_key = field_eq(other, keys[i]);
			if (!same_key || mine->type == MDB_CT_DOUBLE || x->orig_tab[t2] || (x->ws ? x->ws->npush[t2] != 0 : 0))
				break;
			k2 = mine;
		}
		struct mdb_table *rt = s->tabs[t2].t;
		const int np = payload_columns(x, t2, k2->col_idx, pcs[nt]);
		int64_t rlo, rhi;
		if (np < 1 || streams + np > 4 || !rt->nrows || mdb_col_range(x->cat, rt, &rt->cols[k2->col_idx], &rlo, &rhi) != MIDORIDB_OK || rlo > rhi)
			break;
		const void *kv = vr;
		const uint64_t *kn = nr;
		if (t2 > t && table_column(x, t2, k2, NULL, rt->nrows, &kv, &kn))
			break;
		if (kn)
			break;
		lo = rlo < lo ? rlo : lo;
		hi = rhi > hi ? rhi : hi;
		right[nt].keys = (const int64_t *)kv;
		right[nt].rows = t2 > t ? rt->nrows : r_rows;
		right[nt].npay = np;
		for (int i = 0; i < np; i++)
			right[nt].pay_in[i] = rt->cols[pcs[nt][i]].d_data;
		keys[nt] = k2;
		tabs[nt] = t2;
		streams += np;
		nt++;
	}
	if (nt < 2 || (uint64_t)hi - (uint64_t)lo >= ((uint64_t)1 << 27))
		return 1;
	for (int i = 0; i < nt; i++)
		for (int c = 0; c < right[i].npay; c++)
			if (!(right[i].out[c] = dalloc(x, x->n * 8)))
				return dev_fail(x, "allocating carried columns");
	const int rc = mdb_dev_join_payload_multi(x->dev, vl, NULL, x->n, right, nt, lo, hi);
	if (rc == 1)
		return 1;
	if (rc)
		return dev_fail(x, "join with payload (several tables on one key)");
	for (int i = 0; i < nt; i++) {
		const int prc = payload_shadow(x, tabs[i], keys[i]->col_idx, vl, pcs[i], right[i].npay, right[i].out);
		if (prc)
			return prc;
		if (i) {
			x->joined_ahead[tabs[i]] = true;
			x->same_col[tabs[i]] = keys[i]->col_idx;
			x->same_as_tbl[tabs[i]] = kl->tbl_idx;
			x->same_as_col[tabs[i]] = kl->col_idx;
		}
	}
	return 0;
}

int join_with_payload(struct exec *x, int t, const struct mdb_expr *kr, const int64_t *vl, const uint64_t *nl, const void *vr,
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
	return payload_shadow(x, t, kr->col_idx, vl, pc, np, out);
}

int join_next_table(struct exec *x, int t, const struct mdb_expr *const *pconj, int npconj)
{
	struct mdb_select *s = x->s;
	struct mdb_table *rt = s->tabs[t].t;
	struct mdb_expr *conj[32];
	int nconj = 0, key = -1;
	const struct mdb_expr *kl = NULL, *kr = NULL;
	uint32_t *pl = NULL, *pr = NULL;
	uint64_t J = 0;
	int rc;

	if (x->joined_ahead[t])		/* (joined together with an earlier table on the same key: join_with_payload_multi) */
		return MIDORIDB_OK;
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
	if (key >= 0 && nconj == 1 && !npconj && !x->cat->dist && x->n && only_key_needed(x, t, kr->col_idx) && join_is_total_by_catalog(x, kl, kr)) {
		/* join elimination: the catalog says every row of the stream has exactly one partner, and only t's key is read */
		const int64_t *vl;
		const uint64_t *nl;
		if ((rc = stream_column(x, kl, &vl, &nl)))
			return rc;
		if (!nl) {
			x->same_col[t] = kr->col_idx;
			x->same_as_tbl[t] = kl->tbl_idx;
			x->same_as_col[t] = kl->col_idx;
			x->joins_eliminated++;
			return payload_shadow(x, t, kr->col_idx, vl, NULL, 0, NULL);
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
			/* no equi-join key (FROM A, B; a general ON expression: reference executor_select.c:1096-1141): nothing says which rank a
			 * row's partners live on - the new table (its rows that passed its own WHERE conjuncts) is replicated on every rank
			 * and each rank pairs ITS stream with all of it: every (l, r) pair is produced exactly once, where l lives */
			const int tabs1[1] = { t };
			uint32_t *rids1[1] = { (uint32_t *)rsel };
			if ((rc = shard_rows(x, tabs1, 1, rids1, r_rows, NULL, NULL, SHARD_BROADCAST, &r_rows)))
				return rc;
			rsel = NULL;
			rt = s->tabs[t].t;
		} else {
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
			if (!rc && kr->type == MDB_CT_VARCHAR) {	/* (rows go where their STRING hashes to: the ranks' common id of it) */
				const int64_t *common = NULL;
				rc = shard_ids(x, kv, r_rows, true, false, &common);
				kv = common;
			}
			if (rc || (rc = shard_rows(x, tabs1, 1, rids1, r_rows, kv, kn, 0, &r_rows)))
				return rc;
			rsel = NULL;
			rt = s->tabs[t].t;
		}
		if (x->npart < 2 * MDB_MAX_TABS)
			x->part[x->npart++] = kr;
		}
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
		if (kl->type != MDB_CT_DOUBLE && kr->type != MDB_CT_DOUBLE)
			op_stats_begin(x, kl, vl, kr, vr);	/* (until the join is done: op_stats_end below / at the early returns) */
		if (x->n && r_rows && !x->cat->dist && !rsel && kl->type != MDB_CT_DOUBLE && kr->type != MDB_CT_DOUBLE) {
			int prc = nconj == 1 ? join_with_payload_multi(x, t, kl, kr, vl, nl, vr, nr, r_rows) : 1;
			if (prc == 1)
				prc = join_with_payload(x, t, kr, vl, nl, vr, nr, r_rows);
			if (prc <= 0)
				op_stats_end(x);
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
			const int jrc = mdb_dev_join_pairs(x->dev, vl, nl, x->n, vr, nr, r_rows, &pl, &pr, &J);
			op_stats_end(x);
			if (jrc)
				return dev_fail(x, "hash join");
			if (pl && track(x, pl))
				return -MIDORIDB_NOMEM;
			if (pr && track(x, pr))
				return -MIDORIDB_NOMEM;
		}
		op_stats_end(x);
	} else {
		/* no equi-join key: FROM A, B (ON 1=1) or a general ON -> all pairs, then the ON predicate */
		const uint64_t total = x->n * r_rows;
		uint64_t too_large = total > (1ull << 28) ? 1u : 0u;
		if (x->cat->dist && mdb_dist_allreduce_sum_u64(x->cat->dist, &too_large, 1)) {	/* (every rank must leave the statement together) */
			snprintf(x->err, x->errlen, "execution phase: %s\n", mdb_dist_last_error(x->cat->dist));
			return -MIDORIDB_INTERNAL;
		}
		if (too_large) {
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
		/* (two result columns may be ONE device buffer - both key columns of SELECT * over an equi-join: freed with the first) */
		bool dup_d = false, dup_n = false;
		for (int k = 0; k < c; k++) {
			dup_d = dup_d || (r->d_data && r->d_data[c] && r->d_data[k] == r->d_data[c]);
			dup_n = dup_n || (r->d_nullbits && r->d_nullbits[c] && r->d_nullbits[k] == r->d_nullbits[c]);
		}
		if (r->d_data && r->d_data[c] && !dup_d)
			mdb_dev_free(r->dev, r->d_data[c]);
		if (r->d_nullbits && r->d_nullbits[c] && !dup_n)
			mdb_dev_free(r->dev, r->d_nullbits[c]);
	}
	mdb_result_legacy_free(r);
	pthread_mutex_destroy(&r->legacy.mutex);
	free(r->data);
	free(r->nullbits);
	free(r->d_data);
	free(r->d_nullbits);
	free(r->colname);
	free(r->coltype);
	free(r->colprec);
	free(r);
}

/* one result column device -> host; a NULL cell reads as 0 through query_column_int64(), like the reference
 * (cpy_cols skips the copy into the zeroed row, executor_select.c:384-387) */
int result_column_to_host(mdb_dev_ctx *dev, struct mdb_result *res, int c, const void *d_vals, const uint64_t *d_nulls)
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
	mdb_result_legacy_rows(r);
	return MIDORIDB_OK;
}

double now_ms(void)
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
int fused_chain(struct mdb_select *s, const struct mdb_expr **keys, bool count_only)
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

/* a two-table INNER JOIN ON l = r whose result columns are all one of the two key columns, nothing else asked of it: kj[0..1] = the key
 * fields (left table's first).  With the order left open by the host (mdb_database_groups_any_order) mdb_dev_join_keys answers; in the
 * reference's order mdb_dev_join_keys_ordered, when the join turns out to be a primary-key join */
bool keys_only_join(const struct mdb_select *s, const struct mdb_catalog *cat, int has_count, const int *key_tbl, const int *key_col,
			   const int *src, int ncols, const struct mdb_expr **kj)
{
	if (cat->dist || s->ntabs != 2 || s->where || s->ngroup || has_count || s->distinct || s->norder || s->having || s->has_limit || !ncols)
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
	if (fused >= 0 && !cat->dist && split_ok) {
		/* join elimination: when the catalog says every right table of the chain is a complete primary key over the first table's key range,
		 * the fused join + GROUP BY operator has nothing to find out - the general plan drops those joins (join_next_table) and groups the
		 * first table's key column alone (the identity when that column holds no value twice) */
		bool all = s->ntabs > 1;
		for (int t = 1; t < s->ntabs && all; t++)
			all = !ws.npush[t] && only_key_needed(&x, t, fkeys[t]->col_idx) && join_is_total_by_catalog(&x, fkeys[0], fkeys[t]);
		if (all)
			fused = -1;
	}
	/* the join of two key columns, nothing else selected: in the reference's order it is tried as a primary-key join first (the keys with
	 * partners in the left table's row order; large tables only: the attempt runs the ordered join + GROUP BY operator), and left to the
	 * materialising join below when a key has several rows on a side */
	bool keys_only = keys_only_join(s, cat, has_count, key_tbl, key_col, src, ncols, kj);
	int64_t *keys_ordered = NULL;
	uint64_t keys_ordered_rows = 0;
	if (keys_only && !cat->groups_any_order) {
		const struct mdb_table *tl = s->tabs[kj[0]->tbl_idx].t, *tr = s->tabs[kj[1]->tbl_idx].t;
		int served = 0;
		int krc = 0;
		if (tl->nrows && !tl->cols[kj[0]->col_idx].d_nullbits && tl->cols[kj[0]->col_idx].d_data && join_is_total_by_catalog(&x, kj[0], kj[1])) {
			/* join elimination: every row of the left table has exactly one partner (the catalog's statistics) - the joined rows' key column
			 * IS the left table's key column, in its order, without a kernel */
			keys_ordered = (int64_t *)tl->cols[kj[0]->col_idx].d_data;
			keys_ordered_rows = tl->nrows;
			served = 2;
			x.joins_eliminated++;
		} else if (tl->nrows + tr->nrows >= (1u << 21)) {
			op_stats_begin(&x, kj[0], tl->cols[kj[0]->col_idx].d_data, kj[1], tr->cols[kj[1]->col_idx].d_data);
			krc = mdb_dev_join_keys_ordered(x.dev, tl->cols[kj[0]->col_idx].d_data, tl->cols[kj[0]->col_idx].d_nullbits, tl->nrows,
							 tr->cols[kj[1]->col_idx].d_data, tr->cols[kj[1]->col_idx].d_nullbits, tr->nrows, &keys_ordered,
							 &keys_ordered_rows, &served);
			op_stats_end(&x);
		}
		if (krc) {
			rc = dev_fail(&x, "join of two key columns");
			goto out;
		}
		keys_only = served != 0;
		/* (served == 2: the joined rows' key column IS the left table's key column - held by the result, never freed here) */
		if (keys_ordered && served != 2 && track(&x, keys_ordered)) {
			rc = -MIDORIDB_NOMEM;
			goto out;
		}
	}
	if (fused >= 0 && (!split_ok || ws.nresidual))
		fused = -1;	/* a conjunct reads several tables: it has to see the joined rows */
	if (fused >= 0 && cat->dist && s->ntabs > 2 && fkeys[0]->type == MDB_CT_VARCHAR)
		fused = -1;	/* (sharded chains over VARCHAR keys: the general plan exchanges the rows by their strings' common ids) */
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
		/* a group of the join needs a key that both sides hold: no more groups than rows or key values of either column (catalog
		 * statistics) - the output columns are sized for that, and become the result's own without a copy when the bound is close */
		if (!cat->dist && fkeys[0]->type != MDB_CT_DOUBLE) {
			cap = op_groups_bound(&x, fkeys[0], cap);
			cap = op_groups_bound(&x, fkeys[1], cap < nr_rows ? cap : (nr_rows ? nr_rows : 1));
		}
		bool multi_done = false, alias_checked = false;
		const bool text_keys = cat->dist && fkeys[0]->type == MDB_CT_VARCHAR;
		if (text_keys) {
			/* sharded mode, VARCHAR join keys: the ids of this process's dictionary mean nothing to the other ranks - the operator
			 * runs on the ranks' common ids (shard_ids) and its group keys are translated back where they arrive */
			const int64_t *cl = NULL, *cr = NULL;
			if ((rc = shard_ids(&x, (const int64_t *)lv, nl_rows, true, false, &cl)) || (rc = shard_ids(&x, (const int64_t *)rv, nr_rows, true, false, &cr)))
				goto out;
			lv = cl;
			rv = cr;
		}
		if (cat->dist && fkeys[0]->type != MDB_CT_DOUBLE && !text_keys && (rc = shard_promise_ranges(&x, fkeys[0], fkeys[1])))
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
								   ((only_count || cat->groups_any_order) ? 0u : MDB_ORDER_FIRST) | MDB_KEYS_MAY_ALIAS, x.d_fused_key,
								   x.d_count, NULL, cap, &G, &J)) {
					rc = dev_fail(&x, "join + group count over several tables");
					goto out;
				}
				multi_done = true;
				alias_checked = true;
			} else {
				if (fkeys[0]->type != MDB_CT_DOUBLE && !text_keys)
					op_stats_begin(&x, fkeys[0], lv, fkeys[1], rv);
				const int frc = mdb_dev_join_group_count(x.dev, lv, ln, nl_rows, rv, rn, nr_rows,
									  /* (a bare COUNT(*) has no group order to keep; nor has a GROUP BY when the database says so) */
									  ((only_count || cat->groups_any_order) ? 0u : MDB_ORDER_FIRST) | MDB_KEYS_MAY_ALIAS,
									  x.d_fused_key, x.d_count, NULL, cap, &G, &J);
				op_stats_end(&x);
				if (frc) {
					rc = dev_fail(&x, "join + group count");
					goto out;
				}
				alias_checked = true;
			}
		}
		if (alias_checked) {
			/* every left row turned out to be a group: the group keys ARE the left key column, in its order - the operator wrote none
			 * (MDB_KEYS_MAY_ALIAS) and the stream reads the column itself; a result kept on the device becomes one more holder of the
			 * table's buffer (0.8 GB less read and written at 10^8 rows) */
			struct mdb_dev_plan_info pi;
			if (mdb_dev_last_plan(x.dev, &pi) == 0 && pi.keys_are_left_column && G == nl_rows)
				x.d_fused_key = (int64_t *)(uintptr_t)lv;
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
		if (text_keys && G && !only_count) {	/* the group keys that arrived: common ids -> ids of this rank's dictionary */
			const int64_t *same = NULL;
			if ((rc = shard_ids(&x, x.d_fused_key, G, false, true, &same)))
				goto out;
		}
		x.fused = true;
		x.n = only_count ? J : G;	/* COUNT(*) without GROUP BY = the stream length = the joined rows */
		x.joined_rows = J;
	} else if (keys_only) {
		/* ---- a two-table equi-join whose select list names nothing but the two key columns (BASELINE configs[3]: SELECT * over
		 *      two key columns), any order allowed (mdb_database_groups_any_order): both sides hold the same value in every
		 *      joined row, so no row has to be identified - the any-order join + GROUP BY pipeline counts every key's partners and
		 *      the key is written COUNT times (mdb_dev_join_keys); the stream is the fused plan's: one key column, no row ids */
		const struct mdb_column *cl = &s->tabs[kj[0]->tbl_idx].t->cols[kj[0]->col_idx], *cr = &s->tabs[kj[1]->tbl_idx].t->cols[kj[1]->col_idx];
		int64_t *jk = keys_ordered;	/* (the reference's order: answered above) */
		uint64_t J = keys_ordered_rows;
		if (cat->groups_any_order) {
			op_stats_begin(&x, kj[0], cl->d_data, kj[1], cr->d_data);
			const int krc2 = mdb_dev_join_keys(x.dev, cl->d_data, cl->d_nullbits, s->tabs[kj[0]->tbl_idx].t->nrows, cr->d_data, cr->d_nullbits,
							   s->tabs[kj[1]->tbl_idx].t->nrows, &jk, &J);
			op_stats_end(&x);
			if (krc2) {
				rc = dev_fail(&x, "join of two key columns");
				goto out;
			}
		}
		if (jk && jk != keys_ordered && track(&x, jk)) {
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
			x.ws = &ws;
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
				op_stats_begin(&x, s->group[0], kv, NULL, NULL);
				const int krc = mdb_dev_group_count_keys(x.dev, kv, NULL, x.n, gk, gc, x.n, &Gk);
				op_stats_end(&x);
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
			/* (no more groups than key values in the column's range - catalog statistics -, plus the NULL group) */
			uint64_t gcap = x.n ? x.n : 1;
			if (s->group[0]->kind == MDB_EX_FIELD && s->group[0]->type != MDB_CT_DOUBLE) {
				const uint64_t b = op_groups_bound(&x, s->group[0], gcap);
				gcap = b + 1 < gcap ? b + 1 : gcap;
			}
			first = dalloc(&x, gcap * 4);
			x.d_count = dalloc(&x, gcap * 8);
			if (!first || !x.d_count) {
				rc = dev_fail(&x, "allocating group outputs");
				goto out;
			}
			if (s->group[0]->kind == MDB_EX_FIELD && s->group[0]->type != MDB_CT_DOUBLE)
				op_stats_begin(&x, s->group[0], kv, NULL, NULL);
			const int grc = x.n ? mdb_dev_group_count(x.dev, kv, kn, x.n, MDB_ORDER_FIRST, first, x.d_count, gcap, &G) : 0;
			op_stats_end(&x);
			if (grc) {
				rc = dev_fail(&x, "group count");
				goto out;
			}
			{
				/* the catalog knew that the column holds no value twice and the operator answered with the identity (group i = row i of the
				 * stream, COUNT 1): the stream stays what it is - no row-id vector to compose, no gather through one in the projection; a
				 * result kept on the device holds the table's columns */
				struct mdb_dev_plan_info pi;
				const bool identity = x.n && G == x.n && mdb_dev_last_plan(x.dev, &pi) == 0 && pi.group_form == 3;
				if (!identity && (rc = stream_select(&x, s->ntabs, first, G)))
					goto out;
			}
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
		res->colprec = calloc((size_t)(ncols ? ncols : 1), sizeof(int));
		res->data = calloc((size_t)(ncols ? ncols : 1), sizeof(int64_t *));
		res->nullbits = calloc((size_t)(ncols ? ncols : 1), sizeof(uint64_t *));
		if (!d_vals || !d_nulls || !res->colname || !res->coltype || !res->colprec || !res->data || !res->nullbits) {
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
			for (int i = 0; s->sel_alias && !s->select_all && i < s->nsel; i++)	/* `item AS name` names the column (the first alias of an item wins) */
				if (s->sel_alias[i][0] && (key_tbl[key] < 0 ? s->sel[i]->kind == MDB_EX_COUNT
									      : s->sel[i]->kind == MDB_EX_FIELD && s->sel[i]->tbl_idx == key_tbl[key] && s->sel[i]->col_idx == key_col[key])) {
					mdb_copy_name(res->colname[c], s->sel_alias[i]);
					break;
				}
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
			res->colprec[c] = col->type == MDB_CT_VARCHAR ? col->precision : 0;
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
				/* the same device column under two result columns (both key columns of SELECT * over an equi-join): ONE buffer of the
				 * result serves both - query_column_data_device() hands out read-only columns - instead of a copy each (0.3 ms per
				 * 10^8-row column) */
				for (int k = 0; k < c && !own; k++) {
					if (d_vals[k] == src && res->d_data[k])
						own = res->d_data[k];
					else if ((const void *)d_nulls[k] == src && res->d_nullbits[k])
						own = res->d_nullbits[k];
				}
				if (own) {
					if (pass)
						res->d_nullbits[c] = own;
					else
						res->d_data[c] = own;
					continue;
				}
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
				/* a base table's own device column, all of its rows in order (no row ids between): the result becomes one more holder
				 * of the buffer instead of copying it (0.04 ms per 10^7-row column, 0.3 per 10^8) - result columns are read-only, rows
				 * appended later lie behind the result's, and an UPDATE copies a column that has other holders before it writes */
				if (!own && !shared && mdb_dev_alloc_size(x.dev, src) >= bytes &&
				    !(mdb_knob("MDB_RESULT_ALIAS") && mdb_knob("MDB_RESULT_ALIAS")[0] == '0')) {
					bool stmt_buf = false;
					for (int i = 0; i < x.bufs.n; i++)
						stmt_buf = stmt_buf || x.bufs.p[i] == src;
					if (!stmt_buf && mdb_dev_retain(x.dev, src) == 0)
						own = (void *)src;
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
	mdb_result_legacy_header(res);		/* the reference's struct table in front of the columnar result (include/mdb_legacy.h) ... */
	mdb_result_legacy_rows(res);		/* ... and, for a small result whose columns are on the host, its rows as datablocks */
	res->exec_ms = now_ms() - t0;
	res->joined_rows = x.joined_rows;
	*out = res;
	res = NULL;
	rc = MIDORIDB_OK;
out:
	cat->joins_eliminated += (uint64_t)x.joins_eliminated;
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
