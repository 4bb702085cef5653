/*
 * mdb_exec_shard.c - query_execute() in sharded mode (one process per GPU, include/mdb_dist.h): where a statement's tuple stream is
 * exchanged between the ranks, and the shadow tables that stand in for what arrived (reference shape: the same
 * executor_run_select_stmt(), src/engine/executor_select.c:1655-1744, over tables whose rows are spread over the ranks).
 * Split off mdb_exec.c in round 4.
 */
#include "mdb_exec_internal.h"

/* ------------------------------------------------------------------ sharded mode: row exchange
 *
 * One process per GPU, every process holds ITS rows of every table (include/mdb_dist.h).  The general plan stays what it
 * is - the reference's phases, executor_select.c:1655-1744 - and gains ONE step: before an operator that must see all the
 * rows of a key together (equi-join, GROUP BY, DISTINCT), the tuple stream is re-distributed so that every row lands on
 * the rank its key hashes to (mdb_dist_shuffle_rows: key + the columns the statement still reads), unless it already is
 * (x->part: the join keys tied together so far).  What arrives replaces the table for the rest of the statement. */

void mark_needed(struct exec *x, const struct mdb_expr *e)
{
	if (!e)
		return;
	if (e->kind == MDB_EX_FIELD && e->tbl_idx >= 0 && e->tbl_idx < MDB_MAX_TABS && e->col_idx >= 0 && e->col_idx < MDB_MAX_COLS)
		x->need[e->tbl_idx][e->col_idx] = true;
	for (int i = 0; i < e->nkids; i++)
		mark_needed(x, e->kids[i]);
}

void mark_needed_all(struct exec *x)
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

bool in_part(const struct exec *x, const struct mdb_expr *f)
{
	for (int i = 0; i < x->npart; i++)
		if (x->part[i]->tbl_idx == f->tbl_idx && x->part[i]->col_idx == f->col_idx)
			return true;
	return false;
}

/* Tables tabs[0..nt) share one tuple stream of n tuples, table tabs[i] read through rid_of[i] (NULL = identity); kv / kn is
 * the stream's partitioning key.  Afterwards each of those tables is a shadow over the rows this rank received, *n_out of
 * them, all read by identity.  Collective: every rank calls it for the same statement at the same point. */
int shard_rows(struct exec *x, const int *tabs, int nt, uint32_t *const *rid_of, uint64_t n, const int64_t *kv, const uint64_t *kn,
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
	if (flags & SHARD_BROADCAST) {
		/* the small side of a join without an equi-join key: every rank's rows to every rank (a statement that reads no column of
		 * the table - SELECT COUNT(*) FROM A, B - only needs their number) */
		if (nc == 0) {
			got = n;
			if (mdb_dist_allreduce_sum_u64(x->cat->dist, &got, 1)) {
				snprintf(x->err, x->errlen, "execution phase: %s\n", mdb_dist_last_error(x->cat->dist));
				return -MIDORIDB_INTERNAL;
			}
		} else if (mdb_dist_broadcast_rows(x->cat->dist, n, cols, nc, ov, on, &got)) {
			snprintf(x->err, x->errlen, "execution phase: sharded broadcast: %s\n", mdb_dist_last_error(x->cat->dist));
			return -MIDORIDB_INTERNAL;
		}
	} else if (mdb_dist_shuffle_rows(x->cat->dist, kv, kn, n, flags, cols, nc, ov, on, &got)) {
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
int shard_stream(struct exec *x, int nt, const struct mdb_expr *f, uint32_t flags)
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
int shard_promise_ranges(struct exec *x, const struct mdb_expr *fl, const struct mdb_expr *fr)
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

void shard_cleanup(struct exec *x)
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
