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
/* ---- strings in sharded mode.  A VARCHAR cell is the id of its string in THIS process's dictionary (mdb_store.c; upstream keeps a heap
 * pointer per cell, src/primitive/column.c:255-293): meaningless on another rank.  Before the first VARCHAR cell of a statement crosses
 * xGMI the ranks make their dictionaries known to each other: every rank announces the strings it has not announced yet (one
 * all-gather of bytes), and every rank interns ALL announcements, in rank order, into its own dictionary and into a second one, the
 * COMMON dictionary - built from the same strings in the same order everywhere, so its ids mean the same on every rank.  Cells travel
 * as common ids (one table lookup per cell on the device before the exchange, one after it): equal strings stay equal ids, joins,
 * GROUP BY and DISTINCT over VARCHAR columns stay INT64 work, and what a rank returns are ids of its own dictionary again. */
static int dict_tables_grow(struct mdb_catalog *cat, uint64_t nl, uint64_t ng)
{
	if (nl + 1 > cat->l2g_cap) {
		uint64_t c = cat->l2g_cap ? cat->l2g_cap : 1024;
		while (c < nl + 1)
			c *= 2;
		int64_t *p = realloc(cat->l2g, c * sizeof(*p));
		if (!p)
			return -MIDORIDB_NOMEM;
		memset(p + cat->l2g_cap, 0, (c - cat->l2g_cap) * sizeof(*p));
		cat->l2g = p;
		cat->l2g_cap = c;
	}
	if (ng + 1 > cat->g2l_cap) {
		uint64_t c = cat->g2l_cap ? cat->g2l_cap : 1024;
		while (c < ng + 1)
			c *= 2;
		int64_t *p = realloc(cat->g2l, c * sizeof(*p));
		if (!p)
			return -MIDORIDB_NOMEM;
		memset(p + cat->g2l_cap, 0, (c - cat->g2l_cap) * sizeof(*p));
		cat->g2l = p;
		cat->g2l_cap = c;
	}
	return MIDORIDB_OK;
}

/* collective: every rank of the statement gets here at the same point (whether a statement moves VARCHAR cells follows from its text and
 * the schema alone); once per statement */
int shard_dict_sync(struct exec *x)
{
	struct mdb_catalog *cat = x->cat;
	if (x->dict_synced)
		return MIDORIDB_OK;
	const int W = mdb_dist_world(cat->dist);
	int lrc = dict_tables_grow(cat, cat->dict.n, cat->gdict.n);
	/* this rank's announcements: [u32 length][bytes] per string it has not announced */
	size_t bytes = 0;
	for (uint64_t id = 1; id <= cat->dict.n && !lrc; id++)
		if (!cat->l2g[id])
			bytes += 4 + cat->dict.len[id - 1];
	char *mine = malloc(bytes ? bytes : 1);
	if (!mine)
		lrc = -MIDORIDB_NOMEM;
	size_t at = 0;
	for (uint64_t id = 1; id <= cat->dict.n && !lrc; id++)
		if (!cat->l2g[id]) {
			const uint32_t len = cat->dict.len[id - 1];
			memcpy(mine + at, &len, 4);
			memcpy(mine + at + 4, cat->dict.str[id - 1], len);
			at += 4 + len;
		}
	void *all = NULL;
	uint64_t counts[512];
	if (W > 512)
		lrc = -MIDORIDB_ERROR;
	/* (a rank in trouble still takes part, with nothing to say: its failure is its own, the collective completes) */
	const int rc = mdb_dist_allgather_bytes(cat->dist, mine, lrc ? 0 : at, &all, counts);
	free(mine);
	if (rc) {
		snprintf(x->err, x->errlen, "execution phase: dictionary exchange: %s\n", mdb_dist_last_error(cat->dist));
		return -MIDORIDB_INTERNAL;
	}
	if (!lrc) {
		const char *p = all;
		for (int r = 0; r < W && !lrc; r++) {
			const char *end = p + counts[r];
			while (p < end && !lrc) {
				uint32_t len;
				memcpy(&len, p, 4);
				const int64_t l = mdb_dict_intern(&cat->dict, p + 4, len), g = mdb_dict_intern(&cat->gdict, p + 4, len);
				if (!l || !g || dict_tables_grow(cat, (uint64_t)l, (uint64_t)g)) {
					lrc = -MIDORIDB_NOMEM;
					break;
				}
				cat->l2g[l] = g;
				cat->g2l[g] = l;
				p += 4 + len;
			}
		}
	}
	free(all);
	/* device copies of the two tables, when they grew */
	if (!lrc && (cat->d_l2g_n != cat->dict.n + 1 || cat->d_g2l_n != cat->gdict.n + 1)) {
		(void)mdb_dev_sync(x->dev);
		(void)mdb_dev_free(x->dev, cat->d_l2g);
		(void)mdb_dev_free(x->dev, cat->d_g2l);
		cat->d_l2g = cat->d_g2l = NULL;
		cat->d_l2g_n = cat->d_g2l_n = 0;
		void *a = NULL, *b = NULL;
		if (mdb_dev_alloc(x->dev, (cat->dict.n + 1) * 8, &a) || mdb_dev_alloc(x->dev, (cat->gdict.n + 1) * 8, &b) ||
		    mdb_dev_h2d(x->dev, a, cat->l2g, (cat->dict.n + 1) * 8) || mdb_dev_h2d(x->dev, b, cat->g2l, (cat->gdict.n + 1) * 8)) {
			(void)mdb_dev_free(x->dev, a);
			(void)mdb_dev_free(x->dev, b);
			lrc = -MIDORIDB_NOMEM;
		} else {
			cat->d_l2g = a;
			cat->d_g2l = b;
			cat->d_l2g_n = cat->dict.n + 1;
			cat->d_g2l_n = cat->gdict.n + 1;
		}
	}
	/* a rank-local failure (memory, interning, the upload) is every rank's: the peers would go on to the statement's next collective
	 * and wait there for a rank that has returned */
	uint64_t failed = lrc ? 1 : 0;
	if (mdb_dist_allreduce_sum_u64(cat->dist, &failed, 1)) {
		snprintf(x->err, x->errlen, "execution phase: dictionary exchange: %s\n", mdb_dist_last_error(cat->dist));
		return -MIDORIDB_INTERNAL;
	}
	if (failed) {
		snprintf(x->err, x->errlen, "execution phase: dictionary exchange: out of memory%s\n", lrc ? "" : " on another rank");
		return lrc ? lrc : -MIDORIDB_NOMEM;
	}
	x->dict_synced = true;
	return MIDORIDB_OK;
}

/* n VARCHAR cells as ids of the common dictionary (to_common) or of this rank's own again: *out = a statement buffer, or `cells` itself
 * when in_place */
int shard_ids(struct exec *x, const int64_t *cells, uint64_t n, bool to_common, bool in_place, const int64_t **out)
{
	int rc = shard_dict_sync(x);
	if (rc)
		return rc;
	int64_t *dst = in_place ? (int64_t *)cells : dalloc(x, (n ? n : 1) * 8);
	if (!dst)
		return dev_fail(x, "allocating translated string ids");
	if (mdb_dev_map_ids(x->dev, cells, n, to_common ? x->cat->d_l2g : x->cat->d_g2l, to_common ? x->cat->d_l2g_n : x->cat->d_g2l_n, dst))
		return dev_fail(x, "translating string ids");
	*out = dst;
	return MIDORIDB_OK;
}

int shard_rows(struct exec *x, const int *tabs, int nt, uint32_t *const *rid_of, uint64_t n, const int64_t *kv, const uint64_t *kn,
		      uint32_t flags, uint64_t *n_out)
{
	struct mdb_select *s = x->s;
	struct mdb_dist_col cols[MDB_DIST_SHUFFLE_MAX_COLS] = { { NULL, NULL, NULL } };
	int col_i[MDB_DIST_SHUFFLE_MAX_COLS], col_c[MDB_DIST_SHUFFLE_MAX_COLS], nc = 0;
	void *ov[MDB_DIST_SHUFFLE_MAX_COLS];
	uint64_t *on[MDB_DIST_SHUFFLE_MAX_COLS];
	bool is_text[MDB_DIST_SHUFFLE_MAX_COLS] = { false };
	uint64_t got = 0;

	for (int i = 0; i < nt; i++) {
		const struct mdb_table *tb = s->tabs[tabs[i]].t;
		for (int c = 0; c < tb->ncols; c++) {
			if (!x->need[tabs[i]][c])
				continue;
			if (nc == MDB_DIST_SHUFFLE_MAX_COLS) {
				snprintf(x->err, x->errlen, "execution phase: sharded mode: more than %d columns in one exchange\n", MDB_DIST_SHUFFLE_MAX_COLS);
				return -MIDORIDB_ERROR;
			}
			cols[nc].values = tb->cols[c].d_data;
			if (tb->cols[c].type == MDB_CT_VARCHAR) {
				/* the column's cells as ids of the ranks' common dictionary (the whole column: the rows that travel are read
				 * from it through their row ids).  Decided from the SCHEMA alone: a rank that holds no row of the table has no
				 * device column (mdb_table_sync_device skips empty tables) and must still take part in the dictionary exchange,
				 * a collective, and translate what arrives. */
				const int64_t *common = NULL;
				const uint64_t have = tb->cols[c].d_data ? (tb->device_only ? tb->dev_rows : tb->nrows) : 0;
				const int trc = shard_ids(x, tb->cols[c].d_data, have, true, false, &common);
				if (trc)
					return trc;
				cols[nc].values = common;
				is_text[nc] = true;
			}
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
	for (int k = 0; k < nc; k++)
		if (is_text[k] && got) {	/* what arrived: common ids -> ids of this rank's dictionary, where it stands */
			const int64_t *same = NULL;
			const int trc = shard_ids(x, ov[k], got, false, true, &same);
			if (trc)
				return trc;
		}
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
	if (f->type == MDB_CT_VARCHAR && (rc = shard_ids(x, kv, x->n, true, false, &kv)))	/* (placement by the string, not by one rank's id of it) */
		return rc;
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
		}	/* (any column type: what travels are the 8-byte cells) */
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
