/*
 * mdb_exec_tail.c - the clauses the reference parses, checks and never executes (midorisql.y:180-196, 203; semantic_select.c:1718-2035):
 * HAVING, DISTINCT, ORDER BY, LIMIT over the finished stream.  Split off mdb_exec.c in round 4.
 */
#include "mdb_exec_internal.h"

/* HAVING, DISTINCT, ORDER BY, LIMIT over the finished stream (after FROM / WHERE / GROUP BY) */
int select_tail(struct exec *x, int has_count)
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
