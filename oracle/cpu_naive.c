/*
 * cpu_naive.c - TEST INFRASTRUCTURE ONLY (oracle).
 *
 * Loop-for-loop C restatement of the reference's join and GROUP BY algorithms over plain
 * columnar arrays, keeping the reference's iteration order and its in-domain semantics:
 *
 *   orc_naive_join_pairs        <- _join_nested_loop_tbl2tbl(), reference
 *                                  src/engine/executor_select.c:1096-1141 (outer = left rows in
 *                                  table order, inner = right rows; ON l = r evaluated per pair;
 *                                  any NULL operand => false, :557-579)
 *   orc_naive_group_count       <- proc_groupby_clause() :1542-1583 with cmp_rows_col_mattbl()
 *                                  :1465-1499 and inc_count_cols() :1501-1524: row i survives,
 *                                  every later equal row j is tombstoned and count_i++
 *   orc_naive_join_group_count  <- the two chained, as executor_run_select_stmt() does (:1683-1707)
 *
 * Deliberate differences, all documented in SURVEY.md 8a: the quadratic GROUP BY here compares
 * row i with EVERY later live row (the reference restarts j at i+1 in every datablock - defect D1,
 * wrong beyond one 4 KiB block); keys are compared as full int64 (defect D5: reference truncates to
 * 32 bits; identical on [-2^31, 2^31)).  What is NOT restated is the per-pair malloc/strcmp
 * bookkeeping (:340-438), which computes nothing.
 *
 * This is also the "port" CPU baseline bench.py times when oracle/_ref is unavailable.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

uint64_t orc_naive_join_pairs(const int64_t *kl, const uint8_t *nl, uint64_t n_l, const int64_t *kr, const uint8_t *nr,
			      uint64_t n_r, uint32_t *out_l, uint32_t *out_r)
{
	uint64_t j = 0;
	for (uint64_t a = 0; a < n_l; a++) {			/* :1096-1106 */
		for (uint64_t b = 0; b < n_r; b++) {		/* :1108-1119 */
			int isnull = (nl && nl[a]) || (nr && nr[b]);
			if (!isnull && kl[a] == kr[b]) {	/* eval_row_cond -> cmp_fieldname_to_fieldname :673-765 */
				if (out_l) {
					out_l[j] = (uint32_t)a;
					out_r[j] = (uint32_t)b;
				}
				j++;				/* table_insert_row :1130 */
			}
		}
	}
	return j;
}

uint64_t orc_naive_group_count(const int64_t *keys, const uint8_t *nulls, uint64_t n, uint32_t *out_first,
			       int64_t *out_count)
{
	uint8_t *deleted = calloc(n ? n : 1, 1);
	int64_t *count = malloc(sizeof(int64_t) * (n ? n : 1));
	uint64_t g = 0;

	for (uint64_t i = 0; i < n; i++)
		count[i] = 1;					/* init_count_cols :324-338 */
	for (uint64_t i = 0; i < n; i++) {
		if (deleted[i])
			continue;				/* :1551-1552 */
		for (uint64_t j = i + 1; j < n; j++) {
			int n1, n2, eq;
			if (deleted[j])
				continue;			/* :1563-1564 */
			n1 = nulls && nulls[i];
			n2 = nulls && nulls[j];
			if (n1 && n2)
				eq = 1;				/* :1477-1478 */
			else if (n1 || n2)
				eq = 0;
			else
				eq = keys[i] == keys[j];	/* :1487-1488 */
			if (eq) {
				deleted[j] = 1;			/* table_delete_row :1568 */
				count[i]++;			/* inc_count_cols :1574 */
			}
		}
	}
	for (uint64_t i = 0; i < n; i++) {			/* table_vacuum: survivors keep their order */
		if (!deleted[i]) {
			out_first[g] = (uint32_t)i;
			out_count[g] = count[i];
			g++;
		}
	}
	free(deleted);
	free(count);
	return g;
}

uint64_t orc_naive_join_group_count(const int64_t *kl, const uint8_t *nl, uint64_t n_l, const int64_t *kr,
				    const uint8_t *nr, uint64_t n_r, int64_t *out_key, int64_t *out_count,
				    uint64_t *joined)
{
	uint64_t j = orc_naive_join_pairs(kl, nl, n_l, kr, nr, n_r, NULL, NULL);
	uint32_t *pl = malloc(sizeof(uint32_t) * (j ? j : 1)), *pr = malloc(sizeof(uint32_t) * (j ? j : 1));
	int64_t *jk = malloc(sizeof(int64_t) * (j ? j : 1));
	uint32_t *first = malloc(sizeof(uint32_t) * (j ? j : 1));
	int64_t *cnt = malloc(sizeof(int64_t) * (j ? j : 1));
	uint64_t g;

	orc_naive_join_pairs(kl, nl, n_l, kr, nr, n_r, pl, pr);
	for (uint64_t k = 0; k < j; k++)
		jk[k] = kl[pl[k]];		/* the early-materialised joined key column (never NULL) */
	g = orc_naive_group_count(jk, NULL, j, first, cnt);
	for (uint64_t k = 0; k < g; k++) {
		out_key[k] = jk[first[k]];
		out_count[k] = cnt[k];
	}
	*joined = j;
	free(pl);
	free(pr);
	free(jk);
	free(first);
	free(cnt);
	return g;
}
