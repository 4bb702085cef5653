/*
 * oracle.h - TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's SELECT hot path.
 *
 * Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so, and only as the checker / the timed CPU
 * baseline - never as the thing measured or shipped.
 *
 * Pinning: cpu_naive.c is checked against the real reference (oracle/_ref, built from
 * /root/reference by `make ref`) on the reference's own golden queries and on randomised
 * in-domain cases (tests/test_oracle_pinning.py); cpu_hash.c is checked against cpu_naive.c.
 *
 * Columns are int64_t arrays with optional NULL flags (uint8_t, 1 = NULL; NULL pointer = none).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include <stddef.h>

/* ---- cpu_naive.c : the reference's algorithms, loop for loop -------------------------------- */

/* INNER JOIN ON l = r by nested loops, left-major / right-minor
 * (reference src/engine/executor_select.c:1096-1141; NULL never matches, :557-579).
 * out_l/out_r may be NULL to only count.  Returns the number of joined rows. */
uint64_t orc_naive_join_pairs(const int64_t *kl, const uint8_t *nl, uint64_t n_l, const int64_t *kr, const uint8_t *nr,
			      uint64_t n_r, uint32_t *out_l, uint32_t *out_r);

/* GROUP BY one field + COUNT(*): every live row deletes each later row with an equal key and
 * counts it (reference executor_select.c:1542-1583, 1465-1524); NULL keys are equal to each other
 * (:1477-1482).  Returns the number of groups; out_first/out_count sized n. */
uint64_t orc_naive_group_count(const int64_t *keys, const uint8_t *nulls, uint64_t n, uint32_t *out_first,
			       int64_t *out_count);

/* The north-star query exactly as the reference runs it: nested-loop join materialising the
 * joined key column, then the quadratic GROUP BY over it.  Returns groups; *joined = join size. */
uint64_t orc_naive_join_group_count(const int64_t *kl, const uint8_t *nl, uint64_t n_l, const int64_t *kr,
				    const uint8_t *nr, uint64_t n_r, int64_t *out_key, int64_t *out_count,
				    uint64_t *joined);

/* ---- cpu_hash.c : same results at any size (hash join / hash aggregation, pthreads) ----------- */

/* Returns 0 on success.  Outputs sized n_l.  Groups in first-occurrence order. */
int orc_hash_join_group_count(const int64_t *kl, const uint8_t *nl, uint64_t n_l, const int64_t *kr, const uint8_t *nr,
			      uint64_t n_r, int nthreads, int64_t *out_key, int64_t *out_count, uint32_t *out_first,
			      uint64_t *out_groups, uint64_t *out_joined);

#endif /* ORACLE_H */
