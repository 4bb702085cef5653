/*
 * cpu_hash.c - TEST INFRASTRUCTURE ONLY (oracle).
 *
 * Hash-join + hash-aggregation CPU implementation of the north-star query with the reference's
 * (intended) semantics at ANY size: the reference's own nested loops (cpu_naive.c; reference
 * src/engine/executor_select.c:1096-1141, 1542-1583) are O(nA*nB) and cannot go past a few
 * thousand rows.  Checked against cpu_naive.c / the real reference on small inputs in
 * tests/test_oracle_pinning.py; used as the large-N checker and as the multi-threaded
 * "reasonable CPU" yardstick in bench.py.
 *
 *   result: for every distinct non-NULL key present on both sides (NULL never joins, :557-579),
 *   COUNT(*) = cntL * cntR (every pair is a joined row, :1096-1141), groups ordered by the first
 *   left row holding the key (first-occurrence order of the survivors, :1542-1583).
 *
 * Parallelisation: both key columns are radix-partitioned by a mixed hash into 1024 partitions
 * (stable, so the first inserted left row of a key is its first occurrence); worker threads own
 * whole partitions (no shared hash table, no atomics on the data path).
 */
#include "oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NPART 1024u
#define PSHIFT 54

static inline uint64_t mix64(uint64_t k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ULL;
	k ^= k >> 33;
	return k;
}

struct job {
	const int64_t *kl, *kr;
	const uint8_t *nl, *nr;
	uint64_t n_l, n_r;
	int nthreads;
	/* partitioned copies */
	int64_t *pl_key, *pr_key;
	uint32_t *pl_idx;
	uint64_t *hist_l, *hist_r;	/* [nthreads][NPART] -> write cursors */
	uint64_t off_l[NPART + 1], off_r[NPART + 1];
	int64_t *dense;			/* [n_l] COUNT(*) at the first left row of each group */
	uint64_t joined[256];
	int next_part;
	pthread_mutex_t lock;
	pthread_barrier_t bar;
};

struct targ {
	struct job *j;
	int tid;
};

static void slice(uint64_t n, int t, int nt, uint64_t *b, uint64_t *e)
{
	*b = n * (uint64_t)t / (uint64_t)nt;
	*e = n * (uint64_t)(t + 1) / (uint64_t)nt;
}

struct slot {
	int64_t key;
	uint32_t cl, cr, first, used;
};

static void *worker(void *argp)
{
	struct targ *a = argp;
	struct job *j = a->j;
	const int t = a->tid, nt = j->nthreads;
	uint64_t b, e;
	uint64_t *hl = j->hist_l + (size_t)t * NPART, *hr = j->hist_r + (size_t)t * NPART;

	/* phase 1: per-thread histograms over contiguous slices */
	memset(hl, 0, sizeof(uint64_t) * NPART);
	memset(hr, 0, sizeof(uint64_t) * NPART);
	slice(j->n_l, t, nt, &b, &e);
	for (uint64_t i = b; i < e; i++)
		if (!(j->nl && j->nl[i]))
			hl[mix64((uint64_t)j->kl[i]) >> PSHIFT]++;
	slice(j->n_r, t, nt, &b, &e);
	for (uint64_t i = b; i < e; i++)
		if (!(j->nr && j->nr[i]))
			hr[mix64((uint64_t)j->kr[i]) >> PSHIFT]++;
	pthread_barrier_wait(&j->bar);

	/* phase 2: thread 0 turns histograms into write cursors (partition-major, thread-minor => stable) */
	if (t == 0) {
		uint64_t run_l = 0, run_r = 0;
		for (uint32_t p = 0; p < NPART; p++) {
			j->off_l[p] = run_l;
			j->off_r[p] = run_r;
			for (int k = 0; k < nt; k++) {
				uint64_t c = j->hist_l[(size_t)k * NPART + p];
				j->hist_l[(size_t)k * NPART + p] = run_l;
				run_l += c;
				c = j->hist_r[(size_t)k * NPART + p];
				j->hist_r[(size_t)k * NPART + p] = run_r;
				run_r += c;
			}
		}
		j->off_l[NPART] = run_l;
		j->off_r[NPART] = run_r;
	}
	pthread_barrier_wait(&j->bar);

	/* phase 3: scatter */
	slice(j->n_l, t, nt, &b, &e);
	for (uint64_t i = b; i < e; i++)
		if (!(j->nl && j->nl[i])) {
			uint64_t pos = hl[mix64((uint64_t)j->kl[i]) >> PSHIFT]++;
			j->pl_key[pos] = j->kl[i];
			j->pl_idx[pos] = (uint32_t)i;
		}
	slice(j->n_r, t, nt, &b, &e);
	for (uint64_t i = b; i < e; i++)
		if (!(j->nr && j->nr[i]))
			j->pr_key[hr[mix64((uint64_t)j->kr[i]) >> PSHIFT]++] = j->kr[i];
	pthread_barrier_wait(&j->bar);

	/* phase 4: per-partition build (left) / probe (right) / emit */
	uint64_t joined = 0;
	struct slot *tab = NULL;
	uint64_t tabcap = 0;
	for (;;) {
		int p;
		pthread_mutex_lock(&j->lock);
		p = j->next_part++;
		pthread_mutex_unlock(&j->lock);
		if (p >= (int)NPART)
			break;
		const uint64_t l0 = j->off_l[p], l1 = j->off_l[p + 1], r0 = j->off_r[p], r1 = j->off_r[p + 1];
		if (l0 == l1 || r0 == r1)
			continue;
		uint64_t cap = 16;
		while (cap < 2 * (l1 - l0))
			cap <<= 1;
		if (cap > tabcap) {
			free(tab);
			tab = malloc(sizeof(struct slot) * cap);
			tabcap = cap;
		}
		memset(tab, 0, sizeof(struct slot) * cap);
		for (uint64_t i = l0; i < l1; i++) {
			const int64_t k = j->pl_key[i];
			uint64_t s = (mix64((uint64_t)k) * 0x9E3779B97F4A7C15ULL) & (cap - 1);
			while (tab[s].used && tab[s].key != k)
				s = (s + 1) & (cap - 1);
			if (!tab[s].used) {
				tab[s].used = 1;
				tab[s].key = k;
				tab[s].first = j->pl_idx[i];	/* stable partition => first insertion = first occurrence */
			}
			tab[s].cl++;
		}
		for (uint64_t i = r0; i < r1; i++) {
			const int64_t k = j->pr_key[i];
			uint64_t s = (mix64((uint64_t)k) * 0x9E3779B97F4A7C15ULL) & (cap - 1);
			while (tab[s].used && tab[s].key != k)
				s = (s + 1) & (cap - 1);
			if (tab[s].used)
				tab[s].cr++;
		}
		for (uint64_t s = 0; s < cap; s++)
			if (tab[s].used && tab[s].cr) {
				const int64_t c = (int64_t)tab[s].cl * (int64_t)tab[s].cr;
				j->dense[tab[s].first] = c;
				joined += (uint64_t)c;
			}
	}
	free(tab);
	j->joined[t] = joined;
	return NULL;
}

int orc_hash_join_group_count(const int64_t *kl, const uint8_t *nl, uint64_t n_l, const int64_t *kr, const uint8_t *nr,
			      uint64_t n_r, int nthreads, int64_t *out_key, int64_t *out_count, uint32_t *out_first,
			      uint64_t *out_groups, uint64_t *out_joined)
{
	struct job j;
	pthread_t th[256];
	struct targ ta[256];
	uint64_t g = 0, joined = 0;

	if (nthreads < 1)
		nthreads = 1;
	if (nthreads > 256)
		nthreads = 256;
	memset(&j, 0, sizeof(j));
	j.kl = kl;
	j.kr = kr;
	j.nl = nl;
	j.nr = nr;
	j.n_l = n_l;
	j.n_r = n_r;
	j.nthreads = nthreads;
	j.pl_key = malloc(sizeof(int64_t) * (n_l ? n_l : 1));
	j.pl_idx = malloc(sizeof(uint32_t) * (n_l ? n_l : 1));
	j.pr_key = malloc(sizeof(int64_t) * (n_r ? n_r : 1));
	j.hist_l = malloc(sizeof(uint64_t) * NPART * (size_t)nthreads);
	j.hist_r = malloc(sizeof(uint64_t) * NPART * (size_t)nthreads);
	j.dense = calloc(n_l ? n_l : 1, sizeof(int64_t));
	if (!j.pl_key || !j.pl_idx || !j.pr_key || !j.hist_l || !j.hist_r || !j.dense)
		return -1;
	pthread_mutex_init(&j.lock, NULL);
	pthread_barrier_init(&j.bar, NULL, (unsigned)nthreads);
	for (int t = 0; t < nthreads; t++) {
		ta[t].j = &j;
		ta[t].tid = t;
		pthread_create(&th[t], NULL, worker, &ta[t]);
	}
	for (int t = 0; t < nthreads; t++) {
		pthread_join(th[t], NULL);
		joined += j.joined[t];
	}
	/* groups in first-occurrence order */
	for (uint64_t i = 0; i < n_l; i++)
		if (j.dense[i]) {
			out_key[g] = kl[i];
			out_count[g] = j.dense[i];
			if (out_first)
				out_first[g] = (uint32_t)i;
			g++;
		}
	*out_groups = g;
	*out_joined = joined;
	pthread_barrier_destroy(&j.bar);
	pthread_mutex_destroy(&j.lock);
	free(j.pl_key);
	free(j.pl_idx);
	free(j.pr_key);
	free(j.hist_l);
	free(j.hist_r);
	free(j.dense);
	return 0;
}
