/*
 * mdb_gen.h - the synthetic benchmark data generator, shared verbatim by the device kernel
 * (k_gen_keys), the CPU oracle and the tests, so every party sees identical tables
 * (SplitMix64-seeded; "permutation" = affine bijection modulo a prime >= N with
 * cycle walking, so 10^8..10^9 keys can be produced on device without a host shuffle).
 *
 * All generated integers are < 2^31 for N <= 2^31, the range on which the reference's 32-bit
 * integer compares agree with 64-bit ones.
 */
#ifndef MDB_GEN_H
#define MDB_GEN_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MDB_HD __host__ __device__
#else
#define MDB_HD
#endif

typedef struct mdb_perm {
	uint64_t n;	/* domain [0, n) */
	uint64_t p;	/* prime >= n */
	uint64_t a;	/* multiplier in [1, min(p-1, 2^30)] */
	uint64_t b;	/* offset in [0, p) */
} mdb_perm;

static inline uint64_t mdb_splitmix64(uint64_t *state)
{
	uint64_t z = (*state += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

/* the i-th output (i = 0, 1, ...) of the SplitMix64 stream seeded with `seed`, without stepping through the stream */
MDB_HD static inline uint64_t mdb_splitmix64_at(uint64_t seed, uint64_t i)
{
	uint64_t z = seed + (i + 1) * 0x9e3779b97f4a7c15ULL;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

static inline int mdb_is_prime_u64(uint64_t x)
{
	if (x < 2)
		return 0;
	if (x % 2 == 0)
		return x == 2;
	for (uint64_t d = 3; d * d <= x; d += 2)
		if (x % d == 0)
			return 0;
	return 1;
}

/* domain n must be <= 2^33 so that a * x + b stays below 2^64 */
static inline mdb_perm mdb_perm_make(uint64_t n, uint64_t seed)
{
	mdb_perm pm;
	uint64_t s = seed, lim;
	pm.n = n;
	pm.p = n < 2 ? 2 : n;
	while (!mdb_is_prime_u64(pm.p))
		pm.p++;
	lim = pm.p - 1 < (1ull << 30) ? pm.p - 1 : (1ull << 30);
	pm.a = 1 + mdb_splitmix64(&s) % lim;
	pm.b = mdb_splitmix64(&s) % pm.p;
	return pm;
}

MDB_HD static inline uint64_t mdb_perm_apply(const mdb_perm *pm, uint64_t i)
{
	uint64_t x = i;
	do {
		x = (pm->a * x + pm->b) % pm->p;
	} while (x >= pm->n);
	return x;
}

#endif /* MDB_GEN_H */
