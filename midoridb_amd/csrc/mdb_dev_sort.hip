/*
 * mdb_dev_sort.hip - ORDER BY on the device: a stable permutation that sorts a tuple stream by up to
 * MDB_SORT_MAX_KEYS columns (ASC / DESC, INT64 or DOUBLE, NULL = smallest value).
 *
 * The reference parses ORDER BY but never executes it; this operator is the
 * device half of the extension.  Two methods: when the value ranges of the columns and the stream length fit one
 * 64-bit word together, the packed path further down (sort_perm_packed); otherwise, and whenever the packed path
 * reports unevenly spread values, the general method: least-significant-digit radix sort, last ORDER BY column first -
 * every pass is stable (mdb_sort_pass: histogram, scan, ballot-ranked scatter), so ties of a later column
 * keep the order the earlier passes produced and ties of all columns keep the stream order.  Per column:
 *
 *   k_sort_load     u[k] = order-preserving 64-bit image of the column value of stream row perm[k]
 *                   (INT64: sign bit flipped; DOUBLE: IEEE total-order trick; DESC: complemented),
 *                   block-reduced min / max / any-NULL
 *   passes          only over the bits in which min and max differ (8-bit digits): small domains cost one
 *                   or two passes instead of eight
 *   NULL pass       one extra 1-bit pass when the column holds NULLs (first for ASC, last for DESC)
 *
 * HBM traffic per 8-bit pass and row: 8 B (histogram read) + 12 B read + 12 B written.
 *
 * Round 6: the packed word's column ranges come from a jittered sample of 2^17 rows first (sort_pack_ranges; the packing kernel checks
 * every row, measured ranges - all columns in one pass, k_sort_ranges - when one lies outside); and GROUP BY / SELECT DISTINCT over several
 * columns do not sort at all where the columns' ranges fit 63 bits together (group_multi_packed): the composite value is a key for
 * mdb_dev_group_count's forms - built from the columns by the band sort (18 ... 25 bits, mdb_dev_bandgroup.hip) or the per-workgroup LDS
 * tables (at most 14 bits, mdb_dev_groupby.hip) as they load them, written as an 8-byte column otherwise.
 */
#include "mdb_dev_internal.h"
#include "mdb_dev_rowjoin.h"	/* (mdb_group_count_banded: the band sort reads the columns of a composite key itself) */

#define SORT_THREADS 256

__device__ static inline uint64_t sort_image(uint64_t bits, int type, int desc)
{
	uint64_t u;
	if (type == MDB_T_DOUBLE)
		u = (bits >> 63) ? ~bits : (bits ^ 0x8000000000000000ull);
	else
		u = bits ^ 0x8000000000000000ull;
	return desc ? ~u : u;
}

/* mm[0] = min image, mm[1] = max image, mm[2] = 1 when a NULL was seen (all over non-NULL rows) */
__global__ __launch_bounds__(SORT_THREADS) void k_sort_load(const uint64_t *__restrict__ values, const uint64_t *__restrict__ nullbits,
							     const uint32_t *__restrict__ rid, const uint32_t *__restrict__ perm, uint64_t n,
							     int type, int desc, uint64_t *__restrict__ u_out, unsigned long long *mm)
{
	__shared__ unsigned long long s_min, s_max;
	__shared__ uint32_t s_null;
	if (threadIdx.x == 0) {
		s_min = ~0ull;
		s_max = 0ull;
		s_null = 0;
	}
	__syncthreads();
	unsigned long long lo = ~0ull, hi = 0ull;
	bool anynull = false;
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS) {
		const uint32_t p = perm ? perm[k] : (uint32_t)k;
		const uint64_t row = rid ? (uint64_t)rid[p] : (uint64_t)p;
		uint64_t u = 0;
		if (nullbits && mdb_bit_is_set(nullbits, row)) {
			anynull = true;
		} else {
			u = sort_image(values[row], type, desc);
			lo = u < lo ? u : lo;
			hi = u > hi ? u : hi;
		}
		if (u_out)
			u_out[k] = u;
	}
	if (lo <= hi) {
		atomicMin(&s_min, lo);
		atomicMax(&s_max, hi);
	}
	if (anynull)
		atomicOr(&s_null, 1u);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_min <= s_max) {
			atomicMin(&mm[0], s_min);
			atomicMax(&mm[1], s_max);
		}
		if (s_null)
			atomicOr(&mm[2], 1ull);
	}
}

/* key of the NULL pass: ASC 0 = NULL, 1 = value; DESC 0 = value, 1 = NULL */
__global__ __launch_bounds__(SORT_THREADS) void k_sort_nullflag(const uint64_t *__restrict__ nullbits, const uint32_t *__restrict__ rid,
								 const uint32_t *__restrict__ perm, uint64_t n, int desc,
								 uint64_t *__restrict__ u_out)
{
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS) {
		const uint32_t p = perm[k];
		const uint64_t row = rid ? (uint64_t)rid[p] : (uint64_t)p;
		const bool isnull = mdb_bit_is_set(nullbits, row);
		u_out[k] = (uint64_t)(isnull == (desc != 0));
	}
}

/* ---- ORDER BY columns (up to 4) whose value ranges and the stream length together fit one word ---------------------
 *
 * word = [ per column: NULL flag | value image - min ] [ stream position ] (left-aligned): the
 * words are unique, so ANY full sort of them is the stable sort of the column.  That lifts the restriction to stable
 * least-significant-digit passes (ballot ranking: issue-bound, one 24-byte-per-row pass per 8 bits): the words are
 * partitioned by their TOP bits with the histogram-free two-level scatter of the join (8 bytes per row and level),
 * and every leaf - at most SORT_LEAF_CAP words - is finished in LDS (k_sort_leaf).  Value distributions that
 * overflow a fixed-capacity region (the top bits are the values themselves, not a hash) are reported by the scatter
 * kernels and sent to the general path. */
#define SORT_LEAF_CAP 4096u
#define SORT_LEAF_THREADS 512
#define SORT_BUCKETS 1024u
#define SORT_BUCKET_MAX 1024u
#define SORT_PACK_MIN_ROWS (1u << 18)

#define SORT_PACK_MAX_KEYS 4
struct sort_pack_args {
	struct mdb_sort_key key[SORT_PACK_MAX_KEYS];	/* first = most significant */
	uint64_t lo[SORT_PACK_MAX_KEYS];		/* smallest image of the column */
	uint32_t kb[SORT_PACK_MAX_KEYS];		/* bits of (largest image - lo) */
	int nkeys;
	uint32_t rb, up;				/* bits of a stream position; left shift that aligns the word */
	int nopos;					/* the composite value alone (group_multi_packed: a key column, not a sort word) */
	uint64_t span[SORT_PACK_MAX_KEYS];		/* largest image - lo the field holds (ranges from a SAMPLE: k_sort_pack checks every row) */
};

/* outside (or NULL): where ranges come from a sample, *outside becomes 1 when an image does not lie in [lo, lo + span] of its column - the
 * words mean nothing then and the caller packs again with measured ranges */
__global__ __launch_bounds__(SORT_THREADS) void k_sort_pack(sort_pack_args a, uint64_t n, uint64_t *__restrict__ w, uint32_t *outside)
{
	bool out = false;
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS) {
		uint64_t v = 0;
		for (int c = 0; c < a.nkeys; c++) {
			const struct mdb_sort_key &key = a.key[c];
			const uint64_t row = key.rid ? (uint64_t)key.rid[k] : k;
			const uint64_t *values = (const uint64_t *)key.values;
			if (key.nullbits) {
				const bool isnull = mdb_bit_is_set(key.nullbits, row);
				/* ASC: NULLs first (flag 0), DESC: NULLs last (flag 1) - as in the general path */
				const uint64_t flag = (uint64_t)(isnull == (key.desc != 0));
				const uint64_t d = isnull ? 0ull : sort_image(values[row], key.type, key.desc) - a.lo[c];
				out = out || d > a.span[c];
				v = (v << (a.kb[c] + 1)) | (flag << a.kb[c]) | d;
			} else {
				const uint64_t d = sort_image(values[row], key.type, key.desc) - a.lo[c];
				out = out || d > a.span[c];
				v = (v << a.kb[c]) | d;
			}
		}
		w[k] = a.nopos ? v : ((v << a.rb) | k) << a.up;
	}
	if (outside && __any(out) && mdb_lane() == 0)
		*outside = 1u;
}

/* the image ranges of all columns of a packed word in ONE pass and one host round trip (a pass and a round trip per column before):
 * mm[3 c] = smallest image, mm[3 c + 1] = largest, over the non-NULL rows of column c (min > max: none) */
__global__ __launch_bounds__(SORT_THREADS) void k_sort_ranges(sort_pack_args a, uint64_t n /* rows looked at */, uint64_t step /* ... every step-th of the stream */,
							       unsigned long long *mm)
{
	__shared__ unsigned long long s_min[SORT_PACK_MAX_KEYS], s_max[SORT_PACK_MAX_KEYS];
	if (threadIdx.x < SORT_PACK_MAX_KEYS) {
		s_min[threadIdx.x] = ~0ull;
		s_max[threadIdx.x] = 0ull;
	}
	__syncthreads();
	unsigned long long lo[SORT_PACK_MAX_KEYS], hi[SORT_PACK_MAX_KEYS];
#pragma unroll
	for (int c = 0; c < SORT_PACK_MAX_KEYS; c++) {
		lo[c] = ~0ull;
		hi[c] = 0ull;
	}
	const uint64_t stride = (uint64_t)gridDim.x * SORT_THREADS;
	for (uint64_t k0 = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k0 < n; k0 += 2 * stride) {	/* (two rows a turn: 2 x nkeys loads in flight) */
		uint64_t u[2][SORT_PACK_MAX_KEYS];
		bool have[2][SORT_PACK_MAX_KEYS];
		uint64_t at[2];
#pragma unroll
		for (int r = 0; r < 2; r++) {
			const uint64_t k = k0 + (uint64_t)r * stride;
			/* (a sample: a row inside the k-th stretch of `step` rows, not its first - a column that repeats with a period the step
			 * shares a factor with would show every second or fourth of its values only) */
			at[r] = k * step + (step > 1 ? ((k * 0x9E3779B97F4A7C15ull) >> 32) % step : 0ull);
#pragma unroll
			for (int c = 0; c < SORT_PACK_MAX_KEYS; c++) {
				have[r][c] = false;
				u[r][c] = 0;
				if (c < a.nkeys && k < n) {
					const struct mdb_sort_key &key = a.key[c];
					const uint64_t row = key.rid ? (uint64_t)key.rid[at[r]] : at[r];
					if (!(key.nullbits && mdb_bit_is_set(key.nullbits, row))) {
						have[r][c] = true;
						u[r][c] = ((const uint64_t *)key.values)[row];
					}
				}
			}
		}
#pragma unroll
		for (int r = 0; r < 2; r++)
#pragma unroll
			for (int c = 0; c < SORT_PACK_MAX_KEYS; c++)
				if (have[r][c]) {
					const uint64_t im = sort_image(u[r][c], a.key[c].type, a.key[c].desc);
					lo[c] = im < lo[c] ? im : lo[c];
					hi[c] = im > hi[c] ? im : hi[c];
				}
	}
#pragma unroll
	for (int c = 0; c < SORT_PACK_MAX_KEYS; c++) {
		if (c >= a.nkeys)
			break;
#pragma unroll
		for (int o = 32; o; o >>= 1) {
			const unsigned long long ol = __shfl_xor(lo[c], o, MDB_WAVE), oh = __shfl_xor(hi[c], o, MDB_WAVE);
			lo[c] = ol < lo[c] ? ol : lo[c];
			hi[c] = oh > hi[c] ? oh : hi[c];
		}
		if (mdb_lane() == 0 && lo[c] <= hi[c]) {
			atomicMin(&s_min[c], lo[c]);
			atomicMax(&s_max[c], hi[c]);
		}
	}
	__syncthreads();
	if (threadIdx.x < (uint32_t)a.nkeys && s_min[threadIdx.x] <= s_max[threadIdx.x]) {
		atomicMin(&mm[3 * threadIdx.x], s_min[threadIdx.x]);
		atomicMax(&mm[3 * threadIdx.x + 1], s_max[threadIdx.x]);
	}
}

/* the columns' image ranges into pa.lo / pa.kb (one launch, one synchronisation); *total += the bits of the composite value.
 * 0 = done, 1 = a column type the packed word does not take, < 0 = error */
#define SORT_RANGE_SAMPLE_MIN ((uint64_t)1 << 22)	/* rows from which the ranges are first taken from a sample */
#define SORT_RANGE_SAMPLE_ROWS ((uint64_t)1 << 17)
static bool sort_ranges_sampled(uint64_t n)	/* MDB_SORT_RANGE_SAMPLE: 0 never, 2 from 2^18 rows on (tests) */
{
	const char *knob = mdb_knob("MDB_SORT_RANGE_SAMPLE");
	if (knob && knob[0] == '0')
		return false;
	return n >= ((knob && knob[0] == '2') ? SORT_RANGE_SAMPLE_ROWS * 2 : SORT_RANGE_SAMPLE_MIN);
}
static int sort_pack_ranges(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, unsigned long long *mm /* 12 words */,
			    sort_pack_args *pa, uint32_t *total, uint32_t limit, uint64_t *vmax, bool sampled = false)
{
	/* sampled: every (n / 2^17)-th row is looked at and the span found is widened by a 1024th on either side (a column of evenly spread
	 * values shows its extremes to within span / 2^17; a column of few values shows them exactly and is not widened at all - widening by
	 * an eighth left the ends of the word's range empty and the sort's fixed-capacity regions overflowed: 2.9 -> 13.7 ms through the general
	 * path); k_sort_pack then checks every row against [lo, lo + span] and the caller comes back with sampled = false when one lies
	 * outside.  0.37 ms per 10^8 rows and two columns less. */
	const uint64_t step = sampled ? n / SORT_RANGE_SAMPLE_ROWS : 1, looked = sampled ? SORT_RANGE_SAMPLE_ROWS : n;
	uint64_t *h = ctx->h_pinned;
	for (int c = 0; c < nkeys; c++) {
		if (keys[c].type != MDB_T_INT64 && keys[c].type != MDB_T_DOUBLE)
			return 1;	/* the general path reports it */
		pa->key[c] = keys[c];
		h[3 * c] = ~0ull;
		h[3 * c + 1] = 0ull;
		h[3 * c + 2] = 0ull;
	}
	pa->nkeys = nkeys;
	const uint32_t grid = (uint32_t)(((looked + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((looked + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	MDB_HIP(ctx, hipMemcpyAsync(mm, h, 24 * (size_t)nkeys, hipMemcpyHostToDevice, ctx->stream));
	MDB_LAUNCH(ctx, sampled ? "orderby_range_sample" : "orderby_range", k_sort_ranges, grid ? grid : 1, SORT_THREADS, *pa, looked, step, mm);
	MDB_HIP(ctx, hipMemcpyAsync(h, mm, 24 * (size_t)nkeys, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (int c = 0; c < nkeys; c++) {
		uint64_t lo = h[3 * c] <= h[3 * c + 1] ? h[3 * c] : 0, hi = h[3 * c] <= h[3 * c + 1] ? h[3 * c + 1] : 0;
		if (sampled) {
			const uint64_t pad = (hi - lo) >> 10;
			lo = lo > pad ? lo - pad : 0;
			hi = hi < ~0ull - pad ? hi + pad : ~0ull;
		}
		uint32_t kb = 0;
		if (hi != lo)
			kb = 64u - (uint32_t)__builtin_clzll(hi - lo);
		const uint32_t width = kb + (keys[c].nullbits ? 1u : 0u);	/* the flag bit is spent whenever the column can hold NULLs */
		*total += width;
		if (*total > limit)
			return 1;
		pa->lo[c] = lo;
		pa->kb[c] = kb;
		pa->span[c] = hi - lo;
		if (vmax)
			*vmax = (*vmax << width) | ((keys[c].nullbits ? (1ull << kb) : 0ull) | (hi - lo));
	}
	return 0;
}

__global__ __launch_bounds__(SORT_LEAF_THREADS) void k_sort_leaf(const uint64_t *__restrict__ w, const uint32_t *__restrict__ cnt,
								  const uint32_t *__restrict__ out_base, uint32_t cap, uint32_t up, uint32_t rmask,
								  uint32_t bshift, uint32_t *status, uint32_t *__restrict__ perm_out,
								  uint64_t *__restrict__ vkey_out, uint32_t rb, uint32_t *__restrict__ hi32_out)
{
	/* The words of a leaf share their top bits.  They are dealt into SORT_BUCKETS buckets by the next bits (counting
	 * with LDS atomics - the order of arrival does not matter, the words are unique), every bucket - one or two words
	 * on average - is put in order by counting ranks, and the leaf is written out.  ~60 KiB of LDS traffic per leaf where
	 * a bitonic network over the 4096 words moved 1.6 MiB (2.5 ms per 10^8 rows, LDS-bandwidth bound).  A bucket longer
	 * than SORT_BUCKET_MAX (values bunched in their low bits) is reported and the caller takes the general path. */
	__shared__ uint64_t s_w[SORT_LEAF_CAP];
	__shared__ uint32_t s_cnt[SORT_BUCKETS + 1];	/* bucket sizes, then bucket starts (+ the total): 37 KiB of LDS, 4 workgroups per CU */
	__shared__ uint32_t s_tmp[32];
	const uint32_t leaf = blockIdx.x;
	uint32_t c = cnt[leaf];
	c = c < cap ? c : cap;		/* an overflowing region has been flagged by the scatter: the result is discarded */
	if (c == 0)
		return;
	for (uint32_t b = threadIdx.x; b < SORT_BUCKETS; b += SORT_LEAF_THREADS)
		s_cnt[b] = 0;
	const uint64_t *src = w + (uint64_t)leaf * cap;
	uint64_t v[SORT_LEAF_CAP / SORT_LEAF_THREADS];
	uint32_t rank[SORT_LEAF_CAP / SORT_LEAF_THREADS];
#pragma unroll
	for (int e = 0; e < (int)(SORT_LEAF_CAP / SORT_LEAF_THREADS); e++) {
		const uint32_t i = threadIdx.x + (uint32_t)e * SORT_LEAF_THREADS;
		v[e] = i < c ? src[i] : 0ull;
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < (int)(SORT_LEAF_CAP / SORT_LEAF_THREADS); e++) {
		const uint32_t i = threadIdx.x + (uint32_t)e * SORT_LEAF_THREADS;
		rank[e] = i < c ? atomicAdd(&s_cnt[(uint32_t)(v[e] >> bshift) & (SORT_BUCKETS - 1)], 1u) : 0u;
	}
	__syncthreads();
	{
		/* SORT_BUCKETS / SORT_LEAF_THREADS consecutive buckets per thread */
		uint32_t mine[SORT_BUCKETS / SORT_LEAF_THREADS], sum = 0;
#pragma unroll
		for (int q = 0; q < (int)(SORT_BUCKETS / SORT_LEAF_THREADS); q++) {
			mine[q] = s_cnt[threadIdx.x * (SORT_BUCKETS / SORT_LEAF_THREADS) + q];
			sum += mine[q];
		}
		uint32_t total;
		uint32_t run = mdb_block_excl_scan(sum, s_tmp, &total);	/* (its barriers: every size has been read) */
#pragma unroll
		for (int q = 0; q < (int)(SORT_BUCKETS / SORT_LEAF_THREADS); q++) {
			s_cnt[threadIdx.x * (SORT_BUCKETS / SORT_LEAF_THREADS) + q] = run;
			run += mine[q];
		}
		if (threadIdx.x == 0)
			s_cnt[SORT_BUCKETS] = total;
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < (int)(SORT_LEAF_CAP / SORT_LEAF_THREADS); e++) {
		const uint32_t i = threadIdx.x + (uint32_t)e * SORT_LEAF_THREADS;
		if (i < c)
			s_w[s_cnt[(uint32_t)(v[e] >> bshift) & (SORT_BUCKETS - 1)] + rank[e]] = v[e];
	}
	__syncthreads();
	/* order inside the buckets: every word finds its rank among the words of its bucket by counting the smaller ones
	 * (all threads busy, no dependent chain; one thread sorting a whole bucket by insertion took 5x as long when the
	 * buckets hold 32 words each - keys with 16 duplicates), then moves to its final place */
	uint32_t fin[SORT_LEAF_CAP / SORT_LEAF_THREADS];
#pragma unroll
	for (int e = 0; e < (int)(SORT_LEAF_CAP / SORT_LEAF_THREADS); e++) {
		const uint32_t i = threadIdx.x + (uint32_t)e * SORT_LEAF_THREADS;
		fin[e] = 0;
		if (i < c) {
			const uint32_t b = (uint32_t)(v[e] >> bshift) & (SORT_BUCKETS - 1);
			const uint32_t st = s_cnt[b], m = s_cnt[b + 1] - st;
			uint32_t r = 0;
			if (m > SORT_BUCKET_MAX) {
				mdb_raise(status, 256u);
			} else {
				for (uint32_t x = 0; x < m; x++)
					r += s_w[st + x] < v[e];
			}
			fin[e] = st + r;
		}
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < (int)(SORT_LEAF_CAP / SORT_LEAF_THREADS); e++) {
		const uint32_t i = threadIdx.x + (uint32_t)e * SORT_LEAF_THREADS;
		if (i < c)
			s_w[fin[e]] = v[e];
	}
	__syncthreads();
	const uint32_t base = out_base[leaf];
	for (uint32_t i = threadIdx.x; i < c; i += SORT_LEAF_THREADS) {
		const uint64_t w = s_w[i] >> up;
		perm_out[base + i] = (uint32_t)w & rmask;
		if (vkey_out)
			vkey_out[base + i] = w >> rb;	/* the composite value: equal for rows that agree on every column */
		if (hi32_out)
			hi32_out[base + i] = (uint32_t)(w >> rb);
	}
}

static void sort_packed_bits(uint64_t n, int *b1, int *b2)
{
	int b = 2;
	while (b < 2 * MDB_MAX_RADIX_BITS && ((uint64_t)2048 << b) < n)
		b++;
	*b1 = (b + 1) / 2;
	*b2 = b - *b1;
}

static size_t sort_packed_arena_bytes(uint64_t n)
{
	if (n < SORT_PACK_MIN_ROWS)
		return 0;
	int b1, b2;
	sort_packed_bits(n, &b1, &b2);
	const size_t leaves = (size_t)1 << (b1 + b2);
	return mdb_partition_raw_arena_bytes(n, b1, b2, SORT_LEAF_CAP, true, 0) + mdb_align_up((leaves + 1) * 4) +
	       mdb_align_up(mdb_scan_scratch_words(leaves + 1) * 4) + mdb_align_up(n * 4) + mdb_align_up(n * 8) + 4096;
}

/* 0 = *perm holds the result, 1 = not applicable (range too wide, too few rows, skewed values): use the general path */
static int sort_perm_packed(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint64_t *u, unsigned long long *mm,
			    uint32_t **perm, uint64_t **vkey)
{
	if (n < SORT_PACK_MIN_ROWS || nkeys > SORT_PACK_MAX_KEYS)
		return 1;
	const uint32_t grid = (uint32_t)(((n + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((n + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	uint64_t *h = ctx->h_pinned;
	sort_pack_args pa;
	memset(&pa, 0, sizeof(pa));
	pa.nkeys = nkeys;
	pa.rb = 1;
	while (pa.rb < 32 && (1ull << pa.rb) < n)
		pa.rb++;
	uint32_t total = 0, up = 0, digits0 = 0;
	const uint32_t rb = pa.rb;
	int b1, b2;
	sort_packed_bits(n, &b1, &b2);
	/* the columns' ranges: from a sample first (sort_pack_ranges), checked by the packing kernel row by row; measured when a row lies outside */
	for (bool sampled = sort_ranges_sampled(n);; sampled = false) {
		uint64_t vmax = 0;	/* the largest composite value: every field at its maximum */
		total = rb;
		const int rrc = sort_pack_ranges(ctx, keys, nkeys, n, mm, &pa, &total, 64, &vmax, sampled);
		if (rrc == 1 && sampled)
			continue;	/* (the widened ranges do not fit the word: the measured ones may) */
		if (rrc)
			return rrc;
		pa.up = up = 64 - total;
		/* first-level digits that can occur: the largest word's top bits */
		const uint64_t wmax = ((vmax << rb) | (n - 1)) << up;
		digits0 = (uint32_t)(wmax >> (64 - b1)) + 1u;
		uint32_t *outside = reinterpret_cast<uint32_t *>(mm + 12);
		MDB_HIP(ctx, hipMemsetAsync(outside, 0, 8, ctx->stream));
		MDB_LAUNCH(ctx, "orderby_pack", k_sort_pack, grid, SORT_THREADS, pa, n, u, sampled ? outside : (uint32_t *)NULL);
		if (!sampled)
			break;
		MDB_HIP(ctx, hipMemcpyAsync(&h[12], outside, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (!(uint32_t)h[12])
			break;
	}
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	mdb_part_result ps;
	int rc = mdb_partition_raw(ctx, u, n, b1, b2, SORT_LEAF_CAP, true, digits0, &ps, false);	/* no gaps: the zero word is row 0 of the smallest value */
	if (rc)
		return rc;
	if (!ps.leaf_cap)
		return 1;	/* the fixed-capacity layout does not apply at this size */
	uint32_t *obase = (uint32_t *)mdb_arena_take(ctx, ((size_t)ps.nleaves + 1) * 4);
	uint32_t *otmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words((uint64_t)ps.nleaves + 1) * 4);
	uint32_t *out = (uint32_t *)mdb_arena_take(ctx, n * 4);
	uint64_t *vk = vkey ? (uint64_t *)mdb_arena_take(ctx, n * 8) : NULL;	/* sorted composite values, for callers that look for runs */
	if (!obase || !otmp || !out || (vkey && !vk))
		return -MIDORIDB_INTERNAL;
	if (ps.nleaves <= MDB_SCAN_FROM_MAX) {
		rc = mdb_scan_u32_small_from(ctx, ps.leaf_cnt, ps.nleaves, obase);
	} else {
		MDB_HIP(ctx, hipMemcpyAsync(obase, ps.leaf_cnt, (size_t)ps.nleaves * 4, hipMemcpyDeviceToDevice, ctx->stream));
		MDB_HIP(ctx, hipMemsetAsync(obase + ps.nleaves, 0, 4, ctx->stream));
		rc = mdb_scan_u32_inplace(ctx, obase, (uint64_t)ps.nleaves + 1, otmp);
	}
	if (rc)
		return rc;
	/* buckets inside a leaf: the 10 bits below the partition bits (the word is left-aligned; `up` >= 1 unused low bits) */
	const uint32_t bshift = 64u - (uint32_t)(b1 + b2) - 10u;
	MDB_LAUNCH(ctx, "orderby_leaf", k_sort_leaf, ps.nleaves, SORT_LEAF_THREADS, (const uint64_t *)ps.hv, (const uint32_t *)ps.leaf_cnt,
		   (const uint32_t *)obase, ps.leaf_cap, up, (uint32_t)((1ull << rb) - 1ull), bshift, ctx->d_status, out, vk, rb, (uint32_t *)NULL);
	MDB_HIP(ctx, hipMemcpyAsync(&h[8], ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if ((uint32_t)h[8] & (2u | 256u))
		return 1;	/* a region or a bucket overflowed: the values are too unevenly spread for this path */
	*perm = out;
	if (vkey)
		*vkey = vk;
	return 0;
}

/* ---- (a, b) pairs of 32-bit ids, unique as pairs, into (a, b) order ------------------------------------------------
 * word = a << bits(b) | b: the same top-bit partition + leaf sort; the sorted words ARE the pairs, nothing is gathered.
 * Used by the join whose unique keys are on the left (mdb_dev_join.hip): its pairs arrive in right-row order. */
__global__ __launch_bounds__(SORT_THREADS) void k_pairs_pack(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n, uint32_t rb,
							      uint32_t up, uint64_t *__restrict__ w)
{
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS)
		w[k] = (((uint64_t)a[k] << rb) | b[k]) << up;
}

/* 0 = done, 1 = the fixed-capacity layout did not hold (skewed a): caller's fallback, < 0 = error.  na / nb: a < na, b < nb. */
int mdb_sort_pairs(mdb_dev_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t n, uint64_t na, uint64_t nb, uint32_t *out_a,
		   uint32_t *out_b)
{
	if (n < SORT_PACK_MIN_ROWS || n >= 0xFFFFFFFFull)
		return 1;
	uint32_t ab = 1, rb = 1;
	while (ab < 32 && (1ull << ab) < na)
		ab++;
	while (rb < 32 && (1ull << rb) < nb)
		rb++;
	const uint32_t total = ab + rb, up = 64 - total;
	int b1, b2;
	sort_packed_bits(n, &b1, &b2);
	const size_t leaves = (size_t)1 << (b1 + b2);
	int rc = mdb_arena_begin(ctx, mdb_align_up(n * 8) + mdb_partition_raw_arena_bytes(n, b1, b2, SORT_LEAF_CAP, true, 0) +
					      mdb_align_up((leaves + 1) * 4) + mdb_align_up(mdb_scan_scratch_words(leaves + 1) * 4) + 8192);
	if (rc)
		return rc;
	uint64_t *w = (uint64_t *)mdb_arena_take(ctx, n * 8);
	if (!w)
		return -MIDORIDB_INTERNAL;
	const uint32_t grid = (uint32_t)(((n + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((n + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	const uint64_t wmax = (((na - 1) << rb) | (nb - 1)) << up;
	const uint32_t digits0 = (uint32_t)(wmax >> (64 - b1)) + 1u;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	MDB_LAUNCH(ctx, "pairs_pack", k_pairs_pack, grid, SORT_THREADS, a, b, n, rb, up, w);
	mdb_part_result ps;
	rc = mdb_partition_raw(ctx, w, n, b1, b2, SORT_LEAF_CAP, true, digits0, &ps, false);
	if (rc)
		return rc;
	if (!ps.leaf_cap)
		return 1;
	uint32_t *obase = (uint32_t *)mdb_arena_take(ctx, ((size_t)ps.nleaves + 1) * 4);
	uint32_t *otmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words((uint64_t)ps.nleaves + 1) * 4);
	if (!obase || !otmp)
		return -MIDORIDB_INTERNAL;
	if (ps.nleaves <= MDB_SCAN_FROM_MAX) {
		rc = mdb_scan_u32_small_from(ctx, ps.leaf_cnt, ps.nleaves, obase);
	} else {
		MDB_HIP(ctx, hipMemcpyAsync(obase, ps.leaf_cnt, (size_t)ps.nleaves * 4, hipMemcpyDeviceToDevice, ctx->stream));
		MDB_HIP(ctx, hipMemsetAsync(obase + ps.nleaves, 0, 4, ctx->stream));
		rc = mdb_scan_u32_inplace(ctx, obase, (uint64_t)ps.nleaves + 1, otmp);
	}
	if (rc)
		return rc;
	const uint32_t bshift = 64u - (uint32_t)(b1 + b2) - 10u;
	MDB_LAUNCH(ctx, "pairs_leaf", k_sort_leaf, ps.nleaves, SORT_LEAF_THREADS, (const uint64_t *)ps.hv, (const uint32_t *)ps.leaf_cnt,
		   (const uint32_t *)obase, ps.leaf_cap, up, (uint32_t)((1ull << rb) - 1ull), bshift, ctx->d_status, out_b, (uint64_t *)NULL, rb, out_a);
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(&h[8], ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return ((uint32_t)h[8] & (2u | 256u)) ? 1 : 0;
}

/* ---- tiny inputs: one workgroup ranks every row by counting ---------------------------------------------------------
 * Up to SORT_TINY_ROWS rows and SORT_PACK_MAX_KEYS columns: the (NULL flag, image) pairs of every column sit in LDS, row
 * i's place is the number of rows that sort before it (ties: the smaller stream position) - n^2 comparisons, all lanes
 * reading the same row j at the same time (LDS broadcast), one launch instead of the ~10 per column of the radix passes.
 * One workgroup per 64 rows to rank (each loads all rows): 2048 rows took 1.16 ms on a single workgroup. */
#define SORT_TINY_ROWS 2048u
#define SORT_TINY_THREADS 1024

struct sort_tiny_args {
	struct mdb_sort_key key[SORT_PACK_MAX_KEYS];
	int nkeys;
	uint32_t n;
};

__global__ __launch_bounds__(SORT_TINY_THREADS) void k_sort_tiny(sort_tiny_args a, uint32_t *__restrict__ perm_out)
{
	__shared__ uint64_t s_img[SORT_PACK_MAX_KEYS][SORT_TINY_ROWS];
	__shared__ uint8_t s_flag[SORT_PACK_MAX_KEYS][SORT_TINY_ROWS];
	for (int c = 0; c < a.nkeys; c++) {
		const struct mdb_sort_key &key = a.key[c];
		for (uint32_t k = threadIdx.x; k < a.n; k += SORT_TINY_THREADS) {
			const uint64_t row = key.rid ? (uint64_t)key.rid[k] : k;
			const bool isnull = key.nullbits && mdb_bit_is_set(key.nullbits, row);
			/* ASC: NULLs first (flag 0), DESC: NULLs last (flag 1) - as in the other paths */
			s_flag[c][k] = (uint8_t)(isnull == (key.desc != 0));
			s_img[c][k] = isnull ? 0ull : sort_image(((const uint64_t *)key.values)[row], key.type, key.desc);
		}
	}
	/* workgroup b ranks rows [64 b, 64 b + 64): lane = row i, wave w counts over the rows j = w, w + 16, ... (every lane of a
	 * wave reads the same row j: LDS broadcast), the 16 partial counts meet in LDS */
	__shared__ uint32_t s_rank[MDB_WAVE];
	if (threadIdx.x < MDB_WAVE)
		s_rank[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * MDB_WAVE + mdb_lane(), wave = threadIdx.x / MDB_WAVE;
	if (i < a.n) {
		uint32_t rank = 0;
		for (uint32_t j = wave; j < a.n; j += SORT_TINY_THREADS / MDB_WAVE) {
			int cmp = 0;	/* -1: j before i, +1: i before j */
			for (int c = 0; c < a.nkeys && cmp == 0; c++) {
				const uint8_t fj = s_flag[c][j], fi = s_flag[c][i];
				const uint64_t vj = s_img[c][j], vi = s_img[c][i];
				cmp = fj != fi ? (fj < fi ? -1 : 1) : (vj != vi ? (vj < vi ? -1 : 1) : 0);
			}
			rank += cmp < 0 || (cmp == 0 && j < i);
		}
		atomicAdd(&s_rank[mdb_lane()], rank);
	}
	__syncthreads();
	if (wave == 0 && i < a.n)
		perm_out[s_rank[mdb_lane()]] = i;
}

static size_t sort_arena_bytes(uint64_t n)
{
	const size_t hist_words = mdb_sort_pass_hist_words(n);
	return 2 * mdb_align_up(n * 8) + 2 * mdb_align_up(n * 4) + mdb_align_up(hist_words * 4) +
	       mdb_align_up(mdb_scan_scratch_words(hist_words) * 4) + mdb_align_up(128) + 4096 + sort_packed_arena_bytes(n);
}

/* sorts inside an arena the caller has begun (sort_arena_bytes(n) available); *perm = the arena buffer that holds
 * the final permutation.  Synchronises per key (the digit range comes back from the device). */
/* vkey (optional): receives the sorted composite values when the packed path ran (rows equal in every column have equal
 * values), NULL otherwise */
static int sort_perm_impl(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t **perm, uint64_t **vkey = NULL)
{
	if (vkey)
		*vkey = NULL;
	const size_t hist_words = mdb_sort_pass_hist_words(n);
	uint64_t *u[2] = { (uint64_t *)mdb_arena_take(ctx, n * 8), (uint64_t *)mdb_arena_take(ctx, n * 8) };
	uint32_t *pm[2] = { (uint32_t *)mdb_arena_take(ctx, n * 4), (uint32_t *)mdb_arena_take(ctx, n * 4) };
	uint32_t *hist = (uint32_t *)mdb_arena_take(ctx, hist_words * 4);
	uint32_t *scan_tmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words(hist_words) * 4);
	unsigned long long *mm = (unsigned long long *)mdb_arena_take(ctx, 128);
	if (!u[0] || !u[1] || !pm[0] || !pm[1] || !hist || !scan_tmp || !mm)
		return -MIDORIDB_INTERNAL;
	if (n <= SORT_TINY_ROWS && nkeys <= SORT_PACK_MAX_KEYS) {
		sort_tiny_args ta;
		memset(&ta, 0, sizeof(ta));
		bool types_ok = true;
		for (int c = 0; c < nkeys; c++) {
			ta.key[c] = keys[c];
			types_ok = types_ok && (keys[c].type == MDB_T_INT64 || keys[c].type == MDB_T_DOUBLE);
		}
		if (types_ok) {		/* (an unknown type is reported by the general path below) */
			ta.nkeys = nkeys;
			ta.n = (uint32_t)n;
			MDB_LAUNCH(ctx, "orderby_tiny", k_sort_tiny, (ta.n + MDB_WAVE - 1) / MDB_WAVE, SORT_TINY_THREADS, ta, pm[0]);
			*perm = pm[0];
			return MIDORIDB_OK;
		}
	}
	{
		const int prc = sort_perm_packed(ctx, keys, nkeys, n, u[0], mm, perm, vkey);
		if (prc <= 0)
			return prc;
		if (vkey)
			*vkey = NULL;
	}
	int rc = mdb_dev_iota32(ctx, pm[0], n);
	if (rc)
		return rc;
	int cur = 0;	/* (u[cur], pm[cur]) hold the current order */
	const uint32_t grid = (uint32_t)(((n + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((n + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	uint64_t *h = ctx->h_pinned;
	for (int j = nkeys - 1; j >= 0; j--) {
		const struct mdb_sort_key *key = &keys[j];
		if (key->type != MDB_T_INT64 && key->type != MDB_T_DOUBLE)
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "sort: key %d has an unknown type", j);
		h[0] = ~0ull;
		h[1] = 0ull;
		h[2] = 0ull;
		MDB_HIP(ctx, hipMemcpyAsync(mm, h, 24, hipMemcpyHostToDevice, ctx->stream));
		MDB_LAUNCH(ctx, "orderby_load", k_sort_load, grid, SORT_THREADS, (const uint64_t *)key->values, key->nullbits, key->rid,
			   (const uint32_t *)pm[cur], n, key->type, key->desc, u[cur], mm);
		MDB_HIP(ctx, hipMemcpyAsync(h, mm, 24, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint64_t lo = h[0], hi = h[1];
		const bool anynull = h[2] != 0;
		uint32_t bits = 0;
		if (lo <= hi && (lo ^ hi))
			bits = 64u - (uint32_t)__builtin_clzll(lo ^ hi);
		for (uint32_t shift = 0; shift < bits; shift += 8) {
			const uint32_t w = bits - shift < 8 ? bits - shift : 8;
			rc = mdb_sort_pass(ctx, u[cur], pm[cur], n, shift, w, u[cur ^ 1], pm[cur ^ 1], hist, scan_tmp);
			if (rc)
				return rc;
			cur ^= 1;
		}
		if (anynull) {
			MDB_LAUNCH(ctx, "orderby_nullflag", k_sort_nullflag, grid, SORT_THREADS, key->nullbits, key->rid, (const uint32_t *)pm[cur], n,
				   key->desc, u[cur]);
			rc = mdb_sort_pass(ctx, u[cur], pm[cur], n, 0, 1, u[cur ^ 1], pm[cur ^ 1], hist, scan_tmp);
			if (rc)
				return rc;
			cur ^= 1;
		}
	}
	*perm = pm[cur];
	return MIDORIDB_OK;
}

static int sort_check(mdb_dev_ctx *ctx, const char *what, int nkeys, uint64_t n)
{
	if (nkeys < 1 || nkeys > MDB_SORT_MAX_KEYS)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "%s: between 1 and %d keys", what, MDB_SORT_MAX_KEYS);
	if (n >= 0xFFFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "%s: too many rows", what);
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_sort_perm(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *perm_out)
{
	int rc = sort_check(ctx, "sort_perm", nkeys, n);
	if (rc || n == 0)
		return rc;
	rc = mdb_arena_begin(ctx, sort_arena_bytes(n));
	if (rc)
		return rc;
	uint32_t *perm = NULL;
	rc = sort_perm_impl(ctx, keys, nkeys, n, &perm);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemcpyAsync(perm_out, perm, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ ORDER BY ... LIMIT k: the first k entries only
 *
 * Sorting 10^7 rows to keep 10 moves every row through the radix passes.  Instead: a strided sample of the first sort
 * key is ordered (one workgroup), the sample element of a rank that leaves - with a wide margin - at least k rows at or
 * before it becomes a threshold, ONE filter pass keeps the rows at or before the threshold in stream order (ties on the
 * first key all included: whatever the further keys say, a row beyond the threshold has k rows before it), and only those
 * candidates go through the stable sort with all keys.  A threshold that kept fewer than k rows (unrepresentative
 * sample) or most of the table falls back to the full sort: the result is the same prefix of the same permutation. */
#define TOPK_SAMPLE 2048u

__global__ __launch_bounds__(256) void k_topk_sample(struct mdb_sort_key key, uint64_t stride, uint32_t s, uint64_t *__restrict__ vals,
						     uint64_t *__restrict__ nullwords)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	bool isnull = false;
	if (i < s) {
		const uint64_t pos = (uint64_t)i * stride;
		const uint64_t row = key.rid ? (uint64_t)key.rid[pos] : pos;
		isnull = key.nullbits && mdb_bit_is_set(key.nullbits, row);
		vals[i] = ((const uint64_t *)key.values)[row];
	}
	const uint64_t m = __ballot(isnull);
	if (mdb_lane() == 0 && i < s)
		nullwords[i >> 6] = m;
}

__global__ void k_topk_pick(const uint32_t *__restrict__ perm, uint32_t rank, const uint64_t *__restrict__ vals,
			    const uint64_t *__restrict__ nullwords, uint64_t *__restrict__ out)
{
	const uint32_t p = perm[rank];
	out[0] = vals[p];
	out[1] = (nullwords[p >> 6] >> (p & 63)) & 1u;
}

/* *sorted: rows that went through a sort at the innermost level */
static int topk_rec(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint64_t k, uint32_t *perm_out, uint64_t *sorted)
{
	int rc = MIDORIDB_OK;
	*sorted = n;
	uint32_t *full = NULL, *sel = NULL, *perm2 = NULL, *rids[MDB_SORT_MAX_KEYS] = { NULL };
	uint64_t *vals = NULL;
	uint64_t m = 0;
	bool done = false;
	/* rank of the threshold in the ordered sample: twice the expected rank of the k-th row plus a margin of 3 sample steps */
	const uint64_t stride = n / TOPK_SAMPLE;
	const uint64_t rank = stride ? 2 * ((k + stride - 1) / stride) + 3 : TOPK_SAMPLE;
	if (n >= 8 * (uint64_t)TOPK_SAMPLE && rank < TOPK_SAMPLE / 4) {
		const uint32_t s = TOPK_SAMPLE;
		if ((rc = mdb_cached_alloc(ctx, (size_t)s * 8 + s / 8 + s * 4 + 16, (void **)&vals)))
			return rc;
		uint64_t *nullwords = vals + s;
		uint32_t *sperm = (uint32_t *)(nullwords + s / 64);
		uint64_t *picked = (uint64_t *)(sperm + s);
		MDB_LAUNCH(ctx, "topk_sample", k_topk_sample, s / 256, 256, keys[0], stride, s, vals, nullwords);
		struct mdb_sort_key sk = keys[0];
		sk.values = vals;
		sk.nullbits = nullwords;
		sk.rid = NULL;
		rc = mdb_dev_sort_perm(ctx, &sk, 1, s, sperm);
		if (rc)
			goto out;
		MDB_LAUNCH(ctx, "topk_pick", k_topk_pick, 1, 1, sperm, (uint32_t)rank, vals, nullwords, picked);
		uint64_t h[2];
		if (hipMemcpyAsync(h, picked, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
			rc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "topk_perm: reading the threshold failed");
			goto out;
		}
		/* the rows at or before the threshold, in the order of the FIRST key (NULL first ascending, last descending) */
		struct mdb_pred_insn prog[5];
		int np = 0;
		const bool thr_null = h[1] != 0, is_double = keys[0].type == MDB_T_DOUBLE;
		double thr_d;
		memcpy(&thr_d, &h[0], 8);
		bool usable = !(is_double && thr_d != thr_d);
		memset(prog, 0, sizeof(prog));
		if (!keys[0].desc) {
			if (keys[0].nullbits) {
				prog[np].op = MDB_P_ISNULL;
				np++;
			}
			if (!thr_null) {
				prog[np].op = MDB_P_CMP_COL_CONST;
				prog[np].cmp = MDB_CMP_LE;
				prog[np].type = keys[0].type;
				prog[np].imm = (int64_t)h[0];
				np++;
				if (np == 2)
					prog[np++].op = MDB_P_OR;
			}
		} else if (thr_null) {
			usable = false;		/* descending and the threshold is a NULL: every row is at or before it */
		} else {
			prog[np].op = MDB_P_CMP_COL_CONST;
			prog[np].cmp = MDB_CMP_GE;
			prog[np].type = keys[0].type;
			prog[np].imm = (int64_t)h[0];
			np++;
		}
		if (usable && np && is_double) {
			/* NaN rows are candidates whatever the threshold: mdb_dev_sort_perm orders DOUBLE keys by their bits (NaNs with
			 * the sign bit first, the others last), the IEEE comparisons above are false for them - without this a NaN that
			 * sorts first would be missing from the prefix.  col <> col is true exactly for NaN */
			prog[np].op = MDB_P_CMP_COL_COL;
			prog[np].cmp = MDB_CMP_NE;
			prog[np].type = MDB_T_DOUBLE;
			prog[np].a = 0;
			prog[np].b = 0;
			np++;
			prog[np++].op = MDB_P_OR;
		}
		if (usable && np) {
			struct mdb_col_binding cb = { keys[0].values, keys[0].nullbits, keys[0].rid };
			if ((rc = mdb_cached_alloc(ctx, n * 4, (void **)&sel)))
				goto out;
			if ((rc = mdb_dev_filter(ctx, prog, np, &cb, 1, n, sel, &m)))
				goto out;
			if (m >= k && m <= n / 2) {
				/* the candidates through the full stable sort: key columns read through row ids composed with `sel` */
				struct mdb_sort_key k2[MDB_SORT_MAX_KEYS];
				for (int i = 0; i < nkeys; i++) {
					k2[i] = keys[i];
					if (!keys[i].rid) {
						k2[i].rid = sel;
						continue;
					}
					int j = 0;
					while (j < i && keys[j].rid != keys[i].rid)
						j++;
					if (j < i) {
						k2[i].rid = k2[j].rid;
						continue;
					}
					if ((rc = mdb_cached_alloc(ctx, m * 4, (void **)&rids[i])))
						goto out;
					if ((rc = mdb_dev_gather32(ctx, keys[i].rid, sel, m, rids[i])))
						goto out;
					k2[i].rid = rids[i];
				}
				/* (the candidates of 10^8 rows are still 2.5 * 10^5: the same again over them leaves a few hundred) */
				if ((rc = mdb_cached_alloc(ctx, k * 4, (void **)&perm2)))
					goto out;
				if ((rc = topk_rec(ctx, k2, nkeys, m, k, perm2, sorted)))
					goto out;
				if ((rc = mdb_dev_gather32(ctx, sel, perm2, k, perm_out)))
					goto out;
				done = true;
			}
		}
	}
	if (!done) {
		if ((rc = mdb_cached_alloc(ctx, n * 4, (void **)&full)))
			goto out;
		if ((rc = mdb_dev_sort_perm(ctx, keys, nkeys, n, full)))
			goto out;
		if (hipMemcpyAsync(perm_out, full, k * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
		    hipStreamSynchronize(ctx->stream) != hipSuccess)
			rc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "topk_perm: copying the prefix failed");
	}
out:
	for (int i = 0; i < nkeys; i++)
		if (rids[i])
			(void)mdb_cached_free(ctx, rids[i]);
	if (perm2)
		(void)mdb_cached_free(ctx, perm2);
	if (sel)
		(void)mdb_cached_free(ctx, sel);
	if (full)
		(void)mdb_cached_free(ctx, full);
	if (vals)
		(void)mdb_cached_free(ctx, vals);
	return rc;
}

extern "C" int mdb_dev_topk_perm(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint64_t k, uint32_t *perm_out,
				 uint64_t *out_candidates)
{
	int rc = sort_check(ctx, "topk_perm", nkeys, n);
	if (out_candidates)
		*out_candidates = n;
	if (rc || n == 0 || k == 0)
		return rc;
	uint64_t sorted = n;
	rc = topk_rec(ctx, keys, nkeys, n, k < n ? k : n, perm_out, &sorted);
	if (out_candidates)
		*out_candidates = sorted;
	return rc;
}

/* ------------------------------------------------------------------ DISTINCT
 *
 * Rows equal in every key column (NULL = NULL, DOUBLE by bits) are adjacent after the stable sort and the first
 * row of every run is the row's first occurrence in the stream: mark it in a dense flag array, compact the flags
 * to ascending positions (the filter's bitmap -> positions kernels). */
struct distinct_args {
	struct mdb_sort_key key[MDB_SORT_MAX_KEYS];
	int nkeys;
};

__global__ __launch_bounds__(SORT_THREADS) void k_distinct_heads(distinct_args a, const uint32_t *__restrict__ perm, uint64_t n,
								  int64_t *__restrict__ flags)
{
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS) {
		const uint32_t p = perm[k];
		bool head = k == 0;
		if (!head) {
			const uint32_t q = perm[k - 1];
			for (int c = 0; c < a.nkeys && !head; c++) {
				const struct mdb_sort_key &key = a.key[c];
				const uint64_t rp = key.rid ? (uint64_t)key.rid[p] : (uint64_t)p;
				const uint64_t rq = key.rid ? (uint64_t)key.rid[q] : (uint64_t)q;
				const bool np = key.nullbits && mdb_bit_is_set(key.nullbits, rp);
				const bool nq = key.nullbits && mdb_bit_is_set(key.nullbits, rq);
				if (np != nq)
					head = true;
				else if (!np && ((const uint64_t *)key.values)[rp] != ((const uint64_t *)key.values)[rq])
					head = true;
			}
		}
		if (head)
			flags[p] = 1;
	}
}

size_t mdb_filter_arena_bytes(uint64_t n);
int mdb_filter_nonzero64(mdb_dev_ctx *ctx, const int64_t *vals, uint64_t n, uint32_t *out_sel, uint32_t **d_total);

__global__ __launch_bounds__(SORT_THREADS) void k_distinct_heads_vkey(const uint64_t *__restrict__ vkey, const uint32_t *__restrict__ perm, uint64_t n,
								       int64_t *__restrict__ flags)
{
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS)
		if (k == 0 || vkey[k] != vkey[k - 1])
			flags[perm[k]] = 1;
}

static int group_multi_packed(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *out_first,
			      int64_t *out_count, uint64_t cap, uint64_t *out_groups);

extern "C" int mdb_dev_distinct_sel(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *out_sel,
				    uint64_t *out_count)
{
	*out_count = 0;
	int rc = sort_check(ctx, "distinct_sel", nkeys, n);
	if (rc || n == 0)
		return rc;
	{
		/* the first rows of the groups of the columns' composite value (group_multi_packed below; the COUNTs go to a scratch column) */
		void *cnt = NULL;
		if (mdb_dev_alloc(ctx, n * 8, &cnt) == MIDORIDB_OK) {
			uint64_t G = 0;
			rc = group_multi_packed(ctx, keys, nkeys, n, out_sel, (int64_t *)cnt, n, &G);
			mdb_dev_free(ctx, cnt);
			if (rc <= 0) {
				*out_count = rc ? 0 : G;
				return rc;
			}
		}
	}
	rc = mdb_arena_begin(ctx, sort_arena_bytes(n) + mdb_align_up(n * 8) + mdb_filter_arena_bytes(n) + 4096);
	if (rc)
		return rc;
	uint32_t *perm = NULL;
	uint64_t *vkey = NULL;
	rc = sort_perm_impl(ctx, keys, nkeys, n, &perm, &vkey);
	if (rc)
		return rc;
	int64_t *flags = (int64_t *)mdb_arena_take(ctx, n * 8);
	if (!flags)
		return -MIDORIDB_INTERNAL;
	MDB_HIP(ctx, hipMemsetAsync(flags, 0, n * 8, ctx->stream));
	distinct_args a;
	memset(&a, 0, sizeof(a));
	for (int c = 0; c < nkeys; c++)
		a.key[c] = keys[c];
	a.nkeys = nkeys;
	const uint32_t grid = (uint32_t)(((n + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((n + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	if (vkey) {
		MDB_LAUNCH(ctx, "distinct_heads", k_distinct_heads_vkey, grid, SORT_THREADS, (const uint64_t *)vkey, (const uint32_t *)perm, n, flags);
	} else {
		MDB_LAUNCH(ctx, "distinct_heads", k_distinct_heads, grid, SORT_THREADS, a, (const uint32_t *)perm, n, flags);
	}
	uint32_t *d_total = NULL;
	rc = mdb_filter_nonzero64(ctx, flags, n, out_sel, &d_total);
	if (rc)
		return rc;
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h, d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_count = (uint32_t)h[0];
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ GROUP BY several columns + COUNT(*)
 *
 * The reference runs its single-field grouping loop once per GROUP BY field, one after the other
 * (executor_select.c:1537-1541), which does not group by the combination of the fields; here rows form a group
 * when they agree on every key column.  Stable sort by all columns -> equal rows are adjacent and the first row of
 * a run is the group's first occurrence -> run lengths are the counts -> the (first row, count) records go through
 * the same ordering path as the single-column operator, so groups come out in first-occurrence order.
 */
__global__ __launch_bounds__(SORT_THREADS) void k_group_heads(distinct_args a, const uint32_t *__restrict__ perm, uint64_t n,
							       int64_t *__restrict__ flags)
{
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS) {
		bool head = k == 0;
		if (!head) {
			const uint32_t p = perm[k], q = perm[k - 1];
			for (int c = 0; c < a.nkeys && !head; c++) {
				const struct mdb_sort_key &key = a.key[c];
				const uint64_t rp = key.rid ? (uint64_t)key.rid[p] : (uint64_t)p;
				const uint64_t rq = key.rid ? (uint64_t)key.rid[q] : (uint64_t)q;
				const bool np = key.nullbits && mdb_bit_is_set(key.nullbits, rp);
				const bool nq = key.nullbits && mdb_bit_is_set(key.nullbits, rq);
				if (np != nq)
					head = true;
				else if (!np && ((const uint64_t *)key.values)[rp] != ((const uint64_t *)key.values)[rq])
					head = true;
			}
		}
		flags[k] = head ? 1 : 0;	/* indexed by SORTED position */
	}
}

/* the same from the sorted composite values of the packed sort: a streaming comparison of neighbours instead of four
 * random column reads per row (4.5 ms -> 0.3 ms per 10^8 rows and two columns) */
__global__ __launch_bounds__(SORT_THREADS) void k_group_heads_vkey(const uint64_t *__restrict__ vkey, uint64_t n, int64_t *__restrict__ flags)
{
	for (uint64_t k = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; k < n; k += (uint64_t)gridDim.x * SORT_THREADS)
		flags[k] = (k == 0 || vkey[k] != vkey[k - 1]) ? 1 : 0;
}

/* head_pos[g] = sorted position of group g's first row (ascending); record g = (perm[head] << (64 - kbits)) | run length */
__global__ __launch_bounds__(SORT_THREADS) void k_group_records(const uint32_t *__restrict__ head_pos, uint64_t G, uint64_t n,
								 const uint32_t *__restrict__ perm, uint32_t kbits, unsigned long long *__restrict__ rec)
{
	for (uint64_t g = (uint64_t)blockIdx.x * SORT_THREADS + threadIdx.x; g < G; g += (uint64_t)gridDim.x * SORT_THREADS) {
		const uint64_t b = head_pos[g], e = g + 1 < G ? (uint64_t)head_pos[g + 1] : n;
		rec[g] = ((unsigned long long)perm[b] << (64 - kbits)) | (unsigned long long)(e - b);
	}
}

/* GROUP BY over columns whose value ranges fit one word TOGETHER: a row's composite value - [NULL flag | image - min] per column, as the
 * packed sort builds it - is a key like any other.  It is written into an 8-byte column and handed to the single-column operator (LDS
 * tables, band sort or partition passes, by the composite's range; groups in first-occurrence order is what that operator delivers)
 * instead of sorting the whole stream and looking for run heads: 10^8 rows in 512 x 300 combinations 2.70 -> 0.9 ms.
 * 0 = served, 1 = not applicable (wider than 63 bits, a column type the sort reports), < 0 = error.  MDB_GROUP_MULTI_PACKED=0: never. */
static int group_multi_packed(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *out_first,
			      int64_t *out_count, uint64_t cap, uint64_t *out_groups)
{
	const char *knob = mdb_knob("MDB_GROUP_MULTI_PACKED");
	if ((knob && knob[0] == '0') || nkeys > SORT_PACK_MAX_KEYS)
		return 1;
	for (int c = 0; c < nkeys; c++)
		if (keys[c].type != MDB_T_INT64 && keys[c].type != MDB_T_DOUBLE)
			return 1;
	int rc = mdb_arena_begin(ctx, 8192);
	if (rc)
		return rc;
	unsigned long long *mm = (unsigned long long *)mdb_arena_take(ctx, 128);
	if (!mm)
		return -MIDORIDB_INTERNAL;
	const uint32_t grid = (uint32_t)(((n + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((n + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	sort_pack_args pa;
	memset(&pa, 0, sizeof(pa));
	pa.nkeys = nkeys;
	pa.nopos = 1;
	bool pa_measured = false, sample_failed = false;
	/* 18 ... 25 bits in all, no row-id vector: the band sort reads the columns itself (k_bg_band_sort<., true>) - no composite column is
	 * written and read back (0.45 ms per 10^8 rows and two columns).  A row outside the sampled ranges, a hot combination that overflows
	 * a region: the composite column below.  MDB_GROUP_MULTI_FUSED=0: never. */
	{
		const char *fk = mdb_knob("MDB_GROUP_MULTI_FUSED");
		bool plain = !(fk && fk[0] == '0') && n >= ((uint64_t)1 << 18);	/* (the band sort: from 2^21 rows on - it says so itself) */
		for (int c = 0; c < nkeys; c++)
			plain = plain && !keys[c].rid && !((uintptr_t)keys[c].values & 15u);
		if (plain) {
			uint32_t total = 0;
			rc = sort_pack_ranges(ctx, keys, nkeys, n, mm, &pa, &total, 25, NULL, sort_ranges_sampled(n));
			if (rc < 0)
				return rc;
			pa_measured = rc == 0 && !sort_ranges_sampled(n);	/* (measured ranges of every column: good for the composite column below as well) */
			if (rc == 0 && (total >= 18 || total <= 14)) {
				struct mdb_bg_comp bc;
				memset(&bc, 0, sizeof(bc));
				bc.nkeys = nkeys;
				for (int c = 0; c < nkeys; c++) {
					bc.values[c] = (const uint64_t *)keys[c].values;
					bc.nullbits[c] = keys[c].nullbits;
					bc.lo[c] = pa.lo[c];
					bc.span[c] = pa.span[c];
					bc.kb[c] = pa.kb[c];
					bc.is_double[c] = keys[c].type == MDB_T_DOUBLE;
					bc.desc[c] = keys[c].desc;
				}
				bool outside = false;
				if (total <= 14)	/* (few combinations: per-workgroup LDS tables, k_group_direct<true>) */
					rc = mdb_group_count_direct_comp(ctx, &bc, n, total, out_first, out_count, cap, out_groups);
				else
					rc = mdb_group_count_banded(ctx, (const int64_t *)keys[0].values, n, 0, total, out_first, out_count, cap, out_groups, &outside, &bc);
				if (rc <= 0)
					return rc;
				*out_groups = 0;
				sample_failed = outside || total <= 14;	/* (a value outside the sampled ranges: the composite column starts from measured ones) */
			}
			/* (the band sort began an arena of its own) */
			if ((rc = mdb_arena_begin(ctx, 8192)))
				return rc;
			if (!(mm = (unsigned long long *)mdb_arena_take(ctx, 128)))
				return -MIDORIDB_INTERNAL;
		}
	}
	void *comp = NULL;
	if ((rc = mdb_dev_alloc(ctx, (n ? n : 1) * 8, &comp)))
		return rc;
	uint64_t *h = ctx->h_pinned;
	auto pack = [&]() -> int {	/* 0 = comp holds the composite values, 1 = they do not fit 63 bits */
		for (bool sampled = !pa_measured && !sample_failed && sort_ranges_sampled(n);; sampled = false) {
			uint32_t total = 0;
			const int rrc = pa_measured ? 0 : sort_pack_ranges(ctx, keys, nkeys, n, mm, &pa, &total, 63, NULL, sampled);
			if (rrc == 1 && sampled)
				continue;
			if (rrc)
				return rrc;
			uint32_t *outside = reinterpret_cast<uint32_t *>(mm + 12);
			MDB_HIP(ctx, hipMemsetAsync(outside, 0, 8, ctx->stream));
			MDB_LAUNCH(ctx, "groupby_pack", k_sort_pack, grid, SORT_THREADS, pa, n, (uint64_t *)comp, sampled ? outside : (uint32_t *)NULL);
			if (!sampled)
				return MIDORIDB_OK;
			MDB_HIP(ctx, hipMemcpyAsync(&h[12], outside, 4, hipMemcpyDeviceToHost, ctx->stream));
			MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
			if (!(uint32_t)h[12])
				return MIDORIDB_OK;
		}
	};
	if ((rc = pack())) {
		mdb_dev_free(ctx, comp);
		return rc;
	}
	rc = mdb_dev_group_count(ctx, (const int64_t *)comp, NULL, n, MDB_ORDER_FIRST, out_first, out_count, cap, out_groups);	/* (synchronous) */
	mdb_dev_free(ctx, comp);
	return rc;
}

extern "C" int mdb_dev_group_count_multi(mdb_dev_ctx *ctx, const struct mdb_sort_key *keys, int nkeys, uint64_t n, uint32_t *out_first,
					 int64_t *out_count, uint64_t cap, uint64_t *out_groups)
{
	*out_groups = 0;
	int rc = sort_check(ctx, "group_count_multi", nkeys, n);
	if (rc || n == 0)
		return rc;
	rc = group_multi_packed(ctx, keys, nkeys, n, out_first, out_count, cap, out_groups);
	if (rc <= 0)
		return rc;
	*out_groups = 0;
	uint32_t kbits = 0;
	const size_t order_bytes = mdb_order_records_arena_bytes(n, n, &kbits);
	if (!order_bytes)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group_count_multi: too many rows");
	rc = mdb_arena_begin(ctx, sort_arena_bytes(n) + 2 * mdb_align_up(n * 8) + mdb_align_up(n * 4) + mdb_filter_arena_bytes(n) + order_bytes + 8192);
	if (rc)
		return rc;
	uint32_t *perm = NULL;
	uint64_t *vkey = NULL;
	rc = sort_perm_impl(ctx, keys, nkeys, n, &perm, &vkey);
	if (rc)
		return rc;
	int64_t *flags = (int64_t *)mdb_arena_take(ctx, n * 8);
	uint32_t *head_pos = (uint32_t *)mdb_arena_take(ctx, n * 4);
	unsigned long long *rec = (unsigned long long *)mdb_arena_take(ctx, n * 8);
	if (!flags || !head_pos || !rec)
		return -MIDORIDB_INTERNAL;
	distinct_args a;
	memset(&a, 0, sizeof(a));
	for (int c = 0; c < nkeys; c++)
		a.key[c] = keys[c];
	a.nkeys = nkeys;
	const uint32_t grid = (uint32_t)(((n + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((n + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	if (vkey) {
		MDB_LAUNCH(ctx, "groupby_heads", k_group_heads_vkey, grid, SORT_THREADS, (const uint64_t *)vkey, n, flags);
	} else {
		MDB_LAUNCH(ctx, "groupby_heads", k_group_heads, grid, SORT_THREADS, a, (const uint32_t *)perm, n, flags);
	}
	uint32_t *d_total = NULL;
	rc = mdb_filter_nonzero64(ctx, flags, n, head_pos, &d_total);
	if (rc)
		return rc;
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h, d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint64_t G = (uint32_t)h[0];
	if (G > cap)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups", (unsigned long long)cap,
				   (unsigned long long)G);
	const uint32_t ggrid = (uint32_t)(((G + SORT_THREADS - 1) / SORT_THREADS) < 2048 ? ((G + SORT_THREADS - 1) / SORT_THREADS) : 2048);
	MDB_LAUNCH(ctx, "groupby_records", k_group_records, ggrid ? ggrid : 1, SORT_THREADS, (const uint32_t *)head_pos, G, n, (const uint32_t *)perm,
		   kbits, rec);
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
	rc = mdb_order_records_by_rowid(ctx, rec, G, n, kbits, out_first, out_count);
	if (rc)
		return rc;
	*out_groups = G;
	return MIDORIDB_OK;
}
