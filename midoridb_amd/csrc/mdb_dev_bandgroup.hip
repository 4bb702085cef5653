/*
 * mdb_dev_bandgroup.hip - GROUP BY key + COUNT(*) of ONE key column, groups in first-row order, through 4-byte row words (round 5).
 *
 * What it replaces: the reference's _group_by: rows sorted by the grouping column, one COUNT per run, groups in first-occurrence order
 * (/root/reference/src/engine/executor_select.c:1465-1524, 1526-1588).  The contract is mdb_dev_group_count's (mdb_dev_join.hip), which
 * calls in here for a NULL-free key column inside a compact window of 2^18 ... 2^25 key values.
 *
 * Why: the partitioned path moves an 8-byte word per row (key hash | row id) through a first level with global cursors - 0.8 GB in,
 * 0.8 GB out, 0.8 GB into the leaf: 0.36 + 0.21 ms at 10^8 rows.  A row id does not have to travel if WHERE a word lies says most of it:
 *
 *   1. band sort (k_bg_band_sort): the table is cut into bands of 2^18 consecutive rows; every band owns one region per key digit.  A
 *      workgroup sorts a tile of 2^BG_TILE_BITS = 16 384 rows by digit in LDS (counting sort on packed 16-bit counters, as mdb_dev_rowjoin.hip's
 *      tile sort; 64 KiB of staging: two workgroups per CU - tiles of 32 768 rows, one workgroup per CU, measured 0.42 ms against 0.27),
 *      reserves room for each digit's run in its band's region with one global atomic per (tile, digit), and writes the runs there:
 *      word = key bits below the digit << 18 | row inside the band.  8 B read, 4 B written per row.
 *   2. leaf (k_bg_group_leaf): one workgroup per digit reads the digit's region of every band - pieces of 2^18 / digits words, whole
 *      lines, 16 bytes per lane - and keeps (smallest row id, COUNT) per key value in two direct-addressed LDS arrays; the groups leave as
 *      one record list, ordered by first row by order_records (mdb_dev_order.hip).
 *
 * A region holds 1.5 x the average + 64 words; a digit that outgrows it (a hot key) raises a flag and the caller's other forms answer.
 * The BG_BAND_TILES = 16 tiles of a band run on ONE XCD at about the same time (block -> tile mapping below), so that runs that share a line
 * meet in that XCD's L2.
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"

#ifndef BG_TILE_BITS
#define BG_TILE_BITS 14u		/* rows per tile: 64 KiB of LDS staging - two workgroups per CU, one loading while the other writes */
#endif
#define BG_TILE (1u << BG_TILE_BITS)
#define BG_THREADS 1024
#define BG_ITEMS (BG_TILE / BG_THREADS)
#define BG_LOADS 8
#define BG_ROW_BITS 18u			/* rows per band: 2^18 - what is left of a word holds the key bits below the digit (<= 14) */
#define BG_BAND_TILES (1u << (BG_ROW_BITS - BG_TILE_BITS))	/* tiles per band */
#define BG_MAX_DBITS 11u		/* digits per tile: the run of a digit averages 16 words at least */
#define BG_SKEW 80u			/* words between two bands' blocks (a leaf asks for the same place of every block: not 2^k bytes apart) */
#define BG_OVF 0x80000000u

struct bg_sort_args {
	const int64_t *keys;
	uint64_t n;
	int64_t base;
	uint32_t kbits, dbits;
	uint32_t *words;	/* [nbands * bstride]: band b, digit d: words[b * bstride + d * cap ...] */
	uint32_t *cur;		/* [nbands * D]: cur[b * D + d] = words of digit d in band b (more than cap: the region overflowed) */
	uint32_t cap, bstride, nfull /* full tiles */, ntiles;
	uint32_t ablate;	/* measurement only (MDB_BG_ABLATE): 1 no cursor atomics, 2 no words written, 4 words written in tile order (no walk) */
	uint32_t *status;
	struct mdb_bg_comp comp;	/* k_bg_band_sort<., true>: the key is the composite value of these columns (keys, base unused) */
};

/* block -> tile: workgroups are dealt to the 8 XCDs round-robin; XCD x takes the bands x, x + 8, ... and runs a band's tiles one after
 * the other */
__device__ static inline uint32_t bg_tile_of_block(uint32_t b)
{
	const uint32_t xcd = b & 7u, k = b >> 3;
	return ((k / BG_BAND_TILES) * 8u + xcd) * BG_BAND_TILES + (k % BG_BAND_TILES);
}

/* (the image of mdb_dev_sort.hip's sort_image: order does not matter here, equality does - the same bits as the packed sort's fields) */
__device__ static inline uint64_t bg_image(uint64_t bits, int is_double, int desc)
{
	const uint64_t u = (is_double && (bits >> 63)) ? ~bits : (bits ^ 0x8000000000000000ull);
	return desc ? ~u : u;
}

template <bool FULL, bool COMP = false>
__global__ __launch_bounds__(BG_THREADS) void k_bg_band_sort(bg_sort_args a)
{
	extern __shared__ uint32_t bg_lds[];
	const uint32_t D = 1u << a.dbits, rem = a.kbits - a.dbits, smask = (1u << rem) - 1u;
	uint32_t *const s_cnt = bg_lds;				/* D / 2 words: two 16-bit counters each; then starts, then cursors = ends */
	uint32_t *const s_dst = bg_lds + (D >> 1);		/* D words: where position j of the r-th digit that occurs goes: words[band block + s_dst[r] + j] */
	uint32_t *const s_bits = s_dst + D;			/* BG_TILE / 32 words: bit j = a digit's run starts at position j of the sorted tile */
	uint32_t *const s_stage = s_bits + BG_TILE / 32u;	/* BG_TILE words */
	__shared__ uint32_t s_tmp[32];
	const uint32_t tile = FULL ? bg_tile_of_block(blockIdx.x) : a.ntiles - 1u;
	if (FULL && tile >= a.nfull)
		return;
	const uint64_t row0 = (uint64_t)tile * BG_TILE;
	const uint32_t cnt = FULL ? BG_TILE : (uint32_t)(a.n - row0);
	const uint32_t band = tile / BG_BAND_TILES, row_hi = (tile % BG_BAND_TILES) << BG_TILE_BITS;
	for (uint32_t i = threadIdx.x; i < (D >> 1); i += BG_THREADS)
		s_cnt[i] = 0u;
	for (uint32_t i = threadIdx.x; i < BG_TILE / 32u; i += BG_THREADS)
		s_bits[i] = 0u;
	__syncthreads();
	const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(a.keys + row0);
	uint32_t hr[BG_ITEMS];
	uint32_t bad = 0;
	const uint32_t kmask = (1u << a.kbits) - 1u;
	if (COMP) {
		/* the composite value of a row, column by column (most significant first): 4 x 16 bytes of a column in flight per thread, the
		 * fields shifted into a 32-bit word (kbits <= 25) - two workgroups per CU as for a single key column */
		constexpr int CL = 4;
#pragma unroll
		for (int jb = 0; jb < (int)BG_ITEMS / 2; jb += CL) {
			uint32_t acc[CL][2];
#pragma unroll
			for (int jj = 0; jj < CL; jj++)
				acc[jj][0] = acc[jj][1] = 0u;
#pragma unroll
			for (int c = 0; c < MDB_BG_COMP_MAX; c++) {
				if (c >= a.comp.nkeys)
					break;
				const uint64_t *const col = a.comp.values[c] + row0;
				const uint64_t *const nb = a.comp.nullbits[c];
				const uint32_t kb = a.comp.kb[c];
				ulonglong2 pre[CL];
#pragma unroll
				for (int jj = 0; jj < CL; jj++) {
					const uint32_t p = (uint32_t)(jb + jj) * BG_THREADS + threadIdx.x;
					if (FULL) {
						pre[jj] = reinterpret_cast<const ulonglong2 *>(col)[p];
					} else {
						pre[jj] = make_ulonglong2(0ull, 0ull);
						if (2u * p < cnt)
							pre[jj].x = col[2u * p];
						if (2u * p + 1u < cnt)
							pre[jj].y = col[2u * p + 1u];
					}
				}
#pragma unroll
				for (int jj = 0; jj < CL; jj++)
#pragma unroll
					for (int e = 0; e < 2; e++) {
						const uint32_t r = 2u * ((uint32_t)(jb + jj) * BG_THREADS + threadIdx.x) + (uint32_t)e;
						const bool live = FULL || r < cnt;
						const bool isnull = nb && live && mdb_bit_is_set(nb, row0 + r);
						uint64_t d = bg_image(e ? pre[jj].y : pre[jj].x, a.comp.is_double[c], a.comp.desc[c]) - a.comp.lo[c];
						d = (isnull || !live) ? 0ull : d;
						bad |= d > a.comp.span[c] ? 1u : 0u;
						const uint32_t flag = nb ? (uint32_t)(isnull == (a.comp.desc[c] != 0)) : 0u;
						acc[jj][e] = nb ? (acc[jj][e] << (kb + 1u)) | (flag << kb) | (uint32_t)d : (acc[jj][e] << kb) | (uint32_t)d;
					}
			}
#pragma unroll
			for (int jj = 0; jj < CL; jj++)
#pragma unroll
				for (int e = 0; e < 2; e++) {
					const uint32_t r = 2u * ((uint32_t)(jb + jj) * BG_THREADS + threadIdx.x) + (uint32_t)e;
					const uint32_t h = mdb_mixk(acc[jj][e] & kmask, a.kbits);
					hr[2 * (jb + jj) + e] = h;
					if (FULL || r < cnt) {
						const uint32_t d = h >> rem;
						atomicAdd(&s_cnt[d >> 1], 1u << (16u * (d & 1u)));
					}
				}
		}
	} else {
#pragma unroll
	for (int jb = 0; jb < (int)BG_ITEMS / 2; jb += BG_LOADS) {
		ulonglong2 pre[BG_LOADS];
#pragma unroll
		for (int jj = 0; jj < BG_LOADS; jj++) {
			const uint32_t p = (uint32_t)(jb + jj) * BG_THREADS + threadIdx.x;
			if (FULL) {
				pre[jj] = src[p];
			} else {
				pre[jj] = make_ulonglong2((unsigned long long)a.base, (unsigned long long)a.base);
				if (2u * p < cnt)
					pre[jj].x = (unsigned long long)a.keys[row0 + 2u * p];
				if (2u * p + 1u < cnt)
					pre[jj].y = (unsigned long long)a.keys[row0 + 2u * p + 1u];
			}
		}
#pragma unroll
		for (int jj = 0; jj < BG_LOADS; jj++) {
			const int j = jb + jj;
#pragma unroll
			for (int e = 0; e < 2; e++) {
				const uint32_t r = 2u * ((uint32_t)j * BG_THREADS + threadIdx.x) + (uint32_t)e;
				const uint64_t v = (e ? pre[jj].y : pre[jj].x) - (uint64_t)a.base;
				bad |= (uint32_t)(v >> a.kbits) | (uint32_t)(v >> 32);
				const uint32_t h = mdb_mixk((uint32_t)v & kmask, a.kbits);
				hr[2 * j + e] = h;
				if (FULL || r < cnt) {
					const uint32_t d = h >> rem;
					atomicAdd(&s_cnt[d >> 1], 1u << (16u * (d & 1u)));
				}
			}
		}
	}
	}
	if (bad)
		mdb_raise(a.status, 128u);
	__syncthreads();
	/* starts of the D digits inside the tile (thread t: digits 2 t, 2 t + 1), and room in the band's regions: one global atomic per digit
	 * that occurs - issued here, looked at behind the sort pass (a round trip to the L2 that nothing waits for) */
	const bool mine = threadIdx.x < (D >> 1);
	uint32_t c0, c1, s0, s1, r0, g0 = 0, g1 = 0;
	{
		const uint32_t w = mine ? s_cnt[threadIdx.x] : 0u;
		c0 = w & 0xFFFFu;
		c1 = w >> 16;
		/* one scan for both: rows before the digit (low half: at most 2^15) and digits that occur before it (high half) */
		uint32_t total;
		const uint32_t ex = mdb_block_excl_scan((c0 + c1) | (((c0 ? 1u : 0u) + (c1 ? 1u : 0u)) << 16), s_tmp, &total);
		s0 = ex & 0xFFFFu;
		s1 = s0 + c0;
		r0 = ex >> 16;
		if (mine) {
			if (c0 && !(a.ablate & 1u))
				g0 = atomicAdd(&a.cur[(size_t)band * D + 2u * threadIdx.x], c0);
			if (c1 && !(a.ablate & 1u))
				g1 = atomicAdd(&a.cur[(size_t)band * D + 2u * threadIdx.x + 1u], c1);
			s_cnt[threadIdx.x] = s0 | (s1 << 16);
			if (c0)
				atomicOr(&s_bits[s0 >> 5], 1u << (s0 & 31u));
			if (c1)
				atomicOr(&s_bits[s1 >> 5], 1u << (s1 & 31u));
		}
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < (int)BG_ITEMS; i++)
		asm volatile("" : "+v"(hr[i]));
#pragma unroll
	for (int j = 0; j < (int)BG_ITEMS / 2; j++) {
#pragma unroll
		for (int e = 0; e < 2; e++) {
			const uint32_t r = 2u * ((uint32_t)j * BG_THREADS + threadIdx.x) + (uint32_t)e;
			if (FULL || r < cnt) {
				const uint32_t h = hr[2 * j + e], d = h >> rem;
				const uint32_t pos = (atomicAdd(&s_cnt[d >> 1], 1u << (16u * (d & 1u))) >> (16u * (d & 1u))) & 0xFFFFu;
				s_stage[pos] = ((h & smask) << BG_ROW_BITS) | row_hi | r;
			}
		}
	}
	if (mine) {
		const uint32_t d0 = 2u * threadIdx.x;
		const bool o0 = g0 + c0 > a.cap, o1 = g1 + c1 > a.cap;
		if (o0 || o1)
			mdb_raise(a.status, 2u);
		if (c0)
			s_dst[r0] = o0 ? BG_OVF : d0 * a.cap + g0 - s0;
		if (c1)
			s_dst[r0 + (c0 ? 1u : 0u)] = o1 ? BG_OVF : (d0 + 1u) * a.cap + g1 - s1;
	}
	__syncthreads();
	/* Wave w writes the sorted positions [w * BG_TILE / 16, (w + 1) * BG_TILE / 16), 64 at a time: consecutive lanes consecutive positions.
	 * Which run a position belongs to = how many runs start at or before it: the run-start bits counted up to the wave's first position
	 * once, then 64 bits - one broadcast read - per step (a lane following its digit's end through the counters instead: 0.09 ms more per
	 * 10^8 rows) */
	{
		const uint32_t lane = mdb_lane(), wave = threadIdx.x >> 6;
		constexpr uint32_t SPAN = BG_TILE / (BG_THREADS / 64);
		uint32_t *const dst = a.words + (size_t)band * a.bstride;
		uint32_t before = 0;
		for (uint32_t i = lane; i < wave * (SPAN / 32u); i += 64u)
			before += (uint32_t)__popc(s_bits[i]);
#pragma unroll
		for (int o = 32; o; o >>= 1)
			before += (uint32_t)__shfl_xor((int)before, o, MDB_WAVE);
		const uint64_t le = mdb_lanemask_lt() | (1ull << lane);
#pragma unroll 4
		for (uint32_t it = 0; it < SPAN / 64u; it++) {
			const uint32_t j0 = wave * SPAN + it * 64u, j = j0 + lane;
			const uint64_t m = (uint64_t)s_bits[j0 >> 5] | ((uint64_t)s_bits[(j0 >> 5) + 1u] << 32);
			const uint32_t rank = before + (uint32_t)__popcll(m & le) - 1u;
			before += (uint32_t)__popcll(m);
			if (j < cnt && !(a.ablate & 2u)) {
				if (a.ablate & 4u) {
					dst[j] = s_stage[j];
					continue;
				}
				const uint32_t o = s_dst[rank];
				if (o != BG_OVF)
					dst[o + j] = s_stage[j];
			}
		}
	}
}

struct bg_leaf_args {
	const uint32_t *words;
	const uint32_t *cur;
	uint32_t nbands, cap, bstride, dbits, sbits, row_bits;
	unsigned long long *rec;	/* the record list: first row << (64 - row_bits) | COUNT */
	uint32_t rec_cap;
	uint32_t *rec_count;
	uint32_t *groups;
	uint32_t *status;
	/* k_bg_group_leaf<NL, true> (nearly unique keys, mdb_dev_dense.hip): one bit per row - cleared here for every row that is not the first
	 * of its key - and the keys with several rows as exceptions (first row << 32 | COUNT); dense_bits = NULL: the pilot, counters only */
	unsigned int *dense_bits;
	unsigned long long *exc;
	uint32_t exc_cap;
	uint32_t *dense_cnt;		/* [0] rows that are not the first of their key, [1] exceptions, [2] rows seen */
};

#ifndef BG_UNITS
#define BG_UNITS 8u	/* 16-byte loads a lane of the leaf has in flight */
#endif

/* NL: 16-byte loads per lane and piece: a region holds cap <= NL * 256 words */
template <uint32_t NL, bool DN = false /* the bit-per-row form of nearly unique keys */>
__global__ __launch_bounds__(1024, 8 /* waves per SIMD: two workgroups per CU */) void k_bg_group_leaf(bg_leaf_args a)
{
	extern __shared__ uint32_t bgl_lds[];
	const uint32_t S = 1u << a.sbits;
	uint32_t *const s_first = bgl_lds, *const s_count = bgl_lds + S;
	__shared__ uint32_t s_tmp[32];
	__shared__ uint32_t s_base;
	const uint32_t d = blockIdx.x, lane = mdb_lane(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
		s_first[i] = 0xFFFFFFFFu;
		s_count[i] = 0u;
	}
	__syncthreads();
	uint32_t dups = 0;
	{
		/* wave w takes the bands [w * per, (w + 1) * per): their word counts first (one read per band, all in flight), then BG_UNITS loads of
		 * 16 bytes per lane at a time - NL per piece (a region of `cap` words is NL x 256 words at most), BG_UNITS / NL pieces side by side:
		 * the kernel is a chain of round trips, and what counts is how few of them a wave needs */
		constexpr uint32_t PIECES = BG_UNITS / NL;
		const uint32_t per = (a.nbands + nwaves - 1u) / nwaves, b0 = wave * per, b1 = (b0 + per < a.nbands) ? b0 + per : a.nbands;
		const uint32_t *const cw = a.cur + d, D = 1u << a.dbits;
		const uint32_t *const reg = a.words + (size_t)d * a.cap;
		for (uint32_t bb = b0; bb < b1; bb += 64u) {
			uint32_t cl = 0;
			if (bb + lane < b1) {
				cl = cw[(size_t)(bb + lane) * D];
				cl = cl < a.cap ? cl : a.cap;
			}
			const uint32_t nb = (b1 - bb) < 64u ? (b1 - bb) : 64u;
			for (uint32_t p0 = 0; p0 < nb; p0 += PIECES) {
				uint32_t pc[PIECES];
#pragma unroll
				for (uint32_t q = 0; q < PIECES; q++)
					pc[q] = (p0 + q < nb) ? (uint32_t)__builtin_amdgcn_readlane((int)cl, (int)(p0 + q)) : 0u;
				uint4 v[BG_UNITS];
#pragma unroll
				for (uint32_t u = 0; u < BG_UNITS; u++) {
					const uint32_t q = u / NL, k = (u % NL) * 256u + 4u * lane;
					v[u] = make_uint4(0u, 0u, 0u, 0u);
					if (k < pc[q])
						v[u] = *reinterpret_cast<const uint4 *>(reg + (size_t)(bb + p0 + q) * a.bstride + k);
				}
#pragma unroll
				for (uint32_t u = 0; u < BG_UNITS; u++) {
					const uint32_t q = u / NL, k = (u % NL) * 256u + 4u * lane;
					const uint32_t w[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
					const uint32_t hi = (bb + p0 + q) << BG_ROW_BITS;
					if (DN) {
						/* (of two rows of a key that meet in the table of smallest rows the larger is settled: not a first row - four
						 * requests go out, then the answers are looked at) */
						uint32_t prev[4], rowe[4];
#pragma unroll
						for (int e = 0; e < 4; e++) {
							prev[e] = 0xFFFFFFFFu;
							rowe[e] = hi | (w[e] & ((1u << BG_ROW_BITS) - 1u));
							if (k + (uint32_t)e < pc[q]) {
								const uint32_t slot = w[e] >> BG_ROW_BITS;
								prev[e] = atomicMin(&s_first[slot], rowe[e]);
								atomicAdd(&s_count[slot], 1u);
							}
						}
#pragma unroll
						for (int e = 0; e < 4; e++)
							if (prev[e] != 0xFFFFFFFFu) {
								const uint32_t loser = prev[e] > rowe[e] ? prev[e] : rowe[e];
								if (a.dense_bits)
									atomicAnd(&a.dense_bits[loser >> 5], ~(1u << (loser & 31u)));
								dups++;
							}
						continue;
					}
#pragma unroll
					for (int e = 0; e < 4; e++)
						if (k + (uint32_t)e < pc[q]) {
							const uint32_t slot = w[e] >> BG_ROW_BITS, row = hi | (w[e] & ((1u << BG_ROW_BITS) - 1u));
							atomicMin(&s_first[slot], row);
							atomicAdd(&s_count[slot], 1u);
						}
				}
			}
		}
	}
	__syncthreads();
	if (DN) {
		/* no record per group: the counters, and the keys with several rows as exceptions */
		__shared__ uint32_t s_sum[4][16];
		__shared__ uint32_t s_ebase;
		uint32_t groups = 0, nexc = 0, rows = 0;
		for (uint32_t i0 = wave * 64u; i0 < S; i0 += blockDim.x) {
			const uint32_t c = s_count[i0 + lane];
			groups += c ? 1u : 0u;
			nexc += c > 1u ? 1u : 0u;
			rows += c;
		}
#pragma unroll
		for (int o = 32; o; o >>= 1) {
			dups += (uint32_t)__shfl_xor((int)dups, o, MDB_WAVE);
			groups += (uint32_t)__shfl_xor((int)groups, o, MDB_WAVE);
			nexc += (uint32_t)__shfl_xor((int)nexc, o, MDB_WAVE);
			rows += (uint32_t)__shfl_xor((int)rows, o, MDB_WAVE);
		}
		if (lane == 0) {
			s_sum[0][wave] = dups;
			s_sum[1][wave] = groups;
			s_sum[2][wave] = rows;
			s_sum[3][wave] = nexc;
		}
		__syncthreads();
		if (threadIdx.x < 4u) {		/* (one atomic per counter and workgroup) */
			uint32_t t = 0;
			for (uint32_t w = 0; w < nwaves; w++)
				t += s_sum[threadIdx.x][w];
			if (threadIdx.x == 3)
				s_ebase = t ? atomicAdd(&a.dense_cnt[1], t) : 0u;
			else if (t)
				atomicAdd(threadIdx.x == 0 ? &a.dense_cnt[0] : threadIdx.x == 1 ? a.groups : &a.dense_cnt[2], t);
		}
		__syncthreads();
		if (!a.exc || !nexc)
			return;
		uint32_t ebase = s_ebase;
		for (uint32_t w = 0; w < wave; w++)
			ebase += s_sum[3][w];
		if (ebase + nexc > a.exc_cap) {
			if (lane == 0)
				mdb_raise(a.status, 16384u);	/* (more keys with several rows than the list holds: the record form) */
			return;
		}
		for (uint32_t i0 = wave * 64u; i0 < S; i0 += blockDim.x) {
			const uint32_t c = s_count[i0 + lane];
			const uint64_t m = __ballot(c > 1u);
			if (c > 1u)
				a.exc[ebase + (uint32_t)__popcll(m & mdb_lanemask_lt())] = ((unsigned long long)s_first[i0 + lane] << 32) | c;
			ebase += (uint32_t)__popcll(m);
		}
		return;
	}
	/* the groups leave side by side: wave by wave, lane by lane */
	uint32_t mine = 0, cmax = 0;
	for (uint32_t i0 = wave * 64u; i0 < S; i0 += blockDim.x) {
		const uint32_t c = s_count[i0 + lane];
		mine += c ? 1u : 0u;
		cmax = c > cmax ? c : cmax;
	}
	if (cmax >> (32u - a.row_bits))		/* (a COUNT that does not fit beside its row id in 32 bits: the ordering sort keeps 8-byte records) */
		mdb_raise(a.status, 512u);
	/* groups of the waves before this one */
	uint32_t wsum = mine;
#pragma unroll
	for (int o = 32; o; o >>= 1)
		wsum += (uint32_t)__shfl_xor((int)wsum, o, MDB_WAVE);
	uint32_t total;
	const uint32_t before = mdb_block_excl_scan(lane == 0 ? wsum : 0u, s_tmp, &total);	/* (lane 0 of a wave: the waves before it) */
	const uint32_t wbase = (uint32_t)__shfl((int)before, 0, MDB_WAVE);
	if (threadIdx.x == 0) {
		uint32_t nb = 0xFFFFFFFFu;
		if (total) {
			nb = atomicAdd(a.rec_count, total);
			if (nb + total > a.rec_cap) {
				mdb_raise(a.status, 8u);	/* (sized for every key value of the window and every row: cannot happen) */
				nb = 0xFFFFFFFFu;
			} else {
				atomicAdd(a.groups, total);
			}
		}
		s_base = nb;
	}
	__syncthreads();
	if (s_base == 0xFFFFFFFFu)
		return;
	uint32_t run = s_base + wbase;
	for (uint32_t i0 = wave * 64u; i0 < S; i0 += blockDim.x) {
		const uint32_t c = s_count[i0 + lane];
		const uint64_t m = __ballot(c != 0u);
		if (c)
			a.rec[run + (uint32_t)__popcll(m & mdb_lanemask_lt())] = ((unsigned long long)s_first[i0 + lane] << (64 - a.row_bits)) | c;
		run += (uint32_t)__popcll(m);
	}
}

static uint32_t bg_dbits(uint32_t kbits)
{
	/* as few digits as the leaf's LDS allows (2^14 key values per digit: 128 KiB), 512 at least: the longer a tile's run per digit, the fewer
	 * partly written lines - 10^8 rows over 2^23 values: 512 digits 0.267 + 0.122 ms (band sort + leaf), 1024: 0.285 + 0.126, 2048: 0.312 + 0.136 */
	uint32_t d = kbits > 14u + 9u ? kbits - 14u : 9u;
	if (d > BG_MAX_DBITS)
		d = BG_MAX_DBITS;
	return d;
}

/* 0 = done: out_first[g] / out_count[g] = the first row and the rows of group g, groups in first-row order; 1 = not served (the caller's
 * other forms answer; *outside: a key lay outside the window); < 0 = error.  Synchronises. */
int mdb_group_count_banded(mdb_dev_ctx *ctx, const int64_t *keys, uint64_t n, int64_t win_lo, uint32_t kbits, uint32_t *out_first, int64_t *out_count,
			   uint64_t cap, uint64_t *out_groups, bool *outside, const struct mdb_bg_comp *comp)
{
	*outside = false;
	if (comp) {
		if (comp->nkeys < 1 || comp->nkeys > MDB_BG_COMP_MAX || ctx->explain)
			return 1;
		for (int c = 0; c < comp->nkeys; c++)
			if ((uintptr_t)comp->values[c] & 15u)
				return 1;
	}
	if (kbits < 18u || kbits > 14u + BG_MAX_DBITS || n < ((uint64_t)1 << 21) || n >= 0xF0000000ull || ((uintptr_t)keys & 15u) ||
	    (mdb_knob("MDB_GROUP_BANDED") && mdb_knob("MDB_GROUP_BANDED")[0] == '0'))
		return 1;
	/* the regions overflowed on this very column last time (a hot key): not tried again for a while */
	if (ctx->ex_keys == keys && ctx->ex_nl == n && ctx->ex_nr == 0 && ctx->ex_uses < GC_HINT_USES)
		return 1;
	const uint32_t dbits = bg_dbits(kbits), sbits = kbits - dbits, D = 1u << dbits;
	if (sbits > 14u)
		return 1;
	uint32_t row_bits = 0;
	int sb1 = 0, sb2 = 0;
	if (!order_bits(n, &row_bits, &sb1, &sb2))
		return 1;
	const uint32_t ntiles = (uint32_t)((n + BG_TILE - 1) / BG_TILE), nfull = (uint32_t)(n / BG_TILE), nbands = (ntiles + BG_BAND_TILES - 1u) / BG_BAND_TILES;
	const uint32_t avg = (1u << BG_ROW_BITS) >> dbits;
	const uint32_t rcap = (avg + avg / 2u + 64u + 3u) & ~3u;		/* words per (band, digit) region: a multiple of 16 bytes */
	if (rcap > 1024u)	/* (checked before anything is queued: the leaf holds a region's words in registers) */
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a band-sorted column: %u words per region", rcap);
	const uint32_t bstride = D * rcap + BG_SKEW;
	const uint64_t values = (uint64_t)1 << kbits, most = (n < values ? n : values) + 1024;	/* groups: at most the rows, at most the window's key values */
	size_t need = mdb_align_up((size_t)nbands * bstride * 4 + 64) + mdb_align_up((size_t)D * nbands * 4) + mdb_align_up(most * 8) +
		      order_records_arena_bytes(most, n, row_bits, sb1, sb2) + 16384;
	/* (nearly unique keys need nearly as many key values as rows: a window with fewer cannot hold them - no pilot) */
	const bool dense_ok = !comp && n >= ((uint64_t)1 << 22) && values >= n - n / 16 && !(mdb_knob("MDB_GROUP_DENSE") && mdb_knob("MDB_GROUP_DENSE")[0] == '0');
	if (dense_ok)
		need += mdb_dense_arena_bytes(n) + mdb_align_up((n / 8 + 4096) * 8);
	if (ctx->explain) {	/* (mdb_dev_explain_group_count: the band sort serves - nothing is launched; nearly unique keys: a pilot decides) */
		ctx->explain->group_form = 1;
		ctx->explain->key_form = 2;
		ctx->explain->key_bits = kbits;
		ctx->explain->from_stats = ctx->explain_as_sample ? 0u : ctx->pl_from_stats;
		ctx->explain->samples = ctx->explain_as_sample ? 1u : 0u;
		ctx->explain->groups_as_bits = dense_ok ? 1u : 0u;
		return MIDORIDB_OK;
	}
	int rc = mdb_arena_begin(ctx, need);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	uint32_t *words = (uint32_t *)mdb_arena_take(ctx, (size_t)nbands * bstride * 4 + 64);
	uint32_t *cur = (uint32_t *)mdb_arena_take(ctx, (size_t)D * nbands * 4);
	unsigned long long *rec = (unsigned long long *)mdb_arena_take(ctx, most * 8);
	if (!words || !cur || !rec)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a band-sorted column: %s", ctx->err);
	MDB_HIP(ctx, hipMemsetAsync(cur, 0, (size_t)D * nbands * 4, ctx->stream));
	bg_sort_args sa;
	memset(&sa, 0, sizeof(sa));
	sa.keys = keys;
	sa.n = n;
	sa.base = win_lo;
	sa.kbits = kbits;
	sa.dbits = dbits;
	sa.words = words;
	sa.cur = cur;
	sa.cap = rcap;
	sa.bstride = bstride;
	sa.nfull = nfull;
	sa.ntiles = ntiles;
	sa.status = ctx->d_status;
	sa.ablate = mdb_knob("MDB_BG_ABLATE") ? (uint32_t)atoi(mdb_knob("MDB_BG_ABLATE")) : 0u;
	if (comp)
		sa.comp = *comp;
	const size_t lds_sort = ((size_t)(D >> 1) + D + BG_TILE / 32u + BG_TILE) * 4;
	if (nfull) {
		const uint32_t full_bands = (nfull + BG_BAND_TILES - 1u) / BG_BAND_TILES, grid = ((full_bands + 7u) & ~7u) * BG_BAND_TILES;	/* (8 XCDs x bands per XCD x tiles per band) */
		if (comp) {
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bg_band_sort<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
			MDB_LAUNCH_LDS(ctx, "group_band_sort_columns", (k_bg_band_sort<true, true>), grid, BG_THREADS, lds_sort, sa);
		} else {
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bg_band_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
			MDB_LAUNCH_LDS(ctx, "group_band_sort", k_bg_band_sort<true>, grid, BG_THREADS, lds_sort, sa);
		}
	}
	if (ntiles > nfull) {
		if (comp) {
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bg_band_sort<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
			MDB_LAUNCH_LDS(ctx, "group_band_sort_columns", (k_bg_band_sort<false, true>), 1, BG_THREADS, lds_sort, sa);
		} else {
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bg_band_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
			MDB_LAUNCH_LDS(ctx, "group_band_sort", k_bg_band_sort<false>, 1, BG_THREADS, lds_sort, sa);
		}
	}
	bg_leaf_args la;
	memset(&la, 0, sizeof(la));
	la.words = words;
	la.cur = cur;
	la.nbands = nbands;
	la.cap = rcap;
	la.bstride = bstride;
	la.dbits = dbits;
	la.sbits = sbits;
	la.row_bits = row_bits;
	la.rec = rec;
	la.rec_cap = (uint32_t)(most > 0xFFFFFFFFull ? 0xFFFFFFFFull : most);
	la.groups = ctx->d_status + 1;
	la.rec_count = ctx->d_status + 2;
	la.status = ctx->d_status;
	const size_t lds_leaf = (size_t)8 << sbits;
	const uint32_t threads = 1024u;
	uint64_t *h = ctx->h_pinned;
#define BG_LAUNCH_LEAF(N, DNF, GRID, NAME)                                                                                                        \
	do {                                                                                                                                      \
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bg_group_leaf<N, DNF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_leaf)); \
		MDB_LAUNCH_LDS(ctx, NAME, (k_bg_group_leaf<N, DNF>), (GRID), threads, lds_leaf, la);                                                 \
	} while (0)
#define BG_LAUNCH_ANY(DNF, GRID, NAME)                                                                                                            \
	do {                                                                                                                                      \
		if (rcap <= 256u)                                                                                                                 \
			BG_LAUNCH_LEAF(1, DNF, GRID, NAME);                                                                                       \
		else if (rcap <= 512u)                                                                                                            \
			BG_LAUNCH_LEAF(2, DNF, GRID, NAME);                                                                                       \
		else                                                                                                                              \
			BG_LAUNCH_LEAF(4, DNF, GRID, NAME);                                                                                       \
	} while (0)
	/* Nearly unique keys?  The pilot - the same leaf over 64 digits, counters only - counts the rows that are not the first of their key:
	 * one in 16 at most, and the groups leave as one bit per row + exceptions (mdb_dev_dense.hip), no record per group, no sort */
	if (dense_ok) {
		la.dense_cnt = ctx->d_status + 4;
		la.dense_bits = NULL;
		la.exc = NULL;
		BG_LAUNCH_ANY(true, D < 64u ? D : 64u, "group_band_leaf_dense");
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint32_t *ps = reinterpret_cast<const uint32_t *>(&h[1]);
		const uint64_t pilot_dups = ps[4], pilot_rows = ps[6];
		const bool bad_table = (ps[0] & (128u | 2u)) != 0;	/* (a key outside the window, a region that overflowed: dealt with below, on the full run's flags) */
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, 7 * sizeof(uint32_t), ctx->stream));
		if (!bad_table && pilot_rows && pilot_dups * 16u <= pilot_rows) {
			unsigned long long *bits = NULL;
			const uint64_t exc_cap = n / 8 + 4096;
			if ((rc = mdb_dense_bits_begin(ctx, n, &bits)))
				return rc;
			la.dense_bits = reinterpret_cast<unsigned int *>(bits);
			la.exc = (unsigned long long *)mdb_arena_take(ctx, exc_cap * 8);
			if (!la.exc)
				return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a band-sorted column: %s", ctx->err);
			la.exc_cap = (uint32_t)exc_cap;
			BG_LAUNCH_ANY(true, D, "group_band_leaf_dense");
			MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream));
			MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
			const uint32_t dstatus = ps[0], dgroups = ps[1], n_exc = ps[5];
			if (mdb_knob("MDB_DEBUG_GROUP"))
				fprintf(stderr, "group_count (band sort, dense): pilot %llu of %llu rows not first; %u groups, %u rows not first, %u exceptions, status %u\n",
					(unsigned long long)pilot_dups, (unsigned long long)pilot_rows, dgroups, ps[4], n_exc, dstatus);
			if (!(dstatus & (16384u | 128u | 2u)) && (uint64_t)dgroups + ps[4] == n) {
				if (dgroups > cap)
					return mdb_set_err(ctx, -MIDORIDB_ERROR, "GROUP BY: %u groups, room for %llu", dgroups, (unsigned long long)cap);
				if ((rc = mdb_dense_emit(ctx, bits, n, la.exc, n_exc, out_first, out_count)))
					return rc;
				MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
				*out_groups = dgroups;
				ctx->pl_key_bits = kbits;
				ctx->pl_group_form = 1;
				ctx->pl_bits = 1;
				return MIDORIDB_OK;
			}
			/* (the exception list overflowed, or the table's flags ask for another path: the record form says which) */
			MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, 7 * sizeof(uint32_t), ctx->stream));
		}
	}
	BG_LAUNCH_ANY(false, D, "group_band_leaf");
#undef BG_LAUNCH_ANY
#undef BG_LAUNCH_LEAF
	MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint32_t *hs = reinterpret_cast<const uint32_t *>(&h[1]);
	const uint32_t status = hs[0], groups = hs[1], list_len = hs[2];
	if (mdb_knob("MDB_DEBUG_GROUP"))
		fprintf(stderr, "group_count (band sort): window 2^%u at %lld, %u digits, %u bands, %u words per region: status %u, %u groups\n", kbits,
			(long long)win_lo, D, nbands, rcap, status, groups);
	if (status & 128u) {
		*outside = true;
		return 1;
	}
	if (status & 2u) {	/* a region overflowed: the caller's other forms; remembered for this column */
		ctx->ex_keys = keys;
		ctx->ex_nl = n;
		ctx->ex_nr = 0;
		ctx->ex_uses = 0;
		return 1;
	}
	if (status & 8u)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a band-sorted column: the record list overflowed");
	if (status & 512u)
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));	/* (the ordering kernels raise flags of their own there) */
	if (groups > cap)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "GROUP BY: %u groups, room for %llu", groups, (unsigned long long)cap);
	const bool rec32 = !(status & 512u) && row_bits < 32u;
	rc = groups ? order_records(ctx, rec, list_len, n, row_bits, sb1, sb2, out_first, out_count, NULL, NULL, NULL, false, rec32, 0, 0, 0, false, groups) : MIDORIDB_OK;
	if (rc)
		return rc;
	*out_groups = groups;
	ctx->pl_key_bits = kbits;
	ctx->pl_group_form = 1;
	return MIDORIDB_OK;
}
