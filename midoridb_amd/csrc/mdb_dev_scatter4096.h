/*
 * mdb_dev_scatter4096.h - the kernels of the 4096-digit first-level pass over a key column (mdb_dev_shard.hip: mdb_scatter4096; the
 * reference work they stand for is the first half of the join and GROUP BY loops, /root/reference/src/engine/executor_select.c:1076-1149,
 * 1526-1588: every row of a table visited once and brought to where its partners are).  A header, because the same-box A/B and
 * ablation harness (profiles/micro/scatter4096_stream.hip) compiles exactly the kernels the library launches.
 *
 * HBM-bound byte work: 8 n read, 2 n (hash words) or 4.5 n (row words + run headers) written.  No MFMA.
 */
#ifndef MDB_DEV_SCATTER4096_H
#define MDB_DEV_SCATTER4096_H

#include "mdb_dev_common.h"
#include <type_traits>

#define SHW_D_BITS 12		/* the wide fan-out form: 4096 first-level digits */

struct shw_scatter_args {
	const long long *keys;
	const unsigned long long *nullbits;
	uint32_t n;
	long long key_lo;
	uint32_t kbits, rem;	/* rem = kbits - 12 <= 15 */
	uint32_t report;	/* 1 (the right table): a key outside the window raises flag 128; 0: rows with key - key_lo > rel_hi are dropped */
	uint32_t rel_hi;
	void *out;		/* 2-byte words, or 4-byte row words (ROWS) */
	uint32_t *cursor;	/* [nsub][4096] */
	uint32_t cap, nsub;
	uint32_t *status;
	uint32_t rows_per_wg;	/* (even) */
	unsigned long long *dbg;	/* (the harness' cycle stamps: ABLATE & 4) */
};

/* a barrier that waits for the wave's LDS operations only: global loads (the next tile's keys) and the cursor atomics stay in
 * flight across it (__syncthreads() waits for every outstanding memory operation) */
__device__ static inline void shw_barrier(void)
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* exclusive scan over the workgroup, one barrier: every wave scans the waves' totals itself */
__device__ static inline uint32_t shw_block_excl_scan(uint32_t v, uint32_t *tmp /* [waves] */, uint32_t nwaves, uint32_t *total)
{
	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane();
	const uint32_t incl = mdb_wave_incl_scan(v);
	if (lane == MDB_WAVE - 1)
		tmp[wave] = incl;
	shw_barrier();
	const uint32_t pi = mdb_wave_incl_scan(lane < nwaves ? tmp[lane] : 0u);
	*total = (uint32_t)__shfl((int)pi, (int)nwaves - 1, MDB_WAVE);
	const uint32_t before = (uint32_t)__shfl((int)pi, wave ? (int)wave - 1 : 0, MDB_WAVE);
	return incl - v + (wave ? before : 0u);
}

/* ROWS: the words carry the ROW as well (the left table of the ordered operator, mdb_dev_join.hip: k_leaf_wide<.., L32>) - not as a
 * 27-bit row id beside the 15 hash bits (8-byte words: 16 384-row tiles, measured 0.65 ms per 10^8 rows - what the two 512-digit levels
 * they were to replace take), but as the row's place INSIDE ITS TILE (15 bits) in a 4-byte word, and one HEADER word in front of every run
 * (tile x digit: ~8 words) that names the tile: bit 31 set, the tile's first row / 2 below.  The reader resolves a word's row as
 * 2 * header + place; a region always begins with a header.  Staged like the 2-byte words (the spare top bit marks a run's first word),
 * 128 KiB for the same 32 768-row tiles. */
template <int THREADS, int RPT /* rows per thread */, bool ROWS = false, bool NT = false /* the keys as non-temporal loads */>
__global__ __launch_bounds__(THREADS, 4 /* waves per SIMD: one workgroup of 1024 or two of 512 per CU */) void k_shard_scatter_wide(shw_scatter_args a)
{
	typedef typename std::conditional<ROWS, uint32_t, uint16_t>::type W;
	constexpr uint32_t TILE = THREADS * RPT, D = 1u << SHW_D_BITS, DPT = D / THREADS, NCHUNK = TILE / 64u, HALF = RPT / 2;
	constexpr uint32_t MARK = ROWS ? 0x80000000u : 0x8000u, HDR = ROWS ? 1u : 0u;	/* words a run takes beyond its rows */
	static_assert(TILE <= 32768u && DPT >= 2 && (DPT & 1) == 0 && (RPT % 4) == 0, "tile shape");
	extern __shared__ __attribute__((aligned(16))) uint32_t shw_lds[];
	uint32_t *const s_cnt = shw_lds;			/* [D / 2] two 16-bit counters per word, then the digits' tile-local starts */
	uint32_t *const s_delta = s_cnt + D / 2;		/* [D] per NON-EMPTY digit, in digit order: where its run goes minus its tile-local start */
	uint32_t *const s_chunk = s_delta + D;			/* [NCHUNK] runs that begin before staged position 64 c */
	uint32_t *const s_bad = s_chunk + NCHUNK;		/* [D / 32] non-empty digits (by ordinal) whose run did not fit its region */
	uint32_t *const s_tmp = s_bad + D / 32;			/* [32] */
	W *const s_stage = reinterpret_cast<W *>(s_tmp + 32);	/* [TILE] */
	__shared__ uint32_t s_any_bad;

	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane(), sub = blockIdx.x % a.nsub;
	const uint32_t wmask = (1u << a.rem) - 1u;
	const uint64_t limit = a.report ? ((1ull << a.kbits) - 1ull) : (uint64_t)a.rel_hi;
	const uint64_t le = mdb_lanemask_lt() | (1ull << lane);
	for (uint32_t i = threadIdx.x; i < D / 32; i += THREADS)
		s_bad[i] = 0u;
	if (threadIdx.x == 0)
		s_any_bad = 0u;

	/* A workgroup takes one contiguous range of rows, tile after tile (one workgroup per CU: 90 KiB of LDS).  Measured per tile
	 * of 32 768 rows (clock64 around the phases, 10^8 rows): load + rank 32 000 cycles, digits 5 000, stage 8 000, write-out
	 * 16 600 - the sum is the kernel; keys requested one tile ahead, and first tiles of unequal length per workgroup (the CUs'
	 * phases spread over the period) both left the total where it was. */
	const uint64_t r_begin = (uint64_t)blockIdx.x * a.rows_per_wg;
	const uint64_t r_end = r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n;
	for (uint64_t row0 = r_begin; row0 < r_end;) {
		const uint32_t len = (uint32_t)((r_end - row0) < TILE ? (r_end - row0) : TILE);
		const bool full = len == TILE;	/* (uniform) */
		/* (the thread's number, made opaque per tile: otherwise the addresses of all its loads and LDS accesses are computed once,
		 * before the loop, and kept in ~70 registers across it - spills) */
		uint32_t tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		for (uint32_t i = tid; i < D / 2; i += THREADS)
			s_cnt[i] = 0u;
		if (tid == 0)
			s_chunk[0] = 0u;
		shw_barrier();

		/* 1. load (16 bytes = two keys per access, a half of the thread's rows in flight at a time), hash, rank inside the digit */
		uint32_t packed[RPT];	/* digit << 16 | rank, or ~0 for a row that is not taken */
		uint32_t word2[HALF];	/* the rows' 2-byte words, two per register */
#pragma unroll
		for (int hblock = 0; hblock < 2; hblock++) {
			ulonglong2 pre[HALF / 2];
#pragma unroll
			for (int r = 0; r < HALF / 2; r++) {
				const uint32_t e0 = 2u * ((uint32_t)(hblock * (HALF / 2) + r) * THREADS + tid);
				if (full || e0 + 1u < len)		/* (row0 is even and the column 16-byte aligned) */
					if (NT) {
						typedef unsigned long long ull2_nt __attribute__((ext_vector_type(2)));
						const ull2_nt v = __builtin_nontemporal_load(reinterpret_cast<const ull2_nt *>(a.keys + row0 + e0));
						pre[r] = make_ulonglong2(v.x, v.y);
					} else {
						pre[r] = *reinterpret_cast<const ulonglong2 *>(a.keys + row0 + e0);
					}
				else if (e0 < len)
					pre[r] = make_ulonglong2((unsigned long long)a.keys[row0 + e0], 0ull);
				else
					pre[r] = make_ulonglong2(0ull, 0ull);
			}
#pragma unroll
			for (int r = 0; r < HALF / 2; r++) {
				const int pr = hblock * (HALF / 2) + r;		/* pair number of this thread */
				const uint32_t e0 = 2u * ((uint32_t)pr * THREADS + tid);	/* tile-relative row of the pair's first key */
				const unsigned long long kk[2] = { pre[r].x, pre[r].y };
				bool ok[2] = { e0 < len, e0 + 1u < len };
				if (a.nullbits && ok[0]) {	/* (row0 + e0 is even: both bits live in one word) */
					const unsigned long long nb = a.nullbits[(row0 + e0) >> 6] >> ((row0 + e0) & 63u);
					ok[0] = !(nb & 1ull);
					ok[1] = ok[1] && !(nb & 2ull);
				}
				uint32_t w2 = 0u;
#pragma unroll
				for (int e = 0; e < 2; e++) {
					const unsigned long long rel = kk[e] - (unsigned long long)a.key_lo;
					const bool take = ok[e] && rel <= limit;
					if (a.report && ok[e] && !take)
						mdb_raise(a.status, 128u);	/* a right key outside the window: the caller's form does not apply */
					uint32_t pk = 0xFFFFFFFFu;
					if (take) {
						const uint32_t h = mdb_mixk((uint32_t)rel, a.kbits);
						const uint32_t dig = h >> a.rem, sh = (dig & 1u) << 4;
						const uint32_t rank = (atomicAdd(&s_cnt[dig >> 1], 1u << sh) >> sh) & 0xFFFFu;
						pk = (dig << 16) | rank;
						w2 |= ((h & wmask) | (rank == 0u ? 0x8000u : 0u)) << (16 * e);
					}
					packed[2 * pr + e] = pk;
				}
				word2[pr] = w2;
			}
		}
		shw_barrier();

		/* 2. digit counts -> tile-local starts (written back over the counters), the ordinal of every non-empty digit, the runs
		 *    that begin before every 64th staged position, and the run's place in its region: one global atomic per (tile,
		 *    non-empty digit), whose round trip the staging below covers */
		uint32_t cnt[DPT], v = 0u;
#pragma unroll
		for (int j = 0; j < (int)DPT / 2; j++) {
			const uint32_t c2 = s_cnt[tid * (DPT / 2) + j];
			cnt[2 * j] = c2 & 0xFFFFu;
			cnt[2 * j + 1] = c2 >> 16;
			v += cnt[2 * j] + cnt[2 * j + 1] + ((cnt[2 * j] ? 1u : 0u) + (cnt[2 * j + 1] ? 1u : 0u)) * 65536u;
		}
		uint32_t tot;
		const uint32_t ex = shw_block_excl_scan(v, s_tmp, THREADS / 64, &tot);	/* rows below bit 16 (<= 32 768), non-empty digits above */
		const uint32_t tile_total = tot & 0xFFFFu;
		uint32_t base[DPT], st0[DPT];
		const uint32_t ord0 = ex >> 16;
		{
			uint32_t start = ex & 0xFFFFu, ord = ord0;
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				const uint32_t d = tid * DPT + (uint32_t)j;
				st0[j] = start;
				base[j] = 0u;
				if (cnt[j]) {
					base[j] = atomicAdd(&a.cursor[sub * D + d], cnt[j] + HDR);
					/* this run is the last one to begin before position 64 c for every c with start < 64 c <= start + count */
					for (uint32_t c = (start >> 6) + 1u; (c << 6) <= start + cnt[j] && c < NCHUNK; c++)
						s_chunk[c] = ord + 1u;
					ord++;
					start += cnt[j];
				}
			}
#pragma unroll
			for (int j = 0; j < (int)DPT / 2; j++)
				s_cnt[tid * (DPT / 2) + j] = st0[2 * j] | (st0[2 * j + 1] << 16);
		}
		shw_barrier();

		/* 3. stage by digit */
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			if (packed[r] != 0xFFFFFFFFu) {
				const uint32_t dig = packed[r] >> 16;
				const uint32_t st = (s_cnt[dig >> 1] >> ((dig & 1u) << 4)) & 0xFFFFu;
				const uint32_t w16 = (word2[r >> 1] >> (16 * (r & 1))) & 0xFFFFu;
				if (ROWS)	/* (the row's place in the tile: pair r / 2 of this thread, element r & 1) */
					s_stage[st + (packed[r] & 0xFFFFu)] = (W)(((w16 & 0x8000u) << 16) | ((2u * ((uint32_t)(r >> 1) * THREADS + tid) + (uint32_t)(r & 1)) << 15) |
										  (w16 & 0x7FFFu));
				else
					s_stage[st + (packed[r] & 0xFFFFu)] = (W)w16;
			}
		}
		{
			uint32_t ord = ord0;
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				if (cnt[j]) {
					const uint32_t d = tid * DPT + (uint32_t)j;
					if (base[j] + cnt[j] + HDR > a.cap) {
						mdb_raise(a.status, 2u);	/* the region is full: reported, the operator takes its exact path */
						atomicOr(&s_bad[ord >> 5], 1u << (ord & 31u));
						s_any_bad = 1u;
					}
					s_delta[ord] = (d * a.nsub + sub) * a.cap + base[j] + HDR - st0[j];
					ord++;
				}
			}
		}
		shw_barrier();

		/* 4. write out: consecutive lanes, consecutive positions of a run; a position's run = the runs that begin before its
		 *    chunk of 64 + the marks up to it inside the chunk */
		const bool any_bad = s_any_bad != 0u;
#pragma unroll
		for (int k = 0; k < RPT; k++) {
			const uint32_t i = (uint32_t)k * THREADS + tid;
			const uint32_t sv = i < tile_total ? s_stage[i] : 0u;
			const uint64_t m = __ballot(sv & MARK);
			if (i >= tile_total)
				continue;
			const uint32_t ord = s_chunk[(uint32_t)k * (THREADS / 64) + wave] + (uint32_t)__popcll(m & le) - 1u;
			if (any_bad && ((s_bad[ord >> 5] >> (ord & 31u)) & 1u))
				continue;
			const uint32_t g = i + s_delta[ord];
			reinterpret_cast<W *>(a.out)[g] = (W)(sv & (MARK - 1u));
			if (ROWS && (sv & MARK))	/* the run's header: the tile (row0 is even) */
				reinterpret_cast<W *>(a.out)[g - 1u] = (W)(0x80000000u | (uint32_t)(row0 >> 1));
		}
		shw_barrier();
		if (any_bad) {		/* (rare: clear the marks of this tile's full regions) */
			for (uint32_t i = tid; i < D / 32; i += THREADS)
				s_bad[i] = 0u;
			if (tid == 0)
				s_any_bad = 0u;
			shw_barrier();
		}
		row0 += len;
	}
}


static inline size_t shw_scatter_lds(uint32_t tile, size_t word_bytes = 2)
{
	const uint32_t D = 1u << SHW_D_BITS;
	return (size_t)4 * (D / 2 + D + tile / 64 + D / 32 + 32) + word_bytes * tile;
}

/* ------------------------------------------------------------------ the same pass as a stream (round 6)
 *
 * k_shard_scatter_wide above runs its four phases one after the other on a CU (one 90 - 158 KiB workgroup: nothing else is resident):
 * per tile of 32 768 rows load + rank 32 000 cycles, digits 5 000, stage 8 000, write-out 16 600 - no load is in flight during the
 * last three, no store during the first, and most of the cycles are not the memory system's: they are LDS round trips and exec-mask
 * branches that every row pays one after the other.  Here the SAME workgroup keeps both the memory system and its own pipes busy:
 *   - a key's state is ONE register (its k-bit hash; the rank inside the digit is not kept: the histogram is a non-returning LDS
 *     atomic, and the staging pass takes its position from a returning atomic on the digit's cursor instead of reading it), so the
 *     registers hold a ring of 8 x 16 bytes of keys in flight per thread beside the tile's state - 128 KiB per CU on their way at
 *     any time, requested as the ring's slots are consumed, across every phase;
 *   - the write-out of tile t and the hashing + counting of tile t + 1 are ONE instruction stream: step k writes 64 staged positions
 *     per wave and counts one key per lane (they touch different LDS arrays; vmcnt is in order, so a key only waits for loads issued
 *     before the stores around it);
 *   - the hot loops are straight-line code: a row that is not taken (outside the window, behind the end of a short tile, NULL)
 *     counts into a dummy digit and is staged into a dummy slot instead of being branched around, so the LDS requests of many rows
 *     are in flight together; only the global stores are predicated;
 *   - run starts are a bitmap written by the digits' owners (one LDS atomic OR per non-empty digit) instead of a spare bit in every
 *     staged word that each write-out step has to ballot; counters are 32-bit words (a digit's counter is one shift away).
 * Words, headers, regions and cursors are k_shard_scatter_wide's - the readers do not change; inside a run the words are in the
 * order the staging atomics were served (the readers never looked at it: a region's runs already arrive in the cursors' order). */
template <int THREADS, int RPT /* rows per thread */, bool ROWS = false, bool NULLS = false, int RING_ = 0 /* 16-byte requests in flight per thread */,
	  int ABLATE = 0 /* harness only, bits: 1 no global stores, 2 keys not loaded (synthetic), 4 cycle stamps per phase, 16 linear stores, 32 nt stores */,
	  int CPOL = 0 /* cache policy of the key loads: 0 plain, else buffer loads with these bits (1 sc0, 2 nt, 16 sc1) */>
__global__ __launch_bounds__(THREADS, 4) void k_scatter4096_stream(shw_scatter_args a)
{
	typedef typename std::conditional<ROWS, uint32_t, uint16_t>::type W;
	constexpr uint32_t TILE = THREADS * RPT, D = 1u << SHW_D_BITS, DPT = D / THREADS, NCHUNK = TILE / 64u, NPAIR = RPT / 2, RING = RING_ ? RING_ : (NPAIR % 4 == 0 ? 4 : (NPAIR % 5 == 0 ? 5 : 3));
	constexpr uint32_t HDR = ROWS ? 1u : 0u;	/* words a run takes beyond its rows */
	constexpr uint32_t WB = RPT % 4 == 0 ? 4 : 2;	/* write-out steps whose LDS reads travel together */
	static_assert((ROWS ? TILE <= 32768u : TILE < 65536u) && DPT == 4 && (RPT & 1) == 0 && NPAIR % RING == 0 && (TILE / 32u) % 2u == 0, "tile shape");
	extern __shared__ __attribute__((aligned(16))) uint32_t shw_lds[];
	uint32_t *const s_cnt = shw_lds;			/* [D + 64] the digits' counts, then their staging cursors; [D] the dummy digit */
	uint32_t *const s_delta = s_cnt + D + 64;		/* [D] per NON-EMPTY digit, in digit order: where its run goes minus its tile-local start */
	uint32_t *const s_chunk = s_delta + D;			/* [NCHUNK] runs that begin before staged position 64 c */
	uint32_t *const s_mark = s_chunk + NCHUNK;		/* [TILE / 32] bit i: a run begins at staged position i */
	uint32_t *const s_tmp = s_mark + TILE / 32u;		/* [32] */
	W *const s_stage = reinterpret_cast<W *>(s_tmp + 32);	/* [TILE + 64]; [TILE] the dummy slot */
	__shared__ uint32_t s_any_bad;

	const uint32_t lane = mdb_lane(), sub = blockIdx.x % a.nsub;
	const uint32_t wmask = (1u << a.rem) - 1u, none = D << a.rem;	/* (the state of a row that is not taken: the dummy digit) */
	const uint64_t limit = a.report ? ((1ull << a.kbits) - 1ull) : (uint64_t)a.rel_hi;
	const uint64_t le = mdb_lanemask_lt() | (1ull << lane);
	for (uint32_t i = threadIdx.x; i < D + 64; i += THREADS)
		s_cnt[i] = 0u;
	if (threadIdx.x == 0)
		s_any_bad = 0u;

	const uint64_t r_begin = (uint64_t)blockIdx.x * a.rows_per_wg;
	const uint64_t r_end = r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n;
	if (r_begin >= r_end)
		return;		/* (uniform) */

	ulonglong2 pre[RING];	/* the keys in flight, a ring: pair p of this thread = rows 2 (p THREADS + tid), + 1 of its tile, in slot p % RING */
	uint32_t hs[RPT];	/* a row's hash, or `none` */
	/* request pair P of the tile of tlen rows at row t0 (even; the column is 16-byte aligned, so the pair of a column's last, odd row ends
	 * inside the column's last 16 bytes): a uniform base and one offset per thread; pairs behind a short tile read its last pair */
#define SHS_REQUEST(P, t0, tlen, tid_)                                                                                   \
	do {                                                                                                             \
		if (ABLATE & 2) {                                                                                        \
			const unsigned long long i_ = (t0) + 2u * ((uint32_t)(P) * THREADS + (tid_));                    \
			pre[(P) % RING] = make_ulonglong2((unsigned long long)a.key_lo + ((i_ * 0x9E3779B1ull) & ((1ull << a.kbits) - 1ull)), \
							  (unsigned long long)a.key_lo + (((i_ + 1u) * 0x9E3779B1ull) & ((1ull << a.kbits) - 1ull))); \
		} else {	/* (one form for full and short tiles: a branch here costs the loads their counted waits) */ \
			const uint32_t last_ = ((tlen) - 1u) & ~1u, e0_ = 2u * ((uint32_t)(P) * THREADS + (tid_));      \
			const ulonglong2 *const ptr_ = reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const char *>(a.keys + (t0)) + (size_t)(e0_ < last_ ? e0_ : last_) * 8u); \
			if (CPOL) {                                                                                      \
				typedef unsigned int u4_ __attribute__((ext_vector_type(4)));                            \
				const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void *)(a.keys + (t0)), 0, (int)(((tlen) + 1u) & ~1u) * 8, 0x00020000); \
				const u4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)((e0_ < last_ ? e0_ : last_) * 8u), 0, CPOL); \
				pre[(P) % RING] = make_ulonglong2((unsigned long long)v_.x | ((unsigned long long)v_.y << 32), (unsigned long long)v_.z | ((unsigned long long)v_.w << 32)); \
			} else {                                                                                         \
				pre[(P) % RING] = *ptr_;                                                                 \
			}                                                                                                \
		}                                                                                                        \
	} while (0)
	/* hash and count row r (element r & 1 of pair r / 2) of the tile of tlen rows at t0, from the keys requested before */
#define SHS_COUNT(r_, t0, tlen, tid_)                                                                                    \
	do {                                                                                                             \
		const uint32_t e_ = 2u * ((uint32_t)((r_) >> 1) * THREADS + (tid_)) + (uint32_t)((r_) & 1);             \
		const unsigned long long kk_ = ((r_) & 1) ? pre[((r_) >> 1) % RING].y : pre[((r_) >> 1) % RING].x;      \
		bool ok_ = e_ < (tlen);                                                                                  \
		if (NULLS && ok_)                                                                                        \
			ok_ = !((a.nullbits[((t0) + e_) >> 6] >> (((t0) + e_) & 63u)) & 1ull);                           \
		const unsigned long long rel_ = kk_ - (unsigned long long)a.key_lo;                                      \
		const bool take_ = ok_ && rel_ <= limit;                                                                 \
		oow |= ok_ && !take_;                                                                                    \
		uint32_t h_ = mdb_mixk((uint32_t)rel_, a.kbits);	/* (hashed whether taken or not: no branch around two multiplications) */ \
		asm volatile("" : "+v"(h_));                                                                             \
		h_ = take_ ? h_ : none;                                                                                  \
		atomicAdd(&s_cnt[h_ >> a.rem], 1u);                                                                      \
		hs[r_] = h_;                                                                                             \
	} while (0)

	/* the tile that is counted next (its first keys are on their way) and the staged one that is written out behind it (none yet) */
	uint64_t row0 = r_begin, staged_row0 = 0;
	uint32_t len = (uint32_t)((r_end - row0) < TILE ? (r_end - row0) : TILE), staged_total = 0u;
	{
		uint32_t tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
#pragma unroll
		for (int p = 0; p < (int)RING; p++)
			SHS_REQUEST(p, row0, len, tid);
	}
	shw_barrier();		/* (the counters are clear) */

	for (;;) {
		/* (the thread's number, made opaque per tile: otherwise the addresses of all its loads and LDS accesses are computed once,
		 * before the loop, and kept in registers across it) */
		uint32_t tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		const uint32_t wave = tid >> 6;
		const uint64_t nrow0 = row0 + len;
		const uint32_t nlen = len ? (uint32_t)((r_end - nrow0) < TILE ? (r_end - nrow0) : TILE) : 0u;	/* 0: no tile behind this one */
		/* (where the ring's requests go once this tile's pairs are all on their way: the next tile - or, behind the last one, this tile again:
		 * a load nobody looks at instead of a branch around a load, which would cost every load of the loop its counted wait) */
		const uint64_t prow0 = nlen ? nrow0 : row0;
		const uint32_t plen = nlen ? nlen : len;
		bool oow = false;
		unsigned long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0;
		if (ABLATE & 4)
			t0_ = __builtin_amdgcn_s_memtime();

		/* 1. hash and count the tile's rows.  The ring slot a pair leaves takes the pair RING places on - of this tile, then of the next
		 *    one: RING requests of 16 bytes per thread in flight all the time, the next tile's first ones across the phases below (they are
		 *    older than the cursor atomics and the stores there: nobody waits for those to read a key) */
		if (len) {	/* (uniform) */
	#pragma unroll
			for (int k = 0; k < RPT; k++) {
				SHS_COUNT(k, row0, len, tid);
				if (k & 1)	/* (the request stays behind the pair it replaces: issued earlier it would need registers of its own) */
					__builtin_amdgcn_sched_barrier(0);
				if ((k & 1) && (k >> 1) + (int)RING < (int)NPAIR)
					SHS_REQUEST((k >> 1) + (int)RING, row0, len, tid);
				else if (k & 1)
					SHS_REQUEST((k >> 1) + (int)RING - (int)NPAIR, prow0, plen, tid);
			}
			if (a.report && oow)
				mdb_raise(a.status, 128u);	/* a right key outside the window: the caller's form does not apply */
		}
		if (ABLATE & 4)
			t3_ = __builtin_amdgcn_s_memtime();
		/* 1b. ... and write out the staged tile BEHIND it - consecutive lanes, consecutive positions of a run; a position's run = the runs
		 *     that begin before its chunk of 64 + the run-start bits up to it inside the chunk.  The order matters: vmcnt counts loads
		 *     and stores in one sequence, so a wave that has stored waits for its stores to be acknowledged (~10 000 cycles with every CU
		 *     storing) before it sees a key it asks for afterwards; here the stores are followed by the digit bookkeeping and the staging,
		 *     which ask the memory for nothing but the cursor atomics' answers, wanted 18 000 cycles later */
		/* (WB steps at a time: their LDS reads go out together, then the dependent ones, then the stores - one step after the other every
		 * step pays its three LDS round trips in full) */
#pragma unroll
		for (int k0 = 0; k0 < RPT; k0 += (int)WB) {
			if ((uint32_t)k0 * THREADS < staged_total) {	/* (uniform) */
				uint64_t m[WB];
				uint32_t ord[WB], g[WB];
				W sv[WB];
#pragma unroll
				for (int u = 0; u < (int)WB; u++) {
					const uint32_t c = (uint32_t)(k0 + u) * (THREADS / 64) + wave;
					m[u] = *reinterpret_cast<const uint64_t *>(&s_mark[2u * c]);	/* (one broadcast read per wave) */
					ord[u] = s_chunk[c];
					sv[u] = s_stage[(uint32_t)(k0 + u) * THREADS + tid];
				}
#pragma unroll
				for (int u = 0; u < (int)WB; u++) {
					ord[u] += (uint32_t)__popcll(m[u] & le) - 1u;
					ord[u] = ord[u] < D ? ord[u] : D - 1u;	/* (a position behind the tile's rows: read something, write nothing) */
					g[u] = s_delta[ord[u]];
				}
#pragma unroll
				for (int u = 0; u < (int)WB; u++) {
					const uint32_t i = (uint32_t)(k0 + u) * THREADS + tid;
					const bool put = i < staged_total;
					uint32_t gi = i + g[u];
					if (ABLATE & 16)	/* (harness: the same words, written where they stand - whole lines, no runs) */
						gi = (uint32_t)(staged_row0 + i);
					if (!(ABLATE & 1)) {
						if (put) {
							if (ABLATE & 32)
								__builtin_nontemporal_store(sv[u], &reinterpret_cast<W *>(a.out)[gi]);
							else
								reinterpret_cast<W *>(a.out)[gi] = sv[u];
							if (ROWS && !(ABLATE & 16) && ((m[u] >> lane) & 1ull))	/* the run's header: the tile (its first row is even) */
								reinterpret_cast<W *>(a.out)[gi - 1u] = (W)(0x80000000u | (uint32_t)(staged_row0 >> 1));
						}
					} else if (put && sv[u] == (W)0xFFFFFFF1u && gi == 0xFFFFFFFFu) {
						reinterpret_cast<W *>(a.out)[0] = sv[u];
					}
				}
			}
		}
		shw_barrier();
		if (ABLATE & 4) {
			t1_ = __builtin_amdgcn_s_memtime();
			if (threadIdx.x == 0) {
				atomicAdd(&a.dbg[0], t3_ - t0_);	/* count */
				atomicAdd(&a.dbg[4], t1_ - t3_);	/* write-out */
			}
		}
		if (!len)
			break;


		/* 2. digit counts -> tile-local starts (written back over the counters: the staging cursors), the ordinal of every non-empty
		 *    digit, the run-start bits, the runs that begin before every 64th staged position, and the run's place in its region:
		 *    one global atomic per (tile, non-empty digit), whose round trip the staging below covers */
		uint32_t cnt[DPT], v = 0u;
		{
			const uint4 c4 = *reinterpret_cast<const uint4 *>(&s_cnt[tid * DPT]);
			cnt[0] = c4.x;
			cnt[1] = c4.y;
			cnt[2] = c4.z;
			cnt[3] = c4.w;
		}
#pragma unroll
		for (int j = 0; j < (int)DPT; j++)
			v += cnt[j] + (cnt[j] ? 65536u : 0u);
		for (uint32_t i = tid; i < TILE / 32u; i += THREADS)
			s_mark[i] = 0u;		/* (the write-out that read them is behind a barrier) */
		if (tid == 0) {
			s_chunk[0] = 0u;
			s_cnt[D] = TILE;	/* the dummy digit's cursor: behind the tile's last position */
		}
		uint32_t tot;
		const uint32_t ex = shw_block_excl_scan(v, s_tmp, THREADS / 64, &tot);	/* rows below bit 16 (<= 32 768), non-empty digits above */
		const uint32_t tile_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tot & 0xFFFFu));
		uint32_t base[DPT], st0[DPT];
		const uint32_t ord0 = ex >> 16;
		{
			uint32_t start = ex & 0xFFFFu;
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				const uint32_t d = tid * DPT + (uint32_t)j;
				st0[j] = start;
				base[j] = 0u;
				if (cnt[j])
					base[j] = atomicAdd(&a.cursor[sub * D + d], cnt[j] + HDR);
				start += cnt[j];
			}
		}
		{
			uint32_t ord = ord0;
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				if (cnt[j]) {
					const uint32_t start = st0[j];
					atomicOr(&s_mark[start >> 5], 1u << (start & 31u));
					/* this run is the last one to begin before position 64 c for every c with start < 64 c <= start + count */
					for (uint32_t c = (start >> 6) + 1u; (c << 6) <= start + cnt[j] && c < NCHUNK; c++)
						s_chunk[c] = ord + 1u;
					ord++;
				}
			}
			*reinterpret_cast<uint4 *>(&s_cnt[tid * DPT]) = make_uint4(st0[0], st0[1], st0[2], st0[3]);
		}
		shw_barrier();
		if (ABLATE & 4)
			t2_ = __builtin_amdgcn_s_memtime();

		/* 3. stage by digit: a row's position is what the returning atomic on its digit's cursor says (the dummy digit's: the dummy slot) */
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			uint32_t pos = atomicAdd(&s_cnt[hs[r] >> a.rem], 1u);
			pos = pos < TILE ? pos : TILE;
			if (ROWS)	/* (the row's place in the tile: pair r / 2 of this thread, element r & 1) */
				s_stage[pos] = (W)(((2u * ((uint32_t)(r >> 1) * THREADS + tid) + (uint32_t)(r & 1)) << 15) | (hs[r] & wmask));
			else
				s_stage[pos] = (W)(hs[r] & wmask);
		}
		{
			uint32_t ord = ord0;
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				if (cnt[j]) {
					const uint32_t d = tid * DPT + (uint32_t)j;
					if (base[j] + cnt[j] + HDR > a.cap) {
						mdb_raise(a.status, 2u);	/* the region is full: reported, the operator takes its exact path */
						s_any_bad = 1u;			/* ... and nothing of this tile is written */
					}
					s_delta[ord] = (d * a.nsub + sub) * a.cap + base[j] + HDR - st0[j];
					ord++;
				}
			}
		}
		shw_barrier();
		/* (every staging cursor has been read: the counters of the next tile; a run that did not fit: the flag is up, the caller drops the
		 * regions - no word of this tile goes out) */
		*reinterpret_cast<uint4 *>(&s_cnt[tid * DPT]) = make_uint4(0u, 0u, 0u, 0u);
		staged_total = __builtin_amdgcn_readfirstlane((int)s_any_bad) ? 0u : tile_total;
		staged_row0 = row0;
		row0 = nrow0;
		len = nlen;
		shw_barrier();
		if (tid == 0)
			s_any_bad = 0u;		/* (read by everybody in front of the barrier; written next behind two more) */
		if (ABLATE & 4) {
			if (threadIdx.x == 0) {
				atomicAdd(&a.dbg[1], t2_ - t1_);	/* digits */
				atomicAdd(&a.dbg[2], __builtin_amdgcn_s_memtime() - t2_);	/* stage */
				atomicAdd(&a.dbg[3], 1ull);
			}
		}
	}
#undef SHS_REQUEST
#undef SHS_COUNT
}

static inline size_t shs_stream_lds(uint32_t tile, size_t word_bytes = 2)
{
	const uint32_t D = 1u << SHW_D_BITS;
	return (size_t)4 * (D + 64 + D + tile / 64 + tile / 32 + 32) + word_bytes * (tile + 64);
}

#endif /* MDB_DEV_SCATTER4096_H */
