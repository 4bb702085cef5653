/*
 * mdb_dev_rowjoin.hip - a join that carries the right table's payload cells to the LEFT table's rows, result in the left
 * table's row order, without one scattered store per joined row (round 5).
 *
 * What it replaces: the reference's nested loop + cpy_cols / _merge_rows (/root/reference/src/engine/executor_select.c:340-438,
 * 1076-1149, 1151-1232) for a join in which every left row has exactly one partner (foreign key -> primary key; BASELINE
 * configs[1], and configs[4]'s join-only form at 10^8 rows per table).  The contract is mdb_dev_join_payload's (mdb_dev_pairs.hip),
 * which calls in here: out[c][i] = payload cell c of left row i's partner, counted, not assumed.
 *
 * Why: a leaf kernel that holds the right table's cells in LDS meets the left rows of a key digit in KEY order; storing each
 * cell at out[left row id] is one scattered 8-byte store per joined row - beyond the Infinity Cache (an 800 MB column at 10^8
 * rows) a read-modify-write of a whole line in HBM: 3.0 ms per join (profiles/r04/README.md).  Here the left table is never
 * scattered by a global cursor at all:
 *
 *   1. tile sort (k_rj_tile_sort): every tile of 32 768 consecutive left rows is sorted by key digit INSIDE ITS OWN BLOCK of the
 *      word array - words_l[tile * 32768 + p] = (slot inside the digit << 15 | row inside the tile) - plus the tile's digit
 *      offsets (u16).  Sequential reads, sequential writes, no global atomics, nothing that can overflow.
 *   2. leaf (k_rj_leaf): one workgroup per key digit of 2^14 key values: the right table's cells of the digit in an LDS table
 *      (direct-addressed by the slot bits: no stored keys, no probing; an occupancy bitmap sees duplicate right keys and left
 *      rows without partner), then the digit's piece of EVERY left tile - a few words each, found through the tile offsets - is
 *      looked up and the cell written at the SAME position of a cell array laid out like the word array (cells_al[tile * 32768 +
 *      p]): consecutive lanes take consecutive words of a piece, so a piece is one request each way, and adjacent digits -
 *      neighbours on the same XCD (blockIdx -> digit mapping below) - complete each other's lines in that XCD's L2.
 *   3. placement (k_rj_place): one workgroup per half tile reads the tile's words and cells (sequential), drops every cell at
 *      LDS[row inside the half tile] and writes the result column coalesced.
 *
 * Per joined row and cell: 8 B key + 4 B word + (2 + 4) B offsets/word re-read + 8 B cell written, read, and written again in
 * row order - all of it in whole lines.
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"

#define RJ_TILE 32768u		/* left rows per tile: 15 bits of a word name the row inside its tile */
#define RJ_TILE_BITS 15u
#define RJ_THREADS 1024
#define RJ_ITEMS (RJ_TILE / RJ_THREADS)		/* 32 rows per thread */
#define RJ_SLOT_BITS 14u	/* key values per digit: 2^14 eight-byte cells = 128 KiB of LDS */
#define RJ_MAX_DBITS 13u	/* up to 8192 digits per tile (16 KiB of packed 16-bit counters) */
#define RJ_LOADS 8		/* 16-byte key loads a thread of the tile sort issues before it consumes the first */

struct rj_sort_args {
	const int64_t *keys;
	uint64_t n;
	int64_t base;		/* window [base, base + 2^kbits) */
	uint32_t kbits, dbits;
	uint32_t *words;	/* [ntiles * RJ_TILE] */
	uint16_t *offs;		/* [ntiles * (D + 8)]: digit starts inside the tile, entry D = rows of the tile */
	uint32_t *status;
};

/* ---- 1. tile sort.  FULL: a tile of exactly RJ_TILE rows - straight-line code, no per-row branch (the generic form spends a dozen scalar
 * branches per row and spills); the table's last, partial tile takes the other instance */
template <bool FULL>
__global__ __launch_bounds__(RJ_THREADS) void k_rj_tile_sort(rj_sort_args a, uint32_t tile0)
{
	extern __shared__ uint32_t rj_lds[];
	const uint32_t D = 1u << a.dbits, rem = a.kbits - a.dbits, smask = (1u << rem) - 1u;
	uint32_t *const s_cnt = rj_lds;				/* D / 2 words: two 16-bit counters each, then the digit starts */
	uint32_t *const s_stage = rj_lds + (D >> 1);		/* RJ_TILE words */
	__shared__ uint32_t s_tmp[32];
	const uint32_t tile = tile0 + blockIdx.x;
	const uint64_t row0 = (uint64_t)tile * RJ_TILE;
	const uint32_t cnt = FULL ? RJ_TILE : (uint32_t)(a.n - row0);
	for (uint32_t i = threadIdx.x; i < (D >> 1); i += RJ_THREADS)
		s_cnt[i] = 0u;
	__syncthreads();
	/* rows 2 * (j * 1024 + tid) and + 1: 16-byte loads, eight issued before the first is used */
	const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(a.keys + row0);
	uint32_t hr[RJ_ITEMS];		/* k-bit hash */
	uint32_t bad = 0;
	const uint32_t kmask = a.kbits >= 32 ? 0xFFFFFFFFu : ((1u << a.kbits) - 1u);
#pragma unroll
	for (int jb = 0; jb < (int)RJ_ITEMS / 2; jb += RJ_LOADS) {
		ulonglong2 pre[RJ_LOADS];
#pragma unroll
		for (int jj = 0; jj < RJ_LOADS; jj++) {
			const uint32_t p = (uint32_t)(jb + jj) * RJ_THREADS + threadIdx.x;
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
		for (int jj = 0; jj < RJ_LOADS; jj++) {
			const int j = jb + jj;
#pragma unroll
			for (int e = 0; e < 2; e++) {
				const uint32_t r = 2u * ((uint32_t)j * RJ_THREADS + threadIdx.x) + (uint32_t)e;
				const uint64_t v = (e ? pre[jj].y : pre[jj].x) - (uint64_t)a.base;
				bad |= (uint32_t)(v >> a.kbits) | (uint32_t)(v >> 32);	/* (rows behind a partial tile's end read as the window base) */
				const uint32_t h = mdb_mixk((uint32_t)v & kmask, a.kbits);
				hr[2 * j + e] = h;
				if (FULL || r < cnt) {	/* count the digit (no rank is kept: the row takes its place from the digit's cursor below) */
					const uint32_t d = h >> rem;
					atomicAdd(&s_cnt[d >> 1], 1u << (16u * (d & 1u)));
				}
			}
		}
	}
	if (bad)
		mdb_raise(a.status, 128u);
	__syncthreads();
	/* exclusive scan of the D counters: thread t owns D / 1024 consecutive digits (D >= 2048: 2, 4 or 8) */
	{
		const uint32_t per = D / RJ_THREADS, w0 = threadIdx.x * (per >> 1);
		uint32_t c[8], sum = 0;
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			if (i < (per >> 1)) {
				const uint32_t w = s_cnt[w0 + i];
				c[2 * i] = w & 0xFFFFu;
				c[2 * i + 1] = w >> 16;
				sum += c[2 * i] + c[2 * i + 1];
			}
		uint32_t total;
		uint32_t run = mdb_block_excl_scan(sum, s_tmp, &total);
		uint16_t *const og = a.offs + (size_t)tile * (D + 8u) + (size_t)threadIdx.x * per;
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			if (i < (per >> 1)) {
				const uint32_t s0 = run, s1 = run + c[2 * i];
				run = s1 + c[2 * i + 1];
				s_cnt[w0 + i] = s0 | (s1 << 16);	/* (a start is <= 32 768: 16 bits) */
				*reinterpret_cast<uint32_t *>(og + 2 * i) = s0 | (s1 << 16);
			}
		if (threadIdx.x == 0)
			a.offs[(size_t)tile * (D + 8u) + D] = (uint16_t)cnt;	/* (32 768 fits) */
	}
	__syncthreads();
	/* (only the hashes live across the scan: what was derived from them for the counting pass - addresses, shifts - is computed again,
	 * or the compiler keeps three more values per row and spills) */
#pragma unroll
	for (int i = 0; i < (int)RJ_ITEMS; i++)
		asm volatile("" : "+v"(hr[i]));
#pragma unroll
	for (int j = 0; j < (int)RJ_ITEMS / 2; j++) {
#pragma unroll
		for (int e = 0; e < 2; e++) {
			const uint32_t r = 2u * ((uint32_t)j * RJ_THREADS + threadIdx.x) + (uint32_t)e;
			if (FULL || r < cnt) {
				/* the digit's start has become its cursor: two 16-bit cursors per word, a cursor ends at most at 32 768 */
				const uint32_t h = hr[2 * j + e], d = h >> rem;
				const uint32_t pos = (atomicAdd(&s_cnt[d >> 1], 1u << (16u * (d & 1u))) >> (16u * (d & 1u))) & 0xFFFFu;
				s_stage[pos] = ((h & smask) << RJ_TILE_BITS) | r;
			}
		}
	}
	__syncthreads();
	uint4 *const dst = reinterpret_cast<uint4 *>(a.words + row0);
	const uint4 *const st = reinterpret_cast<const uint4 *>(s_stage);
	for (uint32_t i = threadIdx.x; 4u * i < cnt; i += RJ_THREADS)
		dst[i] = st[i];		/* (the words behind a partial last tile's rows are never read) */
}

/* ---- 2. leaf */
struct rj_leaf_args {
	/* the right table in the two-level fixed-capacity layout of mdb_partition_table (leaves of 2^12 key values; the cell of hv_r[i]
	 * is pay_r[i]): a digit of 2^14 values = 4 consecutive leaves */
	const uint64_t *hv_r;
	const uint64_t *pay_r;
	const uint32_t *cnt_r;
	uint32_t cap_r, shift_r /* 32 - kbits */, rem_r /* 12 */;
	const uint32_t *words_l;
	const uint16_t *offs_l;
	uint32_t ntiles, dbits;
	uint64_t *cells_al;	/* [ntiles * RJ_TILE]: cells_al[i] = the cell of the left row that words_l[i] names */
	uint32_t count_pairs;	/* the first cell's pass counts the joined rows */
	unsigned long long *joined;
	uint32_t *status;
};

#define RJ_LEAF_THREADS 1024
#define RJ_BATCH 256u		/* word indices a wave lists per round */

/* blockIdx -> digit: workgroups are dealt to the 8 XCDs round-robin; XCD x walks the digits [x * D / 8, (x + 1) * D / 8) in
 * order, so the digits that share a line of a tile's words / cells / offsets are neighbours in time on ONE L2 */
__device__ static inline uint32_t rj_digit_of_block(uint32_t b, uint32_t D)
{
	return (b & 7u) * (D >> 3) + (b >> 3);
}

__global__ __launch_bounds__(RJ_LEAF_THREADS) void k_rj_leaf(rj_leaf_args a)
{
	extern __shared__ uint64_t rj_cell[];					/* 2^14 cells */
	uint32_t *const s_occ = reinterpret_cast<uint32_t *>(rj_cell + (1u << RJ_SLOT_BITS));	/* 2^14 bits */
	uint32_t *const s_list = s_occ + (1u << RJ_SLOT_BITS) / 32;		/* per wave: RJ_BATCH word indices */
	__shared__ unsigned long long s_red[RJ_LEAF_THREADS / 64];
	__shared__ uint32_t s_dup;
	const uint32_t D = 1u << a.dbits, d = rj_digit_of_block(blockIdx.x, D);
	const uint32_t lane = mdb_lane(), wave = threadIdx.x >> 6, nwaves = RJ_LEAF_THREADS / 64;
	for (uint32_t w = threadIdx.x; w < (1u << RJ_SLOT_BITS) / 32; w += RJ_LEAF_THREADS)
		s_occ[w] = 0u;
	if (threadIdx.x == 0)
		s_dup = 0u;
	__syncthreads();
	/* build: the digit's right rows, 2^(14 - rem_r) leaves one after the other */
	const uint32_t lpd = 1u << (RJ_SLOT_BITS - a.rem_r), rmask = (1u << a.rem_r) - 1u;
	for (uint32_t q = 0; q < lpd; q++) {
		const uint32_t leaf = d * lpd + q, c0 = a.cnt_r[leaf], c = c0 < a.cap_r ? c0 : a.cap_r;
		const size_t b = (size_t)leaf * a.cap_r;
		for (uint32_t i0 = 0; i0 < c; i0 += 2u * RJ_LEAF_THREADS) {
			const uint32_t i = i0 + 2u * threadIdx.x, ic = i < c ? i : 0u;
			const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.hv_r + b + ic);
			const ulonglong2 p = *reinterpret_cast<const ulonglong2 *>(a.pay_r + b + ic);
#pragma unroll
			for (int k = 0; k < 2; k++)
				if (i + (uint32_t)k < c) {
					const uint32_t slot = (q << a.rem_r) | (((uint32_t)((k ? v.y : v.x) >> 32) >> a.shift_r) & rmask);
					const uint32_t old = atomicOr(&s_occ[slot >> 5], 1u << (slot & 31u));
					if (old & (1u << (slot & 31u)))
						s_dup = 1u;
					rj_cell[slot] = k ? p.y : p.x;
				}
		}
	}
	__syncthreads();
	if (s_dup) {	/* a right key occurs twice */
		if (threadIdx.x == 0)
			mdb_raise(a.status, 32u);
		return;
	}
	/* probe: lane = one left tile's piece of this digit; the pieces' words are listed per wave (RJ_BATCH at a time) so that
	 * consecutive lanes then take consecutive words */
	uint32_t *const list = s_list + wave * RJ_BATCH;
	unsigned long long pairs = 0;
	uint32_t miss = 0;
	const size_t ostride = (size_t)D + 8u;
	for (uint32_t t0 = wave * 64u; t0 < a.ntiles; t0 += nwaves * 64u) {
		const uint32_t t = t0 + lane;
		uint32_t s = 0, len = 0;
		if (t < a.ntiles) {
			const uint16_t *o = a.offs_l + (size_t)t * ostride + d;
			s = o[0];
			len = (uint32_t)o[1] - s;	/* (entry D = the tile's rows) */
		}
		uint32_t done = 0;
		while (__ballot(done < len)) {		/* (uniform over the wave) */
			const uint32_t left = len - done;
			const uint32_t incl = mdb_wave_incl_scan(left), before = incl - left;
			const uint32_t total = (uint32_t)__shfl((int)incl, 63, MDB_WAVE);
			uint32_t take = before < RJ_BATCH ? RJ_BATCH - before : 0u;
			take = take < left ? take : left;
			const uint32_t base = t * RJ_TILE + s + done;
			for (uint32_t j = 0; j < take; j++)
				list[before + j] = base + j;
			done += take;
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_s_waitcnt(0xc07f);	/* lgkmcnt(0): the wave's LDS writes have landed */
			const uint32_t m = total < RJ_BATCH ? total : RJ_BATCH;
			for (uint32_t k = lane; k < m; k += 64u) {
				const uint32_t idx = list[k];
				const uint32_t slot = a.words_l[idx] >> RJ_TILE_BITS;
				if ((s_occ[slot >> 5] >> (slot & 31u)) & 1u) {
					a.cells_al[idx] = rj_cell[slot];
					pairs++;
				} else {
					miss = 1u;
				}
			}
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_s_waitcnt(0xc07f);	/* ... and its reads are done before the list is rewritten */
		}
	}
	if (miss)
		mdb_raise(a.status, 4u);	/* a left row without partner */
	if (a.count_pairs) {
		pairs = lw_block_sum(pairs, s_red);
		if (threadIdx.x == 0 && pairs)
			atomicAdd(a.joined, pairs);
	}
}

/* ---- 3. placement */
struct rj_place_args {
	const uint32_t *words_l;
	const uint64_t *cells_al;
	uint64_t *out;
	uint64_t n;
	uint32_t ntiles;
};

__global__ __launch_bounds__(RJ_THREADS) void k_rj_place(rj_place_args a)
{
	extern __shared__ uint64_t rj_rows[];	/* RJ_TILE / 2 cells */
	/* both halves of a tile on one XCD, one right after the other: the second finds the tile's lines in that L2 */
	const uint32_t b = blockIdx.x, tile = (b >> 4) * 8u + (b & 7u), half = (b >> 3) & 1u;
	if (tile >= a.ntiles)
		return;
	const uint64_t row0 = (uint64_t)tile * RJ_TILE;
	const uint32_t cnt = a.n - row0 < RJ_TILE ? (uint32_t)(a.n - row0) : RJ_TILE;
	if (half * (RJ_TILE / 2) >= cnt)
		return;
	const uint4 *const wsrc = reinterpret_cast<const uint4 *>(a.words_l + row0);
	for (uint32_t i = threadIdx.x; 4u * i < cnt; i += RJ_THREADS) {
		const uint4 w = wsrc[i];
		const uint32_t ws[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const uint32_t r = ws[e] & (RJ_TILE - 1u);
			if (4u * i + (uint32_t)e < cnt && (r >> (RJ_TILE_BITS - 1u)) == half)
				rj_rows[r & (RJ_TILE / 2 - 1u)] = a.cells_al[row0 + 4u * i + (uint32_t)e];
		}
	}
	__syncthreads();
	const uint32_t r0 = half * (RJ_TILE / 2), m = cnt - r0 < RJ_TILE / 2 ? cnt - r0 : RJ_TILE / 2;
	ulonglong2 *const dst = reinterpret_cast<ulonglong2 *>(a.out + row0 + r0);
	const ulonglong2 *const st = reinterpret_cast<const ulonglong2 *>(rj_rows);
	for (uint32_t i = threadIdx.x; 2u * i + 1u < m; i += RJ_THREADS)
		dst[i] = st[i];
	if ((m & 1u) && threadIdx.x == 0)
		a.out[row0 + r0 + m - 1u] = rj_rows[m - 1u];
}

/* ---- host */
bool mdb_rowjoin_serves(uint64_t n_l, uint64_t n_r, uint32_t kbits, const void *keys_l, const void *null_l, void *const *out, int npay)
{
	if (null_l)	/* (a NULL left key has no partner: not this operator's join) */
		return false;
	if (getenv("MDB_ROWJOIN") && getenv("MDB_ROWJOIN")[0] == '0')
		return false;
	if (kbits < RJ_SLOT_BITS + 11u || kbits > RJ_SLOT_BITS + RJ_MAX_DBITS)	/* 2048 ... 8192 digits (windows of 2^25 ... 2^27 values) */
		return false;
	if (n_l >= 0xF0000000ull || n_r >= 0xF0000000ull || ((uintptr_t)keys_l & 15u))
		return false;
	for (int c = 0; c < npay; c++)
		if ((uintptr_t)out[c] & 15u)
			return false;
	return true;
}

size_t mdb_rowjoin_arena_bytes(uint64_t n_l, uint32_t kbits)
{
	const uint32_t dbits = kbits - RJ_SLOT_BITS;
	const uint64_t ntiles = (n_l + RJ_TILE - 1) / RJ_TILE;
	return mdb_align_up(ntiles * RJ_TILE * 4) + mdb_align_up(ntiles * (((size_t)1 << dbits) + 8u) * 2) + mdb_align_up(ntiles * RJ_TILE * 8) + 4096;
}

/* the right table has been partitioned (pr: two levels, leaves of 2^12 values, cells beside the words); the arena holds
 * mdb_rowjoin_arena_bytes() more; ctx->d_status has been cleared by the caller.  Queues everything; no host sync.  Flags in
 * d_status[0]: 4 a left row without partner, 32 duplicate right key, 64 NULL left key, 128 key outside the window; joined rows
 * (u64) at d_status[2]. */
int mdb_rowjoin_run(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, int64_t win_lo, uint32_t kbits,
		    const mdb_part_result *pr, uint32_t rem_r, int npay, void *const *out)
{
	const uint32_t dbits = kbits - RJ_SLOT_BITS, D = 1u << dbits;
	const uint32_t ntiles = (uint32_t)((n_l + RJ_TILE - 1) / RJ_TILE);
	uint32_t *words = (uint32_t *)mdb_arena_take(ctx, (size_t)ntiles * RJ_TILE * 4);
	uint16_t *offs = (uint16_t *)mdb_arena_take(ctx, (size_t)ntiles * (D + 8u) * 2);
	uint64_t *cells_al = (uint64_t *)mdb_arena_take(ctx, (size_t)ntiles * RJ_TILE * 8);
	if (!words || !offs || !cells_al)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %s", ctx->err);
	rj_sort_args sa;
	memset(&sa, 0, sizeof(sa));
	sa.keys = keys_l;
	sa.n = n_l;
	sa.base = win_lo;
	sa.kbits = kbits;
	sa.dbits = dbits;
	sa.words = words;
	sa.offs = offs;
	sa.status = ctx->d_status;
	const size_t lds_sort = (size_t)(D >> 1) * 4 + (size_t)RJ_TILE * 4;
	const uint32_t nfull = (uint32_t)(n_l / RJ_TILE);
	if (nfull) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_tile_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
		MDB_LAUNCH_LDS(ctx, "rowjoin_tile_sort", k_rj_tile_sort<true>, nfull, RJ_THREADS, lds_sort, sa, 0u);
	}
	if (nfull < ntiles) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_tile_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
		MDB_LAUNCH_LDS(ctx, "rowjoin_tile_sort", k_rj_tile_sort<false>, 1u, RJ_THREADS, lds_sort, sa, nfull);
	}
	const size_t lds_leaf = ((size_t)8 << RJ_SLOT_BITS) + ((size_t)1 << RJ_SLOT_BITS) / 8 + (size_t)(RJ_LEAF_THREADS / 64) * RJ_BATCH * 4;
	const size_t lds_place = (size_t)(RJ_TILE / 2) * 8;
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_leaf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_leaf));
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_place), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_place));
	for (int c = 0; c < npay; c++) {
		rj_leaf_args la;
		memset(&la, 0, sizeof(la));
		la.hv_r = pr->hv;
		la.pay_r = pr->pay[c];
		la.cnt_r = pr->leaf_cnt;
		la.cap_r = pr->leaf_cap;
		la.shift_r = 32u - kbits;
		la.rem_r = rem_r;
		la.words_l = words;
		la.offs_l = offs;
		la.ntiles = ntiles;
		la.dbits = dbits;
		la.cells_al = cells_al;
		la.count_pairs = c == 0;
		la.joined = (unsigned long long *)(ctx->d_status + 2);
		la.status = ctx->d_status;
		MDB_LAUNCH_LDS(ctx, "rowjoin_leaf", k_rj_leaf, D, RJ_LEAF_THREADS, lds_leaf, la);
		rj_place_args pa;
		memset(&pa, 0, sizeof(pa));
		pa.words_l = words;
		pa.cells_al = cells_al;
		pa.out = reinterpret_cast<uint64_t *>(out[c]);
		pa.n = n_l;
		pa.ntiles = ntiles;
		const uint32_t grid = ((ntiles + 7u) / 8u) * 16u;
		MDB_LAUNCH_LDS(ctx, "rowjoin_place", k_rj_place, grid, RJ_THREADS, lds_place, pa);
	}
	return MIDORIDB_OK;
}
