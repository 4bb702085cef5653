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
 *      offsets (u16).  Sequential reads, sequential writes, no global atomics, nothing that can overflow.  A right table is sorted the
 *      same way into 12-byte records { word, payload cell } - one record array per payload column (round 6): a (tile, digit) piece of a
 *      right table is ONE request for the leaf.  The offsets are transposed and paired (start | end << 16 per digit and tile).
 *   2. leaf (k_rj_leaf): workgroups that stay on their CU and walk their XCD's key digits of 2^14 key values; per digit one round per
 *      (right table, payload column) stream - up to four: several primary-key tables joined on one key in ONE call,
 *      mdb_dev_join_payload_multi -: the stream's cells of the digit in an LDS table (direct-addressed by the slot bits: no stored keys,
 *      no probing; an occupancy bitmap sees duplicate right keys and left rows without partner), then the digit's piece of EVERY left
 *      tile - a few words each, found through the offsets - is looked up and the cell written at the SAME position of a cell array laid
 *      out like the word array (cells_al[s][tile * 32768 + p]): consecutive lanes take consecutive words of a piece, so a piece is one
 *      request each way, and adjacent digits - neighbours on the same XCD - complete each other's lines in that XCD's L2.
 *   3. placement (k_rj_place): one workgroup per half tile reads the tile's words once and, per carried column, its cells (sequential),
 *      drops every cell at LDS[row inside the half tile] and writes the result column coalesced.
 *
 * Per joined row and cell: 8 B key + 4 B word + 12 B record written and read + 4 B word re-read + 8 B cell written, read, and written
 * again in row order - all of it in whole lines.  What bounds the leaf was measured in round 6 (profiles/r06/README.md section 7): the
 * instructions it issues (vector memory and LDS), not requests, latency or the TLB.
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"

#define RJ_TILE 32768u		/* left rows per tile: 15 bits of a word name the row inside its tile */
#define RJ_TILE_BITS 15u
#ifndef RJ_SKEW
#define RJ_SKEW 80u		/* words (cells) between the end of one tile's block and the start of the next: see RJ_STRIDE */
#endif
/* A tile's block of the word / cell arrays starts at tile * RJ_STRIDE: blocks exactly 2^17 bytes apart would put digit d's piece of
 * EVERY tile - the same place in each block, give or take a few words - on the same L2 / memory channel, and a leaf workgroup asks for
 * exactly those, 3052 of them in a row */
#define RJ_STRIDE (RJ_TILE + RJ_SKEW)
#define RJ_THREADS 1024
#define RJ_ITEMS (RJ_TILE / RJ_THREADS)		/* 32 rows per thread */
#define RJ_SLOT_BITS 14u	/* key values per digit: 2^14 eight-byte cells = 128 KiB of LDS */
#define RJ_MAX_DBITS 13u	/* up to 8192 digits per tile (16 KiB of packed 16-bit counters) */
#define RJ_LOADS 8		/* 16-byte key loads a thread of the tile sort issues before it consumes the first */

struct rj_rec {
	uint32_t word, lo, hi;	/* slot inside the digit << 15 | row inside the tile; the cell's halves */
};

struct rj_sort_args {
	const int64_t *keys;
	uint64_t n;
	int64_t base;		/* window [base, base + 2^kbits) */
	uint32_t kbits, dbits;
	uint32_t *words;	/* [ntiles * RJ_STRIDE]: slot inside the digit << 15 | row inside the tile */
	uint16_t *offs;		/* [ntiles * (D + 8)]: digit starts inside the tile, entry D = rows of the tile */
	/* CELLS (a right table): up to two payload columns travel with the words - recs[c][tile * RJ_STRIDE + p] = { word p of the tile, the
	 * row's cell of column c }: ONE 12-byte record per row and column, so that the leaf asks for a (tile, digit) piece of a right table
	 * once instead of once for the words and once for the cells (it is bound by requests, not bytes: round 6); `words` is not written */
	const uint64_t *pay_in[2];
	rj_rec *recs[2];
	uint32_t npay;
	uint32_t *status;
};

/* ---- 1. tile sort.  FULL: a tile of exactly RJ_TILE rows - straight-line code, no per-row branch (the generic form spends a dozen scalar
 * branches per row and spills); the table's last, partial tile takes the other instance */
template <bool FULL, bool CELLS>
__global__ __launch_bounds__(RJ_THREADS) void k_rj_tile_sort(rj_sort_args a, uint32_t tile0)
{
	extern __shared__ uint32_t rj_lds[];
	const uint32_t D = 1u << a.dbits, rem = a.kbits - a.dbits, smask = (1u << rem) - 1u;
	uint32_t *const s_cnt = rj_lds;				/* D / 2 words: two 16-bit counters each, then the digit starts */
	uint32_t *const s_stage = rj_lds + (D >> 1);		/* RJ_TILE words */
	__shared__ uint32_t s_tmp[32];
	const uint32_t tile = tile0 + blockIdx.x;
	const uint64_t row0 = (uint64_t)tile * RJ_TILE, blk0 = (uint64_t)tile * RJ_STRIDE;
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
	/* exclusive scan of the D counters: thread t owns 1, 2 or 4 consecutive words of two counters each (D = 8 ... 8192; with fewer than
	 * 2048 digits the first D / 2 threads own one word each) */
	{
		const uint32_t nw = D >> 1, wpt = nw >= RJ_THREADS ? nw / RJ_THREADS : 1u, w0 = threadIdx.x * wpt;
		const bool mine = w0 < nw;
		uint32_t c[8], sum = 0;
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			if (i < wpt && mine) {
				const uint32_t w = s_cnt[w0 + i];
				c[2 * i] = w & 0xFFFFu;
				c[2 * i + 1] = w >> 16;
				sum += c[2 * i] + c[2 * i + 1];
			}
		uint32_t total;
		uint32_t run = mdb_block_excl_scan(sum, s_tmp, &total);
		uint16_t *const og = a.offs + (size_t)tile * (D + 8u) + (size_t)w0 * 2u;
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			if (i < wpt && mine) {
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
				if (CELLS)
					hr[2 * j + e] = ((h & smask) << RJ_TILE_BITS) | pos;	/* (slot | place: the row's records are built from it below) */
				else
					s_stage[pos] = ((h & smask) << RJ_TILE_BITS) | r;
			}
		}
	}
	__syncthreads();
	if (!CELLS) {
		uint4 *const dst = reinterpret_cast<uint4 *>(a.words + blk0);
		const uint4 *const st = reinterpret_cast<const uint4 *>(s_stage);
		for (uint32_t i = threadIdx.x; 4u * i < cnt; i += RJ_THREADS)
			dst[i] = st[i];		/* (the words behind a partial last tile's rows are never read) */
		return;
	}
	/* the records: a column's cells loaded in row order like the keys, every row's { word, cell } dropped at its place in LDS - a quarter of
	 * the tile's places at a time (8192 x 12 bytes of the staging area): four rounds per column -, written out in order, 16 bytes per lane
	 * (a quarter's 98 304 bytes and a tile's block start are multiples of 16) */
	rj_rec *const s_rec = reinterpret_cast<rj_rec *>(s_stage);
	constexpr uint32_t RJ_Q = RJ_TILE / 4, RJ_QBITS = RJ_TILE_BITS - 2u;
#pragma unroll 1
	for (uint32_t c = 0; c < a.npay; c++) {		/* (uniform) */
		const ulonglong2 *csrc = reinterpret_cast<const ulonglong2 *>(a.pay_in[c] + row0);
		ulonglong2 cv[RJ_ITEMS / 2];
#pragma unroll
		for (int j = 0; j < (int)RJ_ITEMS / 2; j++) {
			const uint32_t p = (uint32_t)j * RJ_THREADS + threadIdx.x;
			if (FULL) {
				cv[j] = csrc[p];
			} else {
				cv[j] = make_ulonglong2(0ull, 0ull);
				if (2u * p < cnt)
					cv[j].x = a.pay_in[c][row0 + 2u * p];
				if (2u * p + 1u < cnt)
					cv[j].y = a.pay_in[c][row0 + 2u * p + 1u];
			}
		}
#pragma unroll 1
		for (uint32_t round = 0; round < 4u; round++) {
			__syncthreads();	/* (the staging area is free: its last readers are done) */
			/* (what a round derives from a row's slot | place - address, word, predicate - is computed in the round: kept across
			 * the rounds it is three more values per row, and the kernel spills) */
#pragma unroll
			for (int i = 0; i < (int)RJ_ITEMS; i++)
				asm volatile("" : "+v"(hr[i]));
#pragma unroll
			for (int j = 0; j < (int)RJ_ITEMS / 2; j++) {
#pragma unroll
				for (int e = 0; e < 2; e++) {
					const uint32_t r = 2u * ((uint32_t)j * RJ_THREADS + threadIdx.x) + (uint32_t)e;
					const uint32_t sp = hr[2 * j + e], pos = sp & (RJ_TILE - 1u);
					if ((FULL || r < cnt) && (pos >> RJ_QBITS) == round) {
						const uint64_t cell = e ? cv[j].y : cv[j].x;
						rj_rec rec;
						rec.word = (sp & ~(RJ_TILE - 1u)) | r;
						rec.lo = (uint32_t)cell;
						rec.hi = (uint32_t)(cell >> 32);
						s_rec[pos & (RJ_Q - 1u)] = rec;
					}
				}
			}
			__syncthreads();
			const uint32_t p0 = round * RJ_Q;
			if (p0 < cnt) {
				const uint32_t m = cnt - p0 < RJ_Q ? cnt - p0 : RJ_Q;	/* records of this round: 3 m words, 16 bytes per lane and step */
				uint4 *const rd = reinterpret_cast<uint4 *>(a.recs[c] + blk0 + p0);
				const uint4 *const st = reinterpret_cast<const uint4 *>(s_rec);
				for (uint32_t i = threadIdx.x; 4u * i < 3u * m; i += RJ_THREADS)
					rd[i] = st[i];	/* (a last odd piece takes undefined words along: inside the tile's block - RJ_SKEW) */
			}
		}
	}
}

/* ---- 1b. the tiles' digit offsets transposed and paired: offT[d * tstride + t] = start of digit d in tile t | its end << 16 (= the start of
 * digit d + 1; entry D = the tile's rows), d = 0 .. D - 1 - so that a leaf workgroup reads its digit's pieces of ALL tiles as one contiguous
 * run of 4-byte words instead of one request per tile, and a lane holds a piece's place in ONE register */
__global__ __launch_bounds__(256) void k_rj_transpose_offs(const uint16_t *offs, uint32_t ntiles, uint32_t D, uint32_t tstride, uint32_t *offT)
{
	__shared__ uint16_t s_t[64][66];
	const uint32_t t0 = blockIdx.x * 64u, d0 = blockIdx.y * 64u;
	for (uint32_t i = threadIdx.x; i < 64u * 65u; i += 256u) {
		const uint32_t tt = i / 65u, dd = i % 65u;
		s_t[tt][dd] = (t0 + tt < ntiles && d0 + dd <= D) ? offs[(size_t)(t0 + tt) * (D + 8u) + d0 + dd] : (uint16_t)0;
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < 64u * 64u; i += 256u) {
		const uint32_t dd = i >> 6, tt = i & 63u;
		if (d0 + dd < D && t0 + tt < tstride)
			offT[(size_t)(d0 + dd) * tstride + t0 + tt] = (uint32_t)s_t[tt][dd] | ((uint32_t)s_t[tt][dd + 1u] << 16);
	}
}

/* ---- 2. leaf */
#define RJ_MAX_STREAMS 4	/* (right table, payload column) pairs one leaf launch serves: two tables of two cells each */
struct rj_leaf_args {
	/* the right tables' payload columns, one STREAM each: sorted tile by tile like the left table, { word, cell } per row */
	uint32_t nstreams;
	const rj_rec *recs[RJ_MAX_STREAMS];
	const uint32_t *offT_r[RJ_MAX_STREAMS];
	uint32_t ntiles_r[RJ_MAX_STREAMS], tstride_r[RJ_MAX_STREAMS];
	uint64_t *cells_al[RJ_MAX_STREAMS];	/* [ntiles * RJ_STRIDE] each: cells_al[s][i] = stream s's cell of the left row that words_l[i] names */
	const uint32_t *words_l;
	const uint32_t *offT_l;
	uint32_t ntiles, tstride, dbits, sbits /* key values per digit: 2^sbits <= 2^14 */;
	uint32_t ablate;	/* measurement only (MDB_RJ_ABLATE): 1 no build, 2 no probe, 4 no cell stores, 8 no right cells read */
	unsigned long long *trace;	/* measurement only (MDB_RJ_TRACE): workgroup 8's wave 0 leaves wall_clock64() stamps at its phase boundaries */
	unsigned long long *joined;	/* += (left row, stream) pairs served: nstreams x the left rows when every row found its partner in every table */
	uint32_t *status;
};

#define RJ_LEAF_THREADS 1024

/* blockIdx -> digit: workgroups are dealt to the 8 XCDs round-robin; XCD x walks the digits [x * D / 8, (x + 1) * D / 8) in
 * order, so the digits that share a line of a tile's words / cells are neighbours in time on ONE L2 */
__device__ static inline uint32_t rj_digit_of_block(uint32_t b, uint32_t D)
{
	return (b & 7u) * (D >> 3) + (b >> 3);
}

/* Digit d's pieces over all tiles of a tile-sorted table.  A wave takes 32 / LPP x 64 tiles per sweep: lane = one tile's piece in each of
 * the 32 / LPP groups (start and end: coalesced loads from the transposed offsets); then LPP consecutive lanes take one piece together -
 * consecutive lanes, consecutive words: a piece is one request each way, 64 / LPP pieces per instruction; the piece's place comes from
 * its lane through the wave's crossbar (no LDS memory).  LPP = twice the average piece, so most pieces are done in one step.
 * The kernel is a chain of dependent memory round trips (offsets -> words -> LDS -> store), so what matters is how many loads a lane
 * has in flight: `load(unit, i)` is called for UNITS pieces' words first (it only ISSUES loads into the caller's registers), then
 * `use(unit, i)` for each of them; words beyond a piece's first LPP take the same two calls, tier by tier (below).
 * All lanes of the wave call it together. */
template <int LPP, int UNITS, bool PER_SWEEP = false /* the lane-derived terms are worked out again in every sweep (callers whose own state leaves them no registers: kept across the
							  * sweeps they are spilled and reloaded BETWEEN the loads of a batch, behind s_waitcnt vmcnt(0) - each such load waits for the one before) */,
	  typename FL, typename FU, typename FD>
__device__ static inline void rj_for_pieces(const uint32_t *offT, uint32_t tstride, uint32_t ntiles, uint32_t d, FL load, FU use, FD done /* after the `use`s of a batch */,
					    bool skip_slow = false /* measurement only: pieces are cut off after LPP words */)
{
	constexpr int RJ_G = LPP >= 32 ? 1 : LPP == 16 ? 2 : 4;	/* tile groups per sweep: 16 (LPP 4), 32 or 64 pieces' steps per lane and sweep */
	static_assert(LPP >= 4 && LPP <= 64 && (RJ_G * LPP) % UNITS == 0, "units per sweep");
	uint32_t lane = mdb_lane();
	/* (called once per stream and phase from a loop: what is derived from the lane - the crossbar's source lanes of every step - is derived
	 * HERE, per call; hoisted out of the caller's loop for all its calls at once it costs two dozen registers and the loads' answers spill) */
	asm volatile("" : "+v"(lane));
	const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	const uint32_t *const o0 = offT + (size_t)d * tstride;
#pragma unroll 1
	for (uint32_t t0 = wave * 64u; t0 < ntiles; t0 += nwaves * 64u * RJ_G) {
		if (PER_SWEEP)
			asm volatile("" : "+v"(lane));
		/* a piece's start | end << 16 stays ONE word until the lanes that walk the piece have it: one crossbar read per step, not two - the
		 * leaf is bound by what it asks of the LDS pipeline (~20 cycles per wave-instruction and CU, profiles/r06/lds_atomics.txt: crossbar
		 * reads, table atomics, cell reads), round 6 */
		uint32_t se[RJ_G];
#pragma unroll
		for (int g = 0; g < RJ_G; g++) {
			const uint32_t t = t0 + (uint32_t)g * nwaves * 64u + lane;	/* (the rows are padded to a multiple of 64 tiles: zeros) */
			se[g] = t < tstride ? o0[t] : 0u;
		}
		uint32_t longest = 0;
#pragma unroll
		for (int u0 = 0; u0 < RJ_G * LPP; u0 += UNITS) {
			uint32_t idx[UNITS];
			bool on[UNITS];
#pragma unroll
			for (int u = 0; u < UNITS; u++) {
				const int g = (u0 + u) / LPP, sb = (u0 + u) % LPP;
				const int src = sb * (64 / LPP) + (int)(lane / LPP);
				/* (measured, not kept: every lane reading its piece's offsets word itself - 8 lanes one address, L1 hits - instead of this
				 * crossbar read: 1.40 -> 1.82 ms per stream; the kernel is short of vector memory instructions before it is short of LDS ones) */
				const uint32_t pse = (uint32_t)__shfl((int)se[g], src, MDB_WAVE), ps = pse & 0xFFFFu, pl = (pse >> 16) - ps;
				idx[u] = (t0 + (uint32_t)g * nwaves * 64u + (uint32_t)src) * RJ_STRIDE + ps + lane % LPP;
				on[u] = lane % LPP < pl;
				longest = pl > longest ? pl : longest;
				if (on[u])
					load(u, idx[u]);
			}
#pragma unroll
			for (int u = 0; u < UNITS; u++)
				if (on[u])
					use(u, idx[u]);
			done();
		}
		if (!skip_slow && __any(longest > (uint32_t)LPP)) {
			/* pieces longer than LPP words (one in 50 at an average of LPP / 2, more than a third at an average of LPP: nearly every sweep
			 * has some).  Tier by tier - words [tier x LPP, (tier + 1) x LPP) of the pieces that have them -, ALL tile groups together
			 * (round 6): per group the owners of such pieces are a ballot; lane group q takes the q-th of them - found by clearing q bits of
			 * the mask, no LDS -, one crossbar read for its offsets word, and the groups' loads go out before the first is used: a round trip
			 * per 64 / LPP pieces and group where walking the 32 steps again, one dependent load at a time, cost a fifth of the leaf
			 * (profiles/r06/rj_ablate.txt: 1.39 ms with, 1.08 without the long pieces; 1.24 with this) */
			constexpr int NQ = 64 / LPP;
			static_assert(RJ_G <= UNITS, "a load slot per tile group");
#pragma unroll 1
			for (uint32_t tier = 1; __any(tier * (uint32_t)LPP < longest); tier++) {
				unsigned long long lm[RJ_G];
				bool more = false;
#pragma unroll
				for (int g = 0; g < RJ_G; g++) {
					lm[g] = __ballot((se[g] >> 16) - (se[g] & 0xFFFFu) > tier * (uint32_t)LPP);
					more = more || lm[g] != 0ull;
				}
				if (!more)
					break;
#pragma unroll 1
				while (more) {
					uint32_t lidx[RJ_G];
					bool lon[RJ_G];
					more = false;
#pragma unroll
					for (int g = 0; g < RJ_G; g++) {
						unsigned long long m = lm[g];
#pragma unroll
						for (int i = 0; i < NQ - 1; i++)
							if ((int)(lane / LPP) > i)
								m &= m - 1ull;
						const int j = m ? __ffsll((long long)m) - 1 : 0;
						const uint32_t pse = (uint32_t)__shfl((int)se[g], j, MDB_WAVE), ps = pse & 0xFFFFu, pl = (pse >> 16) - ps;
						lidx[g] = (t0 + (uint32_t)g * nwaves * 64u + (uint32_t)j) * RJ_STRIDE + ps + tier * (uint32_t)LPP + lane % LPP;
						lon[g] = m && tier * (uint32_t)LPP + lane % LPP < pl;
						if (lon[g])
							load(g, lidx[g]);
						/* (the NQ pieces this pass took leave the mask) */
#pragma unroll
						for (int i = 0; i < NQ; i++)
							lm[g] &= lm[g] - 1ull;
						more = more || lm[g] != 0ull;
					}
#pragma unroll
					for (int g = 0; g < RJ_G; g++)
						if (lon[g])
							use(g, lidx[g]);
					done();
				}
			}
		}
	}
}

template <int LPP, int UNITS, bool PER_SWEEP = false, typename FL, typename FU>
__device__ static inline void rj_for_pieces(const uint32_t *offT, uint32_t tstride, uint32_t ntiles, uint32_t d, FL load, FU use, bool skip_slow = false)
{
	rj_for_pieces<LPP, UNITS, PER_SWEEP>(offT, tstride, ntiles, d, load, use, [] {}, skip_slow);
}

/* The same for pieces of 32 words and more (windows of up to 2^24 values: 1024 digits and fewer): the whole wave walks one piece after
 * the other, 64 x UNITS words of it in flight (LPP = 64 selects it) */
template <int UNITS, typename FL, typename FU>
__device__ static inline void rj_for_long_pieces(const uint32_t *offT, uint32_t tstride, uint32_t ntiles, uint32_t d, FL load, FU use)
{
	const uint32_t lane = mdb_lane(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	const uint32_t *const o0 = offT + (size_t)d * tstride;
	for (uint32_t t0 = wave * 64u; t0 < ntiles; t0 += nwaves * 64u) {
		const uint32_t t = t0 + lane;		/* (the rows are padded to a multiple of 64 tiles: zeros) */
		const uint32_t se = o0[t], s = se & 0xFFFFu, len = (se >> 16) - s;
		for (int j = 0; j < 64; j++) {
			const uint32_t ps = (uint32_t)__builtin_amdgcn_readlane((int)s, j), pl = (uint32_t)__builtin_amdgcn_readlane((int)len, j);
			const uint32_t base = (t0 + (uint32_t)j) * RJ_STRIDE + ps;
			for (uint32_t k0 = 0; k0 < pl; k0 += 64u * UNITS) {		/* (uniform) */
#pragma unroll
				for (int u = 0; u < UNITS; u++)
					if (k0 + (uint32_t)u * 64u + lane < pl)
						load(u, base + k0 + (uint32_t)u * 64u + lane);
#pragma unroll
				for (int u = 0; u < UNITS; u++)
					if (k0 + (uint32_t)u * 64u + lane < pl)
						use(u, base + k0 + (uint32_t)u * 64u + lane);
			}
		}
	}
}

/* a barrier that waits for the wave's LDS operations only: the leaf's phases hand each other LDS contents, never global memory - the
 * cell stores stay in flight across it (__syncthreads() waits for every outstanding memory operation) */
__device__ static inline void rj_barrier(void)
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int LPP, bool LONG>
__global__ __launch_bounds__(RJ_LEAF_THREADS) void k_rj_leaf(rj_leaf_args a)
{
	extern __shared__ uint64_t rj_cell[];					/* 2^sbits cells */
	uint32_t *const s_occ = reinterpret_cast<uint32_t *>(rj_cell + (1u << a.sbits));	/* 2^sbits bits (at least one word) */
	__shared__ unsigned long long s_red[RJ_LEAF_THREADS / 64];
	__shared__ uint32_t s_dup;
	/* blockIdx -> digits: workgroups are dealt to the 8 XCDs round-robin; XCD x walks the digits [x * D / 8, (x + 1) * D / 8) in order, its
	 * `per` workgroups taking consecutive ones - the digits that share a line of a tile's words / cells are neighbours in time on ONE L2.
	 * A workgroup stays and takes its XCD's next digit (round 6: a workgroup's launch - 16 waves, 130 KiB of LDS that the one before must
	 * have given back - is not paid 32 times per CU) */
	const uint32_t D = 1u << a.dbits, xcd = blockIdx.x & 7u, per = gridDim.x >> 3;
	uint32_t pairs = 0;	/* (a thread serves fewer than 2^32 rows) */
	uint32_t miss = 0;
	uint32_t unset = 0;	/* right rows put - occupancy bits seen set (summed over the workgroup's threads: the right rows whose key had been put before) */
	bool first = true;
	if (threadIdx.x == 0)
		s_dup = 0u;
	constexpr int LP = LPP == 64 ? 32 : LPP;	/* (lanes per piece of the piece walker; LPP 64: the build walks whole-wave pieces) */
	uint32_t tr_n = 0;
	const bool tracing = a.trace && blockIdx.x == 8u && threadIdx.x == 0;
#define RJ_STAMP()                                              \
	do {                                                    \
		if (tracing && tr_n < 60u)                      \
			a.trace[tr_n++] = wall_clock64();       \
	} while (0)
#pragma unroll 1
	for (uint32_t dl = blockIdx.x >> 3; dl < (D >> 3); dl += per) {
	const uint32_t d = xcd * (D >> 3) + dl;
	/* one round per stream: the digit's cells of that right column into the table, then the digit's left rows looked up (a second
	 * stream's round finds the left words and offsets of the first one in its XCD's L2) */
#pragma unroll 1
	for (uint32_t s = 0; s < a.nstreams; s++) {		/* (uniform) */
		if (!first)
			rj_barrier();	/* (the last round's lookups are done) */
		first = false;
		RJ_STAMP();	/* 0: round begins */
		for (uint32_t w = threadIdx.x; w < ((1u << a.sbits) + 31u) / 32u; w += blockDim.x)
			s_occ[w] = (a.ablate & 1u) ? 0xFFFFFFFFu : 0u;
		rj_barrier();
		RJ_STAMP();	/* 1: table cleared */
		/* build: the digit's right rows - their cells dropped at their slots.  A right key twice: two rows, one bit - the bits set are
		 * counted against the rows put once the table stands (no atomic that has to come back with the old word), the difference kept
		 * until the workgroup leaves: what a call with such a table wrote is not used */
		{
			if (!(a.ablate & 1u)) {
				constexpr int UB = 8;	/* (16 measured: 8 spills, 1.26 against 1.24 ms) */
				rj_rec rv[UB] = {};	/* (defined here on every path: not carried around the loops as "whatever they held") */
				const rj_rec *const recs = a.recs[s];
				auto put = [&](const rj_rec &r) {
					const uint32_t slot = r.word >> RJ_TILE_BITS;
					atomicOr(&s_occ[slot >> 5], 1u << (slot & 31u));
					unset++;
					rj_cell[slot] = (a.ablate & 8u) ? 0ull : ((uint64_t)r.hi << 32 | r.lo);
				};
				auto ld = [&](int u, uint32_t idx) { rv[u] = recs[idx]; };
				auto us = [&](int u, uint32_t) { put(rv[u]); };
				if (LPP == 64)
					rj_for_long_pieces<UB>(a.offT_r[s], a.tstride_r[s], a.ntiles_r[s], d, ld, us);
				else
					rj_for_pieces<LP, UB, (LPP != 8)>(a.offT_r[s], a.tstride_r[s], a.ntiles_r[s], d, ld, us, (a.ablate & 16u) != 0);
			}
		}
		RJ_STAMP();	/* 2: this wave's share of the build done */
		rj_barrier();
		RJ_STAMP();	/* 3: every wave's */
		for (uint32_t w = threadIdx.x; w < ((1u << a.sbits) + 31u) / 32u && !(a.ablate & 1u); w += blockDim.x)
			unset -= (uint32_t)__popc(s_occ[w]);
		/* probe: every left row of the digit picks its partner's cell up and leaves it at its word's place */
		{
			constexpr int UP = 16;
			uint32_t w[UP] = {};
			uint64_t *const cells_al = a.cells_al[s];
			auto take = [&](uint32_t word, uint32_t idx) {
				const uint32_t slot = word >> RJ_TILE_BITS;
				if ((s_occ[slot >> 5] >> (slot & 31u)) & 1u) {
					if (!(a.ablate & 4u))
						cells_al[idx] = rj_cell[slot];
					pairs++;
				} else {
					miss = 1u;
				}
			};
			auto ld = [&](int u, uint32_t idx) { w[u] = (a.ablate & 2u) ? 0u : a.words_l[idx]; };
			auto us = [&](int u, uint32_t idx) {
				if (a.ablate & 2u)
					pairs++;
				else
					take(w[u], idx);
			};
			if (LONG)
				rj_for_long_pieces<UP>(a.offT_l, a.tstride, a.ntiles, d, ld, us);
			else
				rj_for_pieces<LPP, UP, (LPP != 8)>(a.offT_l, a.tstride, a.ntiles, d, ld, us, (a.ablate & 16u) != 0);
		}
		RJ_STAMP();	/* 4: this wave's share of the probe done */
	}
	}
#undef RJ_STAMP
	if (miss)
		mdb_raise(a.status, 4u);	/* a left row without partner */
	{
#pragma unroll
		for (int o = 32; o; o >>= 1)
			unset += (uint32_t)__shfl_down((int)unset, o, MDB_WAVE);
		if (mdb_lane() == 0 && unset)
			atomicAdd(&s_dup, unset);
	}
	unsigned long long wpairs = pairs;
#pragma unroll
	for (int o = 32; o; o >>= 1)
		wpairs += __shfl_down(wpairs, o, MDB_WAVE);
	if (mdb_lane() == 0)
		s_red[threadIdx.x >> 6] = wpairs;
	rj_barrier();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
		for (uint32_t w = 0; w < (blockDim.x >> 6); w++)
			t += s_red[w];
		if (t)
			atomicAdd(a.joined, t);
		if (s_dup)	/* a right key occurs twice */
			mdb_raise(a.status, 32u);
	}
}

/* ---- 3. placement */
struct rj_place_args {
	const uint32_t *words_l;
	uint32_t ncols;
	const uint64_t *cells_al[RJ_MAX_STREAMS];
	uint64_t *out[RJ_MAX_STREAMS];
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
	const uint64_t row0 = (uint64_t)tile * RJ_TILE, blk0 = (uint64_t)tile * RJ_STRIDE;
	const uint32_t cnt = a.n - row0 < RJ_TILE ? (uint32_t)(a.n - row0) : RJ_TILE;
	if (half * (RJ_TILE / 2) >= cnt)
		return;
	/* four words and their four cells per lane and step: 16-byte loads, every cell of the tile read by both halves' workgroups (the second
	 * finds the lines in its XCD's L2) - a vector memory instruction costs its ~70 cycles whatever it carries (profiles/r05/piece_loads.txt):
	 * 8-byte loads of the one half's cells alone were twice the instructions (0.45 -> 0.30 ms per 10^8 rows).  The tile's words are read
	 * ONCE for all the columns the join carries (round 6) */
	const uint4 *const wsrc = reinterpret_cast<const uint4 *>(a.words_l + blk0);
	uint4 wv[RJ_ITEMS / 4];
#pragma unroll
	for (int k = 0; k < (int)RJ_ITEMS / 4; k++) {
		const uint32_t i = (uint32_t)k * RJ_THREADS + threadIdx.x;
		wv[k] = 4u * i < cnt ? wsrc[i] : make_uint4(0u, 0u, 0u, 0u);
	}
	const uint32_t r0 = half * (RJ_TILE / 2), m = cnt - r0 < RJ_TILE / 2 ? cnt - r0 : RJ_TILE / 2;
	for (uint32_t c = 0; c < a.ncols; c++) {	/* (uniform) */
		const ulonglong2 *const csrc = reinterpret_cast<const ulonglong2 *>(a.cells_al[c] + blk0);
		if (c)
			__syncthreads();	/* (the last column has left the staging area) */
#pragma unroll
		for (int k = 0; k < (int)RJ_ITEMS / 4; k++) {
			const uint32_t i = (uint32_t)k * RJ_THREADS + threadIdx.x;
			if (4u * i < cnt) {
				const ulonglong2 c01 = csrc[2u * i], c23 = csrc[2u * i + 1u];	/* (inside the tile's block: RJ_STRIDE leaves room behind a partial tile) */
				const uint32_t ws[4] = { wv[k].x, wv[k].y, wv[k].z, wv[k].w };
				const uint64_t cs[4] = { c01.x, c01.y, c23.x, c23.y };
#pragma unroll
				for (int e = 0; e < 4; e++) {
					const uint32_t r = ws[e] & (RJ_TILE - 1u);
					if (4u * i + (uint32_t)e < cnt && (r >> (RJ_TILE_BITS - 1u)) == half)
						rj_rows[r & (RJ_TILE / 2 - 1u)] = cs[e];
				}
			}
		}
		__syncthreads();
		ulonglong2 *const dst = reinterpret_cast<ulonglong2 *>(a.out[c] + row0 + r0);
		const ulonglong2 *const st = reinterpret_cast<const ulonglong2 *>(rj_rows);
		for (uint32_t i = threadIdx.x; 2u * i + 1u < m; i += RJ_THREADS)
			dst[i] = st[i];
		if ((m & 1u) && threadIdx.x == 0)
			a.out[c][row0 + r0 + m - 1u] = rj_rows[m - 1u];
	}
}

/* ---- host */
static size_t rj_tiles(uint64_t n) { return (size_t)((n + RJ_TILE - 1) / RJ_TILE); }

/* digits of the tile sort: as many as a tile's counters allow (8192: pieces of 4 words, a leaf table of 2^(kbits - 13) <= 2^14 cells -
 * small tables keep several leaf workgroups on a CU), at least 2^8 key values per digit, never fewer than 8 digits (one per XCD) */
uint32_t mdb_rowjoin_dbits(uint32_t kbits)
{
	uint32_t d = kbits > 8u + 3u ? kbits - 8u : 3u;
	d = d > RJ_MAX_DBITS ? RJ_MAX_DBITS : d;
	const char *knob = mdb_knob("MDB_RJ_DBITS");	/* (measurement: fewer digits, longer pieces - never fewer than the leaf's table allows) */
	if (knob && atoi(knob) >= 3 && (uint32_t)atoi(knob) < d && kbits - (uint32_t)atoi(knob) <= RJ_SLOT_BITS)
		d = (uint32_t)atoi(knob);
	return d;
}

bool mdb_rowjoin_serves(uint64_t n_l, uint64_t n_r, uint32_t kbits, const void *keys_l, const void *null_l, const void *keys_r, const void *null_r,
			const void *const *pay_in, void *const *out, int npay)
{
	if (null_l || null_r)	/* (a NULL left key has no partner: not this operator's join; a nullable right key column: the older forms) */
		return false;
	/* MDB_ROWJOIN: 0 never, 2 whenever the form applies; default: left tables of 2^24 rows and more - up to there a result column
	 * (128 MB) stays in the Infinity Cache and the older forms' scattered 8-byte stores land in it (10^7 x 10^7 rows: 0.44 ms against
	 * 0.56 here; 10^8 x 10^8: 6.0 ms against 2.9) */
	const char *knob = mdb_knob("MDB_ROWJOIN");
	if (knob && knob[0] == '0')
		return false;
	if (n_l < ((uint64_t)1 << 24) && !(knob && knob[0] == '2'))
		return false;
	if (kbits < 15u || kbits > RJ_SLOT_BITS + RJ_MAX_DBITS)	/* 8 ... 8192 digits (windows of up to 2^27 values) */
		return false;
	if (n_l >= 0xF0000000ull || n_r >= 0xF0000000ull || ((uintptr_t)keys_l & 15u) || ((uintptr_t)keys_r & 15u))
		return false;
	for (int c = 0; c < npay; c++)
		if (((uintptr_t)out[c] & 15u) || ((uintptr_t)pay_in[c] & 15u))
			return false;
	return true;
}

/* (+ 64: a 16-byte load may start at a block's last word or cell) */
static size_t rj_left_arena_bytes(uint64_t n_l, size_t ostride, int streams)
{
	return mdb_align_up(rj_tiles(n_l) * RJ_STRIDE * 4 + 64) + 3 * mdb_align_up((rj_tiles(n_l) + 64) * ostride * 2) +
	       (size_t)streams * mdb_align_up(rj_tiles(n_l) * RJ_STRIDE * 8 + 64);
}

static size_t rj_right_arena_bytes(uint64_t n_r, size_t ostride, int npay)
{
	return 3 * mdb_align_up((rj_tiles(n_r) + 64) * ostride * 2) + (size_t)npay * mdb_align_up(rj_tiles(n_r) * RJ_STRIDE * 12 + 64);
}

size_t mdb_rowjoin_arena_bytes(uint64_t n_l, uint64_t n_r, uint32_t kbits, int npay)
{
	const size_t ostride = ((size_t)1 << mdb_rowjoin_dbits(kbits)) + 8u;
	return rj_left_arena_bytes(n_l, ostride, npay) + rj_right_arena_bytes(n_r, ostride, npay) + 8192;
}

size_t mdb_rowjoin_arena_bytes_multi(uint64_t n_l, const struct mdb_rowjoin_right *rt, int nrt, uint32_t kbits)
{
	const size_t ostride = ((size_t)1 << mdb_rowjoin_dbits(kbits)) + 8u;
	size_t need = 8192;
	int streams = 0;
	for (int t = 0; t < nrt; t++) {
		need += rj_right_arena_bytes(rt[t].n, ostride, rt[t].npay);
		streams += rt[t].npay;
	}
	return need + rj_left_arena_bytes(n_l, ostride, streams);
}

/* the tile sort of one table + its offsets transposed: *offT_out = [D][*tstride_out] (start | end << 16) */
template <bool CELLS>
static int rj_sort_table(mdb_dev_ctx *ctx, const rj_sort_args &sa, const char *name, uint32_t **offT_out, uint32_t *tstride_out)
{
	const uint32_t D = 1u << sa.dbits;
	const size_t lds = (size_t)(D >> 1) * 4 + (size_t)RJ_TILE * 4;
	const uint32_t nfull = (uint32_t)(sa.n / RJ_TILE), ntiles = (uint32_t)rj_tiles(sa.n), tstride = (ntiles + 63u) & ~63u;
	uint32_t *offT = (uint32_t *)mdb_arena_take(ctx, (size_t)D * tstride * 4);
	if (!offT)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %s", ctx->err);
	if (nfull) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_tile_sort<true, CELLS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, name, (k_rj_tile_sort<true, CELLS>), nfull, RJ_THREADS, lds, sa, 0u);
	}
	if (nfull < ntiles) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_tile_sort<false, CELLS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, name, (k_rj_tile_sort<false, CELLS>), 1u, RJ_THREADS, lds, sa, nfull);
	}
	MDB_LAUNCH(ctx, "rowjoin_offsets", k_rj_transpose_offs, dim3(tstride / 64u, (D + 63u) / 64u), 256, sa.offs, ntiles, D, tstride, offT);
	*offT_out = offT;
	*tstride_out = tstride;
	return MIDORIDB_OK;
}

/* One left table against up to RJ_MAX_STREAMS (right table, payload column) pairs on ONE key (round 6): the left table is sorted once, every
 * right table once, ONE leaf launch walks a digit's left rows once per stream and ONE placement pass reads a tile's words once for all the
 * columns.  Before, SELECT * over three tables on one key (BASELINE configs[4]'s join-only form; the reference's _join_nested_loop_tbl2mat,
 * /root/reference/src/engine/executor_select.c:1151-1232) sorted the left table's tiles twice and ran two leaves with two streams each
 * for the right table's words and cells. */
int mdb_rowjoin_run_multi(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const struct mdb_rowjoin_right *rt, int nrt, int64_t win_lo, uint32_t kbits)
{
	const uint32_t dbits = mdb_rowjoin_dbits(kbits), D = 1u << dbits;
	const uint32_t ntiles = (uint32_t)rj_tiles(n_l);
	const size_t ostride = (size_t)D + 8u;
	rj_leaf_args la;
	rj_place_args pa;
	memset(&la, 0, sizeof(la));
	memset(&pa, 0, sizeof(pa));
	for (int t = 0; t < nrt; t++) {
		if (rt[t].npay < 1 || rt[t].npay > 2 || la.nstreams + (uint32_t)rt[t].npay > RJ_MAX_STREAMS)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %d payload columns of table %d", rt[t].npay, t);
		const uint32_t ntiles_r = (uint32_t)rj_tiles(rt[t].n);
		rj_sort_args sa;
		memset(&sa, 0, sizeof(sa));
		sa.offs = (uint16_t *)mdb_arena_take(ctx, (size_t)ntiles_r * ostride * 2);
		if (!sa.offs)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %s", ctx->err);
		for (int c = 0; c < rt[t].npay; c++) {
			sa.pay_in[c] = reinterpret_cast<const uint64_t *>(rt[t].pay_in[c]);
			sa.recs[c] = (rj_rec *)mdb_arena_take(ctx, (size_t)ntiles_r * RJ_STRIDE * 12 + 64);
			if (!sa.recs[c])
				return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %s", ctx->err);
		}
		sa.keys = rt[t].keys;
		sa.n = rt[t].n;
		sa.base = win_lo;
		sa.kbits = kbits;
		sa.dbits = dbits;
		sa.npay = (uint32_t)rt[t].npay;
		sa.status = ctx->d_status;
		uint32_t *offT_r = NULL;
		uint32_t tstride_r = 0;
		const int rc = rj_sort_table<true>(ctx, sa, "rowjoin_tile_sort_r", &offT_r, &tstride_r);
		if (rc)
			return rc;
		for (int c = 0; c < rt[t].npay; c++) {
			const uint32_t s = la.nstreams++;
			la.recs[s] = sa.recs[c];
			la.offT_r[s] = offT_r;
			la.ntiles_r[s] = ntiles_r;
			la.tstride_r[s] = tstride_r;
			pa.out[s] = reinterpret_cast<uint64_t *>(rt[t].out[c]);
		}
	}
	uint32_t *words = (uint32_t *)mdb_arena_take(ctx, (size_t)ntiles * RJ_STRIDE * 4 + 64);
	uint16_t *offs = (uint16_t *)mdb_arena_take(ctx, (size_t)ntiles * ostride * 2);
	if (!words || !offs)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %s", ctx->err);
	for (uint32_t s = 0; s < la.nstreams; s++) {
		la.cells_al[s] = (uint64_t *)mdb_arena_take(ctx, (size_t)ntiles * RJ_STRIDE * 8 + 64);
		if (!la.cells_al[s])
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "row-order join: %s", ctx->err);
		pa.cells_al[s] = la.cells_al[s];
	}
	uint32_t *offT_l = NULL;
	uint32_t tstride_l = 0;
	rj_sort_args sl;
	memset(&sl, 0, sizeof(sl));
	sl.keys = keys_l;
	sl.n = n_l;
	sl.base = win_lo;
	sl.kbits = kbits;
	sl.dbits = dbits;
	sl.words = words;
	sl.offs = offs;
	sl.status = ctx->d_status;
	const int rc = rj_sort_table<false>(ctx, sl, "rowjoin_tile_sort", &offT_l, &tstride_l);
	if (rc)
		return rc;
	const uint32_t sbits = kbits - dbits;
	const uint32_t leaf_threads = sbits >= 14u ? 1024u : sbits == 13u ? 512u : 256u;
	const size_t lds_leaf = ((size_t)8 << sbits) + ((((size_t)1 << sbits) + 31) / 32) * 4;	/* cells, occupancy bits */
	const size_t lds_place = (size_t)(RJ_TILE / 2) * 8;
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_place), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_place));
	/* lanes per piece = twice the average piece (32 768 rows of a tile over D digits); fewer than 1024 digits (windows below 2^18 values):
	 * the whole wave walks one piece after the other */
	/* workgroups of the leaf: as many threads as the table's LDS leaves room for several of on a CU (a digit is a chain of dependent
	 * round trips - offsets, words, cells -: what hides them is another digit on the same CU) */
	const int lpp = dbits >= 13 ? 8 : dbits == 12 ? 16 : dbits == 11 ? 32 : dbits == 10 ? 64 : 0;
	/* as many workgroups as the device holds at a time (a multiple of 8: the XCDs), each walking its share of its XCD's digits */
	uint32_t leaf_grid = D;
	if (!(mdb_knob("MDB_RJ_PERSIST") && mdb_knob("MDB_RJ_PERSIST")[0] == '0')) {
		const uint32_t per_cu = (uint32_t)(((size_t)160 << 10) / (lds_leaf + 1024)), by_threads = 2048u / leaf_threads;
		const uint32_t held = ((uint32_t)ctx->num_cus * (per_cu < by_threads ? (per_cu ? per_cu : 1u) : by_threads)) & ~7u;
		if (held >= 8u && held < D)
			leaf_grid = held;
	}
	la.words_l = words;
	la.offT_l = offT_l;
	la.ntiles = ntiles;
	la.tstride = tstride_l;
	la.dbits = dbits;
	la.sbits = sbits;
	la.ablate = mdb_knob("MDB_RJ_ABLATE") ? (uint32_t)atoi(mdb_knob("MDB_RJ_ABLATE")) : 0u;
	la.joined = (unsigned long long *)(ctx->d_status + 2);
	la.status = ctx->d_status;
	if (mdb_knob("MDB_RJ_TRACE")) {
		la.trace = (unsigned long long *)mdb_arena_take(ctx, 64 * 8);
		if (la.trace)
			MDB_HIP(ctx, hipMemsetAsync(la.trace, 0, 64 * 8, ctx->stream));
	}
#define RJ_LAUNCH_LEAF(L, LONG)                                                                                                                   \
	do {                                                                                                                                      \
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_leaf<L, LONG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_leaf)); \
		MDB_LAUNCH_LDS(ctx, "rowjoin_leaf", (k_rj_leaf<L, LONG>), leaf_grid, leaf_threads, lds_leaf, la);                                    \
	} while (0)
	if (lpp == 8)
		RJ_LAUNCH_LEAF(8, false);
	else if (lpp == 16)
		RJ_LAUNCH_LEAF(16, false);
	else if (lpp == 32)
		RJ_LAUNCH_LEAF(32, false);
	else if (lpp == 64)
		RJ_LAUNCH_LEAF(64, false);
	else
		RJ_LAUNCH_LEAF(64, true);
#undef RJ_LAUNCH_LEAF
	if (la.trace) {		/* measurement only */
		unsigned long long h[64];
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		MDB_HIP(ctx, hipMemcpy(h, la.trace, sizeof(h), hipMemcpyDeviceToHost));
		fprintf(stderr, "rowjoin_leaf trace (workgroup 8, wave 0; 10 ns ticks since the first stamp; 0 round begins, 1 table cleared, 2 own build done, 3 all built, 4 own probe done):\n");
		for (int i = 0; i < 60 && h[i]; i++)
			fprintf(stderr, "%s%llu", i % 5 ? " " : i ? "\n  " : "  ", h[i] - h[0]);
		fprintf(stderr, "\n");
	}
	pa.words_l = words;
	pa.ncols = la.nstreams;
	pa.n = n_l;
	pa.ntiles = ntiles;
	const uint32_t grid = ((ntiles + 7u) / 8u) * 16u;
	MDB_LAUNCH_LDS(ctx, "rowjoin_place", k_rj_place, grid, RJ_THREADS, lds_place, pa);
	return MIDORIDB_OK;
}

int mdb_rowjoin_run(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const int64_t *keys_r, uint64_t n_r, const void *const *pay_in,
		    int64_t win_lo, uint32_t kbits, int npay, void *const *out)
{
	struct mdb_rowjoin_right rt;
	memset(&rt, 0, sizeof(rt));
	rt.keys = keys_r;
	rt.n = n_r;
	rt.npay = npay;
	for (int c = 0; c < npay && c < 2; c++) {
		rt.pay_in[c] = pay_in[c];
		rt.out[c] = out[c];
	}
	return mdb_rowjoin_run_multi(ctx, keys_l, n_l, &rt, 1, win_lo, kbits);
}

/* ------------------------------------------------------------------ GROUP BY key + COUNT(*) over a tile-sorted column (round 5)
 *
 * The reference's proc_groupby_clause (/root/reference/src/engine/executor_select.c:1526-1588, cmp_rows_col_mattbl :1465-1499, inc_count_cols
 * :1501-1524) keeps the FIRST row of every key and counts the others into it: (first row id, COUNT) per key, in first-row order.  The key
 * column goes through the tile sort above (one sequential pass, 4-byte words that name the row inside its tile), then one workgroup per key
 * digit walks the digit's piece of every tile and keeps, per key value, the smallest row id and the number of rows in two LDS arrays
 * (direct-addressed by the slot bits).  The groups leave either straight into the ordering kernel's ranges of 2^16 first row ids
 * (k_order_leaf_sparse: few groups) or as one record list for the ordering sort.  Before: two 8-byte partition levels + a leaf that
 * reads 12 bytes per row. */
struct rg_group_args {
	const uint32_t *words;
	const uint32_t *offT;
	uint32_t ntiles, tstride, dbits, sbits;
	uint32_t row_bits;		/* a record = first row id << (64 - row_bits) | COUNT */
	unsigned long long *rg_rec;	/* ranged emit (NULL: the list): region r of rg_cap records at rg_rec + r * rg_cap, its fill in rg_cnt[r] */
	uint32_t *rg_cnt;
	uint32_t rg_cap, rg_shift, rg_n;
	unsigned long long *rec;	/* the list */
	uint32_t *rec_count;
	uint32_t rec_cap;
	uint32_t *groups;
	uint32_t *status;
	/* k_rj_group_leaf_dense (nearly unique keys, mdb_dev_dense.hip): one bit per row - cleared here for every row that is not the first of
	 * its key - and the keys with more than one row as exceptions (first row << 32 | COUNT) */
	unsigned int *dense_bits;	/* NULL: the pilot - duplicates are only counted */
	unsigned long long *exc;
	uint32_t exc_cap;
	uint32_t *dense_cnt;		/* [0] rows that are not the first of their key, [1] exceptions, [2] rows seen */
};

/* ---- the same walk for a column whose keys are nearly unique: no record per group.  A row that meets an earlier row of its key in the
 * table of smallest row ids is, or makes that one, a row that is not the key's first: min(old, new) stays, max(old, new) is settled - its
 * bit is cleared (every row but the final smallest is settled exactly once).  Launched over the first `gridDim.x` digits only with
 * dense_bits = NULL it is the pilot that says how many rows of a sample of the KEYS are duplicates. */
template <int LPP>
__global__ __launch_bounds__(RJ_LEAF_THREADS) void k_rj_group_leaf_dense(rg_group_args a)
{
	extern __shared__ uint32_t rg_lds[];
	const uint32_t S = 1u << a.sbits;
	uint32_t *const s_first = rg_lds, *const s_count = rg_lds + S;
	const uint32_t D = 1u << a.dbits, d = rj_digit_of_block(blockIdx.x, D), lane = mdb_lane(), wave = threadIdx.x >> 6;
	for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
		s_first[i] = 0xFFFFFFFFu;
		s_count[i] = 0u;
	}
	__syncthreads();
	uint32_t dups = 0;
	{
		/* (the table's answer - the smallest row so far - is needed here: a batch's 16 requests go out together, then the answers are
		 * looked at; one at a time every row would wait for its own round trip to the LDS) */
		constexpr int UG = LPP >= 32 ? 8 : 16;		/* (three registers per request in flight) */
		uint32_t w[UG], rw[UG], pv[UG];
		uint32_t live = 0;
		auto settle = [&](uint32_t prev, uint32_t row) {
			if (prev != 0xFFFFFFFFu) {
				const uint32_t loser = prev > row ? prev : row;
				if (a.dense_bits)
					atomicAnd(&a.dense_bits[loser >> 5], ~(1u << (loser & 31u)));
				dups++;
			}
		};
		auto ask = [&](int u, uint32_t word, uint32_t idx) {
			const uint32_t slot = word >> RJ_TILE_BITS;
			rw[u] = (idx / RJ_STRIDE) * RJ_TILE + (word & (RJ_TILE - 1u));
			pv[u] = atomicMin(&s_first[slot], rw[u]);
			atomicAdd(&s_count[slot], 1u);
			live |= 1u << u;
		};
		rj_for_pieces<LPP, UG, true>(a.offT, a.tstride, a.ntiles, d, [&](int u, uint32_t idx) { w[u] = a.words[idx]; },
				       [&](int u, uint32_t idx) { ask(u, w[u], idx); },
				       [&] {
#pragma unroll
					       for (int u = 0; u < UG; u++)
						       if (live & (1u << u))
							       settle(pv[u], rw[u]);
					       live = 0;
				       });
	}
	__syncthreads();
	uint32_t groups = 0, nexc = 0, rows = 0;
	for (uint32_t i0 = wave * 64u; i0 < S; i0 += blockDim.x) {
		const uint32_t c = i0 + lane < S ? s_count[i0 + lane] : 0u;
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
	/* one atomic per counter and WORKGROUP (every wave on its own: 5 x 10^5 atomics on three addresses - 2 ms of a 3 ms kernel) */
	__shared__ uint32_t s_sum[4][RJ_LEAF_THREADS / 64];
	__shared__ uint32_t s_ebase;
	const uint32_t nwaves = blockDim.x >> 6;
	if (lane == 0) {
		s_sum[0][wave] = dups;
		s_sum[1][wave] = groups;
		s_sum[2][wave] = rows;
		s_sum[3][wave] = nexc;
	}
	__syncthreads();
	if (threadIdx.x < 4u) {
		uint32_t t = 0;
		for (uint32_t w = 0; w < nwaves; w++)
			t += s_sum[threadIdx.x][w];
		if (t) {
			if (threadIdx.x == 0)
				atomicAdd(&a.dense_cnt[0], t);
			else if (threadIdx.x == 1)
				atomicAdd(a.groups, t);
			else if (threadIdx.x == 2)
				atomicAdd(&a.dense_cnt[2], t);
			else
				s_ebase = atomicAdd(&a.dense_cnt[1], t);
		} else if (threadIdx.x == 3) {
			s_ebase = 0u;
		}
	}
	__syncthreads();
	uint32_t ebase = s_ebase;
	for (uint32_t w = 0; w < wave; w++)
		ebase += s_sum[3][w];
	if (!a.exc || !nexc)
		return;
	if (ebase + nexc > a.exc_cap) {		/* (more keys with several rows than the list holds: the caller takes the record form) */
		if (lane == 0)
			mdb_raise(a.status, 16384u);
		return;
	}
	for (uint32_t i0 = wave * 64u; i0 < S; i0 += blockDim.x) {
		const uint32_t c = i0 + lane < S ? s_count[i0 + lane] : 0u;
		const uint64_t m = __ballot(c > 1u);
		if (c > 1u)
			a.exc[ebase + (uint32_t)__popcll(m & mdb_lanemask_lt())] = ((unsigned long long)s_first[i0 + lane] << 32) | c;
		ebase += (uint32_t)__popcll(m);
	}
}

template <int LPP>
__global__ __launch_bounds__(RJ_LEAF_THREADS) void k_rj_group_leaf(rg_group_args a)
{
	extern __shared__ uint32_t rg_lds[];
	const uint32_t S = 1u << a.sbits;
	uint32_t *const s_first = rg_lds, *const s_count = rg_lds + S, *const s_rg = s_count + S;
	__shared__ uint32_t s_tmp[32];
	__shared__ uint32_t s_base;
	const uint32_t D = 1u << a.dbits, d = rj_digit_of_block(blockIdx.x, D);
	for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
		s_first[i] = 0xFFFFFFFFu;
		s_count[i] = 0u;
	}
	for (uint32_t r = threadIdx.x; r < (a.rg_rec ? a.rg_n : 0u); r += blockDim.x)
		s_rg[r] = 0u;
	__syncthreads();
	{
		constexpr int UG = 16;
		uint32_t w[UG];
		auto take = [&](uint32_t word, uint32_t idx) {
			const uint32_t slot = word >> RJ_TILE_BITS, row = (idx / RJ_STRIDE) * RJ_TILE + (word & (RJ_TILE - 1u));
			atomicMin(&s_first[slot], row);
			atomicAdd(&s_count[slot], 1u);
		};
		rj_for_pieces<LPP, UG, true>(a.offT, a.tstride, a.ntiles, d, [&](int u, uint32_t idx) { w[u] = a.words[idx]; },
				       [&](int u, uint32_t idx) { take(w[u], idx); });
	}
	__syncthreads();
	if (a.rg_rec) {
		/* the ordering kernel's ranges of 2^rg_shift first row ids are filled here: the digit's groups per range are counted, a place for
		 * them reserved with one global atomic per (digit, range), and every record written there */
		uint32_t mine = 0;
		for (uint32_t i = threadIdx.x; i < S; i += blockDim.x)
			if (s_count[i]) {
				atomicAdd(&s_rg[s_first[i] >> a.rg_shift], 1u);
				mine++;
			}
		uint32_t total;
		(void)mdb_block_excl_scan(mine, s_tmp, &total);		/* (two barriers: s_rg is complete behind it) */
		if (threadIdx.x == 0 && total)
			atomicAdd(a.groups, total);
		for (uint32_t r = threadIdx.x; r < a.rg_n; r += blockDim.x) {
			const uint32_t c = s_rg[r];
			if (!c)
				continue;
			uint32_t at = atomicAdd(&a.rg_cnt[r], c);
			if (at + c > a.rg_cap) {
				mdb_raise(a.status, 8192u);	/* a range outgrew its region: the caller takes the record list and its sort */
				at = a.rg_cap;
			}
			s_rg[r] = at;
		}
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
			const uint32_t c = s_count[i];
			if (!c)
				continue;
			const uint32_t first = s_first[i], r = first >> a.rg_shift;
			const uint32_t at = atomicAdd(&s_rg[r], 1u);
			if (at < a.rg_cap)
				a.rg_rec[(size_t)r * a.rg_cap + at] = ((unsigned long long)first << (64 - a.row_bits)) | c;
		}
		return;
	}
	/* one list: the workgroup's groups side by side, wherever the list's cursor stands */
	uint32_t mine = 0, cmax = 0;
	for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
		mine += s_count[i] ? 1u : 0u;
		cmax = s_count[i] > cmax ? s_count[i] : cmax;
	}
	if (cmax >> (32u - a.row_bits))		/* (a COUNT that does not fit beside its row id in 32 bits: the ordering sort keeps 8-byte records) */
		mdb_raise(a.status, 512u);
	uint32_t total;
	uint32_t at = mdb_block_excl_scan(mine, s_tmp, &total);
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
	at += s_base;
	for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
		const uint32_t c = s_count[i];
		if (c)
			a.rec[at++] = ((unsigned long long)s_first[i] << (64 - a.row_bits)) | c;
	}
}

static uint32_t rg_group_dbits(uint32_t kbits)
{
	/* a digit's two LDS arrays hold 2^(kbits - dbits) <= 2^14 key values; at least 1024 digits: pieces of at most 32 words on average,
	 * several of them in flight per wave */
	const uint32_t d = kbits > 13u + 10u ? kbits - 13u : 10u;
	return d > RJ_MAX_DBITS ? RJ_MAX_DBITS : d;
}

/* 0 = done: out_first[g] / out_count[g] = the first row and the rows of group g, groups in first-row order; 1 = not served (the caller's
 * other forms answer); < 0 = error.  NULL-free key column, keys inside [win_lo, win_lo + 2^kbits) - verified: a key outside -> 1 with
 * *outside = true.  Synchronises. */
int mdb_group_count_tiled(mdb_dev_ctx *ctx, const int64_t *keys, uint64_t n, int64_t win_lo, uint32_t kbits, uint32_t *out_first, int64_t *out_count,
			  uint64_t cap, uint64_t *out_groups, bool *outside)
{
	*outside = false;
	/* Taken for windows of 2^26 and 2^27 values (below, the band sort of mdb_dev_bandgroup.hip serves; MDB_GROUP_TILED=1: every window from
	 * 2^13 values on - the parity tests; =0: never).  At 10^8 rows (profiles/r05/group_forms.json): 2.5 x 10^7 groups spread over 2^27
	 * values 1.20 ms against the partitioned path's 1.83; 10^8 unique keys 2.20 against 2.06 - the leaf's walk over 3052 tiles' pieces of
	 * four words is bound by requests, and every row leaves it as a group record */
	const char *const knob = mdb_knob("MDB_GROUP_TILED");
	const bool on = knob ? knob[0] == '1' : (kbits >= 26u && n >= ((uint64_t)1 << 24));	/* (measured at 10^8 rows only: large tables) */
	if (!on || kbits < 13u || kbits > 14u + RJ_MAX_DBITS || n >= 0xF0000000ull || ((uintptr_t)keys & 15u) || n < ((uint64_t)1 << 21))
		return 1;
	const uint32_t dbits = rg_group_dbits(kbits), sbits = kbits - dbits, D = 1u << dbits;
	if (sbits > 14u)
		return 1;
	uint32_t row_bits = 0;
	int sb1 = 0, sb2 = 0;
	if (!order_bits(n, &row_bits, &sb1, &sb2))
		return 1;
	if (ctx->explain) {	/* (mdb_dev_explain_group_count: the tile sort serves - nothing is launched) */
		ctx->explain->group_form = 2;
		ctx->explain->key_form = 2;
		ctx->explain->key_bits = kbits;
		ctx->explain->from_stats = ctx->explain_as_sample ? 0u : ctx->pl_from_stats;
		ctx->explain->samples = ctx->explain_as_sample ? 1u : 0u;
		return MIDORIDB_OK;
	}
	const uint32_t ntiles = (uint32_t)rj_tiles(n);
	const size_t ostride = (size_t)D + 8u;
	const uint64_t values = (uint64_t)1 << kbits, most = (n < values ? n : values) + 1024;	/* groups: at most the rows, at most the window's key values */
	uint32_t rg_n = 0;
	const bool ranged = order_ranges_apply(n, row_bits, most < ((uint64_t)1 << 23) ? most : ((uint64_t)1 << 22), &rg_n) && values <= ((uint64_t)1 << 23) &&
			    !(mdb_knob("MDB_ORDER_RANGES") && mdb_knob("MDB_ORDER_RANGES")[0] == '0');
	size_t need = mdb_align_up((size_t)ntiles * RJ_STRIDE * 4 + 64) + 3 * mdb_align_up(((size_t)ntiles + 64) * ostride * 2) + mdb_align_up(most * 8) +
		      order_records_arena_bytes(most, n, row_bits, sb1, sb2) + 16384;
	if (ranged)
		need += mdb_align_up((size_t)rg_n * ORDER_RANGE_CAP * 8) + mdb_align_up((size_t)rg_n * 4);
	/* (nearly unique keys need nearly as many key values as rows: a window with fewer cannot hold them - no pilot) */
	const bool dense_ok = n >= ((uint64_t)1 << 22) && values >= n - n / 16 && !(mdb_knob("MDB_GROUP_DENSE") && mdb_knob("MDB_GROUP_DENSE")[0] == '0');
	if (dense_ok)
		need += mdb_dense_arena_bytes(n) + mdb_align_up((n / 8 + 4096) * 8);
	int rc = mdb_arena_begin(ctx, need);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	uint32_t *words = (uint32_t *)mdb_arena_take(ctx, (size_t)ntiles * RJ_STRIDE * 4 + 64);
	uint16_t *offs = (uint16_t *)mdb_arena_take(ctx, (size_t)ntiles * ostride * 2);
	if (!words || !offs)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a tile-sorted column: %s", ctx->err);
	rj_sort_args sa;
	memset(&sa, 0, sizeof(sa));
	sa.keys = keys;
	sa.n = n;
	sa.base = win_lo;
	sa.kbits = kbits;
	sa.dbits = dbits;
	sa.words = words;
	sa.offs = offs;
	sa.status = ctx->d_status;
	uint32_t *offT = NULL;
	uint32_t tstride = 0;
	rc = rj_sort_table<false>(ctx, sa, "group_tile_sort", &offT, &tstride);
	if (rc)
		return rc;
	rg_group_args ga;
	memset(&ga, 0, sizeof(ga));
	ga.words = words;
	ga.offT = offT;
	ga.ntiles = ntiles;
	ga.tstride = tstride;
	ga.dbits = dbits;
	ga.sbits = sbits;
	ga.row_bits = row_bits;
	ga.groups = ctx->d_status + 1;
	ga.rec_count = ctx->d_status + 2;
	ga.status = ctx->d_status;
	const uint32_t threads = sbits >= 14u ? 1024u : sbits == 13u ? 512u : 256u;
	const int lpp = dbits >= 13 ? 8 : dbits == 12 ? 16 : dbits == 11 ? 32 : 64;
	uint64_t *h = ctx->h_pinned;
#define RJ_LAUNCH_DENSE(L, GRID)                                                                                                                  \
	do {                                                                                                                                      \
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_group_leaf_dense<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)8 << sbits))); \
		MDB_LAUNCH_LDS(ctx, "group_tile_leaf_dense", (k_rj_group_leaf_dense<L>), (GRID), threads, (size_t)8 << sbits, ga);                     \
	} while (0)
#define RJ_LAUNCH_DENSE_ANY(GRID)                                                                                                                 \
	do {                                                                                                                                      \
		if (lpp == 8)                                                                                                                     \
			RJ_LAUNCH_DENSE(8, GRID);                                                                                                 \
		else if (lpp == 16)                                                                                                               \
			RJ_LAUNCH_DENSE(16, GRID);                                                                                                \
		else if (lpp == 32)                                                                                                               \
			RJ_LAUNCH_DENSE(32, GRID);                                                                                                \
		else                                                                                                                              \
			RJ_LAUNCH_DENSE(64, GRID);                                                                                                \
	} while (0)
	/* Nearly unique keys?  A pilot over 64 of the digits - a fair sample of the KEYS: all rows of a key are in one digit - counts the rows
	 * that are not the first of their key.  One in 16 at most: the groups leave as one bit per row + exceptions (mdb_dev_dense.hip)
	 * instead of a record each and a sort of the records.  MDB_GROUP_DENSE=0: never. */
	if (dense_ok) {
		ga.dense_cnt = ctx->d_status + 4;
		ga.dense_bits = NULL;
		ga.exc = NULL;
		RJ_LAUNCH_DENSE_ANY(D < 64u ? D : 64u);
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint32_t *ps = reinterpret_cast<const uint32_t *>(&h[1]);
		if (ps[0] & 128u) {
			*outside = true;
			return 1;
		}
		const uint64_t pilot_dups = ps[4], pilot_rows = ps[6];
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, 7 * sizeof(uint32_t), ctx->stream));
		if (pilot_rows && pilot_dups * 16u <= pilot_rows) {
			unsigned long long *bits = NULL;
			const uint64_t exc_cap = n / 8 + 4096;
			if ((rc = mdb_dense_bits_begin(ctx, n, &bits)))
				return rc;
			ga.dense_bits = reinterpret_cast<unsigned int *>(bits);
			ga.exc = (unsigned long long *)mdb_arena_take(ctx, exc_cap * 8);
			if (!ga.exc)
				return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a tile-sorted column: %s", ctx->err);
			ga.exc_cap = (uint32_t)exc_cap;
			RJ_LAUNCH_DENSE_ANY(D);
			MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 32, hipMemcpyDeviceToHost, ctx->stream));
			MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
			const uint32_t dstatus = ps[0], groups = ps[1], n_exc = ps[5];
			if (mdb_knob("MDB_DEBUG_GROUP"))
				fprintf(stderr, "group_count (tile sort, dense): pilot %llu of %llu rows not first; %u groups, %u rows not first, %u exceptions, status %u\n",
					(unsigned long long)pilot_dups, (unsigned long long)pilot_rows, groups, ps[4], n_exc, dstatus);
			if (!(dstatus & 16384u) && (uint64_t)groups + ps[4] == n) {
				if (groups > cap)
					return mdb_set_err(ctx, -MIDORIDB_ERROR, "GROUP BY: %u groups, room for %llu", groups, (unsigned long long)cap);
				if ((rc = mdb_dense_emit(ctx, bits, n, ga.exc, n_exc, out_first, out_count)))
					return rc;
				MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
				*out_groups = groups;
				ctx->pl_key_bits = kbits;
				ctx->pl_group_form = 2;
				ctx->pl_bits = 1;
				return MIDORIDB_OK;
			}
			/* (the exception list overflowed - the pilot's digits were not typical: the record form below) */
			MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 8 * sizeof(uint32_t), ctx->stream));
		}
	}
#undef RJ_LAUNCH_DENSE_ANY
#undef RJ_LAUNCH_DENSE
	for (int attempt = ranged ? 0 : 1; attempt < 2; attempt++) {
		if (attempt == 0) {
			ga.rg_rec = (unsigned long long *)mdb_arena_take(ctx, (size_t)rg_n * ORDER_RANGE_CAP * 8);
			ga.rg_cnt = (uint32_t *)mdb_arena_take(ctx, (size_t)rg_n * 4);
			if (!ga.rg_rec || !ga.rg_cnt)
				return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a tile-sorted column: %s", ctx->err);
			MDB_HIP(ctx, hipMemsetAsync(ga.rg_cnt, 0, (size_t)rg_n * 4, ctx->stream));
			ga.rg_cap = ORDER_RANGE_CAP;
			ga.rg_shift = ORDER_RANGE_BITS;
			ga.rg_n = rg_n;
		} else {
			ga.rg_rec = NULL;
			ga.rg_n = 0;
			ga.rec = (unsigned long long *)mdb_arena_take(ctx, most * 8);
			if (!ga.rec)
				return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a tile-sorted column: %s", ctx->err);
			ga.rec_cap = (uint32_t)(most > 0xFFFFFFFFull ? 0xFFFFFFFFull : most);
			if (ranged)	/* (behind a ranged attempt that overflowed - its flags and counters go; NOT before the first attempt: the tile
					 * sort's "key outside the window" flag is in there) */
				MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
		}
		const size_t lds = ((size_t)8 << sbits) + (ga.rg_rec ? (size_t)rg_n * 4 : 0);
#define RJ_LAUNCH_GROUP(L)                                                                                                                        \
	do {                                                                                                                                      \
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rj_group_leaf<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
		MDB_LAUNCH_LDS(ctx, "group_tile_leaf", (k_rj_group_leaf<L>), D, threads, lds, ga);                                                   \
	} while (0)
		if (lpp == 8)
			RJ_LAUNCH_GROUP(8);
		else if (lpp == 16)
			RJ_LAUNCH_GROUP(16);
		else if (lpp == 32)
			RJ_LAUNCH_GROUP(32);
		else
			RJ_LAUNCH_GROUP(64);
#undef RJ_LAUNCH_GROUP
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint32_t *hs = reinterpret_cast<const uint32_t *>(&h[1]);
		const uint32_t status = hs[0], groups = hs[1], list_len = hs[2];
		if (status & 128u) {
			*outside = true;
			return 1;
		}
		if (status & 8u)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "GROUP BY over a tile-sorted column: the record list overflowed");
		if (status & 512u)
			MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));	/* (the ordering kernels raise flags of their own there) */
		if (attempt == 0 && (status & 8192u))
			continue;	/* more groups in a range of first rows than its region holds: the list and its sort */
		const bool rec32 = attempt == 1 && !(status & 512u) && row_bits < 32u;
		if (groups > cap)
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "GROUP BY: %u groups, room for %llu", groups, (unsigned long long)cap);
		if (attempt == 0) {
			rc = order_presorted(ctx, ga.rg_rec, ga.rg_cnt, rg_n, row_bits, out_first, out_count, NULL, NULL, false, 0, 0, 0);
		} else {
			rc = groups ? order_records(ctx, ga.rec, list_len, n, row_bits, sb1, sb2, out_first, out_count, NULL, NULL, NULL, false, rec32, 0, 0, 0, false, groups) : MIDORIDB_OK;
		}
		if (rc)
			return rc;
		*out_groups = groups;
		ctx->pl_key_bits = kbits;
		ctx->pl_group_form = 2;
		return MIDORIDB_OK;
	}
	return 1;
}

