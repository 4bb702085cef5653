// Same-box A/B + ablations of the 4096-digit first-level pass (midoridb_amd/csrc/mdb_dev_scatter4096.h): the kernels the library
// launches, compiled from the library's own header, over 10^8 unique keys in a window of 2^27 values (variant U's tables), with
// every output region walked and checked (each row exactly once, under the digit and hash bits its key says).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imidoridb_amd/csrc -o profiles/micro/scatter4096_stream profiles/micro/scatter4096_stream.hip
//   ./profiles/micro/scatter4096_stream [rows] [reps]
#include "mdb_dev_scatter4096.h"
#include <cstdlib>
#include <vector>
#include <algorithm>

/* ------------------------------------------------------------------ k_scatter4096_stream with writer waves (round 6; measured here, NOT kept in the library)
 *
 * What k_scatter4096_stream's stamps say (profiles/r06/README.md): with loads and stores switched off the four phases take 41 000 cycles
 * per tile; loads on top cost nothing (the ring hides them); stores make the write-out 19 000 - 28 000 cycles instead of 9 900 - the store
 * path of a CU takes ~4 cycles per 64-byte piece a wave-instruction touches, and a run is a piece - and, vmcnt being ONE in-order counter
 * for loads and stores, every wave that has just stored waits for its stores to drain before it sees the next key.
 * So the roles are split: of the workgroup's 16 waves, 12 load, hash, count and stage (40 rows per thread and tile), and 4 do nothing
 * but write the PREVIOUS tile out while the 12 count the next one - their vmcnt only ever holds stores and nobody waits for it.
 * Digit bookkeeping is done by all 1024 threads.  One stage buffer: the write-out of tile t ends before tile t + 1 is staged. */
template <int CW /* computing waves */, int RPT /* rows per computing thread */, bool ROWS = false, bool NULLS = false, int RING_ = 0,
	  int ABLATE = 0 /* harness only, bits: 1 no global stores, 2 keys not loaded (synthetic), 4 cycle stamps per phase */>
__global__ __launch_bounds__(1024, 4) void k_scatter4096_ws(shw_scatter_args a)
{
	typedef typename std::conditional<ROWS, uint32_t, uint16_t>::type W;
	constexpr uint32_t THREADS = 1024, CT = CW * 64u, WT = THREADS - CT, TILE = CT * RPT, D = 1u << SHW_D_BITS, DPT = D / THREADS, NCHUNK = TILE / 64u,
			   NPAIR = RPT / 2, RING = RING_ ? RING_ : (NPAIR % 8 == 0 ? 8 : (NPAIR % 5 == 0 ? 5 : 4)), WSTEPS = TILE / WT, WUNROLL = 8;
	constexpr uint32_t HDR = ROWS ? 1u : 0u;	/* words a run takes beyond its rows */
	static_assert(TILE <= 32768u && DPT == 4 && (RPT & 1) == 0 && NPAIR % RING == 0 && (TILE / 32u) % 2u == 0 && TILE % WT == 0 && WSTEPS % WUNROLL == 0 &&
			      WT % 64u == 0, "tile shape");
	extern __shared__ __attribute__((aligned(16))) uint32_t shw_lds[];
	uint32_t *const s_cnt = shw_lds;			/* [D + 64] the digits' counts, then their staging cursors; [D] the dummy digit */
	uint32_t *const s_delta = s_cnt + D + 64;		/* [D] per NON-EMPTY digit, in digit order: where its run goes minus its tile-local start */
	uint32_t *const s_chunk = s_delta + D;			/* [NCHUNK] runs that begin before staged position 64 c */
	uint32_t *const s_mark = s_chunk + NCHUNK;		/* [TILE / 32] bit i: a run begins at staged position i */
	uint32_t *const s_tmp = s_mark + TILE / 32u;		/* [32] */
	W *const s_stage = reinterpret_cast<W *>(s_tmp + 32);	/* [TILE + 64]; [TILE] the dummy slot */
	__shared__ uint32_t s_any_bad;

	const uint32_t lane = mdb_lane(), sub = blockIdx.x % a.nsub;
	const bool computing = threadIdx.x < CT;	/* (uniform per wave) */
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

	ulonglong2 pre[RING];	/* the keys in flight, a ring: pair p of this thread = rows 2 (p CT + tid), + 1 of its tile, in slot p % RING */
	uint32_t hs[RPT];	/* a row's hash, or `none` */
#define SHS_REQUEST(P, t0, tlen, tid_)                                                                                   \
	do {                                                                                                             \
		if (ABLATE & 2) {                                                                                        \
			const unsigned long long i_ = (t0) + 2u * ((uint32_t)(P) * CT + (tid_));                         \
			pre[(P) % RING] = make_ulonglong2((unsigned long long)a.key_lo + ((i_ * 0x9E3779B1ull) & ((1ull << a.kbits) - 1ull)), \
							  (unsigned long long)a.key_lo + (((i_ + 1u) * 0x9E3779B1ull) & ((1ull << a.kbits) - 1ull))); \
		} else {	/* (one form for full and short tiles: a branch here costs the loads their counted waits) */ \
			const uint32_t last_ = ((tlen) - 1u) & ~1u, e0_ = 2u * ((uint32_t)(P) * CT + (tid_));           \
			const ulonglong2 *const ptr_ = reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const char *>(a.keys + (t0)) + (size_t)(e0_ < last_ ? e0_ : last_) * 8u); \
			if (ABLATE & 8) {                                                                                \
				typedef unsigned long long ull2_ __attribute__((ext_vector_type(2)));                    \
				const ull2_ v_ = __builtin_nontemporal_load(reinterpret_cast<const ull2_ *>(ptr_));      \
				pre[(P) % RING] = make_ulonglong2(v_.x, v_.y);                                           \
			} else {                                                                                         \
				pre[(P) % RING] = *ptr_;                                                                 \
			}                                                                                                \
		}                                                                                                        \
	} while (0)
#define SHS_COUNT(r_, t0, tlen, tid_)                                                                                    \
	do {                                                                                                             \
		const uint32_t e_ = 2u * ((uint32_t)((r_) >> 1) * CT + (tid_)) + (uint32_t)((r_) & 1);                  \
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

	/* the tile that is counted next (its keys are on their way) and the staged one that is written out meanwhile (none yet) */
	uint64_t row0 = r_begin, staged_row0 = 0;
	uint32_t len = (uint32_t)((r_end - row0) < TILE ? (r_end - row0) : TILE), staged_total = 0u;
	if (computing) {
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
		const uint64_t nrow0 = row0 + len;
		const uint32_t nlen = len ? (uint32_t)((r_end - nrow0) < TILE ? (r_end - nrow0) : TILE) : 0u;	/* 0: no tile behind this one */
		unsigned long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0;
		if (ABLATE & 4)
			t0_ = __builtin_amdgcn_s_memtime();

		if (computing) {
			/* 1a. hash and count the tile's rows.  The ring slot a pair leaves takes the pair RING places on - of this tile, then of the next
			 *     one (behind the last tile: this tile again - a load nobody looks at instead of a branch around a load, which would cost
			 *     every load of the loop its counted wait) */
			if (len) {	/* (uniform) */
				const uint64_t prow0 = nlen ? nrow0 : row0;
				const uint32_t plen = nlen ? nlen : len;
				bool oow = false;
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
		} else if (staged_total) {
			/* 1b. the writer waves: the staged tile - consecutive lanes, consecutive positions of a run; a position's run = the runs that
			 *     begin before its chunk of 64 + the run-start bits up to it inside the chunk */
			const uint32_t wt = tid - CT, ww = wt >> 6;
			for (uint32_t k0 = 0; k0 < WSTEPS && k0 * WT < staged_total; k0 += WUNROLL) {
				/* (four waves cannot hide each other's LDS round trips: WUNROLL steps' reads go out together, then their dependent reads,
				 * then the stores) */
				uint64_t m[WUNROLL];
				uint32_t ord[WUNROLL], g[WUNROLL];
				W sv[WUNROLL];
#pragma unroll
				for (uint32_t u = 0; u < WUNROLL; u++) {
					const uint32_t c = (k0 + u) * (WT / 64u) + ww;
					m[u] = *reinterpret_cast<const uint64_t *>(&s_mark[2u * c]);	/* (one broadcast read per wave) */
					ord[u] = s_chunk[c];
					sv[u] = s_stage[(k0 + u) * WT + wt];
				}
#pragma unroll
				for (uint32_t u = 0; u < WUNROLL; u++) {
					ord[u] += (uint32_t)__popcll(m[u] & le) - 1u;
					ord[u] = ord[u] < D ? ord[u] : D - 1u;	/* (a position behind the tile's rows: read something, write nothing) */
					g[u] = s_delta[ord[u]];
				}
#pragma unroll
				for (uint32_t u = 0; u < WUNROLL; u++) {
					const uint32_t i = (k0 + u) * WT + wt;
					const bool put = i < staged_total;
					if (!(ABLATE & 1)) {
						if (put) {
							reinterpret_cast<W *>(a.out)[i + g[u]] = sv[u];
							if (ROWS && ((m[u] >> lane) & 1ull))	/* the run's header: the tile (its first row is even) */
								reinterpret_cast<W *>(a.out)[i + g[u] - 1u] = (W)(0x80000000u | (uint32_t)(staged_row0 >> 1));
						}
					} else if (put && sv[u] == (W)0xFFFFFFF1u && g[u] == 0xFFFFFFFFu) {
						reinterpret_cast<W *>(a.out)[0] = sv[u];
					}
				}
			}
		}
		shw_barrier();
		if (!len)
			break;
		if (ABLATE & 4)
			t1_ = __builtin_amdgcn_s_memtime();

		/* 2. all threads: digit counts -> tile-local starts (written back over the counters: the staging cursors), the ordinal of every
		 *    non-empty digit, the run-start bits, the runs that begin before every 64th staged position, and the run's place in its region:
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

		/* 3. the computing waves stage by digit: a row's position is what the returning atomic on its digit's cursor says (the dummy
		 *    digit's: the dummy slot) */
		if (computing) {
#pragma unroll
			for (int r = 0; r < RPT; r++) {
				uint32_t pos = atomicAdd(&s_cnt[hs[r] >> a.rem], 1u);
				pos = pos < TILE ? pos : TILE;
				if (ROWS)	/* (the row's place in the tile: pair r / 2 of this thread, element r & 1) */
					s_stage[pos] = (W)(((2u * ((uint32_t)(r >> 1) * CT + tid) + (uint32_t)(r & 1)) << 15) | (hs[r] & wmask));
				else
					s_stage[pos] = (W)(hs[r] & wmask);
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
						s_any_bad = 1u;			/* ... and nothing of this tile is written */
					}
					s_delta[ord] = (d * a.nsub + sub) * a.cap + base[j] + HDR - st0[j];
					ord++;
				}
			}
		}
		shw_barrier();
		if (ABLATE & 4)
			t3_ = __builtin_amdgcn_s_memtime();
		/* (every staging cursor has been read: the counters of the next tile) */
		*reinterpret_cast<uint4 *>(&s_cnt[tid * DPT]) = make_uint4(0u, 0u, 0u, 0u);
		/* (a run that does not fit: the flag is up, the caller drops the regions - no word of this tile goes out) */
		staged_total = __builtin_amdgcn_readfirstlane((int)s_any_bad) ? 0u : tile_total;
		staged_row0 = row0;
		row0 = nrow0;
		len = nlen;
		shw_barrier();
		if (tid == 0)
			s_any_bad = 0u;		/* (read by everybody in front of the barrier; written next behind two more) */
		if (ABLATE & 4) {
			if (threadIdx.x == 0) {
				atomicAdd(&a.dbg[0], t1_ - t0_);	/* count (wave 0) */
				atomicAdd(&a.dbg[1], t2_ - t1_);	/* digits */
				atomicAdd(&a.dbg[2], t3_ - t2_);	/* stage */
				atomicAdd(&a.dbg[3], 1ull);
			}
		}
	}
#undef SHS_REQUEST
#undef SHS_COUNT
}


#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define NSUB 8u
#define KBITS 27u
#define KEY_LO 1000ll
#define MULT 0x9E3779B1ull

__global__ void k_gen(long long *keys, uint32_t n)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		keys[i] = KEY_LO + (long long)(((unsigned long long)i * MULT) & ((1ull << KBITS) - 1ull));
}

// one thread per region: every word parsed like the leaf kernels do; seen[row] set once; errors counted
template <bool ROWS>
__global__ void k_check(const void *out, const uint32_t *cursor, uint32_t cap, const long long *keys, uint32_t n, uint32_t rem, uint32_t *seen,
			unsigned long long *stats /* [0] rows, [1] errors, [2] words incl. headers */)
{
	const uint32_t D = 4096u, r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= D * NSUB)
		return;
	const uint32_t d = r / NSUB, sub = r % NSUB;
	uint32_t cnt = cursor[sub * D + d];
	if (cnt > cap) {
		atomicAdd(&stats[1], 1ull);
		cnt = cap;
	}
	unsigned long long rows = 0, err = 0;
	if (ROWS) {
		const uint32_t *w = reinterpret_cast<const uint32_t *>(out) + (size_t)r * cap;
		uint32_t row2 = 0xFFFFFFFFu;
		for (uint32_t i = 0; i < cnt; i++) {
			const uint32_t x = w[i];
			if (x >> 31) {
				row2 = x << 1;
				continue;
			}
			if (row2 == 0xFFFFFFFFu) {
				err++;
				continue;
			}
			const uint32_t row = row2 + ((x >> 15) & 0x7FFFu);
			if (row >= n) {
				err++;
				continue;
			}
			const uint32_t h = mdb_mixk((uint32_t)(keys[row] - KEY_LO), KBITS);
			if ((h >> rem) != d || (h & ((1u << rem) - 1u)) != (x & 0x7FFFu))
				err++;
			if (atomicOr(&seen[row >> 5], 1u << (row & 31u)) & (1u << (row & 31u)))
				err++;
			rows++;
		}
	} else {
		const uint16_t *w = reinterpret_cast<const uint16_t *>(out) + (size_t)r * cap;
		for (uint32_t i = 0; i < cnt; i++) {
			const uint32_t h = (d << rem) | w[i];
			const uint32_t rel = mdb_unmixk(h, KBITS);
			// rel = (row * MULT) mod 2^27  ->  row = rel * MULT^-1 mod 2^27
			unsigned long long inv = 1;
			for (int it = 0; it < 6; it++)
				inv *= 2ull - MULT * inv;	/* Newton: the inverse of an odd number modulo 2^64 */
			const uint32_t row = (uint32_t)((rel * inv) & ((1ull << KBITS) - 1ull));
			if (w[i] >> rem || row >= n) {
				err++;
				continue;
			}
			if (atomicOr(&seen[row >> 5], 1u << (row & 31u)) & (1u << (row & 31u)))
				err++;
			rows++;
		}
	}
	atomicAdd(&stats[0], rows);
	atomicAdd(&stats[1], err);
	atomicAdd(&stats[2], (unsigned long long)cnt);
}

struct bufs {
	long long *keys;
	void *out;
	uint32_t *cursor, *status, *seen;
	unsigned long long *stats;
	uint32_t n, cap;
};

template <typename K>
static double run(const char *name, K kernel, size_t lds, int threads, uint32_t tile, bool rows, bufs &b, int reps, bool check, int num_cus, int wg_per_cu = 1)
{
	shw_scatter_args a;
	memset(&a, 0, sizeof(a));
	a.keys = b.keys;
	a.n = b.n;
	a.key_lo = KEY_LO;
	a.kbits = KBITS;
	a.rem = KBITS - SHW_D_BITS;
	a.report = rows ? 0u : 1u;
	a.rel_hi = (uint32_t)((1ull << KBITS) - 1ull);
	a.out = b.out;
	a.cursor = b.cursor;
	a.cap = b.cap;
	a.nsub = NSUB;
	a.status = b.status;
	a.dbg = b.stats + 4;
	const uint32_t ntiles = (b.n + tile - 1) / tile;
	uint32_t grid = (uint32_t)num_cus * wg_per_cu;
	if (grid > ntiles)
		grid = ntiles;
	a.rows_per_wg = (uint32_t)((((uint64_t)b.n + grid - 1) / grid + 1) & ~1ull);
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	std::vector<float> ms(reps);
	for (int r = -1; r < reps; r++) {
		CK(hipMemsetAsync(b.cursor, 0, 4096u * NSUB * 4, 0));
		CK(hipMemsetAsync(b.status, 0, 256, 0));
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, a);
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		CK(hipGetLastError());
		if (r >= 0)
			CK(hipEventElapsedTime(&ms[r], e0, e1));
	}
	std::sort(ms.begin(), ms.end());
	const double med = ms[reps / 2];
	const double io = 8.0 * b.n + (rows ? 4.5 : 2.0) * b.n;
	printf("%-44s min %.4f med %.4f max %.4f ms   own I/O %.2f TB/s", name, ms[0], med, ms[reps - 1], io / med * 1e-9);
	if (check) {
		uint32_t st[4];
		unsigned long long h[3];
		CK(hipMemset(b.seen, 0, ((size_t)b.n + 31) / 32 * 4));
		CK(hipMemset(b.stats, 0, 24));
		if (rows)
			hipLaunchKernelGGL(k_check<true>, dim3(4096 * NSUB / 64), dim3(64), 0, 0, b.out, b.cursor, b.cap, b.keys, b.n, a.rem, b.seen, b.stats);
		else
			hipLaunchKernelGGL(k_check<false>, dim3(4096 * NSUB / 64), dim3(64), 0, 0, b.out, b.cursor, b.cap, b.keys, b.n, a.rem, b.seen, b.stats);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h, b.stats, 24, hipMemcpyDeviceToHost));
		CK(hipMemcpy(st, b.status, 16, hipMemcpyDeviceToHost));
		printf("   check: rows %llu / %u, errors %llu, words %llu, status %u -> %s", h[0], b.n, h[1], h[2], st[0], (h[0] == b.n && h[1] == 0 && st[0] == 0) ? "OK" : "WRONG");
	}
	{
		unsigned long long d[5];
		CK(hipMemcpy(d, b.stats + 4, 40, hipMemcpyDeviceToHost));
		if (d[3])
			printf("   cycles per tile: count %llu, digits %llu, stage %llu, write-out %llu (%llu tiles)", d[0] / d[3], d[1] / d[3], d[2] / d[3], d[4] / d[3], d[3]);
		CK(hipMemset(b.stats + 4, 0, 40));
	}
	printf("\n");
	fflush(stdout);
	return med;
}

static uint32_t cap_for(uint64_t n, bool row_words, int num_cus)
{
	const uint64_t regions = 4096ull * NSUB, tile = 30720;
	uint64_t cap = n * 17 / 16 / regions + 320;
	if (row_words) {
		const uint64_t ntiles = (n + tile - 1) / tile, grid = ntiles < (uint64_t)num_cus ? (ntiles ? ntiles : 1) : (uint64_t)num_cus;
		const uint64_t rows_per_wg = (n + grid - 1) / grid + 1;
		cap += ((grid + NSUB - 1) / NSUB) * ((rows_per_wg + tile - 1) / tile) + 8;
	}
	return (uint32_t)((cap + 63) & ~63ull);
}

int main(int argc, char **argv)
{
	const uint32_t n = argc > 1 ? (uint32_t)strtoull(argv[1], NULL, 10) : 100000000u;
	const int reps = argc > 2 ? atoi(argv[2]) : 9;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	bufs b;
	b.n = n;
	CK(hipMalloc(&b.keys, (size_t)n * 8 + 64));
	const uint32_t cap_l = cap_for(n, true, cus), cap_r = cap_for(n, false, cus);
	CK(hipMalloc(&b.out, (size_t)4096 * NSUB * cap_l * 4 + 4096));
	CK(hipMalloc(&b.cursor, 4096u * NSUB * 4));
	CK(hipMalloc(&b.status, 256));
	CK(hipMalloc(&b.seen, ((size_t)n + 31) / 32 * 4 + 64));
	CK(hipMalloc(&b.stats, 128));
	CK(hipMemset(b.stats, 0, 128));
	hipLaunchKernelGGL(k_gen, dim3(cus * 8), dim3(256), 0, 0, b.keys, n);
	CK(hipDeviceSynchronize());
	printf("%u rows, %d CUs, caps %u (hash words) / %u (row words), %d reps\n", n, cus, cap_r, cap_l, reps);
	for (int round = 0; round < 2; round++) {
		const bool chk = round == 0;
		b.cap = cap_r;
		run("r  old   k_shard_scatter_wide<1024,32>", k_shard_scatter_wide<1024, 32, false>, shw_scatter_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  old, loads nt", k_shard_scatter_wide<1024, 32, false, true>, shw_scatter_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   k_scatter4096_stream<1024,32> ring 4", k_scatter4096_stream<1024, 32, false>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   k_scatter4096_stream<1024,32> ring 4", k_scatter4096_stream<1024, 32, false, false, 4>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  ws    k_scatter4096_ws<12,40> ring 5", k_scatter4096_ws<12, 40, false>, shs_stream_lds(30720, 2), 1024, 30720, false, b, reps, chk, cus);
		run("r  ws    k_scatter4096_ws<12,40> ring 4", k_scatter4096_ws<12, 40, false, false, 4>, shs_stream_lds(30720, 2), 1024, 30720, false, b, reps, chk, cus);
		run("r  ws    k_scatter4096_ws<12,40> nt loads", k_scatter4096_ws<12, 40, false, false, 0, 8>, shs_stream_lds(30720, 2), 1024, 30720, false, b, reps, chk, cus);
		run("r  new   ring 4, loads nt", k_scatter4096_stream<1024, 32, false, false, 0, 0, 2>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 4, loads sc0 sc1 nt", k_scatter4096_stream<1024, 32, false, false, 0, 0, 19>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 4, loads sc1", k_scatter4096_stream<1024, 32, false, false, 0, 0, 16>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 4, loads sc0 sc1", k_scatter4096_stream<1024, 32, false, false, 0, 0, 17>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 4, nt stores", k_scatter4096_stream<1024, 32, false, false, 0, 32>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 16, loads nt", k_scatter4096_stream<1024, 32, false, false, 16, 0, 2>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 8, loads plain", k_scatter4096_stream<1024, 32, false, false, 8, 0, 0>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		run("r  new   ring 8, loads nt", k_scatter4096_stream<1024, 32, false, false, 8, 0, 2>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, chk, cus);
		b.cap = cap_l;
		run("l  old   k_shard_scatter_wide<1024,32,ROWS>", k_shard_scatter_wide<1024, 32, true>, shw_scatter_lds(32768, 4), 1024, 32768, true, b, reps, chk, cus);
		run("l  old, loads nt", k_shard_scatter_wide<1024, 32, true, true>, shw_scatter_lds(32768, 4), 1024, 32768, true, b, reps, chk, cus);
		run("l  new   k_scatter4096_stream<1024,30,ROWS> ring 5", k_scatter4096_stream<1024, 30, true>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  new   k_scatter4096_stream<1024,30,ROWS> ring 3", k_scatter4096_stream<1024, 30, true, false, 3>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  ws    k_scatter4096_ws<12,40,ROWS> ring 5", k_scatter4096_ws<12, 40, true>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  ws    k_scatter4096_ws<12,40,ROWS> ring 4", k_scatter4096_ws<12, 40, true, false, 4>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  new   ring 3, loads nt", k_scatter4096_stream<1024, 30, true, false, 3, 0, 2>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  new   ring 5, loads nt", k_scatter4096_stream<1024, 30, true, false, 5, 0, 2>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  new   ring 15, loads nt", k_scatter4096_stream<1024, 30, true, false, 15, 0, 2>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  new   ring 3, loads sc0 sc1 nt", k_scatter4096_stream<1024, 30, true, false, 3, 0, 19>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  new   ring 3, loads sc1", k_scatter4096_stream<1024, 30, true, false, 3, 0, 16>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
		run("l  ws    k_scatter4096_ws<12,40,ROWS> nt loads", k_scatter4096_ws<12, 40, true, false, 0, 8>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, chk, cus);
	}
	b.cap = cap_r;
	run("r  ws, stamped", k_scatter4096_ws<12, 40, false, false, 0, 4>, shs_stream_lds(30720, 2), 1024, 30720, false, b, 3, false, cus);
	run("r  ws, no stores, no loads, stamped", k_scatter4096_ws<12, 40, false, false, 0, 7>, shs_stream_lds(30720, 2), 1024, 30720, false, b, 3, false, cus);
	run("r  ws, no stores", k_scatter4096_ws<12, 40, false, false, 0, 1>, shs_stream_lds(30720, 2), 1024, 30720, false, b, reps, false, cus);
	run("r  ws, no loads", k_scatter4096_ws<12, 40, false, false, 0, 2>, shs_stream_lds(30720, 2), 1024, 30720, false, b, reps, false, cus);
	run("r  new, stores linear (whole lines), stamped", k_scatter4096_stream<1024, 32, false, false, 0, 20>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, false, cus);
	run("r  new, stamped", k_scatter4096_stream<1024, 32, false, false, 0, 4>, shs_stream_lds(32768, 2), 1024, 32768, false, b, 3, false, cus);
	run("r  new, no stores, no loads, stamped", k_scatter4096_stream<1024, 32, false, false, 0, 7>, shs_stream_lds(32768, 2), 1024, 32768, false, b, 3, false, cus);
	run("r  new, no stores, no loads", k_scatter4096_stream<1024, 32, false, false, 0, 3>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, false, cus);
	run("r  new, no global stores (ablation)", k_scatter4096_stream<1024, 32, false, false, 0, 1>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, false, cus);
	run("r  new, keys not loaded (ablation)", k_scatter4096_stream<1024, 32, false, false, 0, 2>, shs_stream_lds(32768, 2), 1024, 32768, false, b, reps, false, cus);
	b.cap = cap_l;
	run("l  ws, stamped", k_scatter4096_ws<12, 40, true, false, 0, 4>, shs_stream_lds(30720, 4), 1024, 30720, true, b, 3, false, cus);
	run("l  ws, no stores, no loads, stamped", k_scatter4096_ws<12, 40, true, false, 0, 7>, shs_stream_lds(30720, 4), 1024, 30720, true, b, 3, false, cus);
	run("l  new, stores linear (whole lines), stamped", k_scatter4096_stream<1024, 30, true, false, 0, 20>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, false, cus);
	run("l  new, stamped", k_scatter4096_stream<1024, 30, true, false, 0, 4>, shs_stream_lds(30720, 4), 1024, 30720, true, b, 3, false, cus);
	run("l  new, no stores, no loads, stamped", k_scatter4096_stream<1024, 30, true, false, 0, 7>, shs_stream_lds(30720, 4), 1024, 30720, true, b, 3, false, cus);
	run("l  new, no global stores (ablation)", k_scatter4096_stream<1024, 30, true, false, 0, 1>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, false, cus);
	run("l  new, keys not loaded (ablation)", k_scatter4096_stream<1024, 30, true, false, 0, 2>, shs_stream_lds(30720, 4), 1024, 30720, true, b, reps, false, cus);
	return 0;
}
