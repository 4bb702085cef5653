// Micro-benchmark: the 4096-digit one-pass scatter (mdb_dev_shard.hip, k_shard_scatter_wide) generalised to wider words -
// what the ORDERED form of the north-star query would need to drop its second partition level:
//   SRC 0  key column -> 2-byte words (hash bits below the digit)                      [the product's kernel, the baseline]
//   SRC 1  key column -> 8-byte words (row id << 16 | hash bits below the digit)       [left table of the ordered form]
//   SRC 2  8-byte group records (row id << (64 - kbits) | COUNT) -> 4-byte words       [the ordering sort in one pass]
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imidoridb_amd/csrc -o profiles/micro/scatter4096_words profiles/micro/scatter4096_words.hip
//   ./profiles/micro/scatter4096_words [rows]
#include "mdb_dev_common.h"
#include <cstdlib>
#include <type_traits>

#define SHW_D_BITS 12
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct args {
	const unsigned long long *in;	// keys (SRC 0, 1) or records (SRC 2)
	uint32_t n;
	long long key_lo;
	uint32_t kbits, rem;		// SRC 0/1: hash bits and bits below the digit; SRC 2: row-id bits, row-id bits below the digit
	uint32_t cbits;			// SRC 2: payload bits kept
	void *out;
	uint32_t *cursor;		// [nsub][4096]
	uint32_t cap, nsub;
	uint32_t *status;
	uint32_t rows_per_wg;
};

__device__ static inline void shw_barrier(void) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ static inline uint32_t shw_block_excl_scan(uint32_t v, uint32_t *tmp, uint32_t nwaves, uint32_t *total)
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

template <int SRC> struct word_of { typedef uint16_t type; };
template <> struct word_of<1> { typedef uint64_t type; };
template <> struct word_of<2> { typedef uint32_t type; };

template <int THREADS, int RPT, int SRC>
__global__ __launch_bounds__(THREADS, 4) void k_scatter4096(args a)
{
	typedef typename word_of<SRC>::type W;
	constexpr uint32_t TILE = THREADS * RPT, D = 1u << SHW_D_BITS, DPT = D / THREADS, NCHUNK = TILE / 64u, HALF = RPT / 2;
	constexpr uint64_t MARK = SRC == 2 ? 0x80000000ull : 0x8000ull;
	static_assert(TILE <= 32768u && DPT >= 2 && (DPT & 1) == 0 && (RPT % 4) == 0, "tile shape");
	extern __shared__ __attribute__((aligned(16))) uint32_t shw_lds[];
	uint32_t *const s_cnt = shw_lds;
	uint32_t *const s_delta = s_cnt + D / 2;
	uint32_t *const s_chunk = s_delta + D;
	uint32_t *const s_bad = s_chunk + NCHUNK;
	uint32_t *const s_tmp = s_bad + D / 32;
	W *const s_stage = reinterpret_cast<W *>(s_tmp + 32);
	__shared__ uint32_t s_any_bad;

	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane(), sub = blockIdx.x % a.nsub;
	const uint32_t wmask = (1u << a.rem) - 1u;
	const uint64_t limit = (1ull << a.kbits) - 1ull;
	const uint64_t le = mdb_lanemask_lt() | (1ull << lane);
	for (uint32_t i = threadIdx.x; i < D / 32; i += THREADS)
		s_bad[i] = 0u;
	if (threadIdx.x == 0)
		s_any_bad = 0u;

	const uint64_t r_begin = (uint64_t)blockIdx.x * a.rows_per_wg;
	const uint64_t r_end = r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n;
	for (uint64_t row0 = r_begin; row0 < r_end;) {
		const uint32_t len = (uint32_t)((r_end - row0) < TILE ? (r_end - row0) : TILE);
		const bool full = len == TILE;
		uint32_t tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		for (uint32_t i = tid; i < D / 2; i += THREADS)
			s_cnt[i] = 0u;
		if (tid == 0)
			s_chunk[0] = 0u;
		shw_barrier();

		uint32_t packed[RPT];				// digit << 16 | rank, or ~0
		uint32_t val[SRC == 2 ? RPT : HALF];		// SRC 0/1: two 16-bit words per register; SRC 2: the 4-byte word
#pragma unroll
		for (int hblock = 0; hblock < 2; hblock++) {
			ulonglong2 pre[HALF / 2];
#pragma unroll
			for (int r = 0; r < HALF / 2; r++) {
				const uint32_t e0 = 2u * ((uint32_t)(hblock * (HALF / 2) + r) * THREADS + tid);
				if (full || e0 + 1u < len)
					pre[r] = *reinterpret_cast<const ulonglong2 *>(a.in + row0 + e0);
				else if (e0 < len)
					pre[r] = make_ulonglong2(a.in[row0 + e0], 0ull);
				else
					pre[r] = make_ulonglong2(0ull, 0ull);
			}
#pragma unroll
			for (int r = 0; r < HALF / 2; r++) {
				const int pr = hblock * (HALF / 2) + r;
				const uint32_t e0 = 2u * ((uint32_t)pr * THREADS + tid);
				const unsigned long long kk[2] = { pre[r].x, pre[r].y };
				const bool ok[2] = { e0 < len, e0 + 1u < len };
				uint32_t w2 = 0u;
#pragma unroll
				for (int e = 0; e < 2; e++) {
					uint32_t pk = 0xFFFFFFFFu;
					if (SRC == 2) {
						const bool take = ok[e] && kk[e] != 0ull;
						uint32_t w = 0u;
						if (take) {
							const uint32_t dig = (uint32_t)(kk[e] >> 52), sh = (dig & 1u) << 4;
							const uint32_t rank = (atomicAdd(&s_cnt[dig >> 1], 1u << sh) >> sh) & 0xFFFFu;
							pk = (dig << 16) | rank;
							w = ((((uint32_t)(kk[e] >> (64u - a.kbits))) & wmask) << a.cbits) | ((uint32_t)kk[e] & ((1u << a.cbits) - 1u)) |
							    (rank == 0u ? 0x80000000u : 0u);
						}
						val[2 * pr + e] = w;
					} else {
						const unsigned long long rel = kk[e] - (unsigned long long)a.key_lo;
						const bool take = ok[e] && rel <= limit;
						if (ok[e] && !take)
							mdb_raise(a.status, 128u);
						if (take) {
							const uint32_t h = mdb_mixk((uint32_t)rel, a.kbits);
							const uint32_t dig = h >> a.rem, sh = (dig & 1u) << 4;
							const uint32_t rank = (atomicAdd(&s_cnt[dig >> 1], 1u << sh) >> sh) & 0xFFFFu;
							pk = (dig << 16) | rank;
							w2 |= ((h & wmask) | (rank == 0u ? 0x8000u : 0u)) << (16 * e);
						}
					}
					packed[2 * pr + e] = pk;
				}
				if (SRC != 2)
					val[pr] = w2;
			}
		}
		shw_barrier();

		uint32_t cnt[DPT], v = 0u;
#pragma unroll
		for (int j = 0; j < (int)DPT / 2; j++) {
			const uint32_t c2 = s_cnt[tid * (DPT / 2) + j];
			cnt[2 * j] = c2 & 0xFFFFu;
			cnt[2 * j + 1] = c2 >> 16;
			v += cnt[2 * j] + cnt[2 * j + 1] + ((cnt[2 * j] ? 1u : 0u) + (cnt[2 * j + 1] ? 1u : 0u)) * 65536u;
		}
		uint32_t tot;
		const uint32_t ex = shw_block_excl_scan(v, s_tmp, THREADS / 64, &tot);
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
					base[j] = atomicAdd(&a.cursor[sub * D + d], cnt[j]);
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

#pragma unroll
		for (int r = 0; r < RPT; r++) {
			if (packed[r] != 0xFFFFFFFFu) {
				const uint32_t dig = packed[r] >> 16;
				const uint32_t st = (s_cnt[dig >> 1] >> ((dig & 1u) << 4)) & 0xFFFFu;
				const uint32_t pos = st + (packed[r] & 0xFFFFu);
				if (SRC == 0)
					s_stage[pos] = (W)(val[r >> 1] >> (16 * (r & 1)));
				else if (SRC == 1) {
					const uint32_t rid = (uint32_t)row0 + 2u * ((uint32_t)(r >> 1) * THREADS + tid) + (uint32_t)(r & 1);
					s_stage[pos] = (W)(((uint64_t)rid << 16) | ((val[r >> 1] >> (16 * (r & 1))) & 0xFFFFu));
				} else
					s_stage[pos] = (W)val[r];
			}
		}
		{
			uint32_t ord = ord0;
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				if (cnt[j]) {
					const uint32_t d = tid * DPT + (uint32_t)j;
					if (base[j] + cnt[j] > a.cap) {
						mdb_raise(a.status, 2u);
						atomicOr(&s_bad[ord >> 5], 1u << (ord & 31u));
						s_any_bad = 1u;
					}
					s_delta[ord] = (d * a.nsub + sub) * a.cap + base[j] - st0[j];
					ord++;
				}
			}
		}
		shw_barrier();

		const bool any_bad = s_any_bad != 0u;
#pragma unroll
		for (int k = 0; k < RPT; k++) {
			const uint32_t i = (uint32_t)k * THREADS + tid;
			const W sv = i < tile_total ? s_stage[i] : (W)0;
			const uint64_t m = __ballot(((uint64_t)sv & MARK) != 0);
			if (i >= tile_total)
				continue;
			const uint32_t ord = s_chunk[(uint32_t)k * (THREADS / 64) + wave] + (uint32_t)__popcll(m & le) - 1u;
			if (any_bad && ((s_bad[ord >> 5] >> (ord & 31u)) & 1u))
				continue;
			reinterpret_cast<W *>(a.out)[i + s_delta[ord]] = (W)((uint64_t)sv & ~MARK);
		}
		shw_barrier();
		if (any_bad) {
			for (uint32_t i = tid; i < D / 32; i += THREADS)
				s_bad[i] = 0u;
			if (tid == 0)
				s_any_bad = 0u;
			shw_barrier();
		}
		row0 += len;
	}
}

// ---- two resident workgroups per CU: nothing per row is kept in registers across the phases - the keys are read TWICE (the second time
// from L2 / the Infinity Cache), counted in the first pass (non-returning LDS atomics), ranked and staged in the second (returning atomics on
// the digits' running positions); a run's first word is marked in a bitmap by the digit's owner.  24 576-row tiles: 78 KiB of LDS.
template <int THREADS, int RPT>
__global__ __launch_bounds__(THREADS, 8) void k_scatter4096_2p(args a)
{
	constexpr uint32_t TILE = THREADS * RPT, D = 1u << SHW_D_BITS, DPT = D / THREADS, NCHUNK = TILE / 64u, BATCH = (RPT % 16) ? 4 : 8;
	static_assert((RPT % (2 * BATCH)) == 0 && DPT == 4, "tile shape");
	extern __shared__ __attribute__((aligned(16))) uint32_t shw_lds[];
	uint32_t *const s_cnt = shw_lds;			// [D / 2] counts, then running positions (16-bit halves)
	uint32_t *const s_delta = s_cnt + D / 2;		// [D] per non-empty digit, in digit order
	uint32_t *const s_chunk = s_delta + D;			// [NCHUNK]
	uint32_t *const s_mark = s_chunk + NCHUNK;		// [TILE / 32] first words of runs
	uint32_t *const s_bad = s_mark + TILE / 32;		// [D / 32]
	uint32_t *const s_tmp = s_bad + D / 32;			// [32]
	uint16_t *const s_stage = reinterpret_cast<uint16_t *>(s_tmp + 32);	// [TILE]
	__shared__ uint32_t s_any_bad;
	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane(), sub = blockIdx.x % a.nsub;
	const uint32_t wmask = (1u << a.rem) - 1u;
	const uint64_t limit = (1ull << a.kbits) - 1ull;
	const uint64_t le = mdb_lanemask_lt() | (1ull << lane);
	for (uint32_t i = threadIdx.x; i < D / 32; i += THREADS)
		s_bad[i] = 0u;
	if (threadIdx.x == 0)
		s_any_bad = 0u;
	const uint64_t r_begin = (uint64_t)blockIdx.x * a.rows_per_wg;
	const uint64_t r_end = r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n;
	for (uint64_t row0 = r_begin; row0 < r_end;) {
		const uint32_t len = (uint32_t)((r_end - row0) < TILE ? (r_end - row0) : TILE);
		const bool full = len == TILE;
		uint32_t tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		for (uint32_t i = tid; i < D / 2; i += THREADS)
			s_cnt[i] = 0u;
		for (uint32_t i = tid; i < TILE / 32; i += THREADS)
			s_mark[i] = 0u;
		if (tid == 0)
			s_chunk[0] = 0u;
		shw_barrier();
		// pass 1: count
#pragma unroll 1
		for (int b0 = 0; b0 < RPT / 2; b0 += BATCH) {
			ulonglong2 pre[BATCH];
#pragma unroll
			for (int r = 0; r < (int)BATCH; r++) {
				const uint32_t e0 = 2u * ((uint32_t)(b0 + r) * THREADS + tid);
				if (full || e0 + 1u < len)
					pre[r] = *reinterpret_cast<const ulonglong2 *>(a.in + row0 + e0);
				else if (e0 < len)
					pre[r] = make_ulonglong2(a.in[row0 + e0], ~0ull);
				else
					pre[r] = make_ulonglong2(~0ull, ~0ull);
			}
#pragma unroll
			for (int r = 0; r < (int)BATCH; r++) {
				const unsigned long long kk[2] = { pre[r].x, pre[r].y };
#pragma unroll
				for (int e = 0; e < 2; e++) {
					const unsigned long long rel = kk[e] - (unsigned long long)a.key_lo;
					if (rel <= limit) {
						const uint32_t dig = mdb_mixk((uint32_t)rel, a.kbits) >> a.rem;
						atomicAdd(&s_cnt[dig >> 1], 1u << ((dig & 1u) << 4));
					}
				}
			}
		}
		shw_barrier();
		// digits: counts -> tile-local starts (the counters become running positions), cursor atomics, chunk table, run marks
		uint32_t cnt[DPT], v = 0u;
#pragma unroll
		for (int j = 0; j < (int)DPT / 2; j++) {
			const uint32_t c2 = s_cnt[tid * (DPT / 2) + j];
			cnt[2 * j] = c2 & 0xFFFFu;
			cnt[2 * j + 1] = c2 >> 16;
			v += cnt[2 * j] + cnt[2 * j + 1] + ((cnt[2 * j] ? 1u : 0u) + (cnt[2 * j + 1] ? 1u : 0u)) * 65536u;
		}
		uint32_t tot;
		const uint32_t ex = shw_block_excl_scan(v, s_tmp, THREADS / 64, &tot);
		const uint32_t tile_total = tot & 0xFFFFu;
		{
			uint32_t start = ex & 0xFFFFu, ord = ex >> 16, st0[DPT];
#pragma unroll
			for (int j = 0; j < (int)DPT; j++) {
				const uint32_t d = tid * DPT + (uint32_t)j;
				st0[j] = start;
				if (cnt[j]) {
					const uint32_t base = atomicAdd(&a.cursor[sub * D + d], cnt[j]);
					for (uint32_t c = (start >> 6) + 1u; (c << 6) <= start + cnt[j] && c < NCHUNK; c++)
						s_chunk[c] = ord + 1u;
					atomicOr(&s_mark[start >> 5], 1u << (start & 31u));
					if (base + cnt[j] > a.cap) {
						mdb_raise(a.status, 2u);
						atomicOr(&s_bad[ord >> 5], 1u << (ord & 31u));
						s_any_bad = 1u;
					}
					s_delta[ord] = (d * a.nsub + sub) * a.cap + base - start;
					ord++;
					start += cnt[j];
				}
			}
#pragma unroll
			for (int j = 0; j < (int)DPT / 2; j++)
				s_cnt[tid * (DPT / 2) + j] = st0[2 * j] | (st0[2 * j + 1] << 16);
		}
		shw_barrier();
		// pass 2: the same keys again, ranked at the digits' running positions and staged
#pragma unroll 1
		for (int b0 = 0; b0 < RPT / 2; b0 += BATCH) {
			ulonglong2 pre[BATCH];
#pragma unroll
			for (int r = 0; r < (int)BATCH; r++) {
				const uint32_t e0 = 2u * ((uint32_t)(b0 + r) * THREADS + tid);
				if (full || e0 + 1u < len)
					pre[r] = *reinterpret_cast<const ulonglong2 *>(a.in + row0 + e0);
				else if (e0 < len)
					pre[r] = make_ulonglong2(a.in[row0 + e0], ~0ull);
				else
					pre[r] = make_ulonglong2(~0ull, ~0ull);
			}
#pragma unroll
			for (int r = 0; r < (int)BATCH; r++) {
				const unsigned long long kk[2] = { pre[r].x, pre[r].y };
#pragma unroll
				for (int e = 0; e < 2; e++) {
					const unsigned long long rel = kk[e] - (unsigned long long)a.key_lo;
					if (rel <= limit) {
						const uint32_t h = mdb_mixk((uint32_t)rel, a.kbits), dig = h >> a.rem, sh = (dig & 1u) << 4;
						const uint32_t pos = (atomicAdd(&s_cnt[dig >> 1], 1u << sh) >> sh) & 0xFFFFu;
						s_stage[pos] = (uint16_t)(h & wmask);
					}
				}
			}
		}
		shw_barrier();
		const bool any_bad = s_any_bad != 0u;
#pragma unroll 4
		for (int k = 0; k < RPT; k++) {
			const uint32_t i = (uint32_t)k * THREADS + tid;
			const uint32_t mk = i < tile_total ? (s_mark[i >> 5] >> (i & 31u)) & 1u : 0u;
			const uint64_t m = __ballot(mk != 0u);
			if (i >= tile_total)
				continue;
			const uint32_t ord = s_chunk[(uint32_t)k * (THREADS / 64) + wave] + (uint32_t)__popcll(m & le) - 1u;
			if (any_bad && ((s_bad[ord >> 5] >> (ord & 31u)) & 1u))
				continue;
			reinterpret_cast<uint16_t *>(a.out)[i + s_delta[ord]] = s_stage[i];
		}
		shw_barrier();
		if (any_bad) {
			for (uint32_t i = tid; i < D / 32; i += THREADS)
				s_bad[i] = 0u;
			if (tid == 0)
				s_any_bad = 0u;
			shw_barrier();
		}
		row0 += len;
	}
}

template <int THREADS, int RPT>
static void run2p(const char *what, const unsigned long long *in, uint32_t n, uint32_t kbits, uint32_t cus, uint32_t per_cu)
{
	const uint32_t D = 1u << SHW_D_BITS, nsub = 8, tile = THREADS * RPT;
	const uint32_t cap = (uint32_t)(((uint64_t)n * 17 / 16 / (D * nsub) + 320 + 63) & ~63ull);
	uint32_t *cur, *status;
	uint16_t *out;
	CK(hipMalloc(&cur, (size_t)D * nsub * 4));
	CK(hipMalloc(&status, 64));
	CK(hipMalloc(&out, (size_t)D * nsub * cap * 2));
	CK(hipMemset(status, 0, 64));
	args a;
	memset(&a, 0, sizeof(a));
	a.in = in;
	a.n = n;
	a.kbits = kbits;
	a.rem = kbits - 12;
	a.out = out;
	a.cursor = cur;
	a.cap = cap;
	a.nsub = nsub;
	a.status = status;
	const uint32_t ntiles = (n + tile - 1) / tile, grid = ntiles < per_cu * cus ? ntiles : per_cu * cus;
	a.rows_per_wg = (uint32_t)((((uint64_t)n + grid - 1) / grid + 1) & ~1ull);
	const size_t lds = (size_t)4 * (D / 2 + D + tile / 64 + tile / 32 + D / 32 + 32) + (size_t)2 * tile;
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter4096_2p<THREADS, RPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	float best = 1e9f, sum = 0;
	const int iters = 6;
	for (int it = 0; it < iters; it++) {
		CK(hipMemsetAsync(cur, 0, (size_t)D * nsub * 4));
		hipEventRecord(e0);
		hipLaunchKernelGGL((k_scatter4096_2p<THREADS, RPT>), dim3(grid), dim3(THREADS), lds, 0, a);
		hipEventRecord(e1);
		CK(hipEventSynchronize(e1));
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		if (it) {
			sum += ms;
			best = ms < best ? ms : best;
		}
	}
	uint32_t st = 0;
	CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
	std::vector<uint32_t> hc((size_t)D * nsub);
	CK(hipMemcpy(hc.data(), cur, hc.size() * 4, hipMemcpyDeviceToHost));
	uint64_t total = 0;
	for (uint32_t c : hc)
		total += c;
	// spot check: the multiset of words of a few digits equals what the keys hash to
	std::vector<unsigned long long> hk(n);
	CK(hipMemcpy(hk.data(), in, (size_t)n * 8, hipMemcpyDeviceToHost));
	uint64_t bad = 0;
	for (uint32_t d : { 0u, 777u, 4095u }) {
		std::vector<uint32_t> want(1u << (kbits - 12), 0), got(1u << (kbits - 12), 0);
		for (uint32_t i = 0; i < n; i++) {
			const uint32_t h = mdb_mixk((uint32_t)hk[i], kbits);
			if ((h >> (kbits - 12)) == d)
				want[h & ((1u << (kbits - 12)) - 1u)]++;
		}
		for (uint32_t sb = 0; sb < nsub; sb++) {
			const uint32_t c = hc[(size_t)sb * D + d];
			std::vector<uint16_t> w(c);
			CK(hipMemcpy(w.data(), out + ((size_t)d * nsub + sb) * cap, (size_t)c * 2, hipMemcpyDeviceToHost));
			for (uint16_t x : w)
				got[x]++;
		}
		for (size_t i = 0; i < want.size(); i++)
			bad += want[i] != got[i];
	}
	printf("%-58s tile %5u lds %6zu B x %u / CU: avg %.3f ms best %.3f ms  rows placed %llu of %u, status %u, digits checked: %llu wrong values\n", what, tile, lds, per_cu,
	       sum / (iters - 1), best, (unsigned long long)total, n, st, (unsigned long long)bad);
	hipFree(cur);
	hipFree(status);
	hipFree(out);
}

static size_t lds_bytes(uint32_t tile, size_t wb)
{
	const uint32_t D = 1u << SHW_D_BITS;
	return (size_t)4 * (D / 2 + D + tile / 64 + D / 32 + 32) + wb * tile;
}

__global__ void k_gen(unsigned long long *keys, uint32_t n, uint32_t total)
{
	// a permutation of [0, total): i * odd mod 2^b folded into range by cycle walking would be exact; a hash modulo total is close enough here
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		keys[i] = mdb_fmix64(i + 12345) % total;
}

__global__ void k_gen_rec(unsigned long long *rec, uint32_t n, uint32_t kbits)
{
	// records of a unique-key join: every row id once, scrambled; COUNT = 1
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t rid = (i * 2654435761ull + 17) % n;	// (n not a multiple of the multiplier's factors: a permutation when gcd = 1)
		rec[i] = (rid << (64 - kbits)) | 1ull;
	}
}


template <int THREADS, int RPT, int SRC>
static void run(const char *what, const unsigned long long *in, uint32_t n, uint32_t kbits, uint32_t rem, uint32_t cbits, uint32_t cus)
{
	typedef typename word_of<SRC>::type W;
	const uint32_t D = 1u << SHW_D_BITS, nsub = 8, tile = THREADS * RPT;
	const uint32_t cap = (uint32_t)(((uint64_t)n * 17 / 16 / (D * nsub) + 320 + 63) & ~63ull);
	uint32_t *cur, *status;
	W *out;
	CK(hipMalloc(&cur, (size_t)D * nsub * 4));
	CK(hipMalloc(&status, 64));
	CK(hipMalloc(&out, (size_t)D * nsub * cap * sizeof(W)));
	CK(hipMemset(status, 0, 64));
	args a;
	memset(&a, 0, sizeof(a));
	a.in = in;
	a.n = n;
	a.kbits = kbits;
	a.rem = rem;
	a.cbits = cbits;
	a.out = out;
	a.cursor = cur;
	a.cap = cap;
	a.nsub = nsub;
	a.status = status;
	const uint32_t ntiles = (n + tile - 1) / tile, grid = ntiles < cus ? ntiles : cus;
	a.rows_per_wg = (uint32_t)((((uint64_t)n + grid - 1) / grid + 1) & ~1ull);
	const size_t lds = lds_bytes(tile, sizeof(W));
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter4096<THREADS, RPT, SRC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	float best = 1e9f, sum = 0;
	const int iters = 6;
	for (int it = 0; it < iters; it++) {
		CK(hipMemsetAsync(cur, 0, (size_t)D * nsub * 4));
		hipEventRecord(e0);
		hipLaunchKernelGGL((k_scatter4096<THREADS, RPT, SRC>), dim3(grid), dim3(THREADS), lds, 0, a);
		hipEventRecord(e1);
		CK(hipEventSynchronize(e1));
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		if (it) {
			sum += ms;
			best = ms < best ? ms : best;
		}
	}
	uint32_t st = 0;
	CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
	std::vector<uint32_t> hc((size_t)D * nsub);
	CK(hipMemcpy(hc.data(), cur, hc.size() * 4, hipMemcpyDeviceToHost));
	uint64_t total = 0;
	uint32_t mx = 0;
	for (uint32_t c : hc) {
		total += c;
		mx = c > mx ? c : mx;
	}
	printf("%-58s tile %5u lds %6zu B: avg %.3f ms best %.3f ms  (%u B words, %.2f GB read+written -> %.0f GB/s)  rows placed %llu of %u, fullest region %u of %u, status %u\n",
	       what, tile, lds, sum / (iters - 1), best, (unsigned)sizeof(W), (8.0 + sizeof(W)) * n / 1e9, (8.0 + sizeof(W)) * n / (sum / (iters - 1) * 1e-3) / 1e9,
	       (unsigned long long)total, n, mx, cap, st);
	// spot check: every word of a few regions belongs there
	if (SRC == 1) {
		std::vector<unsigned long long> hk(n);
		CK(hipMemcpy(hk.data(), in, (size_t)n * 8, hipMemcpyDeviceToHost));
		uint64_t bad = 0, seen = 0;
		for (uint32_t d : { 0u, 1u, 777u, 4095u })
			for (uint32_t s = 0; s < nsub; s++) {
				const uint32_t c = hc[(size_t)s * D + d];
				std::vector<uint64_t> w(c);
				CK(hipMemcpy(w.data(), out + ((size_t)d * nsub + s) * cap, (size_t)c * 8, hipMemcpyDeviceToHost));
				for (uint64_t x : w) {
					const uint32_t rid = (uint32_t)(x >> 16), h = mdb_mixk((uint32_t)hk[rid], kbits);
					bad += (h >> rem) != d || (h & ((1u << rem) - 1u)) != (x & 0xFFFFu);
					seen++;
				}
			}
		printf("    checked %llu words: %llu wrong\n", (unsigned long long)seen, (unsigned long long)bad);
	}
	if (SRC == 2) {
		uint64_t bad = 0, seen = 0;
		for (uint32_t d : { 0u, 1u, 777u, 3000u })
			for (uint32_t s = 0; s < nsub; s++) {
				const uint32_t c = hc[(size_t)s * D + d];
				std::vector<uint32_t> w(c);
				CK(hipMemcpy(w.data(), out + ((size_t)d * nsub + s) * cap, (size_t)c * 4, hipMemcpyDeviceToHost));
				for (uint32_t x : w) {
					bad += (x & 1u) != 1u || (x >> (cbits + rem)) != 0u;
					seen++;
				}
			}
		printf("    checked %llu words: %llu wrong\n", (unsigned long long)seen, (unsigned long long)bad);
	}
	hipFree(cur);
	hipFree(status);
	hipFree(out);
}

int main(int argc, char **argv)
{
	const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 100000000u;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const uint32_t cus = (uint32_t)prop.multiProcessorCount;
	uint32_t kbits = 1;
	while ((1ull << kbits) < n)
		kbits++;
	unsigned long long *keys, *rec;
	CK(hipMalloc(&keys, (size_t)n * 8));
	CK(hipMalloc(&rec, (size_t)n * 8));
	k_gen<<<4096, 256>>>(keys, n, n);
	k_gen_rec<<<4096, 256>>>(rec, n, kbits);
	CK(hipDeviceSynchronize());
	printf("%u rows, %u CUs, kbits %u (digits 4096, %u bits below)\n", n, cus, kbits, kbits - 12);
	run<1024, 32, 0>("key -> 2-byte words (the product's kernel)", keys, n, kbits, kbits - 12, 0, cus);
	run2p<1024, 32>("two passes over the keys, 32 768-row tiles, 1 / CU", keys, n, kbits, cus, 1);
	run2p<1024, 16>("two passes over the keys, 16 384-row tiles, 2 / CU", keys, n, kbits, cus, 2);
	run2p<1024, 16>("two passes over the keys, 16 384-row tiles, 1 / CU", keys, n, kbits, cus, 1);
	run2p<1024, 24>("two passes over the keys, 24 576-row tiles, 2 / CU", keys, n, kbits, cus, 2);
	run<1024, 16, 0>("key -> 2-byte words, 16 384-row tiles", keys, n, kbits, kbits - 12, 0, cus);
	run<1024, 16, 0>("key -> 2-byte words, 16 384-row tiles, 2 workgroups / CU", keys, n, kbits, kbits - 12, 0, 2 * cus);
	run<512, 32, 0>("key -> 2-byte words, 16 384-row tiles, 512 thr, 2 / CU", keys, n, kbits, kbits - 12, 0, 2 * cus);
	run<512, 16, 0>("key -> 2-byte words, 8192-row tiles, 512 thr, 3 / CU", keys, n, kbits, kbits - 12, 0, 3 * cus);
	run<1024, 16, 1>("key -> 8-byte words (row id | hash bits)", keys, n, kbits, kbits - 12, 0, cus);
	run<512, 16, 1>("key -> 8-byte words, 8192-row tiles, 512 threads", keys, n, kbits, kbits - 12, 0, 2 * cus);
	run<1024, 32, 2>("8-byte records -> 4-byte words by row id", rec, n, kbits, kbits - 12, 5, cus);
	run<1024, 16, 2>("8-byte records -> 4-byte words, 16 384-row tiles", rec, n, kbits, kbits - 12, 5, cus);
	run<512, 32, 2>("8-byte records -> 4-byte words, 16 384-row tiles, 512 thr", rec, n, kbits, kbits - 12, 5, 2 * cus);
	return 0;
}
