/*
 * mdb_dev_filter.hip - predicate filter: wavefront ballot + ordered stream compaction.
 *
 * Replaces the reference's WHERE pass (reference src/engine/executor_select.c:1435-1463), which
 * walks every early-materialised row, re-resolves each operand's column by string compare
 * (:776-789) and tombstones the rows that fail.  Here:
 *
 *   k_pred_bits   each lane evaluates the predicate program for one tuple; a 64-lane __ballot
 *                 turns the wave's results into one 64-bit word of the pass bitmap (n/8 bytes),
 *                 and the per-block pass counts are accumulated
 *   (scan)        exclusive scan of the block counts -> output offset of every block
 *   k_bits_to_sel re-reads only the bitmap: popcount prefix inside the block gives each passing
 *                 tuple its output slot; tuple positions are written in ascending order
 *
 * so the input columns are read exactly once (coalesced when the stream is a base table) and the
 * result order equals the reference's survivor order.
 *
 * NULL semantics as the reference: a comparison with a NULL operand is false (:557-579, :629-631),
 * IS [NOT] NULL reads the NULL bit (:965), AND/OR/XOR combine plain booleans (:1041-1058).
 */
#include "mdb_dev_internal.h"

#define FILT_THREADS 256
#define FILT_WAVES (FILT_THREADS / MDB_WAVE)
#define FILT_WORDS_PER_WAVE 16
#define FILT_WORDS_PER_BLOCK (FILT_WAVES * FILT_WORDS_PER_WAVE)	/* 64 words */
#define FILT_TUPLES_PER_BLOCK (FILT_WORDS_PER_BLOCK * 64)		/* 4096 tuples */

struct pred_args {
	mdb_pred_insn insn[MDB_PRED_MAX_INSNS];
	mdb_col_binding cols[MDB_PRED_MAX_SLOTS];
	int32_t n_insns;
	int32_t pad;
};

__device__ static inline bool pred_load(const mdb_col_binding &c, uint64_t k, uint64_t *v)
{
	const uint64_t row = c.rid ? (uint64_t)c.rid[k] : k;
	if (c.nullbits && mdb_bit_is_set(c.nullbits, row))
		return false;
	*v = reinterpret_cast<const uint64_t *>(c.values)[row];
	return true;
}

__device__ static inline bool pred_cmp(int32_t cmp, int32_t type, uint64_t x, uint64_t y)
{
	if (type == MDB_T_DOUBLE) {
		const double a = __longlong_as_double((long long)x), b = __longlong_as_double((long long)y);
		switch (cmp) {
		case MDB_CMP_LT: return a < b;
		case MDB_CMP_GT: return a > b;
		case MDB_CMP_NE: return a != b;
		case MDB_CMP_EQ: return a == b;
		case MDB_CMP_LE: return a <= b;
		case MDB_CMP_GE: return a >= b;
		}
		return false;
	}
	const int64_t a = (int64_t)x, b = (int64_t)y;
	switch (cmp) {
	case MDB_CMP_LT: return a < b;
	case MDB_CMP_GT: return a > b;
	case MDB_CMP_NE: return a != b;
	case MDB_CMP_EQ: return a == b;
	case MDB_CMP_LE: return a <= b;
	case MDB_CMP_GE: return a >= b;
	}
	return false;
}

__device__ static inline bool pred_eval(const pred_args &p, uint64_t k)
{
	uint64_t st = 0;	/* boolean stack, top = bit 0 */
	for (int i = 0; i < p.n_insns; i++) {
		const mdb_pred_insn &in = p.insn[i];
		bool b = false;
		uint64_t x, y;
		switch (in.op) {
		case MDB_P_CMP_COL_CONST:
			b = pred_load(p.cols[in.a], k, &x) && pred_cmp(in.cmp, in.type, x, (uint64_t)in.imm);
			st = (st << 1) | (uint64_t)b;
			break;
		case MDB_P_CMP_CONST_COL:
			b = pred_load(p.cols[in.a], k, &x) && pred_cmp(in.cmp, in.type, (uint64_t)in.imm, x);
			st = (st << 1) | (uint64_t)b;
			break;
		case MDB_P_CMP_COL_COL: {
			const bool okx = pred_load(p.cols[in.a], k, &x);
			const bool oky = pred_load(p.cols[in.b], k, &y);
			b = okx && oky && pred_cmp(in.cmp, in.type, x, y);
			st = (st << 1) | (uint64_t)b;
			break;
		}
		case MDB_P_ISNULL:
			b = !pred_load(p.cols[in.a], k, &x);
			b = b != (in.cmp != 0);
			st = (st << 1) | (uint64_t)b;
			break;
		case MDB_P_IN_BITS: {
			const bool okx = pred_load(p.cols[in.a], k, &x);
			const bool in_table = okx && (x >> 6) < (uint64_t)(uint32_t)in.b;
			const bool hit = in_table && ((reinterpret_cast<const uint64_t *>((uintptr_t)in.imm)[x >> 6] >> (x & 63u)) & 1u);
			b = okx && (hit != (in.cmp != 0));
			st = (st << 1) | (uint64_t)b;
			break;
		}
		case MDB_P_CONST:
			st = (st << 1) | (uint64_t)(in.imm != 0);
			break;
		case MDB_P_AND:
		case MDB_P_OR:
		case MDB_P_XOR: {
			const uint64_t r = st & 1, l = (st >> 1) & 1;
			const uint64_t v = in.op == MDB_P_AND ? (l & r) : (in.op == MDB_P_OR ? (l | r) : (l ^ r));
			st = ((st >> 2) << 1) | v;
			break;
		}
		default:
			break;
		}
	}
	return st & 1;
}

/* Two adjacent tuples (k0 even) of a stream that IS a base table (no row-id indirection): every operand is one
 * 16-byte load (8-byte accesses reach only ~0.6x of that rate on gfx950, MI355X_MICROARCH.md). */
__device__ static inline void pred_load_pair(const mdb_col_binding &c, uint64_t k0, bool both, uint64_t v[2], bool ok[2])
{
	const uint64_t *vals = reinterpret_cast<const uint64_t *>(c.values);
	if (both) {
		const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(vals + k0);
		v[0] = q.x;
		v[1] = q.y;
	} else {
		v[0] = vals[k0];
		v[1] = 0;
	}
	ok[0] = true;
	ok[1] = both;
	if (c.nullbits) {
		const uint64_t w = c.nullbits[k0 >> 6] >> (k0 & 63);	/* k0 is even: both bits live in one word */
		ok[0] = !(w & 1ull);
		ok[1] = both && !(w & 2ull);
	}
}

/* x0 / ok0 = the pair of column slot 0, loaded by the caller ahead of time (several tuples' loads in flight) */
__device__ static inline void pred_eval_pair(const pred_args &p, uint64_t k0, bool both, const uint64_t x0[2], const bool ok0[2], bool out[2])
{
	uint64_t st0 = 0, st1 = 0;
	auto load = [&](int slot, uint64_t v[2], bool ok[2]) {
		if (slot == 0) {
			v[0] = x0[0];
			v[1] = x0[1];
			ok[0] = ok0[0];
			ok[1] = ok0[1];
		} else {
			pred_load_pair(p.cols[slot], k0, both, v, ok);
		}
	};
	for (int i = 0; i < p.n_insns; i++) {
		const mdb_pred_insn &in = p.insn[i];
		uint64_t x[2], y[2];
		bool okx[2], oky[2], b0 = false, b1 = false;
		switch (in.op) {
		case MDB_P_CMP_COL_CONST:
			load(in.a, x, okx);
			b0 = okx[0] && pred_cmp(in.cmp, in.type, x[0], (uint64_t)in.imm);
			b1 = okx[1] && pred_cmp(in.cmp, in.type, x[1], (uint64_t)in.imm);
			break;
		case MDB_P_CMP_CONST_COL:
			load(in.a, x, okx);
			b0 = okx[0] && pred_cmp(in.cmp, in.type, (uint64_t)in.imm, x[0]);
			b1 = okx[1] && pred_cmp(in.cmp, in.type, (uint64_t)in.imm, x[1]);
			break;
		case MDB_P_CMP_COL_COL:
			load(in.a, x, okx);
			load(in.b, y, oky);
			b0 = okx[0] && oky[0] && pred_cmp(in.cmp, in.type, x[0], y[0]);
			b1 = okx[1] && oky[1] && pred_cmp(in.cmp, in.type, x[1], y[1]);
			break;
		case MDB_P_ISNULL:
			load(in.a, x, okx);
			b0 = (!okx[0]) != (in.cmp != 0);
			b1 = (!okx[1]) != (in.cmp != 0);
			break;
		case MDB_P_IN_BITS: {
			load(in.a, x, okx);
			const uint64_t *const tab = reinterpret_cast<const uint64_t *>((uintptr_t)in.imm);
			const bool h0 = okx[0] && (x[0] >> 6) < (uint64_t)(uint32_t)in.b && ((tab[x[0] >> 6] >> (x[0] & 63u)) & 1u);
			const bool h1 = okx[1] && (x[1] >> 6) < (uint64_t)(uint32_t)in.b && ((tab[x[1] >> 6] >> (x[1] & 63u)) & 1u);
			b0 = okx[0] && (h0 != (in.cmp != 0));
			b1 = okx[1] && (h1 != (in.cmp != 0));
			break;
		}
		case MDB_P_CONST:
			b0 = b1 = in.imm != 0;
			break;
		default: {	/* AND / OR / XOR */
			const uint64_t r0 = st0 & 1, l0 = (st0 >> 1) & 1, r1 = st1 & 1, l1 = (st1 >> 1) & 1;
			st0 = ((st0 >> 2) << 1) | (in.op == MDB_P_AND ? (l0 & r0) : (in.op == MDB_P_OR ? (l0 | r0) : (l0 ^ r0)));
			st1 = ((st1 >> 2) << 1) | (in.op == MDB_P_AND ? (l1 & r1) : (in.op == MDB_P_OR ? (l1 | r1) : (l1 ^ r1)));
			continue;
		}
		}
		st0 = (st0 << 1) | (uint64_t)b0;
		st1 = (st1 << 1) | (uint64_t)b1;
	}
	out[0] = st0 & 1;
	out[1] = both && (st1 & 1);
}

/* bitmap -> compacted output columns, in row order: the surviving rows of up to FP_MAX_COLS base columns are written at
 * their final positions (block offset + rank inside the block); NULL bits are rebuilt one output word at a time. */
#define FP_MAX_COLS 16
struct fp_args {
	const uint64_t *src[FP_MAX_COLS];
	const uint64_t *src_null[FP_MAX_COLS];
	uint64_t *dst[FP_MAX_COLS];
	unsigned long long *dst_null[FP_MAX_COLS];	/* zero-filled before the launch: rows of one word may come from two blocks */
	int ncols;
	uint32_t *sel_out;	/* != NULL: also (or only, ncols == 0) the surviving rows' positions - a selection vector (mdb_dev_filter) */
};


/* ---- single-pass scan + WHERE + projection (mdb_dev_filter_project) --------------------------------------------
 * The predicate kernels below can finish the job themselves: a workgroup that knows how many of its 4096 rows pass
 * learns how many pass in all the row blocks before it (decoupled look-back over one 8-byte word per block: flag +
 * count in ONE word, stored and polled with agent-scope atomics - the granule form of the inter-workgroup hand-off,
 * cdna_hip_programming.md Guideline 16) and writes its surviving rows of every projected column straight to their final
 * positions.  The column is read once and the output written once: no bitmap pass, no scan launches, no selection
 * vector.  Row blocks are handed out by a ticket counter, so every block a workgroup waits for has started before it;
 * the poll is bounded all the same (status bit 8 on time-out: reported as an error, never a hang). */
struct fp_fused {
	fp_args cols;
	unsigned long long *state;	/* [row blocks]: 0 = nothing yet; FZ_AGG | rows of the block; FZ_PFX | rows up to and including it */
	uint32_t *ticket;
	uint32_t *status;
};
#define FZ_AGG (1ull << 62)
#define FZ_PFX (2ull << 62)
#define FZ_VAL ((1ull << 62) - 1ull)
#define FZ_TIMEOUT 256u
#define FZ_TILES 1u			/* row blocks (of 4096 rows) per workgroup in the single-pass form (8 per workgroup, i.e. fewer look-backs, measured 25 % slower: less parallelism) */

__device__ static inline uint32_t filt_block_id(const fp_fused &fz)
{
	__shared__ uint32_t s_bid;
	if (!fz.state)
		return blockIdx.x;
	if (threadIdx.x == 0)
		s_bid = atomicAdd(fz.ticket, 1u);
	__syncthreads();
	return s_bid;
}

__device__ static inline uint32_t filt_wave_sum_u32(uint32_t v)
{
#pragma unroll
	for (int d = 1; d < MDB_WAVE; d <<= 1)
		v += (uint32_t)__shfl_xor((int)v, d, MDB_WAVE);
	return v;
}

/* cnt = rows of the workgroup's FZ_TILES row blocks (bid * FZ_TILES ...) that pass (uniform); their bitmap words are in
 * bits[] (written by this workgroup before the barrier the caller has just passed) */
__device__ static inline void filt_fused_tail(const fp_fused &fz, uint32_t bid, uint32_t cnt, const uint64_t *bits, uint64_t n)
{
	__shared__ uint32_t s_base;
	__shared__ uint32_t s_woff[FZ_TILES * FILT_WORDS_PER_BLOCK];
	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane();
	const uint64_t nwords = (n + 63) >> 6;
	const uint64_t bword0 = (uint64_t)bid * FZ_TILES * FILT_WORDS_PER_BLOCK;
	if (wave == 0) {
		if (lane == 0)
			__hip_atomic_store(&fz.state[bid], (bid == 0 ? FZ_PFX : FZ_AGG) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		/* the words' offsets inside the workgroup's rows, while the predecessors' counts travel */
		uint32_t carry = 0;
		for (uint32_t i = 0; i < FZ_TILES * FILT_WORDS_PER_BLOCK; i += MDB_WAVE) {
			const uint64_t w = bword0 + i + lane;
			const uint32_t c = w < nwords ? (uint32_t)__popcll(bits[w]) : 0;
			const uint32_t incl = mdb_wave_incl_scan(c);
			s_woff[i + lane] = carry + incl - c;
			carry += (uint32_t)__shfl((int)incl, MDB_WAVE - 1, MDB_WAVE);
		}
		uint32_t excl = 0;
		if (bid > 0) {
			int64_t look = (int64_t)bid - 1;
			uint32_t spins = 0;
			for (;;) {
				const int64_t idx = look - (int64_t)lane;	/* lane 0 = the nearest predecessor */
				unsigned long long v = FZ_PFX;			/* in front of block 0: a prefix of zero rows */
				for (;;) {
					if (idx >= 0)
						v = __hip_atomic_load(&fz.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (!__ballot((v >> 62) == 0))
						break;
					if (++spins > (1u << 22)) {
						if (lane == 0)
							atomicOr(fz.status, FZ_TIMEOUT);
						v = FZ_PFX;
						break;
					}
					__builtin_amdgcn_s_sleep(2);
				}
				const uint64_t pm = __ballot((v >> 62) == 2);	/* predecessors that already know their inclusive prefix */
				if (pm) {
					const uint32_t first = (uint32_t)__ffsll((long long)pm) - 1u;
					excl += filt_wave_sum_u32(lane <= first ? (uint32_t)(v & FZ_VAL) : 0u);
					break;
				}
				excl += filt_wave_sum_u32((uint32_t)(v & FZ_VAL));
				look -= MDB_WAVE;
			}
			if (lane == 0)
				__hip_atomic_store(&fz.state[bid], FZ_PFX | (unsigned long long)(excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (lane == 0)
			s_base = excl;
	}
	__syncthreads();
	const uint32_t base_off = s_base;
	const uint64_t lt = mdb_lanemask_lt();
	const fp_args &a = fz.cols;
	for (uint32_t sub = 0; sub < FZ_TILES; sub++) {
#pragma unroll 2
		for (int r = 0; r < FILT_WORDS_PER_WAVE; r++) {
			const uint32_t wl = sub * FILT_WORDS_PER_BLOCK + wave * FILT_WORDS_PER_WAVE + r;
			const uint64_t w = bword0 + wl;
			if (w >= nwords)
				break;
			const uint64_t m = bits[w];
			const bool keep = (m >> lane) & 1ull;
			const uint64_t row = (w << 6) + lane;
			const uint32_t pos = base_off + s_woff[wl] + (uint32_t)__popcll(m & lt);
			for (int c = 0; c < a.ncols; c++) {
				bool isnull = false;
				if (keep) {
					a.dst[c][pos] = a.src[c][row];
					if (a.src_null[c])
						isnull = mdb_bit_is_set(a.src_null[c], row);
				}
				if (a.dst_null[c] && isnull)
					atomicOr(&a.dst_null[c][pos >> 6], 1ull << (pos & 63));
			}
		}
	}
}

/* bit i of `even` -> bit 2i, bit i of `odd` -> bit 2i+1 (32 bits each) */
__device__ static inline uint64_t filt_interleave32(uint32_t even, uint32_t odd)
{
	auto spread = [](uint64_t x) {
		x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
		x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
		x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
		x = (x | (x << 2)) & 0x3333333333333333ull;
		x = (x | (x << 1)) & 0x5555555555555555ull;
		return x;
	};
	return spread(even) | (spread(odd) << 1);
}

/* Same result as k_pred_bits, for streams without row-id indirection: a lane evaluates the two adjacent tuples
 * 2*lane, 2*lane+1 of a 128-tuple span with 16-byte loads; two ballots, bit-interleaved, give the span's two
 * bitmap words.  MODE 0: predicate program, MODE 1: vals != 0. */
template <int MODE>
__global__ __launch_bounds__(FILT_THREADS) void k_pred_bits_pair(pred_args p, const int64_t *__restrict__ vals, uint64_t n,
								 uint64_t *__restrict__ bits, uint32_t *__restrict__ block_counts, fp_fused fz)
{
	const uint32_t bid = filt_block_id(fz);
	__shared__ uint32_t s_cnt;
	if (threadIdx.x == 0)
		s_cnt = 0;
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6;
	uint32_t cnt = 0;
	const uint32_t tiles = fz.state ? FZ_TILES : 1u;	/* single pass: FZ_TILES consecutive row blocks per workgroup, one look-back */
	for (uint32_t sub = 0; sub < tiles; sub++) {
		const uint64_t word0 = ((uint64_t)bid * tiles + sub) * FILT_WORDS_PER_BLOCK + (uint64_t)wave * FILT_WORDS_PER_WAVE;
		constexpr int AHEAD = 4;	/* spans whose first operand is requested before any is evaluated */
		for (int r4 = 0; r4 < FILT_WORDS_PER_WAVE; r4 += 2 * AHEAD) {
			uint64_t x0[AHEAD][2];
			bool ok0[AHEAD][2];
#pragma unroll
			for (int u = 0; u < AHEAD; u++) {
				const uint64_t k0 = ((word0 + r4 + 2 * u) << 6) + 2ull * mdb_lane();
				x0[u][0] = x0[u][1] = 0;
				ok0[u][0] = ok0[u][1] = false;
				if (k0 < n) {
					if (MODE == 1) {
						if (k0 + 1 < n) {
							const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(vals + k0);
							x0[u][0] = q.x;
							x0[u][1] = q.y;
						} else {
							x0[u][0] = (uint64_t)vals[k0];
						}
					} else {
						pred_load_pair(p.cols[0], k0, k0 + 1 < n, x0[u], ok0[u]);
					}
				}
			}
#pragma unroll
			for (int u = 0; u < AHEAD; u++) {
				const uint64_t word = word0 + r4 + 2 * u;
				const uint64_t k0 = (word << 6) + 2ull * mdb_lane();
				bool pass[2] = { false, false };
				if (k0 < n) {
					if (MODE == 1) {
						pass[0] = x0[u][0] != 0;
						pass[1] = x0[u][1] != 0;
					} else {
						pred_eval_pair(p, k0, k0 + 1 < n, x0[u], ok0[u], pass);
					}
				}
				const uint64_t m0 = __ballot(pass[0]), m1 = __ballot(pass[1]);
				if ((word << 6) < n) {
					const uint64_t wa = filt_interleave32((uint32_t)m0, (uint32_t)m1);
					const uint64_t wb = filt_interleave32((uint32_t)(m0 >> 32), (uint32_t)(m1 >> 32));
					if (mdb_lane() == 0) {
						bits[word] = wa;
						if (((word + 1) << 6) < n)
							bits[word + 1] = wb;
					}
					cnt += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
				}
			}
		}
	}
	if (mdb_lane() == 0 && cnt)
		atomicAdd(&s_cnt, cnt);
	__syncthreads();
	if (fz.state) {
		filt_fused_tail(fz, bid, s_cnt, bits, n);
		return;
	}
	if (threadIdx.x == 0)
		block_counts[bid] = s_cnt;
}

/* The commonest predicate - ONE comparison of an INT64 base-table column with a constant (BASELINE config 1:
 * WHERE v > 500000) - without the program interpreter: the interpreter costs ~200 vector instructions per 128
 * tuples, which caps it near 3 TB/s, while a read-only stream reaches 6 TB/s on this chip
 * (profiles/micro/read_bw.hip).  Same bitmap / block-count output as k_pred_bits_pair. */
template <int CMP>
__global__ __launch_bounds__(FILT_THREADS) void k_pred_bits_cmp1(const int64_t *__restrict__ vals, const uint64_t *__restrict__ nullbits,
								 int64_t imm, uint64_t n, uint64_t *__restrict__ bits,
								 uint32_t *__restrict__ block_counts, fp_fused fz)
{
	const uint32_t bid = filt_block_id(fz);
	__shared__ uint32_t s_cnt;
	if (threadIdx.x == 0)
		s_cnt = 0;
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6;
	uint32_t cnt = 0;
	const uint32_t tiles = fz.state ? FZ_TILES : 1u;	/* single pass: FZ_TILES consecutive row blocks per workgroup, one look-back */
	for (uint32_t sub = 0; sub < tiles; sub++) {
		const uint64_t word0 = ((uint64_t)bid * tiles + sub) * FILT_WORDS_PER_BLOCK + (uint64_t)wave * FILT_WORDS_PER_WAVE;
		auto cmp = [&](int64_t a) -> bool {
			return CMP == MDB_CMP_LT ? a < imm : CMP == MDB_CMP_GT ? a > imm : CMP == MDB_CMP_NE ? a != imm : CMP == MDB_CMP_EQ ? a == imm
			       : CMP == MDB_CMP_LE ? a <= imm : a >= imm;
		};
		constexpr int SPANS = FILT_WORDS_PER_WAVE / 2;
		longlong2 q[SPANS];
#pragma unroll
		for (int u = 0; u < SPANS; u++) {	/* every load of the wave's 8 spans is in flight before the first compare */
			const uint64_t k0 = ((word0 + 2 * u) << 6) + 2ull * mdb_lane();
			q[u] = make_longlong2(0, 0);
			if (k0 + 1 < n)
				q[u] = *reinterpret_cast<const longlong2 *>(vals + k0);
			else if (k0 < n)
				q[u].x = vals[k0];
		}
#pragma unroll
		for (int u = 0; u < SPANS; u++) {
			const uint64_t word = word0 + 2 * u;
			const uint64_t k0 = (word << 6) + 2ull * mdb_lane();
			bool p0 = k0 < n && cmp(q[u].x), p1 = k0 + 1 < n && cmp(q[u].y);
			if (nullbits && k0 < n) {
				const uint64_t w = nullbits[k0 >> 6] >> (k0 & 63);
				p0 = p0 && !(w & 1ull);
				p1 = p1 && !(w & 2ull);
			}
			const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
			if ((word << 6) < n) {
				const uint64_t wa = filt_interleave32((uint32_t)m0, (uint32_t)m1);
				const uint64_t wb = filt_interleave32((uint32_t)(m0 >> 32), (uint32_t)(m1 >> 32));
				if (mdb_lane() == 0) {
					bits[word] = wa;
					if (((word + 1) << 6) < n)
						bits[word + 1] = wb;
				}
				cnt += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
			}
		}
	}
	if (mdb_lane() == 0 && cnt)
		atomicAdd(&s_cnt, cnt);
	__syncthreads();
	if (fz.state) {
		filt_fused_tail(fz, bid, s_cnt, bits, n);
		return;
	}
	if (threadIdx.x == 0)
		block_counts[bid] = s_cnt;
}

/* Single-pass scan + WHERE + projection for that commonest predicate: the workgroup's 4096 rows are loaded ONCE with
 * 16-byte loads and stay in registers while it learns - by look-back over the row blocks in front of it - where its
 * survivors go; every projected column is then written at its final position (the predicate's own column from the
 * registers, the others from 16-byte loads of the same rows).  Traffic = the algorithmic bytes: the column read once,
 * the survivors written once.  1024-thread workgroups (16384 rows each): the prefix travels along the row blocks at about
 * 64 blocks per L2 round trip, so with 4096-row blocks it - not HBM - bounded the kernel (0.50 ms for 10^8 rows). */
#define SP_THREADS 1024
#define SP_WAVES (SP_THREADS / MDB_WAVE)

template <int SPANS, typename Pred>	/* SPANS = 128-row spans per wave: 8 = 16384 rows per workgroup; cmp(value) = the predicate on a non-NULL value */
__device__ static inline void scan_project_body(const int64_t *__restrict__ vals, const uint64_t *__restrict__ nullbits, bool null_passes,
						uint64_t n, const fp_fused &fz, Pred cmp)
{
	const uint32_t bid = filt_block_id(fz);
	__shared__ uint32_t s_wcnt[SP_WAVES];
	__shared__ uint32_t s_base;
	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane();
	const uint64_t word0 = ((uint64_t)bid * SP_WAVES + wave) * (2 * SPANS);
	longlong2 q[SPANS];
	/* a workgroup whose rows all exist issues its loads together; behind the per-row range tests every load sits in a branch
	 * of its own and the compiler waits for it (s_waitcnt vmcnt(0)) before it issues the next: one load in flight per thread */
	if (((((uint64_t)bid + 1) * SP_WAVES * (2 * SPANS)) << 6) <= n) {	/* (uniform) */
#pragma unroll
		for (int u = 0; u < SPANS; u++)
			q[u] = *reinterpret_cast<const longlong2 *>(vals + (((word0 + 2 * u) << 6) + 2ull * lane));
	} else {
#pragma unroll
		for (int u = 0; u < SPANS; u++) {
			const uint64_t k0 = ((word0 + 2 * u) << 6) + 2ull * lane;
			q[u] = make_longlong2(0, 0);
			if (k0 + 1 < n)
				q[u] = *reinterpret_cast<const longlong2 *>(vals + k0);
			else if (k0 < n)
				q[u].x = vals[k0];
		}
	}
	/* the predicate is evaluated ONCE per row; a lane keeps its 2 x SPANS verdicts as bits of one register (sixteen 64-bit
	 * ballot masks kept beside the rows cost 150 registers - two thirds of the occupancy -, evaluating again for the count and
	 * for every output column is cheap for one comparison but not for a term list: 0.56 ms instead of 0.33) */
	uint32_t pbits = 0, cnt = 0;
#pragma unroll
	for (int u = 0; u < SPANS; u++) {
		const uint64_t k0 = ((word0 + 2 * u) << 6) + 2ull * lane;
		bool p0 = k0 < n && cmp(q[u].x), p1 = k0 + 1 < n && cmp(q[u].y);
		if (nullbits && k0 < n) {
			const uint64_t w = nullbits[k0 >> 6] >> (k0 & 63);
			if (null_passes) {	/* an OR list with "IS NULL" among its terms */
				p0 = p0 || (w & 1ull);
				p1 = k0 + 1 < n && (p1 || (w & 2ull));
			} else {
				p0 = p0 && !(w & 1ull);
				p1 = p1 && !(w & 2ull);
			}
		}
		pbits |= (p0 ? 1u : 0u) << (2 * u) | (p1 ? 2u : 0u) << (2 * u);
		cnt += (uint32_t)__popcll(__ballot(p0)) + (uint32_t)__popcll(__ballot(p1));
	}
	auto survivors = [&](int u, bool *p0, bool *p1) {
		*p0 = (pbits >> (2 * u)) & 1u;
		*p1 = (pbits >> (2 * u + 1)) & 1u;
	};
	if (lane == 0)
		s_wcnt[wave] = cnt;
	__syncthreads();
	if (wave == 0) {
		uint32_t total = 0;
#pragma unroll
		for (int w = 0; w < SP_WAVES; w++)
			total += s_wcnt[w];
		if (lane == 0)
			__hip_atomic_store(&fz.state[bid], (bid == 0 ? FZ_PFX : FZ_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		uint32_t excl = 0;
		if (bid > 0) {
			int64_t look = (int64_t)bid - 1;
			uint32_t spins = 0;
			for (;;) {
				const int64_t idx = look - (int64_t)lane;
				unsigned long long v = FZ_PFX;
				for (;;) {
					if (idx >= 0)
						v = __hip_atomic_load(&fz.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (!__ballot((v >> 62) == 0))
						break;
					if (++spins > (1u << 22)) {
						if (lane == 0)
							atomicOr(fz.status, FZ_TIMEOUT);
						v = FZ_PFX;
						break;
					}
					__builtin_amdgcn_s_sleep(2);
				}
				const uint64_t pm = __ballot((v >> 62) == 2);
				if (pm) {
					const uint32_t first = (uint32_t)__ffsll((long long)pm) - 1u;
					excl += filt_wave_sum_u32(lane <= first ? (uint32_t)(v & FZ_VAL) : 0u);
					break;
				}
				excl += filt_wave_sum_u32((uint32_t)(v & FZ_VAL));
				look -= MDB_WAVE;
			}
			if (lane == 0)
				__hip_atomic_store(&fz.state[bid], FZ_PFX | (unsigned long long)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (lane == 0)
			s_base = excl;
	}
	__syncthreads();
	uint32_t run = s_base;
	for (uint32_t w = 0; w < wave; w++)
		run += s_wcnt[w];
	const uint64_t lt = mdb_lanemask_lt();
	const fp_args &a = fz.cols;
	for (int c = 0; c < a.ncols; c++) {
		const bool same = a.src[c] == reinterpret_cast<const uint64_t *>(vals);	/* (uniform) the predicate's own column: still in registers */
		uint32_t at = run;
#pragma unroll
		for (int u = 0; u < SPANS; u++) {
			const uint64_t k0 = ((word0 + 2 * u) << 6) + 2ull * lane;
			bool p0, p1;
			survivors(u, &p0, &p1);
			const uint64_t m0u = __ballot(p0), m1u = __ballot(p1);
			if (m0u | m1u) {
				longlong2 x = q[u];
				if (!same) {
					x = make_longlong2(0, 0);
					if (k0 + 1 < n)
						x = *reinterpret_cast<const longlong2 *>(reinterpret_cast<const int64_t *>(a.src[c]) + k0);
					else if (k0 < n)
						x.x = (int64_t)a.src[c][k0];
				}
				const uint32_t pos0 = at + (uint32_t)__popcll(m0u & lt) + (uint32_t)__popcll(m1u & lt), pos1 = pos0 + (p0 ? 1u : 0u);
				if (p0)
					a.dst[c][pos0] = (uint64_t)x.x;
				if (p1)
					a.dst[c][pos1] = (uint64_t)x.y;
				if (a.dst_null[c]) {	/* (a projected column other than the predicate's may hold NULLs in surviving rows) */
					const uint64_t nw = a.src_null[c][k0 >> 6] >> (k0 & 63);
					if (p0 && (nw & 1ull))
						atomicOr(&a.dst_null[c][pos0 >> 6], 1ull << (pos0 & 63));
					if (p1 && (nw & 2ull))
						atomicOr(&a.dst_null[c][pos1 >> 6], 1ull << (pos1 & 63));
				}
			}
			at += (uint32_t)__popcll(m0u) + (uint32_t)__popcll(m1u);
		}
	}
	if (a.sel_out) {
		uint32_t at = run;
#pragma unroll
		for (int u = 0; u < SPANS; u++) {
			const uint64_t k0 = ((word0 + 2 * u) << 6) + 2ull * lane;
			bool p0, p1;
			survivors(u, &p0, &p1);
			const uint64_t m0u = __ballot(p0), m1u = __ballot(p1);
			const uint32_t pos0 = at + (uint32_t)__popcll(m0u & lt) + (uint32_t)__popcll(m1u & lt), pos1 = pos0 + (p0 ? 1u : 0u);
			if (p0)
				a.sel_out[pos0] = (uint32_t)k0;
			if (p1)
				a.sel_out[pos1] = (uint32_t)k0 + 1u;
			at += (uint32_t)__popcll(m0u) + (uint32_t)__popcll(m1u);
		}
	}
}

template <int CMP, int SPANS>
__global__ __launch_bounds__(SP_THREADS, 4) void k_scan_project_cmp1(const int64_t *__restrict__ vals, const uint64_t *__restrict__ nullbits,
								    int64_t imm, uint64_t n, fp_fused fz)
{
	scan_project_body<SPANS>(vals, nullbits, false, n, fz, [imm](int64_t a) -> bool {
		return CMP == MDB_CMP_LT ? a < imm : CMP == MDB_CMP_GT ? a > imm : CMP == MDB_CMP_NE ? a != imm : CMP == MDB_CMP_EQ ? a == imm
		       : CMP == MDB_CMP_LE ? a <= imm : a >= imm;
	});
}

/* The next commonest predicates - several comparisons of ONE INT64 base-table column with constants, all joined by AND
 * (ranges: fa >= 3 AND fa <= 900 AND fa <> 5) or all joined by OR (IN lists) - also without the interpreter: the column
 * is loaded once, the terms are evaluated in registers (uniform switch per term). */
#define FILT_MAX_TERMS 8
struct filt_terms {
	int32_t cmp[FILT_MAX_TERMS];
	int64_t imm[FILT_MAX_TERMS];
	int32_t n;
	int32_t null_passes;	/* OR list with "col IS NULL" among its terms: a NULL row passes (otherwise it fails every term) */
};

/* ... in ONE pass when the output is the compacted rows / their positions (scan_project_body) */
template <bool IS_OR, int SPANS>
__global__ __launch_bounds__(SP_THREADS, 4) void k_scan_project_terms(const int64_t *__restrict__ vals, const uint64_t *__restrict__ nullbits,
								     filt_terms t, uint64_t n, fp_fused fz)
{
	scan_project_body<SPANS>(vals, nullbits, t.null_passes != 0, n, fz, [&t](int64_t a) -> bool {
		bool r = !IS_OR;
		for (int k = 0; k < t.n; k++) {
			const int64_t imm = t.imm[k];
			const int c = t.cmp[k];
			const bool b = c == MDB_CMP_LT ? a < imm : c == MDB_CMP_GT ? a > imm : c == MDB_CMP_NE ? a != imm : c == MDB_CMP_EQ ? a == imm
				       : c == MDB_CMP_LE ? a <= imm : a >= imm;
			r = IS_OR ? (r || b) : (r && b);
		}
		return r;
	});
}

template <bool IS_OR>
__global__ __launch_bounds__(FILT_THREADS) void k_pred_bits_terms(const int64_t *__restrict__ vals, const uint64_t *__restrict__ nullbits,
								  filt_terms t, uint64_t n, uint64_t *__restrict__ bits,
								  uint32_t *__restrict__ block_counts, fp_fused fz)
{
	const uint32_t bid = filt_block_id(fz);
	__shared__ uint32_t s_cnt;
	if (threadIdx.x == 0)
		s_cnt = 0;
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6;
	uint32_t cnt = 0;
	const uint32_t tiles = fz.state ? FZ_TILES : 1u;	/* single pass: FZ_TILES consecutive row blocks per workgroup, one look-back */
	for (uint32_t sub = 0; sub < tiles; sub++) {
		const uint64_t word0 = ((uint64_t)bid * tiles + sub) * FILT_WORDS_PER_BLOCK + (uint64_t)wave * FILT_WORDS_PER_WAVE;
		auto eval = [&](int64_t a) -> bool {
			bool r = !IS_OR;
			for (int k = 0; k < t.n; k++) {
				const int64_t imm = t.imm[k];
				const int c = t.cmp[k];
				const bool b = c == MDB_CMP_LT ? a < imm : c == MDB_CMP_GT ? a > imm : c == MDB_CMP_NE ? a != imm : c == MDB_CMP_EQ ? a == imm
					       : c == MDB_CMP_LE ? a <= imm : a >= imm;
				r = IS_OR ? (r || b) : (r && b);
			}
			return r;
		};
		constexpr int SPANS = FILT_WORDS_PER_WAVE / 2;
		longlong2 q[SPANS];
#pragma unroll
		for (int u = 0; u < SPANS; u++) {
			const uint64_t k0 = ((word0 + 2 * u) << 6) + 2ull * mdb_lane();
			q[u] = make_longlong2(0, 0);
			if (k0 + 1 < n)
				q[u] = *reinterpret_cast<const longlong2 *>(vals + k0);
			else if (k0 < n)
				q[u].x = vals[k0];
		}
#pragma unroll
		for (int u = 0; u < SPANS; u++) {
			const uint64_t word = word0 + 2 * u;
			const uint64_t k0 = (word << 6) + 2ull * mdb_lane();
			bool p0 = k0 < n && eval(q[u].x), p1 = k0 + 1 < n && eval(q[u].y);
			if (nullbits && k0 < n) {
				const uint64_t w = nullbits[k0 >> 6] >> (k0 & 63);
				if (t.null_passes) {
					p0 = p0 || (w & 1ull);
					p1 = (k0 + 1 < n) && (p1 || (w & 2ull));
				} else {
					p0 = p0 && !(w & 1ull);
					p1 = p1 && !(w & 2ull);
				}
			}
			const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
			if ((word << 6) < n) {
				const uint64_t wa = filt_interleave32((uint32_t)m0, (uint32_t)m1);
				const uint64_t wb = filt_interleave32((uint32_t)(m0 >> 32), (uint32_t)(m1 >> 32));
				if (mdb_lane() == 0) {
					bits[word] = wa;
					if (((word + 1) << 6) < n)
						bits[word + 1] = wb;
				}
				cnt += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
			}
		}
	}
	if (mdb_lane() == 0 && cnt)
		atomicAdd(&s_cnt, cnt);
	__syncthreads();
	if (fz.state) {
		filt_fused_tail(fz, bid, s_cnt, bits, n);
		return;
	}
	if (threadIdx.x == 0)
		block_counts[bid] = s_cnt;
}

/* is the program "terms on ONE INT64 column, all ANDed or all ORed"?  (a NULL makes every term false - and with it both the
 * conjunction and the disjunction, which is what the interpreter computes: NULL <cmp> x is false) */
static bool filter_one_column_terms(const pred_args *p, filt_terms *t, int *slot, bool *is_or)
{
	int n_and = 0, n_or = 0, col = -1, n_notnull = 0, n_isnull = 0;
	t->n = 0;
	t->null_passes = 0;
	for (int i = 0; i < p->n_insns; i++) {
		const mdb_pred_insn &in = p->insn[i];
		if (in.op == MDB_P_AND) {
			n_and++;
		} else if (in.op == MDB_P_OR) {
			n_or++;
		} else if (in.op == MDB_P_ISNULL) {	/* IS NOT NULL in an AND list adds nothing (a NULL fails the comparisons), IS NULL in an OR list lets NULLs pass */
			if (col >= 0 && in.a != col)
				return false;
			col = in.a;
			if (in.cmp)
				n_notnull++;
			else
				n_isnull++;
		} else if ((in.op == MDB_P_CMP_COL_CONST || in.op == MDB_P_CMP_CONST_COL) && in.type == MDB_T_INT64) {
			if (col >= 0 && in.a != col)
				return false;
			col = in.a;
			if (t->n == FILT_MAX_TERMS)
				return false;
			int c = in.cmp;
			if (in.op == MDB_P_CMP_CONST_COL)	/* imm <cmp> col  ==  col <flipped cmp> imm */
				c = c == MDB_CMP_LT ? MDB_CMP_GT : c == MDB_CMP_GT ? MDB_CMP_LT : c == MDB_CMP_LE ? MDB_CMP_GE : c == MDB_CMP_GE ? MDB_CMP_LE : c;
			t->cmp[t->n] = c;
			t->imm[t->n] = in.imm;
			t->n++;
		} else {
			return false;
		}
	}
	if (t->n < 1 || (n_and && n_or) || n_and + n_or != t->n + n_notnull + n_isnull - 1 || n_and + n_or == 0)
		return false;
	if ((n_notnull && !n_and) || (n_isnull && !n_or) || n_isnull > 1)
		return false;	/* IS NOT NULL only among ANDed terms, IS NULL (once) only among ORed ones */
	t->null_passes = n_isnull ? 1 : 0;
	*slot = col;
	*is_or = n_or > 0;
	return true;
}

/* MODE 0: general predicate program; MODE 1: "vals[k] != 0" over an int64 array */
template <int MODE>
__global__ __launch_bounds__(FILT_THREADS) void k_pred_bits(pred_args p, const int64_t *__restrict__ vals, uint64_t n,
							    uint64_t *__restrict__ bits, uint32_t *__restrict__ block_counts, fp_fused fz)
{
	const uint32_t bid = filt_block_id(fz);
	__shared__ uint32_t s_cnt;
	if (threadIdx.x == 0)
		s_cnt = 0;
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6;
	uint32_t cnt = 0;
	const uint32_t tiles = fz.state ? FZ_TILES : 1u;	/* single pass: FZ_TILES consecutive row blocks per workgroup, one look-back */
	for (uint32_t sub = 0; sub < tiles; sub++) {
		const uint64_t word0 = ((uint64_t)bid * tiles + sub) * FILT_WORDS_PER_BLOCK + (uint64_t)wave * FILT_WORDS_PER_WAVE;
#pragma unroll 4
		for (int r = 0; r < FILT_WORDS_PER_WAVE; r++) {
			const uint64_t word = word0 + r;
			const uint64_t k = (word << 6) + mdb_lane();
			bool pass = false;
			if (k < n)
				pass = MODE == 1 ? (vals[k] != 0) : pred_eval(p, k);
			const uint64_t m = __ballot(pass);
			if ((word << 6) < n) {
				if (mdb_lane() == 0)
					bits[word] = m;
				cnt += (uint32_t)__popcll(m);
			}
		}
	}
	if (mdb_lane() == 0 && cnt)
		atomicAdd(&s_cnt, cnt);
	__syncthreads();
	if (fz.state) {
		filt_fused_tail(fz, bid, s_cnt, bits, n);
		return;
	}
	if (threadIdx.x == 0)
		block_counts[bid] = s_cnt;
}

__global__ __launch_bounds__(FILT_THREADS) void k_bits_to_sel(const uint64_t *__restrict__ bits, uint64_t n,
							      const uint32_t *__restrict__ block_off, uint32_t *__restrict__ out_sel)
{
	__shared__ uint32_t s_woff[FILT_WORDS_PER_BLOCK];
	const uint64_t nwords = (n + 63) >> 6;
	const uint64_t bword0 = (uint64_t)blockIdx.x * FILT_WORDS_PER_BLOCK;
	if (threadIdx.x < FILT_WORDS_PER_BLOCK) {	/* exactly wave 0 */
		const uint64_t w = bword0 + threadIdx.x;
		const uint32_t c = w < nwords ? (uint32_t)__popcll(bits[w]) : 0;
		s_woff[threadIdx.x] = mdb_wave_incl_scan(c) - c;
	}
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t base_off = block_off[blockIdx.x];
	const uint64_t lt = mdb_lanemask_lt();
#pragma unroll 4
	for (int r = 0; r < FILT_WORDS_PER_WAVE; r++) {
		const uint32_t wl = wave * FILT_WORDS_PER_WAVE + r;
		const uint64_t w = bword0 + wl;
		if (w >= nwords)
			break;
		const uint64_t m = bits[w];
		if ((m >> mdb_lane()) & 1ull)
			out_sel[base_off + s_woff[wl] + (uint32_t)__popcll(m & lt)] = (uint32_t)((w << 6) + mdb_lane());
	}
}

static inline uint32_t filt_blocks(uint64_t n) { return (uint32_t)((n + FILT_TUPLES_PER_BLOCK - 1) / FILT_TUPLES_PER_BLOCK); }

size_t mdb_filter_arena_bytes(uint64_t n)
{
	const uint64_t nb = filt_blocks(n) + 1;
	return mdb_align_up(((n + 63) / 64) * 8) + mdb_align_up(nb * 4) + mdb_align_up(mdb_scan_scratch_words(nb) * 4) + mdb_align_up(nb * 8 + 64) + 1024;
}

/* Shared tail: bitmap + block counts (arena) -> scan -> selection vector.  *d_total = device
 * address of the number of selected tuples.  No host sync. */
static int filter_run(mdb_dev_ctx *ctx, int mode, const pred_args *p, int n_cols, const int64_t *vals, uint64_t n, uint32_t *out_sel,
		      uint32_t **d_total, const fp_fused *fuse = nullptr, uint32_t *last_block = nullptr)
{
	fp_fused fz;
	memset(&fz, 0, sizeof(fz));
	if (fuse)
		fz = *fuse;	/* single pass: the predicate kernel also compacts the projected columns (mdb_dev_filter_project) */
	const uint32_t nb = fuse ? (filt_blocks(n) + FZ_TILES - 1) / FZ_TILES : filt_blocks(n);	/* workgroups */
	uint64_t *bits = (uint64_t *)mdb_arena_take(ctx, ((n + 63) / 64) * 8);
	uint32_t *bc = (uint32_t *)mdb_arena_take(ctx, ((size_t)nb + 1) * 4);
	uint32_t *scan_tmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words((uint64_t)nb + 1) * 4);
	if (!bits || !bc || !scan_tmp)
		return -MIDORIDB_INTERNAL;
	MDB_HIP(ctx, hipMemsetAsync(bc + nb, 0, 4, ctx->stream));
	if (mode == 1) {
		pred_args empty;
		memset(&empty, 0, sizeof(empty));
		if (((uintptr_t)vals & 15) == 0) {
			MDB_LAUNCH(ctx, "compact_nonzero_bits", k_pred_bits_pair<1>, nb, FILT_THREADS, empty, vals, n, bits, bc, fz);
		} else {
			MDB_LAUNCH(ctx, "compact_nonzero_bits", k_pred_bits<1>, nb, FILT_THREADS, empty, vals, n, bits, bc, fz);
		}
	} else {
		/* base-table streams (no row-id vector, 16-byte aligned columns) take the paired 16-byte-load form */
		bool direct = n_cols >= 1;	/* (a program of constants only has no column to stream) */
		for (int c = 0; c < n_cols; c++)
			direct = direct && !p->cols[c].rid && ((uintptr_t)p->cols[c].values & 15) == 0;
		const mdb_pred_insn &i0 = p->insn[0];
		int tslot = 0;
		bool tor = false;
		if (fuse && direct && p->n_insns == 1 && i0.op == MDB_P_CMP_COL_CONST && i0.type == MDB_T_INT64) {
			/* single pass, rows kept in registers across the look-back */
			const int64_t *v = (const int64_t *)p->cols[i0.a].values;
			const uint64_t *nbits = p->cols[i0.a].nullbits;
			/* 16384 rows per workgroup (8 spans per wave) measured 0.33 ms for 10^8 rows at 50 %, 8192 rows 0.36 ms
			 * (profiles/micro/scan_project_exp.py) */
			const uint64_t rows_per_block = (uint64_t)SP_WAVES * FILT_WORDS_PER_WAVE * 64;
			const uint32_t nb = (uint32_t)((n + rows_per_block - 1) / rows_per_block);
			if (last_block)
				*last_block = nb - 1;
			switch (i0.cmp) {
			case MDB_CMP_LT: MDB_LAUNCH(ctx, "scan_project", (k_scan_project_cmp1<MDB_CMP_LT, 8>), nb, SP_THREADS, v, nbits, i0.imm, n, fz); break;
			case MDB_CMP_GT: MDB_LAUNCH(ctx, "scan_project", (k_scan_project_cmp1<MDB_CMP_GT, 8>), nb, SP_THREADS, v, nbits, i0.imm, n, fz); break;
			case MDB_CMP_NE: MDB_LAUNCH(ctx, "scan_project", (k_scan_project_cmp1<MDB_CMP_NE, 8>), nb, SP_THREADS, v, nbits, i0.imm, n, fz); break;
			case MDB_CMP_EQ: MDB_LAUNCH(ctx, "scan_project", (k_scan_project_cmp1<MDB_CMP_EQ, 8>), nb, SP_THREADS, v, nbits, i0.imm, n, fz); break;
			case MDB_CMP_LE: MDB_LAUNCH(ctx, "scan_project", (k_scan_project_cmp1<MDB_CMP_LE, 8>), nb, SP_THREADS, v, nbits, i0.imm, n, fz); break;
			default: MDB_LAUNCH(ctx, "scan_project", (k_scan_project_cmp1<MDB_CMP_GE, 8>), nb, SP_THREADS, v, nbits, i0.imm, n, fz); break;
			}
		} else if (direct && p->n_insns == 1 && i0.op == MDB_P_CMP_COL_CONST && i0.type == MDB_T_INT64) {
			const int64_t *v = (const int64_t *)p->cols[i0.a].values;
			const uint64_t *nbits = p->cols[i0.a].nullbits;
			switch (i0.cmp) {
			case MDB_CMP_LT: MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_cmp1<MDB_CMP_LT>, nb, FILT_THREADS, v, nbits, i0.imm, n, bits, bc, fz); break;
			case MDB_CMP_GT: MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_cmp1<MDB_CMP_GT>, nb, FILT_THREADS, v, nbits, i0.imm, n, bits, bc, fz); break;
			case MDB_CMP_NE: MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_cmp1<MDB_CMP_NE>, nb, FILT_THREADS, v, nbits, i0.imm, n, bits, bc, fz); break;
			case MDB_CMP_EQ: MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_cmp1<MDB_CMP_EQ>, nb, FILT_THREADS, v, nbits, i0.imm, n, bits, bc, fz); break;
			case MDB_CMP_LE: MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_cmp1<MDB_CMP_LE>, nb, FILT_THREADS, v, nbits, i0.imm, n, bits, bc, fz); break;
			default: MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_cmp1<MDB_CMP_GE>, nb, FILT_THREADS, v, nbits, i0.imm, n, bits, bc, fz); break;
			}
		} else if (filt_terms ft; fuse && direct && filter_one_column_terms(p, &ft, &tslot, &tor)) {
			/* ranges / IN lists over one column: the same single pass */
			const int64_t *v = (const int64_t *)p->cols[tslot].values;
			const uint64_t *nbits = p->cols[tslot].nullbits;
			const uint64_t rows_per_block = (uint64_t)SP_WAVES * FILT_WORDS_PER_WAVE * 64;
			const uint32_t nb = (uint32_t)((n + rows_per_block - 1) / rows_per_block);
			if (last_block)
				*last_block = nb - 1;
			if (tor) {
				MDB_LAUNCH(ctx, "scan_project", (k_scan_project_terms<true, 8>), nb, SP_THREADS, v, nbits, ft, n, fz);
			} else {
				MDB_LAUNCH(ctx, "scan_project", (k_scan_project_terms<false, 8>), nb, SP_THREADS, v, nbits, ft, n, fz);
			}
		} else if (filt_terms ft; direct && filter_one_column_terms(p, &ft, &tslot, &tor)) {
			const int64_t *v = (const int64_t *)p->cols[tslot].values;
			const uint64_t *nbits = p->cols[tslot].nullbits;
			if (tor) {
				MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_terms<true>, nb, FILT_THREADS, v, nbits, ft, n, bits, bc, fz);
			} else {
				MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_terms<false>, nb, FILT_THREADS, v, nbits, ft, n, bits, bc, fz);
			}
		} else if (direct) {
			MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits_pair<0>, nb, FILT_THREADS, *p, (const int64_t *)NULL, n, bits, bc, fz);
		} else {
			MDB_LAUNCH(ctx, "filter_pred_bits", k_pred_bits<0>, nb, FILT_THREADS, *p, (const int64_t *)NULL, n, bits, bc, fz);
		}
	}
	if (fuse && last_block && *last_block == 0xFFFFFFFFu)
		*last_block = nb - 1;
	if (fuse)
		return MIDORIDB_OK;
	int rc = mdb_scan_u32_inplace(ctx, bc, (uint64_t)nb + 1, scan_tmp);
	if (rc)
		return rc;
	MDB_LAUNCH(ctx, "filter_bits_to_sel", k_bits_to_sel, nb, FILT_THREADS, bits, n, bc, out_sel);
	*d_total = bc + nb;
	return MIDORIDB_OK;
}

int mdb_filter_nonzero64(mdb_dev_ctx *ctx, const int64_t *vals, uint64_t n, uint32_t *out_sel, uint32_t **d_total)
{
	return filter_run(ctx, 1, NULL, 0, vals, n, out_sel, d_total);
}

static int filter_prepare(mdb_dev_ctx *ctx, const struct mdb_pred_insn *prog, int n_insns, const struct mdb_col_binding *cols, int n_cols,
			  uint64_t n, pred_args *p);

extern "C" int mdb_dev_filter(mdb_dev_ctx *ctx, const struct mdb_pred_insn *prog, int n_insns, const struct mdb_col_binding *cols,
			      int n_cols, uint64_t n, uint32_t *out_sel, uint64_t *out_count)
{
	*out_count = 0;
	if (n == 0)
		return MIDORIDB_OK;
	pred_args p;
	int rc = filter_prepare(ctx, prog, n_insns, cols, n_cols, n, &p);
	if (rc)
		return rc;
	rc = mdb_arena_begin(ctx, mdb_filter_arena_bytes(n));
	if (rc)
		return rc;
	/* one comparison of an INT64 base-table column with a constant (pushed-down WHERE conjuncts, DELETE / UPDATE, the
	 * threshold filter of ORDER BY ... LIMIT): ONE pass - scan_project_body with the positions as its only output - instead
	 * of bitmap, scan and bitmap -> positions (10^8 rows at 50 %: 0.41 -> 0.30 ms) */
	/* (term lists - ranges, IN lists - take the three passes here: their evaluation, not the passes, is what costs - 0.43 ms in
	 * one pass against 0.41 in three at 10^8 rows; mdb_dev_filter_project does use the one-pass term kernel: it saves the gather) */
	const bool one_pass = n >= (1u << 18) && n_cols >= 1 && p.n_insns == 1 && p.insn[0].op == MDB_P_CMP_COL_CONST && p.insn[0].type == MDB_T_INT64 &&
			      !p.cols[p.insn[0].a].rid && ((uintptr_t)p.cols[p.insn[0].a].values & 15) == 0;
	if (one_pass) {
		const uint64_t rows_per_block = (uint64_t)SP_WAVES * FILT_WORDS_PER_WAVE * 64;
		const uint32_t nblk = (uint32_t)((n + rows_per_block - 1) / rows_per_block);
		fp_fused fz;
		memset(&fz, 0, sizeof(fz));
		fz.cols.sel_out = out_sel;
		fz.state = (unsigned long long *)mdb_arena_take(ctx, (size_t)nblk * 8 + 64);
		if (!fz.state)
			return -MIDORIDB_INTERNAL;
		fz.ticket = (uint32_t *)(fz.state + nblk);
		fz.status = ctx->d_status;
		MDB_HIP(ctx, hipMemsetAsync(fz.state, 0, (size_t)nblk * 8 + 64, ctx->stream));
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
		uint32_t *unused = NULL, last = 0xFFFFFFFFu;
		rc = filter_run(ctx, 0, &p, n_cols, NULL, n, NULL, &unused, &fz, &last);
		if (rc)
			return rc;
		uint64_t *h64 = ctx->h_pinned;
		MDB_HIP(ctx, hipMemcpyAsync(&h64[0], fz.state + last, 8, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipMemcpyAsync(&h64[1], ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (((uint32_t)h64[1] & FZ_TIMEOUT) || (h64[0] >> 62) != 2)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "filter: the look-back over the row blocks did not complete");
		*out_count = h64[0] & FZ_VAL;
		return MIDORIDB_OK;
	}
	uint32_t *d_total = NULL;
	rc = filter_run(ctx, 0, &p, n_cols, NULL, n, out_sel, &d_total);
	if (rc)
		return rc;
	uint32_t *h = (uint32_t *)ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h, d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_count = h[0];
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_filter_project(mdb_dev_ctx *ctx, const struct mdb_pred_insn *prog, int n_insns, const struct mdb_col_binding *cols,
				      int n_cols, uint64_t n, const struct mdb_project_col *proj, int n_proj, uint64_t *out_count)
{
	if (!out_count || n_proj < 0 || n_proj > FP_MAX_COLS || (n_proj && !proj))
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter_project: between 0 and %d projected columns", FP_MAX_COLS);
	*out_count = 0;
	for (int c = 0; c < n_proj; c++) {
		if (!proj[c].values || !proj[c].out_values || (proj[c].nullbits && !proj[c].out_nullbits))
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter_project: column %d lacks values or an output slot", c);
		*proj[c].out_values = NULL;
		if (proj[c].out_nullbits)
			*proj[c].out_nullbits = NULL;
	}
	if (n == 0)
		return MIDORIDB_OK;
	pred_args p;
	int rc = filter_prepare(ctx, prog, n_insns, cols, n_cols, n, &p);
	if (rc)
		return rc;
	rc = mdb_arena_begin(ctx, mdb_filter_arena_bytes(n));
	if (rc)
		return rc;
	/* The outputs are sized for every row (the number of survivors is known only when the single pass is over); the
	 * buffers come from the context's recycling allocator, the caller reads *out_count rows of them. */
	const uint32_t nb = (filt_blocks(n) + FZ_TILES - 1) / FZ_TILES;	/* workgroups = look-back words */
	fp_fused fz;
	memset(&fz, 0, sizeof(fz));
	fz.cols.ncols = n_proj;
	for (int c = 0; c < n_proj && !rc; c++) {
		rc = mdb_dev_alloc(ctx, n * 8, proj[c].out_values);
		if (!rc && proj[c].nullbits) {
			const size_t bytes = (size_t)((n + 63) / 64) * 8;
			rc = mdb_dev_alloc(ctx, bytes, (void **)proj[c].out_nullbits);
			if (!rc)
				rc = mdb_dev_memset(ctx, *proj[c].out_nullbits, 0, bytes);	/* rows of one word may come from two row blocks: the bits are OR-ed in */
		}
		fz.cols.src[c] = (const uint64_t *)proj[c].values;
		fz.cols.src_null[c] = proj[c].nullbits;
		fz.cols.dst[c] = rc ? NULL : (uint64_t *)*proj[c].out_values;
		fz.cols.dst_null[c] = (rc || !proj[c].nullbits) ? NULL : (unsigned long long *)*proj[c].out_nullbits;
	}
	uint64_t *h = ctx->h_pinned;
	if (!rc) {
		fz.state = (unsigned long long *)mdb_arena_take(ctx, (size_t)nb * 8 + 64);
		if (!fz.state)
			rc = -MIDORIDB_INTERNAL;
	}
	if (!rc) {
		fz.ticket = (uint32_t *)(fz.state + nb);
		fz.status = ctx->d_status;
		uint32_t *d_total = NULL;
		if (hipMemsetAsync(fz.state, 0, (size_t)nb * 8 + 64, ctx->stream) != hipSuccess ||
		    hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream) != hipSuccess)
			rc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "filter_project: clearing the look-back words failed");
		uint32_t last = 0xFFFFFFFFu;	/* the look-back word that holds the grand total: depends on the kernel's row-block size */
		if (!rc)
			rc = filter_run(ctx, 0, &p, n_cols, NULL, n, NULL, &d_total, &fz, &last);
		if (!rc && (hipMemcpyAsync(&h[0], fz.state + last, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
			    hipMemcpyAsync(&h[1], ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
			    hipStreamSynchronize(ctx->stream) != hipSuccess))
			rc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "filter_project: %s", hipGetErrorString(hipGetLastError()));
		if (!rc && (((uint32_t)h[1] & FZ_TIMEOUT) || (h[0] >> 62) != 2))
			rc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "filter_project: the look-back over the row blocks did not complete");
		if (!rc)
			*out_count = h[0] & FZ_VAL;
	}
	if (rc) {
		for (int c = 0; c < n_proj; c++) {
			if (*proj[c].out_values)
				mdb_dev_free(ctx, *proj[c].out_values);
			*proj[c].out_values = NULL;
			if (proj[c].out_nullbits && *proj[c].out_nullbits) {
				mdb_dev_free(ctx, *proj[c].out_nullbits);
				*proj[c].out_nullbits = NULL;
			}
		}
	}
	return rc;
}

static int filter_prepare(mdb_dev_ctx *ctx, const struct mdb_pred_insn *prog, int n_insns, const struct mdb_col_binding *cols, int n_cols,
			  uint64_t n, pred_args *pp)
{
	if (n >= 0xFFFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: too many tuples");
	if (n_insns <= 0 || n_insns > MDB_PRED_MAX_INSNS || n_cols < 0 || n_cols > MDB_PRED_MAX_SLOTS)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: predicate program too large (%d insns, %d columns)", n_insns, n_cols);
	/* validate the program: stack discipline and slot numbers */
	int depth = 0;
	for (int i = 0; i < n_insns; i++) {
		const mdb_pred_insn &in = prog[i];
		switch (in.op) {
		case MDB_P_CMP_COL_COL:
			if (in.b < 0 || in.b >= n_cols)
				return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: bad column slot");
			/* fallthrough */
		case MDB_P_CMP_COL_CONST:
		case MDB_P_CMP_CONST_COL:
		case MDB_P_ISNULL:
		case MDB_P_IN_BITS:
			if (in.op == MDB_P_IN_BITS && (!in.imm || in.b < 0))
				return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: a bit-table test without a table");
			if (in.a < 0 || in.a >= n_cols)
				return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: bad column slot");
			depth++;
			break;
		case MDB_P_CONST:
			depth++;
			break;
		case MDB_P_AND:
		case MDB_P_OR:
		case MDB_P_XOR:
			if (depth < 2)
				return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: malformed predicate program");
			depth--;
			break;
		default:
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: unknown predicate opcode %d", in.op);
		}
		if (depth > 60)
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: predicate nesting too deep");
	}
	if (depth != 1)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "filter: malformed predicate program");

	pred_args &p = *pp;
	memset(&p, 0, sizeof(p));
	memcpy(p.insn, prog, sizeof(mdb_pred_insn) * (size_t)n_insns);
	if (n_cols)
		memcpy(p.cols, cols, sizeof(mdb_col_binding) * (size_t)n_cols);
	p.n_insns = n_insns;
	return MIDORIDB_OK;
}
