/*
 * mdb_dev_dense.hip - the groups of a GROUP BY whose keys are nearly unique, without sorting a record per group (round 5).
 *
 * The reference's proc_groupby_clause keeps the FIRST row of every key and counts the others into it, groups in first-row order
 * (/root/reference/src/engine/executor_select.c:1526-1588).  When nearly every row is a group of its own, (first row, COUNT) per group in
 * first-row order is nearly the identity: what says it all is ONE BIT per row - "this row is the first of its key" - and a short list of
 * exceptions (first row, COUNT) for the keys that have more than one row.  The leaf kernel that meets a key's rows clears the bit of every
 * row that turns out not to be the first (one global atomic per DUPLICATE row, none per group) and appends the exceptions; here the bits
 * are expanded: out_first = the set bits' positions, out_count = 1, then the exceptions' COUNTs dropped at their ranks.  Sequential
 * reads and writes of 12 bytes per group instead of two scatter levels and a leaf over 10^8 records (0.9 ms of a 2.0 ms GROUP BY).
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"

#define DN_THREADS 128
#define DN_BLOCK_ROWS (DN_THREADS * 64u)	/* rows per workgroup: one 64-bit word of bits per thread */

__device__ static inline unsigned long long dn_word(const unsigned long long *bits, uint64_t n, uint64_t idx)
{
	const uint64_t row0 = idx * 64u;
	if (row0 >= n)
		return 0ull;
	unsigned long long w = bits[idx];
	if (n - row0 < 64u)
		w &= (1ull << (n - row0)) - 1ull;	/* (the bits behind the table's last row were never cleared) */
	return w;
}

__global__ __launch_bounds__(DN_THREADS) void k_dense_count(const unsigned long long *__restrict__ bits, uint64_t n, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t s_tmp[32];
	const uint32_t c = (uint32_t)__popcll(dn_word(bits, n, (uint64_t)blockIdx.x * DN_THREADS + threadIdx.x));
	uint32_t total;
	(void)mdb_block_excl_scan(c, s_tmp, &total);
	if (threadIdx.x == 0)
		cnt[blockIdx.x] = total;
}

/* base[b] = groups before block b.  wordbase[w] = groups before the 64 rows of word w (the exceptions find their ranks with it). */
__global__ __launch_bounds__(DN_THREADS) void k_dense_expand(const unsigned long long *__restrict__ bits, uint64_t n, const uint32_t *__restrict__ base,
							      uint32_t *__restrict__ wordbase, uint32_t *__restrict__ out_first, int64_t *__restrict__ out_count,
							      const int64_t *__restrict__ keys, uint32_t keys32, int64_t *__restrict__ out_key)
{
	__shared__ uint32_t s_tmp[32];
	__shared__ unsigned long long s_w[DN_THREADS];
	__shared__ uint32_t s_b[DN_THREADS];
	const uint64_t idx = (uint64_t)blockIdx.x * DN_THREADS + threadIdx.x;
	const unsigned long long w = dn_word(bits, n, idx);
	uint32_t total;
	const uint32_t wb = base[blockIdx.x] + mdb_block_excl_scan((uint32_t)__popcll(w), s_tmp, &total);
	if (idx * 64u < n)
		wordbase[idx] = wb;
	s_w[threadIdx.x] = w;
	s_b[threadIdx.x] = wb;
	__syncthreads();
	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane();
	const uint64_t below = mdb_lanemask_lt();
	for (uint32_t k = 0; k < 64u; k++) {
		const unsigned long long ww = s_w[wave * 64u + k];	/* (the same word for the whole wave: one broadcast read) */
		if ((ww >> lane) & 1ull) {
			const uint32_t pos = s_b[wave * 64u + k] + (uint32_t)__popcll(ww & below);
			const uint32_t row = (uint32_t)(((uint64_t)blockIdx.x * DN_THREADS + wave * 64u + k) * 64u + lane);
			if (out_first)
				out_first[pos] = row;
			if (out_count)
				out_count[pos] = 1;
			if (out_key)		/* (the join's group key: the left table's key of the group's first row - consecutive lanes, consecutive rows) */
				out_key[pos] = keys32 ? (int64_t)reinterpret_cast<const int32_t *>(keys)[row] : keys[row];
		}
	}
}

/* exception e = first row << 32 | COUNT: the group's place is the number of first rows before its own */
__global__ __launch_bounds__(256) void k_dense_patch(const unsigned long long *__restrict__ exc, uint32_t n_exc, const unsigned long long *__restrict__ bits,
						      const uint32_t *__restrict__ wordbase, int64_t *__restrict__ out_count)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_exc; i += gridDim.x * 256u) {
		const unsigned long long e = exc[i];
		const uint32_t first = (uint32_t)(e >> 32), w = first >> 6, b = first & 63u;
		const uint32_t pos = wordbase[w] + (uint32_t)__popcll(bits[w] & ((1ull << b) - 1ull));
		out_count[pos] = (int64_t)(uint32_t)e;
	}
}

size_t mdb_dense_arena_bytes(uint64_t n)
{
	const uint64_t nwords = (n + 63) / 64, nblocks = (nwords + DN_THREADS - 1) / DN_THREADS;
	return mdb_align_up(nwords * 8 + 64) + mdb_align_up(nwords * 4 + 64) + 2 * mdb_align_up((nblocks + 2) * 4) + mdb_align_up(mdb_scan_scratch_words(nblocks + 1) * 4) + 4096;
}

/* the bitmap the leaf kernel clears bits in: every row a first row to begin with */
int mdb_dense_bits_begin(mdb_dev_ctx *ctx, uint64_t n, unsigned long long **bits)
{
	const uint64_t nwords = (n + 63) / 64;
	*bits = (unsigned long long *)mdb_arena_take(ctx, nwords * 8 + 64);
	if (!*bits)
		return -MIDORIDB_INTERNAL;
	MDB_HIP(ctx, hipMemsetAsync(*bits, 0xFF, nwords * 8, ctx->stream));
	return MIDORIDB_OK;
}

/* out_first[g] / out_count[g] of the `groups` set bits among the first n, COUNT 1 but for the n_exc exceptions.  No host sync; the set
 * bits are the caller's count of groups (the leaf kernel counted them): checked by the caller against its output capacity before. */
int mdb_dense_emit(mdb_dev_ctx *ctx, const unsigned long long *bits, uint64_t n, const unsigned long long *exc, uint32_t n_exc, uint32_t *out_first,
		   int64_t *out_count, const int64_t *keys, bool keys32, int64_t *out_key)
{
	const uint64_t nwords = (n + 63) / 64, nblocks = (nwords + DN_THREADS - 1) / DN_THREADS;
	if (nblocks >= 0x7FFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "dense group emit: too many rows");
	if (!out_first && !out_count && !out_key)
		return MIDORIDB_OK;	/* (nothing to write: the caller takes the left key column for the group keys and every COUNT is 1) */
	uint32_t *wordbase = (uint32_t *)mdb_arena_take(ctx, nwords * 4 + 64);
	uint32_t *cnt = (uint32_t *)mdb_arena_take(ctx, (nblocks + 2) * 4);
	uint32_t *base = (uint32_t *)mdb_arena_take(ctx, (nblocks + 2) * 4);
	uint32_t *tmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words(nblocks + 1) * 4);
	if (!wordbase || !cnt || !base || !tmp)
		return -MIDORIDB_INTERNAL;
	MDB_LAUNCH(ctx, "dense_count", k_dense_count, (uint32_t)nblocks, DN_THREADS, bits, n, cnt);
	int rc;
	if (nblocks <= MDB_SCAN_FROM_MAX) {
		rc = mdb_scan_u32_small_from(ctx, cnt, (uint32_t)nblocks, base);
	} else {
		MDB_HIP(ctx, hipMemcpyAsync(base, cnt, nblocks * 4, hipMemcpyDeviceToDevice, ctx->stream));
		MDB_HIP(ctx, hipMemsetAsync(base + nblocks, 0, 4, ctx->stream));
		rc = mdb_scan_u32_inplace(ctx, base, nblocks + 1, tmp);
	}
	if (rc)
		return rc;
	MDB_LAUNCH(ctx, "dense_expand", k_dense_expand, (uint32_t)nblocks, DN_THREADS, bits, n, base, wordbase, out_first, out_count, keys, keys32 ? 1u : 0u,
		   out_key);
	if (n_exc && out_count) {
		const uint32_t grid = (n_exc + 255u) / 256u;
		MDB_LAUNCH(ctx, "dense_patch", k_dense_patch, grid < 4096u ? grid : 4096u, 256, exc, n_exc, bits, wordbase, out_count);
	}
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ GROUP BY over a column the catalog knows to hold no value twice
 * (mdb_dev_group_count): group i = row i, COUNT 1 */
__global__ __launch_bounds__(256) void k_group_identity(uint64_t n, uint32_t *__restrict__ out_first, long long *__restrict__ out_count)
{
	const uint64_t i4 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
	if (i4 + 3u < n && !((uintptr_t)out_first & 15u) && !((uintptr_t)out_count & 15u)) {
		*reinterpret_cast<uint4 *>(out_first + i4) = make_uint4((uint32_t)i4, (uint32_t)i4 + 1u, (uint32_t)i4 + 2u, (uint32_t)i4 + 3u);
		*reinterpret_cast<longlong2 *>(out_count + i4) = make_longlong2(1ll, 1ll);
		*reinterpret_cast<longlong2 *>(out_count + i4 + 2u) = make_longlong2(1ll, 1ll);
	} else {
		for (uint64_t i = i4; i < n && i < i4 + 4u; i++) {
			out_first[i] = (uint32_t)i;
			out_count[i] = 1ll;
		}
	}
}

int mdb_group_identity(mdb_dev_ctx *ctx, uint64_t n, uint32_t *out_first, int64_t *out_count)
{
	if (!n)
		return MIDORIDB_OK;
	const uint64_t blocks = (n + 1023u) / 1024u;
	MDB_LAUNCH(ctx, "group_identity", k_group_identity, (uint32_t)blocks, 256, n, out_first, reinterpret_cast<long long *>(out_count));
	return MIDORIDB_OK;
}

