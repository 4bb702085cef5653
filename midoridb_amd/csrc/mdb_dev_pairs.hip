/*
 * mdb_dev_pairs.hip - materialising INNER JOIN (_join_nested_loop_tbl2tbl, executor_select.c:1076-1149): pairs in the reference's left-major / right-minor order
 * (split off mdb_dev_join.hip; what the files share: mdb_dev_join_internal.h).  Hand-written HIP for gfx950, HBM-bound
 * integer work: no MFMA.
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"

/* ------------------------------------------------------------------ materialising join: count phase */

/* ------------------------------------------------------------------ materialising join, unique right keys
 *
 * The common shape (primary key on the right: BASELINE configs 2 and 5): every left row matches at most one right
 * row, so a pair is fully described by ONE 64-bit record (left row id, right row id + 1) - exactly the shape of the
 * group records above.  One persistent kernel builds the per-leaf table (key -> right row id), probes it with the
 * left rows and appends the records; the ordering sort + k_order_leaf then deliver (l, r) in left-row order.  No
 * match-count array, no scan, no second table build.  A duplicate right key (or any overflow) is flagged and the
 * caller falls back to the general count / scan / emit path.
 */
struct pu_args {
	const uint64_t *hv_l;
	const uint32_t *rid_l;
	const uint32_t *off_l;
	const uint32_t *cnt_l;
	uint32_t cap_l;
	const uint64_t *hv_r;
	const uint32_t *rid_r;
	const uint32_t *off_r;
	const uint32_t *cnt_r;
	uint32_t cap_r;
	unsigned long long *rec;
	uint32_t *rec_count;		/* list slots handed out (chunked, zero-filled gaps) */
	uint32_t *rec_valid;		/* pairs */
	uint32_t rec_cap;
	uint32_t kbits;
	uint32_t *status;		/* bit 0 table overflow, bit 3 list exhausted, bit 5 duplicate right key */
	uint32_t nleaves;
};

/* NARROW: both sides travel as hash32 << 32 | row id words (no row-id arrays; see the narrow form of the group count) */
template <bool NARROW>
__global__ __launch_bounds__(GC_THREADS, 8) void k_leaf_pairs_unique(pu_args a)
{
	__shared__ unsigned long long s_key[GC_SLOTS];
	__shared__ uint32_t s_val[GC_SLOTS + 1];	/* right row id + 1; [GC_SLOTS] = the key whose hash is 0 (0 = absent) */
	__shared__ uint32_t s_chunk[4];			/* [0] base [1] used [2] size [3] pairs */
	__shared__ uint32_t s_abort;			/* a duplicate right key (or a full table) was met: this is not a unique-key join */

	for (uint32_t s = threadIdx.x; s <= GC_SLOTS; s += GC_THREADS) {
		if (s < GC_SLOTS)
			s_key[s] = 0ull;
		s_val[s] = 0;
	}
	if (threadIdx.x < 4)
		s_chunk[threadIdx.x] = 0;
	if (threadIdx.x == 0)
		s_abort = 0;
	uint32_t npairs = 0;
	__syncthreads();
	for (uint32_t leaf = blockIdx.x; leaf < a.nleaves; leaf += gridDim.x) {
		/* The verdict "not unique" is raised ONCE per workgroup and ends its work (one global atomic per duplicate row -
		 * 10^8 of them on one address for a right table with 16 rows per key - made this failed attempt cost 18 ms).
		 * Workgroups that meet no duplicate themselves run to the end: looking at the global flag once per leaf put an
		 * uncached round trip on every leaf's critical path (0.24 -> 0.45 ms per 10^7 x 10^7 join). */
		if (s_abort) {		/* uniform: read after the barriers that ended the previous leaf */
			if (threadIdx.x == 0)
				atomicOr(a.status, s_abort);
			break;
		}
		uint32_t l0, l1, r0, r1;
		gc_leaf_range(a.off_l, a.cnt_l, a.cap_l, leaf, &l0, &l1);
		gc_leaf_range(a.off_r, a.cnt_r, a.cap_r, leaf, &r0, &r1);
		if (l0 == l1 || r0 == r1)
			continue;	/* uniform */
		/* request the first batch of both sides before anything else */
		uint64_t hr[LEAF_BATCH], hl[LEAF_BATCH];
		uint32_t rr[LEAF_BATCH], rl[LEAF_BATCH];
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint32_t j = r0 + (uint32_t)u * GC_THREADS + threadIdx.x;
			const uint32_t i = l0 + (uint32_t)u * GC_THREADS + threadIdx.x;
			hr[u] = j < r1 ? a.hv_r[j] : 0;
			hl[u] = i < l1 ? a.hv_l[i] : 0;
			rr[u] = rl[u] = 0;
			if (!NARROW) {
				rr[u] = j < r1 ? a.rid_r[j] : 0;
				rl[u] = i < l1 ? a.rid_l[i] : 0;
			}
		}
		{	/* list space for at most one record per left row (chunked reservation as in k_leaf_group_count) */
			const uint32_t need = (l1 - l0) + 1;
			const uint32_t base = s_chunk[0], used = s_chunk[1], size = s_chunk[2];
			if (used + need > size) {
				for (uint32_t i = used + threadIdx.x; i < size; i += GC_THREADS)
					a.rec[base + i] = 0ull;
				__syncthreads();
				if (threadIdx.x == 0) {
					const uint32_t want = need > GC_REC_CHUNK ? need : GC_REC_CHUNK;
					const uint32_t nb = atomicAdd(a.rec_count, want);
					if (nb + want > a.rec_cap) {
						mdb_raise(a.status, 8u);
						s_chunk[0] = 0;
						s_chunk[2] = 0;
					} else {
						s_chunk[0] = nb;
						s_chunk[2] = want;
					}
					s_chunk[1] = 0;
				}
			}
		}
		/* build: right rows (unique keys expected) */
		uint32_t own[LEAF_BATCH];
		const bool by_owner = (r1 - r0) <= GC_THREADS * LEAF_BATCH;
		for (uint32_t base = r0; base < r1; base += GC_THREADS * LEAF_BATCH) {
#pragma unroll
			for (int u = 0; u < LEAF_BATCH; u++) {
				const uint32_t j = base + (uint32_t)u * GC_THREADS + threadIdx.x;
				if (base != r0) {
					hr[u] = j < r1 ? a.hv_r[j] : 0;
					if (!NARROW)
						rr[u] = j < r1 ? a.rid_r[j] : 0;
				}
				if (base == r0)
					own[u] = 0xFFFFFFFFu;
				if (j >= r1)
					continue;
				/* the words are decoded here, where they are used (not at load time: the first batch is in flight) */
				const uint64_t key_r = NARROW ? gc_narrow_hv(hr[u]) : hr[u];
				const uint32_t rid_r = NARROW ? (uint32_t)hr[u] : rr[u];
				if (key_r == 0) {
					if (atomicExch(&s_val[GC_SLOTS], rid_r + 1u) != 0)
						s_abort = 32u;
					continue;
				}
				bool created = false;
				const uint32_t s = leaf_insert(s_key, GC_SLOTS, key_r, &created);
				if (s == 0xFFFFFFFFu) {
					s_abort = 1u;
				} else if (!created) {
					s_abort = 32u;		/* the key is already there: not a unique-key join */
				} else {
					s_val[s] = rid_r + 1u;
					if (base == r0)
						own[u] = s;
				}
			}
		}
		__syncthreads();
		/* probe: left rows -> records */
		const uint32_t cbase = s_chunk[0], csize = s_chunk[2];
		for (uint32_t base = l0; base < l1; base += GC_THREADS * LEAF_BATCH) {
#pragma unroll
			for (int u = 0; u < LEAF_BATCH; u++) {
				const uint32_t i = base + (uint32_t)u * GC_THREADS + threadIdx.x;
				if (base != l0) {
					hl[u] = i < l1 ? a.hv_l[i] : 0;
					if (!NARROW)
						rl[u] = i < l1 ? a.rid_l[i] : 0;
				}
				unsigned long long recv = 0;
				if (i < l1) {
					const uint64_t key_l = NARROW ? gc_narrow_hv(hl[u]) : hl[u];
					const uint32_t rid_l = NARROW ? (uint32_t)hl[u] : rl[u];
					uint32_t s = GC_SLOTS;
					if (key_l != 0)
						s = leaf_find(s_key, GC_SLOTS, key_l);
					const uint32_t v = s != 0xFFFFFFFFu ? s_val[s] : 0u;
					if (v)
						recv = ((unsigned long long)rid_l << (64 - a.kbits)) | v;
				}
				const uint64_t m = __ballot(recv != 0ull);
				if (m) {
					const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
					uint32_t wbase = 0;
					if (mdb_lane() == leader)
						wbase = atomicAdd(&s_chunk[1], (uint32_t)__popcll(m));
					wbase = __shfl(wbase, (int)leader, MDB_WAVE);
					if (recv) {
						const uint32_t pos = wbase + (uint32_t)__popcll(m & mdb_lanemask_lt());
						if (pos < csize)
							a.rec[cbase + pos] = recv;
						npairs++;
					}
				}
			}
		}
		__syncthreads();
		/* clear what this leaf wrote */
		if (by_owner) {
#pragma unroll
			for (int u = 0; u < LEAF_BATCH; u++)
				if (own[u] != 0xFFFFFFFFu) {
					s_key[own[u]] = 0ull;
					s_val[own[u]] = 0;
				}
			if (threadIdx.x == 0)
				s_val[GC_SLOTS] = 0;
		} else {
			for (uint32_t s = threadIdx.x; s <= GC_SLOTS; s += GC_THREADS) {
				if (s < GC_SLOTS)
					s_key[s] = 0ull;
				s_val[s] = 0;
			}
		}
		__syncthreads();
	}
	{
		const uint32_t base = s_chunk[0], used = s_chunk[1], size = s_chunk[2];
		for (uint32_t i = used + threadIdx.x; i < size; i += GC_THREADS)
			a.rec[base + i] = 0ull;
		if (npairs)
			atomicAdd(&s_chunk[3], npairs);
	}
	__syncthreads();
	if (threadIdx.x == 0 && s_abort)
		atomicOr(a.status, s_abort);	/* (also when it was raised by the workgroup's last leaf) */
	if (threadIdx.x == 0 && s_chunk[3])
		atomicAdd(a.rec_valid, s_chunk[3]);
}

struct pj_args {
	const uint64_t *hv_l;
	const uint32_t *rid_l;
	const uint32_t *off_l;		/* exact leaf offsets, or (cnt, cap) of the fixed-capacity layout, as in gc_args */
	const uint32_t *cnt_l;
	uint32_t cap_l;
	const uint64_t *hv_r;
	const uint32_t *rid_r;
	const uint32_t *off_r;
	const uint32_t *cnt_r;
	uint32_t cap_r;
	uint32_t *match;	/* [n_l + 1]: count phase writes matches per left row; scanned into offsets */
	uint32_t n_l;		/* a row id read from a region is checked against it before it indexes match[]: a fixed-capacity region that overflowed
				 * (status bit 1: the join is redone with exact regions) holds slots nobody wrote */
	uint32_t *out_l;
	uint32_t *out_r;
	uint32_t *status;
	unsigned long long *total64;	/* 64-bit sum of all match counts (guards the 32-bit offsets) */
	uint32_t nleaves;
};

__global__ __launch_bounds__(LEAF_THREADS) void k_leaf_pairs_count(pj_args a)
{
	__shared__ unsigned long long s_key[PJ_SLOTS];
	__shared__ uint32_t s_cnt[PJ_SLOTS + 1];	/* [PJ_SLOTS] = the key with hash 0 */
	__shared__ unsigned long long s_total;

	const uint32_t leaf = blockIdx.x;
	uint32_t l0, l1, r0, r1;
	gc_leaf_range(a.off_l, a.cnt_l, a.cap_l, leaf, &l0, &l1);
	gc_leaf_range(a.off_r, a.cnt_r, a.cap_r, leaf, &r0, &r1);
	if (l0 == l1 || r0 == r1)
		return;
	if (threadIdx.x == 0 && r1 - r0 > PJ_CHUNK)
		mdb_raise(a.status, 16u);	/* the emit kernel will sweep this leaf's right rows in several chunks: they must be in row-id order */
	for (uint32_t s = threadIdx.x; s <= PJ_SLOTS; s += LEAF_THREADS) {
		if (s < PJ_SLOTS)
			s_key[s] = 0ull;
		s_cnt[s] = 0;
	}
	if (threadIdx.x == 0)
		s_total = 0ull;
	__syncthreads();
	for (uint32_t j = r0 + threadIdx.x; j < r1; j += LEAF_THREADS) {
		const uint64_t hv = a.hv_r[j];
		uint32_t s = PJ_SLOTS;
		if (hv != 0) {
			s = leaf_insert(s_key, PJ_SLOTS, hv);
			if (s == 0xFFFFFFFFu) {
				mdb_raise(a.status, 1u);
				continue;
			}
		}
		atomicAdd(&s_cnt[s], 1u);
	}
	__syncthreads();
	unsigned long long mine = 0;
	for (uint32_t i = l0 + threadIdx.x; i < l1; i += LEAF_THREADS) {
		const uint64_t hv = a.hv_l[i];
		uint32_t s = PJ_SLOTS;
		if (hv != 0)
			s = leaf_find(s_key, PJ_SLOTS, hv);
		if (s != 0xFFFFFFFFu) {
			const uint32_t m = s_cnt[s];
			if (m && a.rid_l[i] < a.n_l) {
				a.match[a.rid_l[i]] = m;
				mine += m;
			}
		}
	}
	/* one global atomic per workgroup (a per-thread atomic on this single address serialised 10^7 updates
	 * and cost 1.5 ms at 10^7 rows - profiles/r01/operators.json history) */
	if (mine)
		atomicAdd(&s_total, mine);
	__syncthreads();
	if (threadIdx.x == 0 && s_total)
		atomicAdd(a.total64, s_total);
}

/* ------------------------------------------------------------------ materialising join: emit phase
 *
 * Per leaf: table of the right side's distinct keys; the right rows are swept in chunks of
 * PJ_CHUNK.  Inside a chunk the row ids of each key are placed contiguously (LDS counting sort by
 * slot; ranks by comparing row ids, so every key's list is ascending = right-minor order), then
 * every left row of the leaf copies its key's list to out[offset(left row) + matches so far].
 */
__global__ __launch_bounds__(LEAF_THREADS) void k_leaf_pairs_emit(pj_args a)
{
	__shared__ unsigned long long s_key[PJ_SLOTS];
	__shared__ uint32_t s_cnt[PJ_SLOTS + 1];	/* matches of the slot inside the current chunk */
	__shared__ uint32_t s_start[PJ_SLOTS + 1];	/* first list position of the slot inside the chunk */
	__shared__ uint32_t s_cur[PJ_SLOTS + 1];
	__shared__ uint32_t s_prior[PJ_SLOTS + 1];	/* matches of the slot in earlier chunks */
	__shared__ uint32_t s_tmp[PJ_CHUNK];
	__shared__ uint32_t s_sorted[PJ_CHUNK];
	__shared__ uint16_t s_eslot[PJ_CHUNK];
	__shared__ uint32_t s_scan[32];

	const uint32_t leaf = blockIdx.x;
	uint32_t l0, l1, r0, r1;
	gc_leaf_range(a.off_l, a.cnt_l, a.cap_l, leaf, &l0, &l1);
	gc_leaf_range(a.off_r, a.cnt_r, a.cap_r, leaf, &r0, &r1);
	if (l0 == l1 || r0 == r1)
		return;
	for (uint32_t s = threadIdx.x; s <= PJ_SLOTS; s += LEAF_THREADS) {
		if (s < PJ_SLOTS)
			s_key[s] = 0ull;
		s_prior[s] = 0;
	}
	__syncthreads();
	for (uint32_t j = r0 + threadIdx.x; j < r1; j += LEAF_THREADS) {
		const uint64_t hv = a.hv_r[j];
		if (hv != 0 && leaf_insert(s_key, PJ_SLOTS, hv) == 0xFFFFFFFFu)
			mdb_raise(a.status, 1u);
	}
	__syncthreads();

	constexpr uint32_t PER_T = (PJ_SLOTS + 1 + LEAF_THREADS - 1) / LEAF_THREADS;	/* slots scanned per thread */
	for (uint32_t c0 = r0; c0 < r1; c0 += PJ_CHUNK) {
		const uint32_t clen = (r1 - c0) < PJ_CHUNK ? (r1 - c0) : PJ_CHUNK;
		for (uint32_t s = threadIdx.x; s <= PJ_SLOTS; s += LEAF_THREADS) {
			s_cnt[s] = 0;
			s_cur[s] = 0;
		}
		__syncthreads();
		/* count the chunk's rows per slot */
		for (uint32_t e = threadIdx.x; e < clen; e += LEAF_THREADS) {
			const uint64_t hv = a.hv_r[c0 + e];
			uint32_t s = PJ_SLOTS;
			if (hv != 0)
				s = leaf_find(s_key, PJ_SLOTS, hv);
			if (s == 0xFFFFFFFFu)
				s = PJ_SLOTS;	/* only after an overflow, which fails the whole call anyway */
			s_eslot[e] = (uint16_t)s;
			atomicAdd(&s_cnt[s], 1u);
		}
		__syncthreads();
		/* exclusive scan of the slot counts -> list starts */
		{
			uint32_t v[PER_T], sum = 0;
#pragma unroll
			for (uint32_t k = 0; k < PER_T; k++) {
				const uint32_t s = threadIdx.x * PER_T + k;
				v[k] = s <= PJ_SLOTS ? s_cnt[s] : 0;
				sum += v[k];
			}
			uint32_t total;
			uint32_t run = mdb_block_excl_scan(sum, s_scan, &total);
#pragma unroll
			for (uint32_t k = 0; k < PER_T; k++) {
				const uint32_t s = threadIdx.x * PER_T + k;
				if (s <= PJ_SLOTS)
					s_start[s] = run;
				run += v[k];
			}
		}
		__syncthreads();
		/* place row ids by slot (arbitrary order inside a list) */
		for (uint32_t e = threadIdx.x; e < clen; e += LEAF_THREADS) {
			const uint32_t s = s_eslot[e];
			const uint32_t pos = s_start[s] + atomicAdd(&s_cur[s], 1u);
			s_tmp[pos] = a.rid_r[c0 + e];
		}
		__syncthreads();
		/* rank every row id inside its list (row ids are distinct) -> ascending lists */
		for (uint32_t e = threadIdx.x; e < clen; e += LEAF_THREADS) {
			const uint32_t s = s_eslot[e];
			const uint32_t rid = a.rid_r[c0 + e];
			const uint32_t b = s_start[s], m = s_cnt[s];
			uint32_t rank = 0;
			for (uint32_t k = 0; k < m; k++)
				rank += s_tmp[b + k] < rid;
			s_sorted[b + rank] = rid;
		}
		__syncthreads();
		/* every left row copies its key's list */
		for (uint32_t i = l0 + threadIdx.x; i < l1; i += LEAF_THREADS) {
			const uint64_t hv = a.hv_l[i];
			uint32_t s = PJ_SLOTS;
			if (hv != 0)
				s = leaf_find(s_key, PJ_SLOTS, hv);
			if (s == 0xFFFFFFFFu)
				continue;
			const uint32_t m = s_cnt[s];
			if (!m)
				continue;
			const uint32_t rid = a.rid_l[i];
			const uint32_t base = a.match[rid] + s_prior[s];
			const uint32_t b = s_start[s];
			for (uint32_t k = 0; k < m; k++) {
				a.out_l[base + k] = rid;
				a.out_r[base + k] = s_sorted[b + k];
			}
		}
		__syncthreads();
		for (uint32_t s = threadIdx.x; s <= PJ_SLOTS; s += LEAF_THREADS)
			s_prior[s] += s_cnt[s];
		__syncthreads();
	}
}

/* 0 = done, 1 = not applicable (overflow: use the general path), 2 = a key outside the window met the narrow form (call
 * again with narrow = false), 3 = a right key occurs more than once (not a unique-key join this way round), < 0 = error */
static int join_pairs_unique(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			     const uint64_t *null_r, uint64_t n_r, bool narrow, int64_t base, uint32_t **out_l, uint32_t **out_r,
			     uint64_t *out_count)
{
	int b1, b2, sb1 = 0, sb2 = 0;
	uint32_t kbits = 0;
	mdb_choose_bits(n_r, GC_TARGET, &b1, &b2);
	if ((n_r >> (b1 + b2)) > (uint64_t)GC_SLOTS * 7 / 10 || !order_bits(n_l, &kbits, &sb1, &sb2))
		return 1;
	const uint64_t rec_cap = gc_rec_capacity(ctx, n_l);
	size_t need = mdb_partition_arena_bytes(n_l, b1, b2, true, true) + mdb_partition_arena_bytes(n_r, b1, b2, true, true) +
		      mdb_align_up(rec_cap * 8) + order_records_arena_bytes(rec_cap, n_l, kbits, sb1, sb2) + 4096;
	int rc = mdb_arena_begin(ctx, need);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	mdb_part_result pl, pr;
	rc = mdb_partition_table(ctx, keys_r, null_r, n_r, b1, b2, !narrow, false, true, &pr, narrow ? 1 : 0, false, base);
	if (rc)
		return rc;
	rc = mdb_partition_table(ctx, keys_l, null_l, n_l, b1, b2, !narrow, false, true, &pl, narrow ? 1 : 0, false, base);
	if (rc)
		return rc;
	unsigned long long *rec = (unsigned long long *)mdb_arena_take(ctx, rec_cap * 8);
	if (!rec)
		return -MIDORIDB_INTERNAL;
	pu_args a;
	a.hv_l = pl.hv;
	a.rid_l = pl.rid;
	a.off_l = pl.leaf_off;
	a.cnt_l = pl.leaf_cnt;
	a.cap_l = pl.leaf_cap;
	a.hv_r = pr.hv;
	a.rid_r = pr.rid;
	a.off_r = pr.leaf_off;
	a.cnt_r = pr.leaf_cnt;
	a.cap_r = pr.leaf_cap;
	a.rec = rec;
	a.rec_count = ctx->d_status + 1;
	a.rec_valid = ctx->d_status + 8;
	a.rec_cap = (uint32_t)rec_cap;
	a.kbits = kbits;
	a.status = ctx->d_status;
	a.nleaves = pl.nleaves;
	{
		const uint32_t resident = 2u * (uint32_t)ctx->num_cus;
		const uint32_t grid = pl.nleaves < resident ? pl.nleaves : resident;
		if (narrow) {
			MDB_LAUNCH(ctx, "leaf_pairs_unique", k_leaf_pairs_unique<true>, grid, GC_THREADS, a);
		} else {
			MDB_LAUNCH(ctx, "leaf_pairs_unique", k_leaf_pairs_unique<false>, grid, GC_THREADS, a);
		}
	}
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint32_t status = (uint32_t)h[1];
	const uint64_t list_len = h[1] >> 32, J = (uint32_t)h[5];
	if (status & 128u)
		return 2;
	if (status & 32u)
		return 3;	/* a right key occurs more than once */
	if (status & (1u | 2u | 8u))
		return 1;
	*out_count = J;
	if (J == 0)
		return 0;
	uint32_t *ol = NULL, *orr = NULL;
	if (mdb_cached_alloc(ctx, J * 4, (void **)&ol) || mdb_cached_alloc(ctx, J * 4, (void **)&orr)) {
		if (ol)
			(void)mdb_cached_free(ctx, ol);
		return mdb_set_err(ctx, -MIDORIDB_NOMEM, "cannot allocate %llu join pairs", (unsigned long long)J);
	}
	rc = order_records(ctx, rec, list_len, n_l, kbits, sb1, sb2, ol, NULL, orr, NULL, NULL);
	if (rc) {
		(void)mdb_cached_free(ctx, ol);
		(void)mdb_cached_free(ctx, orr);
		return rc;
	}
	*out_l = ol;
	*out_r = orr;
	return 0;
}

/* ------------------------------------------------------------------ unique right keys in a window of at most 2^24 values: ONE level
 *
 * The primary-key join of BASELINE configs[1] (10^7 x 10^7 rows, 4 result columns).  Through the path above it is two
 * partition levels per table, a hashed leaf table, one 8-byte record per pair and an ordering sort of the records by left row
 * id: 0.46 ms of kernels before the projection.  When the key sample offers a compact window of at most 2^24 values (the
 * k-bit bijection of the compact narrow form) both tables are partitioned ONCE by 9 bits (mdb_part_filter.level0_only) and
 * one 1024-thread workgroup joins a whole digit: a table of 2^(k-9) <= 2^15 LDS words indexed by the remaining hash bits
 * holds right row id + 1 - plain stores, no atomics: that the right keys are unique is checked afterwards (the number of
 * occupied entries must equal the number of right rows; otherwise status bit 5 and the caller takes the other paths).  A
 * left row reads its entry and writes the partner to match[left row id]: a random 4-byte write, but into an array of 4 n_L
 * bytes that the Infinity Cache holds (the path is taken up to 2^25 left rows).  The pairs in the reference's order - left
 * row ascending - are then the non-zero entries of match[] in index order: a count / scan / emit compaction instead of a
 * sort.  (The scatter is what the kernel's time is made of: 0.16 ms per 10^7 x 10^7 rows where the two streams need 0.06;
 * non-temporal stores took 0.33 - the L2s merge the writes of neighbouring rows, which a digit's sub-regions deliver in
 * roughly ascending order.) */
#define PW_THREADS 1024
#define PW_MIN_REM 10u
#define PW_MAX_REM 15u
#define PW_UNROLL 4
#define PW_MAX_LEFT (1ull << 25)
#define MC_THREADS 256
#define MC_PER_THREAD 16
#define MC_BLOCK (MC_THREADS * MC_PER_THREAD)

struct pw_args {
	const uint64_t *hv_l, *hv_r;		/* first-level output: hash32 << 32 | row id */
	const uint32_t *cnt_l, *cnt_r;		/* rows per sub-region: [sub * nleaves + digit] */
	uint32_t cap_l, cap_r, nleaves, nsub;
	uint32_t *match;			/* [n_l], zeroed: right row id + 1 of the left row's partner */
	uint32_t n_l;				/* a word's row id is checked against it before it indexes match[]: a region that overflowed (status
						 * bit 1: the operator is redone) holds slots nobody wrote - whatever an earlier call left there */
	unsigned long long *joined;
	uint32_t *status;
};

__global__ __launch_bounds__(PW_THREADS) void k_leaf_pairs_wide(pw_args a, uint32_t rem, uint32_t shift)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t pw_tab[];
	__shared__ unsigned long long s_red[PW_THREADS / 64];
	const uint32_t T = 1u << rem, mask = T - 1u, leaf = blockIdx.x;
	for (uint32_t s = threadIdx.x; s < T; s += PW_THREADS)
		pw_tab[s] = 0u;
	__syncthreads();
	uint32_t rows_r = 0;
	for (uint32_t sub = 0; sub < a.nsub; sub++) {
		const uint32_t c0 = a.cnt_r[sub * a.nleaves + leaf], c = c0 < a.cap_r ? c0 : a.cap_r;
		const uint64_t *const src = a.hv_r + (size_t)(leaf * a.nsub + sub) * a.cap_r;
		rows_r += c;
		for (uint32_t i0 = 0; i0 < c; i0 += 2u * PW_THREADS * PW_UNROLL) {	/* uniform trip count */
			ulonglong2 v[PW_UNROLL];
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * PW_THREADS + threadIdx.x);
				v[u] = make_ulonglong2(0ull, 0ull);
				if (i < c)
					v[u] = *reinterpret_cast<const ulonglong2 *>(src + i);
			}
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * PW_THREADS + threadIdx.x);
				if (i < c)
					pw_tab[((uint32_t)(v[u].x >> 32) >> shift) & mask] = (uint32_t)v[u].x + 1u;
				if (i + 1 < c)
					pw_tab[((uint32_t)(v[u].y >> 32) >> shift) & mask] = (uint32_t)v[u].y + 1u;
			}
		}
	}
	__syncthreads();
	/* unique right keys: every right row has its own entry */
	unsigned long long occupied = 0;
	for (uint32_t s = threadIdx.x; s < T; s += PW_THREADS)
		occupied += pw_tab[s] != 0u;
	occupied = lw_block_sum(occupied, s_red);
	if (occupied != rows_r) {
		if (threadIdx.x == 0)
			mdb_raise(a.status, 32u);
		return;
	}
	unsigned long long pairs = 0;
	for (uint32_t sub = 0; sub < a.nsub; sub++) {
		const uint32_t c0 = a.cnt_l[sub * a.nleaves + leaf], c = c0 < a.cap_l ? c0 : a.cap_l;
		const uint64_t *const src = a.hv_l + (size_t)(leaf * a.nsub + sub) * a.cap_l;
		for (uint32_t i0 = 0; i0 < c; i0 += 2u * PW_THREADS * PW_UNROLL) {
			ulonglong2 v[PW_UNROLL];
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * PW_THREADS + threadIdx.x);
				v[u] = make_ulonglong2(0ull, 0ull);
				if (i < c)
					v[u] = *reinterpret_cast<const ulonglong2 *>(src + i);
			}
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * PW_THREADS + threadIdx.x);
				const unsigned long long w[2] = { v[u].x, v[u].y };
#pragma unroll
				for (int k = 0; k < 2; k++)
					if (i + k < c) {
						const uint32_t r = pw_tab[((uint32_t)(w[k] >> 32) >> shift) & mask];
						if (r && (uint32_t)w[k] < a.n_l) {
							a.match[(uint32_t)w[k]] = r;
							pairs++;
						}
					}
			}
		}
	}
	pairs = lw_block_sum(pairs, s_red);
	if (threadIdx.x == 0 && pairs)
		atomicAdd(a.joined, pairs);
}

/* non-zero entries per block of MC_BLOCK entries (lane-interleaved 16-byte loads: a wave reads 1 KiB per instruction) */
__global__ __launch_bounds__(MC_THREADS) void k_match_count(const uint32_t *__restrict__ match, uint32_t n, uint32_t *__restrict__ blk)
{
	__shared__ uint32_t s_tmp[32];
	uint32_t c = 0;
	if ((uint64_t)(blockIdx.x + 1) * MC_BLOCK <= n) {	/* (uniform) a full block: its loads are issued together */
		uint4 v[MC_PER_THREAD / 4];
#pragma unroll
		for (int q = 0; q < MC_PER_THREAD / 4; q++)
			v[q] = *reinterpret_cast<const uint4 *>(match + blockIdx.x * MC_BLOCK + ((uint32_t)q * MC_THREADS + threadIdx.x) * 4u);
#pragma unroll
		for (int q = 0; q < MC_PER_THREAD / 4; q++)
			c += (v[q].x != 0u) + (v[q].y != 0u) + (v[q].z != 0u) + (v[q].w != 0u);
	} else {
#pragma unroll
		for (int q = 0; q < MC_PER_THREAD / 4; q++) {
			const uint32_t i = blockIdx.x * MC_BLOCK + ((uint32_t)q * MC_THREADS + threadIdx.x) * 4u;
			if (i + 3 < n) {
				const uint4 v = *reinterpret_cast<const uint4 *>(match + i);
				c += (v.x != 0u) + (v.y != 0u) + (v.z != 0u) + (v.w != 0u);
			} else {
				for (uint32_t k = i; k < n && k < i + 4; k++)
					c += match[k] != 0u;
			}
		}
	}
	uint32_t total;
	(void)mdb_block_excl_scan(c, s_tmp, &total);
	if (threadIdx.x == 0)
		blk[blockIdx.x] = total;
}

/* the pairs in left-row order: (i, match[i] - 1) for every non-zero entry.  The block's entries are loaded lane-interleaved
 * (chunk q = entries [q * 1024, q * 1024 + 1024) of the block, 4 consecutive ones per thread), ranked chunk by chunk, staged
 * in LDS at their ranks and written with consecutive threads on consecutive pairs (thread-contiguous loads and stores - 64
 * scattered 4-byte accesses per instruction - took 0.10 ms per 10^7 entries instead of 0.03). */
__global__ __launch_bounds__(MC_THREADS) void k_match_emit(const uint32_t *__restrict__ match, uint32_t n, const uint32_t *__restrict__ blk_start,
							   uint32_t *__restrict__ out_l, uint32_t *__restrict__ out_r)
{
	__shared__ uint32_t s_tmp[32];
	__shared__ uint32_t s_l[MC_BLOCK], s_r[MC_BLOCK];
	uint32_t m[MC_PER_THREAD];
	uint32_t run = 0;
	const bool full = (uint64_t)(blockIdx.x + 1) * MC_BLOCK <= n;	/* (uniform) */
	if (full) {
#pragma unroll
		for (int q = 0; q < MC_PER_THREAD / 4; q++) {
			const uint4 v = *reinterpret_cast<const uint4 *>(match + blockIdx.x * MC_BLOCK + ((uint32_t)q * MC_THREADS + threadIdx.x) * 4u);
			m[4 * q] = v.x;
			m[4 * q + 1] = v.y;
			m[4 * q + 2] = v.z;
			m[4 * q + 3] = v.w;
		}
	}
#pragma unroll
	for (int q = 0; q < MC_PER_THREAD / 4 && !full; q++) {
		const uint32_t i = blockIdx.x * MC_BLOCK + ((uint32_t)q * MC_THREADS + threadIdx.x) * 4u;
		uint4 v = make_uint4(0u, 0u, 0u, 0u);
		if (i + 3 < n) {
			v = *reinterpret_cast<const uint4 *>(match + i);
		} else {
			if (i < n)
				v.x = match[i];
			if (i + 1 < n)
				v.y = match[i + 1];
			if (i + 2 < n)
				v.z = match[i + 2];
		}
		m[4 * q] = v.x;
		m[4 * q + 1] = v.y;
		m[4 * q + 2] = v.z;
		m[4 * q + 3] = v.w;
	}
#pragma unroll
	for (int q = 0; q < MC_PER_THREAD / 4; q++) {
		const uint32_t c = (m[4 * q] != 0u) + (m[4 * q + 1] != 0u) + (m[4 * q + 2] != 0u) + (m[4 * q + 3] != 0u);
		uint32_t total;
		uint32_t pos = run + mdb_block_excl_scan(c, s_tmp, &total);
		const uint32_t i = blockIdx.x * MC_BLOCK + ((uint32_t)q * MC_THREADS + threadIdx.x) * 4u;
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (m[4 * q + k]) {
				s_l[pos] = i + (uint32_t)k;
				s_r[pos] = m[4 * q + k] - 1u;
				pos++;
			}
		run += total;
	}
	__syncthreads();
	const uint32_t start = blk_start[blockIdx.x];
	for (uint32_t p = threadIdx.x; p < run; p += MC_THREADS) {
		out_l[start + p] = s_l[p];
		out_r[start + p] = s_r[p];
	}
}

/* 0 = done, 1 = not applicable (a first-level region overflowed), 2 = a key outside the window, 3 = a right key occurs more
 * than once, < 0 = error */
static int join_pairs_unique_wide(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
				  const uint64_t *null_r, uint64_t n_r, uint32_t kbits, int64_t lo, uint32_t **out_l, uint32_t **out_r,
				  uint64_t *out_count)
{
	const int b1 = 9;
	const uint32_t rem = kbits - (uint32_t)b1, shift = 32u - kbits;
	const uint32_t nb = (uint32_t)((n_l + MC_BLOCK - 1) / MC_BLOCK);
	const size_t need = mdb_partition_level0_arena_bytes(n_l, b1) + mdb_partition_level0_arena_bytes(n_r, b1) + mdb_align_up(n_l * 4) +
			    mdb_align_up(((size_t)nb + 2) * 4) + mdb_align_up(mdb_scan_scratch_words((uint64_t)nb + 1) * 4) + 8192;
	int rc = mdb_arena_begin(ctx, need);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	mdb_part_filter flt;
	memset(&flt, 0, sizeof(flt));
	flt.level0_only = true;
	mdb_part_result pl, pr;
	rc = mdb_partition_table(ctx, keys_r, null_r, n_r, b1, 0, false, false, true, &pr, 1, false, lo, kbits, &flt);
	if (rc)
		return rc;
	rc = mdb_partition_table(ctx, keys_l, null_l, n_l, b1, 0, false, false, true, &pl, 1, false, lo, kbits, &flt);
	if (rc)
		return rc;
	uint32_t *match = (uint32_t *)mdb_arena_take(ctx, n_l * 4);
	uint32_t *blk = (uint32_t *)mdb_arena_take(ctx, ((size_t)nb + 2) * 4);
	uint32_t *blk_tmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words((uint64_t)nb + 1) * 4);
	if (!match || !blk || !blk_tmp)
		return -MIDORIDB_INTERNAL;
	if (!pl.nsub || !pr.nsub || pl.nsub != pr.nsub || pl.nleaves != pr.nleaves || pl.w32 || pr.w32)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "one-level unique-key join: the tables are not in the first-level layout");
	MDB_HIP(ctx, hipMemsetAsync(match, 0, n_l * 4, ctx->stream));
	pw_args a;
	a.hv_l = pl.hv;
	a.hv_r = pr.hv;
	a.cnt_l = pl.leaf_cnt;
	a.cnt_r = pr.leaf_cnt;
	a.cap_l = pl.leaf_cap;
	a.cap_r = pr.leaf_cap;
	a.nleaves = pl.nleaves;
	a.nsub = pl.nsub;
	a.match = match;
	a.n_l = (uint32_t)n_l;
	a.joined = (unsigned long long *)(ctx->d_status + 2);
	a.status = ctx->d_status;
	const size_t lds = (size_t)4 << rem;
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_pairs_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	MDB_LAUNCH_LDS(ctx, "leaf_pairs_wide", k_leaf_pairs_wide, pl.nleaves, PW_THREADS, lds, a, rem, shift);
	/* the compaction's first half needs nothing from the host */
	MDB_LAUNCH(ctx, "match_count", k_match_count, nb, MC_THREADS, match, (uint32_t)n_l, blk);
	MDB_HIP(ctx, hipMemsetAsync(blk + nb, 0, 4, ctx->stream));
	rc = mdb_scan_u32_inplace(ctx, blk, (uint64_t)nb + 1, blk_tmp);
	if (rc)
		return rc;
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint32_t status = (uint32_t)h[1];
	const uint64_t J = h[2];
	if (status & 128u)
		return 2;
	if (status & 32u)
		return 3;
	if (status & 2u)
		return 1;
	*out_count = J;
	if (J == 0)
		return 0;
	uint32_t *ol = NULL, *orr = NULL;
	if (mdb_cached_alloc(ctx, J * 4, (void **)&ol) || mdb_cached_alloc(ctx, J * 4, (void **)&orr)) {
		if (ol)
			(void)mdb_cached_free(ctx, ol);
		return mdb_set_err(ctx, -MIDORIDB_NOMEM, "cannot allocate %llu join pairs", (unsigned long long)J);
	}
	MDB_LAUNCH(ctx, "match_emit", k_match_emit, nb, MC_THREADS, match, (uint32_t)n_l, blk, ol, orr);
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_l = ol;
	*out_r = orr;
	return 0;
}

/* ------------------------------------------------------------------ the same join when EVERY left row has its one partner, with the
 * right table's payload carried to the leaf (BASELINE configs[1]: A(id_a, f1) JOIN B(id_b, f2) on primary keys, SELECT *)
 *
 * Through mdb_dev_join_pairs + the projection, B's payload column is a random gather through the right row ids: 10^7 8-byte
 * reads that each move a 128-byte line (PMC: 1.87 GB read for 0.32 GB algorithmic, 0.22 of the 0.58 ms), after a 4-byte scatter
 * of the partners (0.34 GB written for 0.04).  Here up to two payload cells of the right table travel through its ONE partition
 * level beside the word (k_part_scatter<pf_key_cf_pay>: read in row order, written in region order - both sequential), the
 * leaf's LDS table holds the right row's PLACE in the digit's regions instead of its row id, and a left row that finds its
 * partner copies the cells from there (the digit's regions: lines the workgroup has just streamed) to out[left row id] - the one
 * scattered access left, an 8-byte store per cell.  When every left row found a partner (referential integrity: counted, not
 * assumed) the outputs ARE the joined rows' payload columns in left-row order, no row ids exist and nothing is compacted;
 * otherwise the call says "not served" and the pairs path answers.  (reference: _join_nested_loop_tbl2tbl + cpy_cols/merge_rows,
 * executor_select.c:1076-1149, 340-438) */
struct pp_args {
	const uint64_t *hv_l, *hv_r;
	const uint64_t *pay_r[2];
	uint64_t *out[2];
	uint32_t npay;
	const uint32_t *cnt_l, *cnt_r;
	uint32_t cap_l, cap_r, nleaves, nsub;
	uint32_t n_l;		/* rows of the left table: a word's row id is checked against it before it indexes out[] (an overflowed region -
				 * status bit 1, the join is then "not served" - holds slots nobody wrote: whatever an earlier call left there) */
	unsigned long long *joined;
	uint32_t *status;
};

/* a digit's rows lie in nsub sub-regions: walked one after the other, every region is a round of its own - load, look up, load, store,
 * each waiting for the one before (8 rounds of mostly idle threads per digit: the kernel took 0.31 ms where its streams need 0.06).
 * Here the sub-regions' 16-byte chunks (pairs of words) form ONE list: pstart[s] = chunks before sub-region s, pstart[nsub] = all */
#define PP_MAX_SUB 8
__device__ static inline void pp_locate(const uint32_t *pstart, uint32_t nsub, uint32_t p, uint32_t *sub, uint32_t *i)
{
	uint32_t s = 0;
#pragma unroll
	for (uint32_t k = 1; k < PP_MAX_SUB; k++)
		s += (k < nsub && p >= pstart[k]) ? 1u : 0u;
	*sub = s;
	*i = 2u * (p - pstart[s]);
}

template <typename E /* table entry: the partner's place among the digit's regions + 1 (uint16_t when nsub * cap_r < 2^16: 64 KiB of LDS at
		       * 2^15 entries, two workgroups per CU - the phases of one overlap the other's) */>
__global__ __launch_bounds__(PW_THREADS) void k_leaf_pairs_payload(pp_args a, uint32_t rem, uint32_t shift)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t pp_lds[];
	E *const tab = reinterpret_cast<E *>(pp_lds);
	__shared__ unsigned long long s_red[PW_THREADS / 64];
	__shared__ uint32_t s_cnt[2][PP_MAX_SUB], s_pstart[2][PP_MAX_SUB + 1];
	const uint32_t T = 1u << rem, mask = T - 1u, leaf = blockIdx.x;
	for (uint32_t s = threadIdx.x; s < T * sizeof(E) / 4u; s += PW_THREADS)
		pp_lds[s] = 0u;
	if (threadIdx.x < 2u * PP_MAX_SUB) {
		const uint32_t side = threadIdx.x / PP_MAX_SUB, sub = threadIdx.x % PP_MAX_SUB;
		uint32_t c = 0;
		if (sub < a.nsub) {
			c = side ? a.cnt_r[sub * a.nleaves + leaf] : a.cnt_l[sub * a.nleaves + leaf];
			const uint32_t cap = side ? a.cap_r : a.cap_l;
			c = c < cap ? c : cap;
		}
		s_cnt[side][sub] = c;
	}
	__syncthreads();
	if (threadIdx.x < 2u) {
		uint32_t run = 0;
		for (uint32_t k = 0; k < PP_MAX_SUB; k++) {
			s_pstart[threadIdx.x][k] = run;
			run += (s_cnt[threadIdx.x][k] + 1u) >> 1;
		}
		s_pstart[threadIdx.x][PP_MAX_SUB] = run;
	}
	__syncthreads();
	/* ---- the right table: every row's place into the table */
	uint32_t rows_r = 0;
	for (uint32_t k = 0; k < a.nsub; k++)
		rows_r += s_cnt[1][k];
	{
		const uint32_t P = s_pstart[1][PP_MAX_SUB];
		for (uint32_t p0 = 0; p0 < P; p0 += PW_THREADS * PW_UNROLL) {	/* uniform trip count */
			ulonglong2 v[PW_UNROLL];
			uint32_t place[PW_UNROLL], left_in_sub[PW_UNROLL];
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				const uint32_t p = p0 + (uint32_t)u * PW_THREADS + threadIdx.x, pc = p < P ? p : P - 1u;
				uint32_t sub, i;
				pp_locate(s_pstart[1], a.nsub, pc, &sub, &i);
				v[u] = *reinterpret_cast<const ulonglong2 *>(a.hv_r + (size_t)(leaf * a.nsub + sub) * a.cap_r + i);
				place[u] = sub * a.cap_r + i + 1u;
				left_in_sub[u] = p < P ? s_cnt[1][sub] - i : 0u;	/* rows of the chunk that exist: 0, 1 or 2+ */
			}
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				if (left_in_sub[u] >= 1u)
					tab[((uint32_t)(v[u].x >> 32) >> shift) & mask] = (E)place[u];
				if (left_in_sub[u] >= 2u)
					tab[((uint32_t)(v[u].y >> 32) >> shift) & mask] = (E)(place[u] + 1u);
			}
		}
	}
	__syncthreads();
	unsigned long long occupied = 0;
	for (uint32_t s = threadIdx.x; s < T; s += PW_THREADS)
		occupied += tab[s] != 0;
	occupied = lw_block_sum(occupied, s_red);
	if (occupied != rows_r) {	/* a right key occurs twice */
		if (threadIdx.x == 0)
			mdb_raise(a.status, 32u);
		return;
	}
	/* ---- the left table: partner's place -> its cells -> out[left row id] */
	const size_t base_r = (size_t)leaf * a.nsub * a.cap_r;
	unsigned long long pairs = 0;
	{
		const uint32_t P = s_pstart[0][PP_MAX_SUB];
		for (uint32_t p0 = 0; p0 < P; p0 += PW_THREADS * PW_UNROLL) {
			ulonglong2 v[PW_UNROLL];
			uint32_t have[PW_UNROLL];
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				const uint32_t p = p0 + (uint32_t)u * PW_THREADS + threadIdx.x, pc = p < P ? p : P - 1u;
				uint32_t sub, i;
				pp_locate(s_pstart[0], a.nsub, pc, &sub, &i);
				v[u] = *reinterpret_cast<const ulonglong2 *>(a.hv_l + (size_t)(leaf * a.nsub + sub) * a.cap_l + i);
				have[u] = p < P ? s_cnt[0][sub] - i : 0u;
			}
			/* the partners' places first, then their cells (loads in flight together; a row without partner reads the digit's first
			 * cell: a load behind a per-row test is waited for before the next one is issued), then the stores */
			uint32_t e[2 * PW_UNROLL];
			uint64_t cell[2][2 * PW_UNROLL];
#pragma unroll
			for (int u = 0; u < PW_UNROLL; u++) {
				e[2 * u] = have[u] >= 1u ? (uint32_t)tab[((uint32_t)(v[u].x >> 32) >> shift) & mask] : 0u;
				e[2 * u + 1] = have[u] >= 2u ? (uint32_t)tab[((uint32_t)(v[u].y >> 32) >> shift) & mask] : 0u;
			}
#pragma unroll
			for (int q = 0; q < 2 * PW_UNROLL; q++)
				cell[0][q] = a.pay_r[0][base_r + (e[q] ? e[q] - 1u : 0u)];
			if (a.npay > 1u) {	/* (uniform) */
#pragma unroll
				for (int q = 0; q < 2 * PW_UNROLL; q++)
					cell[1][q] = a.pay_r[1][base_r + (e[q] ? e[q] - 1u : 0u)];
			}
#pragma unroll
			for (int q = 0; q < 2 * PW_UNROLL; q++)
				if (e[q]) {
					const uint32_t lrid = (uint32_t)((q & 1) ? v[q >> 1].y : v[q >> 1].x);
					if (lrid >= a.n_l)
						continue;
					a.out[0][lrid] = cell[0][q];
					if (a.npay > 1u)
						a.out[1][lrid] = cell[1][q];
					pairs++;
				}
		}
	}
	pairs = lw_block_sum(pairs, s_red);
	if (threadIdx.x == 0 && pairs)
		atomicAdd(a.joined, pairs);
}

/* ONE payload column: the cells themselves sit in the LDS table (8 bytes per key value, 2^14 of them: 128 KiB) - a right row drops
 * its cell at its key's slot, a left row picks it up: no global lookup at all, the only scattered access left is the store to
 * out[left row id].  A digit of 2^15 key values (windows of 2^24) is joined in two passes over its words, one per half of the
 * slots - the second pass reads what the first has just pulled through the L2.  (The variant above looks the cells up in the digit's
 * regions: 10^7 scattered 8-byte loads fetch 32-byte sectors that leave the 4 MiB L2 before their other cells are asked for -
 * PMC: 0.98 GB fetched for 0.24 GB - and cost 0.08 ms of its 0.30.)  Duplicate right keys are seen by the occupancy bitmap. */
#define PC_SLOT_BITS 14u
__global__ __launch_bounds__(PW_THREADS) void k_leaf_pairs_cell(pp_args a, uint32_t rem, uint32_t shift)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t pp_lds[];
	__shared__ unsigned long long s_red[PW_THREADS / 64];
	__shared__ uint32_t s_cnt[2][PP_MAX_SUB], s_pstart[2][PP_MAX_SUB + 1];
	__shared__ uint32_t s_dup;
	const uint32_t lowbits = rem < PC_SLOT_BITS ? rem : PC_SLOT_BITS, npass = 1u << (rem - lowbits), S = 1u << lowbits, leaf = blockIdx.x;
	const uint32_t mask = (1u << rem) - 1u;
	uint64_t *const cellt = reinterpret_cast<uint64_t *>(pp_lds);		/* [S] */
	uint32_t *const occ = pp_lds + 2u * S;					/* [S / 32] */
	if (threadIdx.x == 0)
		s_dup = 0u;
	if (threadIdx.x < 2u * PP_MAX_SUB) {
		const uint32_t side = threadIdx.x / PP_MAX_SUB, sub = threadIdx.x % PP_MAX_SUB;
		uint32_t c = 0;
		if (sub < a.nsub) {
			c = side ? a.cnt_r[sub * a.nleaves + leaf] : a.cnt_l[sub * a.nleaves + leaf];
			const uint32_t cap = side ? a.cap_r : a.cap_l;
			c = c < cap ? c : cap;
		}
		s_cnt[side][sub] = c;
	}
	__syncthreads();
	if (threadIdx.x < 2u) {
		uint32_t run = 0;
		for (uint32_t k = 0; k < PP_MAX_SUB; k++) {
			s_pstart[threadIdx.x][k] = run;
			run += (s_cnt[threadIdx.x][k] + 1u) >> 1;
		}
		s_pstart[threadIdx.x][PP_MAX_SUB] = run;
	}
	__syncthreads();
	unsigned long long pairs = 0;
	for (uint32_t pass = 0; pass < npass; pass++) {		/* (uniform) */
		for (uint32_t w = threadIdx.x; w < (S >> 5 ? S >> 5 : 1u); w += PW_THREADS)
			occ[w] = 0u;
		__syncthreads();
		{	/* ---- right rows of this pass's slots: cell into the table, occupancy bit set */
			const uint32_t P = s_pstart[1][PP_MAX_SUB];
			for (uint32_t p0 = 0; p0 < P; p0 += PW_THREADS * PW_UNROLL) {
				ulonglong2 v[PW_UNROLL], cv[PW_UNROLL];
				uint32_t have[PW_UNROLL];
#pragma unroll
				for (int u = 0; u < PW_UNROLL; u++) {
					const uint32_t p = p0 + (uint32_t)u * PW_THREADS + threadIdx.x, pc = p < P ? p : P - 1u;
					uint32_t sub, i;
					pp_locate(s_pstart[1], a.nsub, pc, &sub, &i);
					const size_t at = (size_t)(leaf * a.nsub + sub) * a.cap_r + i;
					v[u] = *reinterpret_cast<const ulonglong2 *>(a.hv_r + at);
					cv[u] = *reinterpret_cast<const ulonglong2 *>(a.pay_r[0] + at);
					have[u] = p < P ? s_cnt[1][sub] - i : 0u;
				}
#pragma unroll
				for (int u = 0; u < PW_UNROLL; u++) {
#pragma unroll
					for (int k = 0; k < 2; k++) {
						const uint32_t slot = ((uint32_t)((k ? v[u].y : v[u].x) >> 32) >> shift) & mask;
						if (have[u] > (uint32_t)k && (slot >> lowbits) == pass) {
							const uint32_t lo = slot & (S - 1u);
							const uint32_t old = atomicOr(&occ[lo >> 5], 1u << (lo & 31u));
							if (old & (1u << (lo & 31u)))
								s_dup = 1u;	/* a right key occurs twice */
							cellt[lo] = k ? cv[u].y : cv[u].x;
						}
					}
				}
			}
		}
		__syncthreads();
		if (s_dup) {
			if (threadIdx.x == 0)
				mdb_raise(a.status, 32u);
			return;
		}
		{	/* ---- left rows of this pass's slots: the partner's cell, if there is a partner, to out[left row id] */
			const uint32_t P = s_pstart[0][PP_MAX_SUB];
			for (uint32_t p0 = 0; p0 < P; p0 += PW_THREADS * PW_UNROLL) {
				ulonglong2 v[PW_UNROLL];
				uint32_t have[PW_UNROLL];
#pragma unroll
				for (int u = 0; u < PW_UNROLL; u++) {
					const uint32_t p = p0 + (uint32_t)u * PW_THREADS + threadIdx.x, pc = p < P ? p : P - 1u;
					uint32_t sub, i;
					pp_locate(s_pstart[0], a.nsub, pc, &sub, &i);
					v[u] = *reinterpret_cast<const ulonglong2 *>(a.hv_l + (size_t)(leaf * a.nsub + sub) * a.cap_l + i);
					have[u] = p < P ? s_cnt[0][sub] - i : 0u;
				}
#pragma unroll
				for (int u = 0; u < PW_UNROLL; u++) {
#pragma unroll
					for (int k = 0; k < 2; k++) {
						const unsigned long long w = k ? v[u].y : v[u].x;
						const uint32_t slot = ((uint32_t)(w >> 32) >> shift) & mask, lo = slot & (S - 1u);
						if (have[u] > (uint32_t)k && (slot >> lowbits) == pass && ((occ[lo >> 5] >> (lo & 31u)) & 1u) && (uint32_t)w < a.n_l) {
							a.out[0][(uint32_t)w] = cellt[lo];
							pairs++;
						}
					}
				}
			}
		}
		__syncthreads();
	}
	pairs = lw_block_sum(pairs, s_red);
	if (threadIdx.x == 0 && pairs)
		atomicAdd(a.joined, pairs);
}

/* Key windows beyond 2^24 values (10^8-row primary-key joins: 27 key bits): two partition levels - the right table's cells travel
 * through both (k_part_scatter<pf_key_cf_pay>, <pf_word_pay>) - down to leaves of 2^rem <= 2^12 key values, whose cells (8 or 16 bytes
 * per value) and occupancy bits sit in LDS; one 256-thread workgroup per leaf, several per CU.  Same contract as above. */
struct pl_args {
	const uint64_t *hv_l, *hv_r;
	const uint64_t *pay_r[2];
	uint64_t *out[2];
	uint32_t npay;
	const uint32_t *cnt_l, *cnt_r;
	uint32_t cap_l, cap_r, nleaves;
	uint32_t n_l;		/* (as in pp_args) */
	unsigned long long *joined;
	uint32_t *status;
};
#define PL_THREADS 256
#define PL_MAX_REM 12u

template <int NP /* payload columns: 32 KiB of LDS each - four (two) workgroups per CU */>
__global__ __launch_bounds__(PL_THREADS) void k_leaf_pairs_cell2(pl_args a, uint32_t rem, uint32_t shift)
{
	__shared__ uint64_t s_cell[NP][1u << PL_MAX_REM];
	__shared__ uint32_t s_occ[(1u << PL_MAX_REM) / 32];
	__shared__ unsigned long long s_red[PL_THREADS / 64];
	__shared__ uint32_t s_dup;
	const uint32_t T = 1u << rem, mask = T - 1u;
	unsigned long long pairs = 0;
	for (uint32_t leaf = blockIdx.x; leaf < a.nleaves; leaf += gridDim.x) {		/* (uniform) */
		for (uint32_t w = threadIdx.x; w < (T >> 5 ? T >> 5 : 1u); w += PL_THREADS)
			s_occ[w] = 0u;
		if (threadIdx.x == 0)
			s_dup = 0u;
		__syncthreads();
		const uint32_t cr0 = a.cnt_r[leaf], cr = cr0 < a.cap_r ? cr0 : a.cap_r, cl0 = a.cnt_l[leaf], cl = cl0 < a.cap_l ? cl0 : a.cap_l;
		const size_t br = (size_t)leaf * a.cap_r, bl = (size_t)leaf * a.cap_l;
		for (uint32_t i0 = 0; i0 < cr; i0 += 2u * PL_THREADS) {
			const uint32_t i = i0 + 2u * threadIdx.x, ic = i < cr ? i : 0u;
			const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.hv_r + br + ic);
			const ulonglong2 c0 = *reinterpret_cast<const ulonglong2 *>(a.pay_r[0] + br + ic);
			ulonglong2 c1 = make_ulonglong2(0ull, 0ull);
			if (NP > 1)
				c1 = *reinterpret_cast<const ulonglong2 *>(a.pay_r[1] + br + ic);
#pragma unroll
			for (int k = 0; k < 2; k++)
				if (i + (uint32_t)k < cr) {
					const uint32_t slot = ((uint32_t)((k ? v.y : v.x) >> 32) >> shift) & mask;
					const uint32_t old = atomicOr(&s_occ[slot >> 5], 1u << (slot & 31u));
					if (old & (1u << (slot & 31u)))
						s_dup = 1u;
					s_cell[0][slot] = k ? c0.y : c0.x;
					if (NP > 1)
						s_cell[1][slot] = k ? c1.y : c1.x;
				}
		}
		__syncthreads();
		if (s_dup) {	/* a right key occurs twice */
			if (threadIdx.x == 0)
				mdb_raise(a.status, 32u);
			return;
		}
		for (uint32_t i0 = 0; i0 < cl; i0 += 2u * PL_THREADS) {
			const uint32_t i = i0 + 2u * threadIdx.x, ic = i < cl ? i : 0u;
			const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.hv_l + bl + ic);
#pragma unroll
			for (int k = 0; k < 2; k++)
				if (i + (uint32_t)k < cl) {
					const unsigned long long w = k ? v.y : v.x;
					const uint32_t slot = ((uint32_t)(w >> 32) >> shift) & mask;
					if (((s_occ[slot >> 5] >> (slot & 31u)) & 1u) && (uint32_t)w < a.n_l) {
						a.out[0][(uint32_t)w] = s_cell[0][slot];
						if (NP > 1)
							a.out[1][(uint32_t)w] = s_cell[1][slot];
						pairs++;
					}
				}
		}
		__syncthreads();
	}
	/* (lw_block_sum is written for 1024-thread workgroups) */
#pragma unroll
	for (int o = 32; o; o >>= 1)
		pairs += __shfl_down(pairs, o, MDB_WAVE);
	if (mdb_lane() == 0)
		s_red[threadIdx.x >> 6] = pairs;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
		for (int w = 0; w < PL_THREADS / 64; w++)
			t += s_red[w];
		if (t)
			atomicAdd(a.joined, t);
	}
}

/* 0 = done: every one of the n_l left rows has its partner and out[c][i] = payload cell c of left row i's partner;
 * 1 = not served (some left row without a partner - NULL keys included -, duplicate right keys, no compact window of at most 2^24
 * values, a region overflow ...: mdb_dev_join_pairs answers); < 0 = error.  Synchronises. */
extern "C" int mdb_dev_join_payload(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
				    const uint64_t *null_r, uint64_t n_r, const void *const *pay_in, int npay, void *const *out)
{
	mdb_plan_scope plan_scope(ctx);
	if (!ctx || !keys_l || !keys_r || !pay_in || !out || npay < 1 || npay > 2)
		return -MIDORIDB_ERROR;
	for (int c = 0; c < npay; c++)
		if (!pay_in[c] || !out[c])
			return -MIDORIDB_ERROR;
	auto explained = [&](uint32_t form, uint32_t kbits, uint32_t levels, size_t arena) {	/* (mdb_dev_explain_join_payload: nothing is launched) */
		ctx->explain->payload_form = form;
		ctx->explain->payload_tables = form ? 1u : 0u;
		ctx->explain->key_form = 2;
		ctx->explain->key_bits = kbits;
		ctx->explain->levels = levels;
		ctx->explain->digits = 512;
		ctx->explain->from_stats = ctx->explain_as_sample ? 0u : ctx->pl_from_stats;
		ctx->explain->samples = ctx->explain_as_sample ? 1u : 0u;
		ctx->explain->arena_mib = (uint32_t)((arena + (1u << 20) - 1) >> 20);
		return MIDORIDB_OK;
	};
	if (n_l == 0 || n_r == 0 || n_l >= 0xFFFFFFFFull || n_r >= 0xFFFFFFFFull || n_l + n_r < (1ull << 20) || ld_disabled() ||
	    (mdb_knob("MDB_JOIN_PAYLOAD") && mdb_knob("MDB_JOIN_PAYLOAD")[0] == '0'))
		return 1;
	mdb_memo_switch(ctx, keys_l, n_l, keys_r, n_r);
	if (ctx->jp_bad_l == keys_l && ctx->jp_bad_nl == n_l && ctx->jp_bad_r == keys_r && ctx->jp_bad_nr == n_r && ++ctx->jp_bad_skips < 32)
		return 1;	/* (these columns were not such a join last time: not tried again for a while) */
	for (int attempt = 0; attempt < 2; attempt++) {
	bool narrow = false;
	int64_t base = 0;
	gc_window win = { 0, 0, false, false, false, false };
	int rc = gc_narrow_guess(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, &narrow, &base, &win);
	if (rc)
		return rc;
	const bool remembered = ctx->guess_remembered;
	if (mdb_knob("MDB_DEBUG_PAYLOAD"))
		fprintf(stderr, "join_payload: narrow %d window 2^%u at %lld (attempt %d, remembered %d)\n", (int)narrow, win.kbits, (long long)win.lo, attempt, (int)remembered);
	if (!narrow || !win.kbits)
		return 1;
	/* ---- round 5: both tables sorted tile by tile, the cells placed in the left table's row order without one scattered store per joined
	 * row (mdb_dev_rowjoin.hip): windows of up to 2^27 values, NULL-free 16-byte-aligned columns */
	if (mdb_rowjoin_serves(n_l, n_r, win.kbits, keys_l, null_l, keys_r, null_r, pay_in, out, npay)) {
		const uint32_t kbits = win.kbits;
		if (ctx->explain)
			return explained(3, kbits, 1, mdb_rowjoin_arena_bytes(n_l, n_r, kbits, npay) + 8192);
		rc = mdb_arena_begin(ctx, mdb_rowjoin_arena_bytes(n_l, n_r, kbits, npay) + 8192);
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
		rc = mdb_rowjoin_run(ctx, keys_l, n_l, keys_r, n_r, pay_in, win.lo, kbits, npay, out);
		if (rc)
			return rc;
		uint64_t *h = ctx->h_pinned;
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint32_t status = (uint32_t)h[1];
		const uint64_t J = h[2];
		if (mdb_knob("MDB_DEBUG_PAYLOAD"))
			fprintf(stderr, "join_payload (row order): k %u status %u J %llu of %llu left rows\n", kbits, status, (unsigned long long)J,
				(unsigned long long)n_l);
		if (status == 0 && J == (uint64_t)npay * n_l) {
			ctx->pl_payload_form = 3;
			ctx->pl_payload_tables = 1;
			return MIDORIDB_OK;
		}
		if ((status & 128u) && remembered && attempt == 0) {
			ctx->nh_result = -1;
			ctx->sr_valid = 0;
			ctx->nh_distrust = 1;
			continue;
		}
		if (status & 128u) {
			ctx->nh_distrust = 8;
		} else {
			ctx->jp_bad_l = keys_l;
			ctx->jp_bad_nl = n_l;
			ctx->jp_bad_r = keys_r;
			ctx->jp_bad_nr = n_r;
			ctx->jp_bad_skips = 0;
		}
		return 1;
	}
	if (win.kbits > 9u + PW_MAX_REM) {
		/* ---- windows of 2^25 ... 2^30 values: two levels, leaves of 2^12 values */
		if (win.kbits > 30u)
			return 1;
		const uint32_t kbits = win.kbits, shift = 32u - kbits;
		const int b1 = 9, b2 = (int)kbits - 9 - (int)PL_MAX_REM;
		const uint32_t rem = kbits - (uint32_t)(b1 + b2);
		if (b2 < 1 || b2 > MDB_MAX_RADIX_BITS || n_l >= 0xF0000000ull || n_r >= 0xF0000000ull)
			return 1;
		if (ctx->explain)
			return explained(2, kbits, 2, mdb_partition_arena_bytes(n_l, b1, b2, false, true) + mdb_partition_arena_bytes(n_r, b1, b2, false, true, npay) + 8192);
		rc = mdb_arena_begin(ctx, mdb_partition_arena_bytes(n_l, b1, b2, false, true) + mdb_partition_arena_bytes(n_r, b1, b2, false, true, npay) + 8192);
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
		mdb_part_filter rflt;
		memset(&rflt, 0, sizeof(rflt));
		rflt.npay = npay;
		for (int c = 0; c < npay; c++)
			rflt.pay_in[c] = pay_in[c];
		mdb_part_result pl, pr;
		memset(&pl, 0, sizeof(pl));
		memset(&pr, 0, sizeof(pr));
		rc = mdb_partition_table(ctx, keys_r, null_r, n_r, b1, b2, false, false, true, &pr, 1, false, win.lo, kbits, &rflt);
		if (rc)
			return rc;
		rc = mdb_partition_table(ctx, keys_l, null_l, n_l, b1, b2, false, false, true, &pl, 1, false, win.lo, kbits, NULL);
		if (rc)
			return rc;
		if (!pl.leaf_cap || !pr.leaf_cap || !pl.leaf_cnt || !pr.leaf_cnt || pl.nleaves != pr.nleaves || pl.w32 || pr.w32 || !pr.pay[0] ||
		    (npay > 1 && !pr.pay[1]))
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "join with payload: the tables are not in the two-level fixed-capacity layout");
		pl_args a;
		memset(&a, 0, sizeof(a));
		a.hv_l = pl.hv;
		a.hv_r = pr.hv;
		a.npay = (uint32_t)npay;
		for (int c = 0; c < npay; c++) {
			a.pay_r[c] = pr.pay[c];
			a.out[c] = reinterpret_cast<uint64_t *>(out[c]);
		}
		a.cnt_l = pl.leaf_cnt;
		a.cnt_r = pr.leaf_cnt;
		a.cap_l = pl.leaf_cap;
		a.cap_r = pr.leaf_cap;
		a.nleaves = pl.nleaves;
		a.n_l = (uint32_t)n_l;
		a.joined = (unsigned long long *)(ctx->d_status + 2);
		a.status = ctx->d_status;
		const uint32_t grid = pl.nleaves < 8u * (uint32_t)ctx->num_cus ? pl.nleaves : 8u * (uint32_t)ctx->num_cus;
		if (npay > 1) {
			MDB_LAUNCH(ctx, "leaf_pairs_payload", k_leaf_pairs_cell2<2>, grid, PL_THREADS, a, rem, shift);
		} else {
			MDB_LAUNCH(ctx, "leaf_pairs_payload", k_leaf_pairs_cell2<1>, grid, PL_THREADS, a, rem, shift);
		}
		uint64_t *h = ctx->h_pinned;
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint32_t status = (uint32_t)h[1];
		const uint64_t J = h[2];
		if (mdb_knob("MDB_DEBUG_PAYLOAD"))
			fprintf(stderr, "join_payload (two levels): k %u b2 %d rem %u status %u J %llu of %llu left rows\n", kbits, b2, rem, status, (unsigned long long)J,
				(unsigned long long)n_l);
		if (status == 0 && J == n_l) {
			ctx->pl_payload_form = 2;
			ctx->pl_payload_tables = 1;
			return MIDORIDB_OK;
		}
		if ((status & 128u) && remembered && attempt == 0) {
			ctx->nh_result = -1;
			ctx->sr_valid = 0;
			ctx->nh_distrust = 1;
			continue;
		}
		if (status & 128u) {
			ctx->nh_distrust = 8;
		} else if (!(status & 2u)) {	/* (a region overflow says nothing about the join) */
			ctx->jp_bad_l = keys_l;
			ctx->jp_bad_nl = n_l;
			ctx->jp_bad_r = keys_r;
			ctx->jp_bad_nr = n_r;
			ctx->jp_bad_skips = 0;
		}
		return 1;
	}
	const int b1 = 9;
	/* (a dimension table of a few thousand keys: the window may be wider than its keys need) */
	const uint32_t kbits = win.kbits < 9u + PW_MIN_REM ? 9u + PW_MIN_REM : win.kbits, rem = kbits - (uint32_t)b1, shift = 32u - kbits;
	if (ctx->explain)
		return explained(1, kbits, 1, mdb_partition_level0_arena_bytes(n_l, b1) + mdb_partition_level0_arena_bytes(n_r, b1, false, npay) + 8192);
	rc = mdb_arena_begin(ctx, mdb_partition_level0_arena_bytes(n_l, b1) + mdb_partition_level0_arena_bytes(n_r, b1, false, npay) + 8192);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	mdb_part_filter flt, rflt;
	memset(&flt, 0, sizeof(flt));
	flt.level0_only = true;
	rflt = flt;
	rflt.npay = npay;
	for (int c = 0; c < npay; c++)
		rflt.pay_in[c] = pay_in[c];
	mdb_part_result pl, pr;
	memset(&pl, 0, sizeof(pl));
	memset(&pr, 0, sizeof(pr));
	rc = mdb_partition_table(ctx, keys_r, null_r, n_r, b1, 0, false, false, true, &pr, 1, false, win.lo, kbits, &rflt);
	if (rc)
		return rc;
	rc = mdb_partition_table(ctx, keys_l, null_l, n_l, b1, 0, false, false, true, &pl, 1, false, win.lo, kbits, &flt);
	if (rc)
		return rc;
	if (!pl.nsub || pl.nsub != pr.nsub || pl.nleaves != pr.nleaves || pl.w32 || pr.w32 || !pr.pay[0] || (npay > 1 && !pr.pay[1]))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "join with payload: the tables are not in the first-level layout");
	pp_args a;
	memset(&a, 0, sizeof(a));
	a.hv_l = pl.hv;
	a.hv_r = pr.hv;
	a.npay = (uint32_t)npay;
	for (int c = 0; c < npay; c++) {
		a.pay_r[c] = pr.pay[c];
		a.out[c] = reinterpret_cast<uint64_t *>(out[c]);
	}
	a.cnt_l = pl.leaf_cnt;
	a.cnt_r = pr.leaf_cnt;
	a.cap_l = pl.leaf_cap;
	a.cap_r = pr.leaf_cap;
	a.nleaves = pl.nleaves;
	a.nsub = pl.nsub;
	a.n_l = (uint32_t)n_l;
	a.joined = (unsigned long long *)(ctx->d_status + 2);
	a.status = ctx->d_status;
	if (pl.nsub > PP_MAX_SUB)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "join with payload: %u sub-regions per digit", pl.nsub);
	if (!(mdb_knob("MDB_PP_CELL") && mdb_knob("MDB_PP_CELL")[0] == '0')) {
		/* the cells IN the leaf's LDS table: one scattered access per joined row.  Two carried columns: the kernel once per column
		 * (round 5: 0.51 -> 2 x 0.22 ms at 10^7 rows - the region-lookup leaf pays two scattered accesses per row and cell pair); the second
		 * launch counts its pairs into a word nobody reads */
		const uint32_t lowbits = rem < PC_SLOT_BITS ? rem : PC_SLOT_BITS;
		const size_t lds = ((size_t)8 << lowbits) + ((size_t)1 << lowbits) / 8 + 64;
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_pairs_cell), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		for (int c = 0; c < npay; c++) {
			pp_args ac = a;
			ac.pay_r[0] = a.pay_r[c];
			ac.out[0] = a.out[c];
			ac.npay = 1;
			if (c)
				ac.joined = (unsigned long long *)(ctx->d_status + 6);
			MDB_LAUNCH_LDS(ctx, "leaf_pairs_payload", k_leaf_pairs_cell, pl.nleaves, PW_THREADS, lds, ac, rem, shift);
		}
	} else if ((uint64_t)pr.nsub * pr.leaf_cap < 0xFFFFull && !(mdb_knob("MDB_PP_E16") && mdb_knob("MDB_PP_E16")[0] == '0')) {
		const size_t lds = (size_t)2 << rem;
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_pairs_payload<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_pairs_payload", k_leaf_pairs_payload<uint16_t>, pl.nleaves, PW_THREADS, lds, a, rem, shift);
	} else {
		const size_t lds = (size_t)4 << rem;
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_pairs_payload<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_pairs_payload", k_leaf_pairs_payload<uint32_t>, pl.nleaves, PW_THREADS, lds, a, rem, shift);
	}
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint32_t status = (uint32_t)h[1];
	const uint64_t J = h[2];
	if (status == 0 && J == n_l) {
		ctx->pl_payload_form = 1;
		ctx->pl_payload_tables = 1;
		return MIDORIDB_OK;
	}
	if ((status & 128u) && remembered && attempt == 0) {
		/* a REMEMBERED window proved wrong: the buffers hold other data than when it was learned (a caller's allocator handed the
		 * same addresses out again) - forget, look at the data itself, once more */
		ctx->nh_result = -1;
		ctx->sr_valid = 0;
		ctx->nh_distrust = 1;
		continue;
	}
	if (status & 128u) {	/* a key outside the window after all: the sample is not trusted for a while */
		ctx->nh_distrust = 8;
	} else {
		ctx->jp_bad_l = keys_l;
		ctx->jp_bad_nl = n_l;
		ctx->jp_bad_r = keys_r;
		ctx->jp_bad_nr = n_r;
		ctx->jp_bad_skips = 0;
	}
	return 1;
	}
	return 1;
}

/* One left key column joined with SEVERAL right tables on that key, each carrying up to two payload columns - SELECT * FROM A JOIN B ON
 * A.k = B.k JOIN C ON A.k = C.k with B.k and C.k primary keys that cover A.k (BASELINE configs[4]'s join-only form; the reference's
 * _join_nested_loop_tbl2tbl followed by _join_nested_loop_tbl2mat, /root/reference/src/engine/executor_select.c:1076-1232): the row-order
 * form sorts the left table's tiles ONCE and serves every right table from one leaf launch and one placement pass (mdb_dev_rowjoin.hip).
 * The key window is the CALLER's statement about the columns ([key_min, key_max] covers every key of every table: the catalog's ranges)
 * and is verified like every promise: a key outside, a left row without partner in some table, a right key twice -> 1, nothing usable
 * written, the caller joins table by table.  Nothing is sampled, nothing remembered.  Synchronises. */
extern "C" int mdb_dev_join_payload_multi(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const struct mdb_dev_payload_right *right,
					  int nright, int64_t key_min, int64_t key_max)
{
	mdb_plan_scope plan_scope(ctx);
	if (!ctx || !keys_l || !right || nright < 1 || nright > 4)
		return -MIDORIDB_ERROR;
	struct mdb_rowjoin_right rt[4];
	int streams = 0;
	memset(rt, 0, sizeof(rt));
	for (int t = 0; t < nright; t++) {
		if (!right[t].keys || right[t].npay < 1 || right[t].npay > 2)
			return -MIDORIDB_ERROR;
		for (int c = 0; c < right[t].npay; c++)
			if (!right[t].pay_in[c] || !right[t].out[c])
				return -MIDORIDB_ERROR;
		streams += right[t].npay;
	}
	if (streams > 4 || key_min > key_max || n_l == 0 || ld_disabled() || (mdb_knob("MDB_JOIN_PAYLOAD") && mdb_knob("MDB_JOIN_PAYLOAD")[0] == '0') ||
	    (mdb_knob("MDB_JOIN_PAYLOAD_MULTI") && mdb_knob("MDB_JOIN_PAYLOAD_MULTI")[0] == '0'))
		return 1;
	const uint64_t span = (uint64_t)key_max - (uint64_t)key_min;
	if (span >= ((uint64_t)1 << 27))
		return 1;
	uint32_t kbits = 15;
	while (((uint64_t)1 << kbits) <= span)
		kbits++;
	for (int t = 0; t < nright; t++) {
		if (!mdb_rowjoin_serves(n_l, right[t].rows, kbits, keys_l, null_l, right[t].keys, right[t].nulls, right[t].pay_in, right[t].out, right[t].npay))
			return 1;
		rt[t].keys = right[t].keys;
		rt[t].n = right[t].rows;
		rt[t].npay = right[t].npay;
		for (int c = 0; c < right[t].npay; c++) {
			rt[t].pay_in[c] = right[t].pay_in[c];
			rt[t].out[c] = right[t].out[c];
		}
	}
	const size_t arena = mdb_rowjoin_arena_bytes_multi(n_l, rt, nright, kbits) + 8192;
	if (ctx->explain) {
		ctx->explain->payload_form = 3;
		ctx->explain->payload_tables = (uint32_t)nright;
		ctx->explain->key_form = 2;
		ctx->explain->key_bits = kbits;
		ctx->explain->levels = 1;
		ctx->explain->digits = 1u << mdb_rowjoin_dbits(kbits);
		ctx->explain->from_stats = 1;
		ctx->explain->arena_mib = (uint32_t)((arena + (1u << 20) - 1) >> 20);
		return MIDORIDB_OK;
	}
	int rc = mdb_arena_begin(ctx, arena);
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	rc = mdb_rowjoin_run_multi(ctx, keys_l, n_l, rt, nright, key_min, kbits);
	if (rc)
		return rc;
	uint64_t *h = ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint32_t status = (uint32_t)h[1];
	const uint64_t J = h[2];
	if (mdb_knob("MDB_DEBUG_PAYLOAD"))
		fprintf(stderr, "join_payload_multi (row order): %d tables, %d columns, k %u status %u J %llu of %llu x %d\n", nright, streams, kbits, status,
			(unsigned long long)J, (unsigned long long)n_l, streams);
	if (status == 0 && J == (uint64_t)streams * n_l) {
		ctx->pl_payload_form = 3;
		ctx->pl_payload_tables = (uint32_t)nright;
		ctx->pl_key_bits = kbits;
		ctx->pl_from_stats = 1;
		return MIDORIDB_OK;
	}
	return 1;
}

#define SORT_SWAP_MIN_ROWS (1u << 18)

/* ------------------------------------------------------------------ tiny materialising join: one kernel, one workgroup
 *
 * The reference's own test cases join a handful of rows (tests/engine/executor_select.c:102-260).  Up to TINY_ROWS rows per
 * table the right keys sit in LDS and every left row simply walks over them - twice: once to count its matches, once,
 * after a prefix sum over the left rows, to write its pairs - which is the reference's nested loop and delivers its
 * left-major / right-minor order by construction.  All lanes read the same right row at the same time (an LDS broadcast). */
#define TINY_PAIRS_CAP 65536u

struct tinyp_args {
	const int64_t *keys_l;
	const uint64_t *null_l;
	uint32_t n_l;
	const int64_t *keys_r;
	const uint64_t *null_r;
	uint32_t n_r;
	uint32_t *out_l, *out_r;
	uint32_t cap;
	uint32_t *status;	/* [0] bit 12: more pairs than cap, [1] pairs */
};

__global__ __launch_bounds__(GC_THREADS) void k_tiny_join_pairs(tinyp_args a)
{
	__shared__ int64_t s_kr[TINY_ROWS];
	__shared__ uint8_t s_nr[TINY_ROWS];
	__shared__ uint32_t s_tmp[32];
	for (uint32_t j = threadIdx.x; j < a.n_r; j += GC_THREADS) {
		s_kr[j] = a.keys_r[j];
		s_nr[j] = a.null_r && mdb_bit_is_set(a.null_r, j);
	}
	__syncthreads();
	int64_t kl[LEAF_BATCH];
	bool ok[LEAF_BATCH];
	uint32_t m[LEAF_BATCH];
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		const uint32_t i = threadIdx.x + (uint32_t)u * GC_THREADS;
		ok[u] = i < a.n_l && !(a.null_l && mdb_bit_is_set(a.null_l, i));
		kl[u] = ok[u] ? a.keys_l[i] : 0;
		m[u] = 0;
	}
	for (uint32_t j = 0; j < a.n_r; j++) {
		const int64_t k = s_kr[j];
		const bool live = !s_nr[j];
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++)
			m[u] += ok[u] && live && kl[u] == k;
	}
	uint32_t base = 0, off[LEAF_BATCH];
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {	/* rows in index order: u = 0 covers rows 0 .. GC_THREADS - 1 */
		uint32_t total;
		off[u] = base + mdb_block_excl_scan(m[u], s_tmp, &total);
		base += total;
	}
	if (base <= a.cap) {
		for (uint32_t j = 0; j < a.n_r; j++) {
			const int64_t k = s_kr[j];
			const bool live = !s_nr[j];
#pragma unroll
			for (int u = 0; u < LEAF_BATCH; u++)
				if (ok[u] && live && kl[u] == k) {
					a.out_l[off[u]] = threadIdx.x + (uint32_t)u * GC_THREADS;
					a.out_r[off[u]] = j;
					off[u]++;
				}
		}
	}
	if (threadIdx.x == 0) {
		a.status[0] = base > a.cap ? 4096u : 0u;
		a.status[1] = base;
	}
}

/* 0 = done, 1 = not applicable (too many rows or pairs), < 0 = error */
static int tiny_join_pairs(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			   const uint64_t *null_r, uint64_t n_r, uint32_t **out_l, uint32_t **out_r, uint64_t *out_count)
{
	if (n_l > TINY_ROWS || n_r > TINY_ROWS)
		return 1;
	const uint64_t worst = n_l * n_r;
	const uint32_t cap = worst < TINY_PAIRS_CAP ? (uint32_t)worst : TINY_PAIRS_CAP;
	uint32_t *ol = NULL, *orr = NULL;
	if (mdb_cached_alloc(ctx, (size_t)cap * 4, (void **)&ol) || mdb_cached_alloc(ctx, (size_t)cap * 4, (void **)&orr)) {
		if (ol)
			(void)mdb_cached_free(ctx, ol);
		return mdb_set_err(ctx, -MIDORIDB_NOMEM, "cannot allocate %u join pairs", cap);
	}
	tinyp_args a;
	a.keys_l = keys_l;
	a.null_l = null_l;
	a.n_l = (uint32_t)n_l;
	a.keys_r = keys_r;
	a.null_r = null_r;
	a.n_r = (uint32_t)n_r;
	a.out_l = ol;
	a.out_r = orr;
	a.cap = cap;
	a.status = ctx->d_status;
	mdb_prof_begin(ctx, "tiny_join_pairs", (const void *)k_tiny_join_pairs);	/* (not MDB_LAUNCH: an error has two buffers to give back) */
	hipLaunchKernelGGL(k_tiny_join_pairs, dim3(1), dim3(GC_THREADS), 0, ctx->stream, a);
	mdb_prof_end(ctx);
	uint32_t *h = (uint32_t *)ctx->h_pinned;
	hipError_t e = hipMemcpyAsync(h, ctx->d_status, 8, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess || (h[0] & 4096u)) {
		(void)mdb_cached_free(ctx, ol);
		(void)mdb_cached_free(ctx, orr);
		if (e != hipSuccess)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "tiny join failed: %s", hipGetErrorString(e));
		return 1;	/* more pairs than the small buffers hold: the general path sizes its output exactly */
	}
	*out_l = ol;
	*out_r = orr;
	*out_count = h[1];
	return 0;
}

/* the unique-key join with its narrow-form decision and retry; result codes of join_pairs_unique() except 2 */
static int join_pairs_unique_auto(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
				  const uint64_t *null_r, uint64_t n_r, uint32_t **out_l, uint32_t **out_r, uint64_t *out_count)
{
	bool narrow = false;
	int64_t base = 0;
	gc_window win = { 0, 0, false, false, false, false };
	int urc = gc_narrow_guess(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, &narrow, &base, &win);
	if (urc)
		return urc;
	/* a compact window of at most 2^24 key values: one partition level, see join_pairs_unique_wide (MDB_ONE_LEVEL=0 switches
	 * it off).  What it cannot do - a key outside the window after all, a first-level region overflow - goes the usual way */
	if (narrow && win.kbits >= 9u + PW_MIN_REM && win.kbits <= 9u + PW_MAX_REM && n_l <= PW_MAX_LEFT && n_l + n_r >= (1ull << 20) &&
	    !(ctx->pw_bad_keys == keys_r && ctx->pw_bad_n == n_r) && !ld_disabled() &&
	    !(mdb_knob("MDB_ONE_LEVEL") && mdb_knob("MDB_ONE_LEVEL")[0] == '0')) {
		urc = join_pairs_unique_wide(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, win.kbits, win.lo, out_l, out_r, out_count);
		if (urc <= 0 || urc == 3)
			return urc;
		ctx->pw_bad_keys = keys_r;
		ctx->pw_bad_n = n_r;
	}
	urc = join_pairs_unique(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, narrow, base, out_l, out_r, out_count);
	if (urc == 2) {	/* the sample (or what was remembered about these columns) missed a wide key */
		if (ctx->narrow_mode == 1) {
			ctx->nh_distrust = 8;
			gc_narrow_note(ctx, keys_l, n_l, keys_r, n_r, false);
		}
		urc = join_pairs_unique(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, false, 0, out_l, out_r, out_count);
	}
	return urc;
}

/* Unique LEFT keys, duplicates on the right (FROM pk_table JOIN fk_table): the unique-key join runs with the sides
 * swapped - it delivers the pairs in right-row order - and a stable sort by left row id puts them into the reference's
 * left-major / right-minor order (for one left row the right rows are already ascending).  About half the time of the
 * general count / scan / emit path.  Same result codes as join_pairs_unique(). */
static int join_pairs_unique_left(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
				  const uint64_t *null_r, uint64_t n_r, uint32_t **out_l, uint32_t **out_r, uint64_t *out_count)
{
	if (n_l >= 0x7FFFFFFFull)
		return 1;
	uint32_t *sr = NULL, *sl = NULL;	/* swapped call: "left" ids are right rows, "right" ids are left rows */
	uint64_t J = 0;
	int rc = join_pairs_unique_auto(ctx, keys_r, null_r, n_r, keys_l, null_l, n_l, &sr, &sl, &J);
	if (rc)
		return rc;
	*out_count = J;
	if (J == 0)
		return 0;
	int64_t *wide = NULL;
	uint32_t *perm = NULL, *ol = NULL, *orr = NULL;
	rc = -MIDORIDB_NOMEM;
	if (mdb_cached_alloc(ctx, J * 8, (void **)&wide) || mdb_cached_alloc(ctx, J * 4, (void **)&perm) ||
	    mdb_cached_alloc(ctx, J * 4, (void **)&ol) || mdb_cached_alloc(ctx, J * 4, (void **)&orr)) {
		(void)mdb_set_err(ctx, -MIDORIDB_NOMEM, "cannot allocate %llu join pairs", (unsigned long long)J);
		goto fail;
	}
	/* the pairs as words, sorted: nothing to widen, no permutation to gather through */
	rc = mdb_sort_pairs(ctx, sl, sr, J, n_l, n_r, ol, orr);
	if (rc < 0)
		goto fail;
	if (rc == 0) {
		(void)mdb_cached_free(ctx, wide);
		(void)mdb_cached_free(ctx, perm);
		(void)mdb_cached_free(ctx, sr);
		(void)mdb_cached_free(ctx, sl);
		*out_l = ol;
		*out_r = orr;
		return 0;
	}
	/* few pairs or unevenly spread left rows: stable sort of a permutation by left row id, two gathers */
	rc = mdb_dev_widen32to64(ctx, (const int32_t *)sl, J, wide);
	if (rc)
		goto fail;
	{
		struct mdb_sort_key key;
		memset(&key, 0, sizeof(key));
		key.values = wide;
		key.type = MDB_T_INT64;
		rc = mdb_dev_sort_perm(ctx, &key, 1, J, perm);
		if (rc)
			goto fail;
	}
	rc = mdb_dev_gather32(ctx, sl, perm, J, ol);
	if (!rc)
		rc = mdb_dev_gather32(ctx, sr, perm, J, orr);
	if (!rc)
		rc = mdb_dev_sync(ctx);
	if (rc)
		goto fail;
	(void)mdb_cached_free(ctx, wide);
	(void)mdb_cached_free(ctx, perm);
	(void)mdb_cached_free(ctx, sr);
	(void)mdb_cached_free(ctx, sl);
	*out_l = ol;
	*out_r = orr;
	return 0;
fail:
	if (wide)
		(void)mdb_cached_free(ctx, wide);
	if (perm)
		(void)mdb_cached_free(ctx, perm);
	if (ol)
		(void)mdb_cached_free(ctx, ol);
	if (orr)
		(void)mdb_cached_free(ctx, orr);
	(void)mdb_cached_free(ctx, sr);
	(void)mdb_cached_free(ctx, sl);
	return rc < 0 ? rc : -MIDORIDB_INTERNAL;
}

extern "C" int mdb_dev_join_pairs(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
				  const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, uint32_t **out_l, uint32_t **out_r,
				  uint64_t *out_count)
{
	mdb_plan_scope plan_scope(ctx);
	*out_l = *out_r = NULL;
	*out_count = 0;
	ctx->last_pairs_identity = 0;
	if (n_l == 0 || n_r == 0)
		return MIDORIDB_OK;
	mdb_memo_switch(ctx, keys_l, n_l, keys_r, n_r);
	{
		const int trc = tiny_join_pairs(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, out_l, out_r, out_count);
		if (trc <= 0)
			return trc;
	}
	/* ---- unique right keys (the usual primary-key join): one record per pair, ordered like group records;
	 *      unique left keys: the same with the sides swapped and a stable sort.  What a column turned out to be is
	 *      remembered (by pointer and length), so that a repeated query does not pay for failed attempts. */
	{
		uint32_t *ul = NULL, *ur = NULL;
		uint64_t uj = 0;
		int urc = 3;
		bool right_dups = ctx->pu_dup_keys == keys_r && ctx->pu_dup_n == n_r;
		bool left_dups = ctx->pu_dupl_keys == keys_l && ctx->pu_dupl_n == n_l;
		if ((right_dups || left_dups) && ++ctx->pu_dup_skips > 32) {	/* the buffers may hold other data by now: look again once in a while */
			right_dups = left_dups = false;
			ctx->pu_dup_keys = ctx->pu_dupl_keys = NULL;
		}
		bool right_unique = false;
		if (!right_dups) {
			urc = join_pairs_unique_auto(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, &ul, &ur, &uj);
			if (urc < 0)
				return urc;
			right_unique = urc == 0;
			if (urc == 3) {
				ctx->pu_dup_keys = keys_r;
				ctx->pu_dup_n = n_r;
				ctx->pu_dup_skips = 0;
			}
		}
		if (urc == 3 && !left_dups && n_l + n_r >= SORT_SWAP_MIN_ROWS) {	/* small tables: the general path has fewer launches */
			urc = join_pairs_unique_left(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, &ul, &ur, &uj);
			if (urc < 0)
				return urc;
			if (urc == 3) {
				ctx->pu_dupl_keys = keys_l;
				ctx->pu_dupl_n = n_l;
				ctx->pu_dup_skips = 0;
			}
		}
		if (urc == 0) {
			*out_l = ul;
			*out_r = ur;
			*out_count = uj;
			/* (unique right keys: a left row has at most one partner; as many pairs as left rows, in left-row order: 0, 1, 2 ...) */
			ctx->last_pairs_identity = right_unique && uj == n_l;
			return MIDORIDB_OK;
		}
		/* duplicates on both sides, or a table / region overflowed: general path below */
	}
	int b1, b2;
	mdb_choose_bits(n_r, PJ_TARGET, &b1, &b2);
	const uint64_t mlen = n_l + 1;
	pj_args a;
	uint64_t J = 0;
	uint64_t *h = ctx->h_pinned;
	/* First with the histogram-free partition layout and the right side in arbitrary order (the emit kernel orders
	 * every key's row ids itself inside one chunk); the exact, stable layout is the fallback when a region
	 * overflows (skew) or a leaf holds more right rows than one chunk (their order across chunks matters). */
	for (int fast = 1; fast >= 0; fast--) {
		size_t need = mdb_partition_arena_bytes(n_l, b1, b2, true, fast != 0) + mdb_partition_arena_bytes(n_r, b1, b2, true, fast != 0) +
			      mdb_align_up(mlen * 4) + mdb_align_up(mdb_scan_scratch_words(mlen) * 4) + 4096;
		int rc = mdb_arena_begin(ctx, need);
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 8 * sizeof(uint32_t), ctx->stream));
		mdb_part_result pl, pr;
		hipStream_t main_stream = NULL;
		rc = mdb_aux_begin(ctx, &main_stream);	/* the right table is partitioned on the auxiliary stream (when enabled) */
		if (rc)
			return rc;
		rc = mdb_partition_table(ctx, keys_r, null_r, n_r, b1, b2, true, !fast, fast != 0, &pr);
		{
			int rc2 = mdb_aux_end(ctx, main_stream);
			if (rc)
				return rc;
			if (rc2)
				return rc2;
		}
		rc = mdb_partition_table(ctx, keys_l, null_l, n_l, b1, b2, true, false, fast != 0, &pl);
		if (rc)
			return rc;
		if ((rc = mdb_aux_join(ctx)))
			return rc;
		uint32_t *match = (uint32_t *)mdb_arena_take(ctx, mlen * 4);
		uint32_t *scan_tmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words(mlen) * 4);
		if (!match || !scan_tmp)
			return -MIDORIDB_INTERNAL;
		MDB_HIP(ctx, hipMemsetAsync(match, 0, mlen * 4, ctx->stream));

		a.hv_l = pl.hv;
		a.rid_l = pl.rid;
		a.off_l = pl.leaf_off;
		a.cnt_l = pl.leaf_cnt;
		a.cap_l = pl.leaf_cap;
		a.hv_r = pr.hv;
		a.rid_r = pr.rid;
		a.off_r = pr.leaf_off;
		a.cnt_r = pr.leaf_cnt;
		a.cap_r = pr.leaf_cap;
		a.match = match;
		a.n_l = (uint32_t)n_l;
		a.out_l = a.out_r = NULL;
		a.status = ctx->d_status;
		a.total64 = (unsigned long long *)(ctx->d_status + 2);
		a.nleaves = pl.nleaves;
		MDB_LAUNCH(ctx, "leaf_pairs_count", k_leaf_pairs_count, pl.nleaves, LEAF_THREADS, a);

		/* offsets are 32-bit; the 64-bit total written by the count kernel guards against N:M blow-ups */
		rc = mdb_scan_u32_inplace(ctx, match, mlen, scan_tmp);
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemcpyAsync(&h[0], match + n_l, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		J = (uint32_t)h[0];
		const uint32_t status = (uint32_t)h[1];
		if (fast && (status & (2u | 16u)))
			continue;	/* region overflow, or a multi-chunk leaf with unordered right rows */
		if (status & 1u)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL,
					   "leaf hash table overflow (more than %u distinct keys in one leaf): unsupported key skew", PJ_SLOTS);
		break;
	}
	if (h[2] != J)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "join produces %llu rows: more than the 2^32-1 a single call can materialise",
				   (unsigned long long)h[2]);
	if (J == 0)
		return MIDORIDB_OK;
	uint32_t *ol = NULL, *orr = NULL;
	if (mdb_cached_alloc(ctx, J * 4, (void **)&ol) || mdb_cached_alloc(ctx, J * 4, (void **)&orr)) {
		if (ol)
			(void)mdb_cached_free(ctx, ol);
		return mdb_set_err(ctx, -MIDORIDB_NOMEM, "cannot allocate %llu join pairs", (unsigned long long)J);
	}
	a.out_l = ol;
	a.out_r = orr;
	MDB_LAUNCH(ctx, "leaf_pairs_emit", k_leaf_pairs_emit, a.nleaves, LEAF_THREADS, a);
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_l = ol;
	*out_r = orr;
	*out_count = J;
	return MIDORIDB_OK;
}


/* ------------------------------------------------------------------ a join whose only output is the key column
 *
 * (include/mdb_dev.h: mdb_dev_join_keys; reference shape: _join_nested_loop_tbl2tbl, executor_select.c:1076-1149, followed by a
 * projection of the two key columns).  Nothing has to say WHICH rows met: (key, COUNT) per key that occurs on both sides - the
 * any-order join + GROUP BY operator - and every key written COUNT times. */
#define XK_THREADS 256

__global__ __launch_bounds__(XK_THREADS) void k_counts_u32(const int64_t *__restrict__ cnt, uint64_t n, uint32_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * XK_THREADS + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * XK_THREADS)
		out[i] = i < n ? (uint32_t)cnt[i] : 0u;		/* (one more word: the scan's grand total lands there) */
}

/* one wave per group run: lane j writes copies j, j + 64, ... (a group's copies are adjacent: coalesced stores whatever COUNT is) */
__global__ __launch_bounds__(XK_THREADS) void k_expand_keys(const int64_t *__restrict__ key, const uint32_t *__restrict__ pos, uint64_t n,
							     int64_t *__restrict__ out)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t wave = ((uint64_t)blockIdx.x * XK_THREADS + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * XK_THREADS) >> 6;
	for (uint64_t base = wave * 64; base < n; base += nwaves * 64) {
		const uint64_t i = base + lane;
		const int64_t k = i < n ? key[i] : 0;
		const uint32_t p = i < n ? pos[i] : 0u, e = i < n ? pos[i + 1] : 0u;
		if (__all(e - p <= 1u)) {		/* (unique partners: one store per lane) */
			if (e > p)
				out[p] = k;
			continue;
		}
		for (uint32_t g = 0; g < 64u; g++) {
			const int64_t kg = __shfl(k, (int)g, 64);
			const uint32_t pg = (uint32_t)__shfl((int)p, (int)g, 64), eg = (uint32_t)__shfl((int)e, (int)g, 64);
			for (uint32_t q = pg + lane; q < eg; q += 64u)
				out[q] = kg;
		}
	}
}

static inline uint32_t xk_grid(uint64_t n)
{
	const uint64_t b = (n + XK_THREADS - 1) / XK_THREADS;
	return (uint32_t)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

int mdb_expand_keys_by_count(mdb_dev_ctx *ctx, const int64_t *key, const int64_t *count, uint64_t groups, uint64_t joined, int64_t **out)
{
	*out = NULL;
	if (joined >= 0xFFFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "%llu joined rows exceed the 32-bit row limit of one GPU", (unsigned long long)joined);
	uint32_t *pos = NULL, *tmp = NULL;
	int64_t *rows = NULL;
	int rc = mdb_dev_alloc(ctx, (groups + 1) * 4, (void **)&pos);
	if (!rc)
		rc = mdb_dev_alloc(ctx, mdb_scan_scratch_words(groups + 1) * 4, (void **)&tmp);
	if (!rc)
		rc = mdb_dev_alloc(ctx, (joined ? joined : 1) * 8, (void **)&rows);
	if (!rc) {
		hipLaunchKernelGGL(k_counts_u32, dim3(xk_grid(groups + 1)), dim3(XK_THREADS), 0, ctx->stream, count, groups, pos);
		rc = mdb_scan_u32_inplace(ctx, pos, groups + 1, tmp);
	}
	if (!rc && groups)
		hipLaunchKernelGGL(k_expand_keys, dim3(xk_grid(groups)), dim3(XK_THREADS), 0, ctx->stream, key, pos, groups, rows);
	if (!rc)
		rc = mdb_dev_sync(ctx);
	(void)mdb_dev_free(ctx, pos);
	(void)mdb_dev_free(ctx, tmp);
	if (rc) {
		(void)mdb_dev_free(ctx, rows);
		return rc;
	}
	*out = rows;
	return MIDORIDB_OK;
}

/* The same in the REFERENCE's row order (left-major, executor_select.c:1096-1141) for the join whose matched keys are unique in the LEFT
 * table: the ordered join + GROUP BY + COUNT(*) operator delivers the keys that have partners in the left table's row order with their
 * COUNTs; J == G says that every COUNT is 1 - these keys ARE the joined rows -, and when its direct-address leaf kernels saw no key with
 * several left rows (status bit GC_ST_LEFT_DUPS) a COUNT is the number of the left row's partners: the key, COUNT times.  *served = 0
 * (nothing allocated) otherwise - the caller's materialising join answers -, remembered for these columns. */
extern "C" int mdb_dev_join_keys_ordered(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
					 const uint64_t *null_r, uint64_t n_r, int64_t **out_key, uint64_t *out_rows, int *served)
{
	mdb_plan_scope plan_scope(ctx);
	if (!ctx || !out_key || !out_rows || !served)
		return -MIDORIDB_ERROR;
	*out_key = NULL;
	*out_rows = 0;
	*served = 0;
	if (!n_l || !n_r)
		return MIDORIDB_OK;
	mdb_memo_switch(ctx, keys_l, n_l, keys_r, n_r);
	if (ctx->jk_dup_l == keys_l && ctx->jk_dup_nl == n_l && ctx->jk_dup_r == keys_r && ctx->jk_dup_nr == n_r && ++ctx->jk_dup_uses < 32)
		return MIDORIDB_OK;	/* (these columns had duplicates last time) */
	int64_t *gk = NULL, *gc = NULL;
	if (mdb_dev_alloc(ctx, n_l * 8, (void **)&gk) || mdb_dev_alloc(ctx, n_l * 8, (void **)&gc)) {
		(void)mdb_dev_free(ctx, gk);
		return -MIDORIDB_NOMEM;
	}
	uint64_t G = 0, J = 0;
	ctx->last_left_dups_known = false;
	/* (no copies where the operator knows for free that none is needed: every left row a group of COUNT 1 - two primary keys - makes the joined
	 * rows' key column the LEFT KEY COLUMN ITSELF, in its order: *served = 2, nothing allocated, nothing written) */
	const bool was_a = ctx->key_alias_ok, was_o = ctx->counts_optional;
	ctx->key_alias_ok = ctx->counts_optional = true;	/* (this call is the outermost operator: its wishes hold for the one it calls) */
	int rc = mdb_dev_join_group_count(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, MDB_ORDER_FIRST | MDB_KEYS_MAY_ALIAS | MDB_COUNTS_OPTIONAL, gk, gc, NULL,
					  n_l, &G, &J);
	ctx->key_alias_ok = was_a;
	ctx->counts_optional = was_o;
	const bool keys_left = !rc && ctx->pl_keys_left != 0, counts_one = !rc && ctx->pl_counts_one != 0;
	if (keys_left && J == G) {
		(void)mdb_dev_free(ctx, gk);
		(void)mdb_dev_free(ctx, gc);
		*out_key = const_cast<int64_t *>(keys_l);
		*out_rows = G;
		*served = 2;
		return MIDORIDB_OK;
	}
	if (!rc && (keys_left || counts_one) && J != G) {
		/* (cannot be: COUNTs that are all 1 join as many rows as there are groups, and a COUNT above 1 keeps the COUNT column) */
		(void)mdb_dev_free(ctx, gk);
		(void)mdb_dev_free(ctx, gc);
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "join of two key columns: %llu joined rows in %llu groups without a COUNT column", (unsigned long long)J,
				   (unsigned long long)G);
	}
	if (!rc && J != G && ctx->last_left_dups_known && !ctx->last_left_dups) {
		/* duplicates on the RIGHT side only (a dimension joined with its facts): a left row's joined rows all carry its key - the key,
		 * COUNT times, in the left table's row order */
		int64_t *rows = NULL;
		rc = mdb_expand_keys_by_count(ctx, gk, gc, G, J, &rows);
		(void)mdb_dev_free(ctx, gk);
		(void)mdb_dev_free(ctx, gc);
		if (rc)
			return rc;
		*out_key = rows;
		*out_rows = J;
		*served = 1;
		return MIDORIDB_OK;
	}
	(void)mdb_dev_free(ctx, gc);
	if (rc || J != G) {
		(void)mdb_dev_free(ctx, gk);
		if (!rc) {
			ctx->jk_dup_l = keys_l;
			ctx->jk_dup_nl = n_l;
			ctx->jk_dup_r = keys_r;
			ctx->jk_dup_nr = n_r;
			ctx->jk_dup_uses = 0;
		}
		return rc;
	}
	*out_key = gk;
	*out_rows = G;
	*served = 1;
	return MIDORIDB_OK;
}

extern "C" int mdb_dev_join_keys(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
				 const uint64_t *null_r, uint64_t n_r, int64_t **out_key, uint64_t *out_rows)
{
	mdb_plan_scope plan_scope(ctx);
	if (!ctx || !out_key || !out_rows)
		return -MIDORIDB_ERROR;
	*out_key = NULL;
	*out_rows = 0;
	const uint64_t cap = n_l ? n_l : 1;	/* (a group per distinct left key at most) */
	int64_t *gk = NULL, *gc = NULL;
	if (mdb_dev_alloc(ctx, cap * 8, (void **)&gk) || mdb_dev_alloc(ctx, cap * 8, (void **)&gc)) {
		(void)mdb_dev_free(ctx, gk);
		return -MIDORIDB_NOMEM;
	}
	uint64_t G = 0, J = 0;
	/* (no MDB_ORDER_FIRST, no first rows: the any-order form where it is served, else the ordered operator - its groups serve as well) */
	/* first WITHOUT the COUNT column (primary-key joins: every COUNT is 1, and writing 10^8 of them is 0.8 GB of the leaf kernel's
	 * 2.0): J == G says that this was right; otherwise - or when these columns had duplicates last time - once more with counts */
	int rc = MIDORIDB_OK;
	mdb_memo_switch(ctx, keys_l, n_l, keys_r, n_r);
	const bool had_dups = ctx->jk_dup_l == keys_l && ctx->jk_dup_nl == n_l && ctx->jk_dup_r == keys_r && ctx->jk_dup_nr == n_r && ++ctx->jk_dup_uses < 32;
	if (n_l && n_r && !had_dups) {
		ctx->unordered_no_counts = 1;
		rc = mdb_dev_join_group_count(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, 0u, gk, gc, NULL, cap, &G, &J);
		ctx->unordered_no_counts = 0;
		if (!rc && J != G) {
			ctx->jk_dup_l = keys_l;
			ctx->jk_dup_nl = n_l;
			ctx->jk_dup_r = keys_r;
			ctx->jk_dup_nr = n_r;
			ctx->jk_dup_uses = 0;
		}
	}
	if (!rc && n_l && n_r && (had_dups || J != G))
		rc = mdb_dev_join_group_count(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, 0u, gk, gc, NULL, cap, &G, &J);
	if (rc) {
		(void)mdb_dev_free(ctx, gk);
		(void)mdb_dev_free(ctx, gc);
		return rc;
	}
	if (J == G) {		/* every COUNT is 1: the group keys ARE the joined rows */
		(void)mdb_dev_free(ctx, gc);
		*out_key = gk;
		*out_rows = G;
		return MIDORIDB_OK;
	}
	int64_t *rows = NULL;
	rc = mdb_expand_keys_by_count(ctx, gk, gc, G, J, &rows);
	(void)mdb_dev_free(ctx, gk);
	(void)mdb_dev_free(ctx, gc);
	if (rc)
		return rc;
	*out_key = rows;
	*out_rows = J;
	return MIDORIDB_OK;
}
