/*
 * mdb_dev_leaf_wide.hip - the leaf kernels of the fused join / GROUP BY + COUNT(*) operators that join a whole first-level DIGIT per
 * workgroup from direct-address tables in LDS (split off mdb_dev_join.hip in round 4; what the files share: mdb_dev_join_internal.h):
 *   k_leaf_wide     one 9-bit level, key windows of 2^15 ... 2^23 values: 8 bytes of LDS per key value (first row + two 16-bit counts)
 *   k_leaf_wide4    the same with 4 bytes per key value + 4-bit left counts: two workgroups per CU (counts up to 31 / 15)
 *   k_leaf_wide12   the digits of ONE 4096-digit pass per table (windows of 2^24 ... 2^27 values), persistent, the left table as row words
 * and their launchers.  The reference's phases they stand for: the join and GROUP BY loops of src/engine/executor_select.c:1076-1149,
 * 1526-1588.  Hand-written HIP for gfx950; HBM- and issue-bound integer work: no MFMA.
 */
#include "mdb_dev_join_internal.h"

/* a region's words are read once: non-temporal, so that they do not push out what the leaf keeps coming back to (the bits of the left rows,
 * the ordering ranges it fills) */
__device__ static inline uint4 lw_load_nt(const void *p)
{
	typedef unsigned int u4_nt __attribute__((ext_vector_type(4)));
	const u4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u4_nt *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}

/* ------------------------------------------------------------------ wide direct-address leaves: ONE partition level
 *
 * Key windows of at most 2^23 values (a dimension table's keys; after R-based pruning the benchmark's variant D: 6.25 * 10^6
 * right keys) need 11 bits of partitioning before k_leaf_direct's 2^12-entry tables fit - two scatter levels, the second one
 * a full read + write of both tables for 3 or 4 bits.  Here the tables are partitioned ONCE, by 9 bits (histogram-free first
 * level, mdb_part_filter.level0_only), and one 1024-thread workgroup joins a whole digit: 2^rem entries, rem <= 14, with the
 * row counts as 16-BIT halves of 32-bit LDS words (8 bytes per entry instead of 12: 128 KiB at rem = 14).  A count that
 * outgrows its half carries into (or out of) the neighbour - every such accident makes the sum of the halves SMALLER than the
 * number of adds, which the emit pass checks: status bit 10, the operator is redone with two levels (and their hot-key path).
 * The digit's rows lie in the PART_NSUB sub-regions the first level wrote; they are streamed with four 16-byte loads in
 * flight per thread.  The records of a digit are counted first and appended with one global atomic: the list has no gaps.
 * Variant D: second-level scatters 0.194 + 0.031 ms and leaf 0.133 ms -> 0.134 ms.  Ablations (same box): the streams alone,
 * no LDS atomics, 0.116 ms; every atomic on one of 32 words 0.13; with the loads of the next step in flight while the current
 * one is counted (unconditional loads, s_waitcnt vmcnt(4..7) instead of 0) 0.143 - not latency: with one workgroup per CU
 * nothing streams while a workgroup clears its 128 KiB or emits. */

template <bool HAS_R, bool R16 = false /* the right table's words are 2 bytes: the hash bits below the digit (mdb_part_result.w16) */>
__global__ __launch_bounds__(LW_THREADS) void k_leaf_wide(gc_args a, uint32_t rem, uint32_t shift, uint32_t nsub)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lw_lds[];
	__shared__ unsigned long long s_red[LW_THREADS / 64];
	__shared__ uint32_t s_scan[32];
	__shared__ uint32_t s_base;
	const uint32_t T = 1u << rem, mask = T - 1u, leaf = blockIdx.x;
	uint32_t *const s_first = lw_lds;			/* first left row per key */
	uint32_t *const s_cl = lw_lds + T;			/* left rows per key (joins: only of keys that have right rows), 16-bit halves */
	uint32_t *const s_cr = lw_lds + T + T / 2;		/* right rows per key, 16-bit halves */
	for (uint32_t s = threadIdx.x; s < T; s += LW_THREADS)
		s_first[s] = 0xFFFFFFFFu;
	for (uint32_t s = threadIdx.x; s < (HAS_R ? T : T / 2); s += LW_THREADS)
		s_cl[s] = 0u;
	__syncthreads();

	uint32_t rows_r = 0;
	if (HAS_R && R16) {
		const uint16_t *const hv_r16 = reinterpret_cast<const uint16_t *>(a.hv_r);
		for (uint32_t sub = 0; sub < nsub; sub++) {
			const uint32_t c0 = a.cnt_r[sub * a.nleaves + leaf], c = c0 < a.cap_r ? c0 : a.cap_r;
			const uint16_t *const src = hv_r16 + (size_t)(leaf * nsub + sub) * a.cap_r;
			rows_r += c;
			for (uint32_t j0 = 0; j0 < c; j0 += 8u * LW_THREADS * LW_UNROLL) {	/* uniform trip count */
				uint4 v[LW_UNROLL];
#pragma unroll
				for (int u = 0; u < LW_UNROLL; u++) {
					const uint32_t j = j0 + 8u * ((uint32_t)u * LW_THREADS + threadIdx.x);
					v[u] = make_uint4(0u, 0u, 0u, 0u);
					if (j < c)
						v[u] = lw_load_nt(src + j);
				}
#pragma unroll
				for (int u = 0; u < LW_UNROLL; u++) {
					const uint32_t j = j0 + 8u * ((uint32_t)u * LW_THREADS + threadIdx.x);
					const uint32_t w[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
					if (j + 8u <= c) {	/* (all but a sub-region's last chunk: no test per word - one workgroup per CU is bound by the instructions it issues) */
#pragma unroll
						for (int k = 0; k < 8; k++) {
							const uint32_t idx = (w[k >> 1] >> (16 * (k & 1))) & mask;
							atomicAdd(&s_cr[idx >> 1], 1u << ((idx & 1u) * 16u));
						}
					} else {
#pragma unroll
						for (int k = 0; k < 8; k++)
							if (j + k < c) {
								const uint32_t idx = (w[k >> 1] >> (16 * (k & 1))) & mask;
								atomicAdd(&s_cr[idx >> 1], 1u << ((idx & 1u) * 16u));
							}
					}
				}
			}
		}
		__syncthreads();
	} else if (HAS_R) {
		const uint32_t *const hv_r32 = reinterpret_cast<const uint32_t *>(a.hv_r);
		for (uint32_t sub = 0; sub < nsub; sub++) {
			const uint32_t c0 = a.cnt_r[sub * a.nleaves + leaf], c = c0 < a.cap_r ? c0 : a.cap_r;
			const uint32_t *const src = hv_r32 + (size_t)(leaf * nsub + sub) * a.cap_r;
			rows_r += c;
			for (uint32_t j0 = 0; j0 < c; j0 += 4u * LW_THREADS * LW_UNROLL) {	/* uniform trip count */
				uint4 v[LW_UNROLL];
#pragma unroll
				for (int u = 0; u < LW_UNROLL; u++) {
					const uint32_t j = j0 + 4u * ((uint32_t)u * LW_THREADS + threadIdx.x);
					v[u] = make_uint4(0u, 0u, 0u, 0u);
					if (j < c)
						v[u] = lw_load_nt(src + j);
				}
#pragma unroll
				for (int u = 0; u < LW_UNROLL; u++) {
					const uint32_t j = j0 + 4u * ((uint32_t)u * LW_THREADS + threadIdx.x);
					const uint32_t w[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
					for (int k = 0; k < 4; k++)
						if (j + k < c) {
							const uint32_t idx = (w[k] >> shift) & mask;
							atomicAdd(&s_cr[idx >> 1], 1u << ((idx & 1u) * 16u));
						}
				}
			}
		}
		__syncthreads();
	}

	uint32_t adds = 0;
	for (uint32_t sub = 0; sub < nsub; sub++) {
		const uint32_t c0 = a.cnt_l[sub * a.nleaves + leaf], c = c0 < a.cap_l ? c0 : a.cap_l;
		const uint64_t *const src = a.hv_l + (size_t)(leaf * nsub + sub) * a.cap_l;
		for (uint32_t i0 = 0; i0 < c; i0 += 2u * LW_THREADS * LW_UNROLL) {
			ulonglong2 v[LW_UNROLL];
#pragma unroll
			for (int u = 0; u < LW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * LW_THREADS + threadIdx.x);
				v[u] = make_ulonglong2(0ull, 0ull);
				if (i < c)
					v[u] = *reinterpret_cast<const ulonglong2 *>(src + i);
			}
#pragma unroll
			for (int u = 0; u < LW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * LW_THREADS + threadIdx.x);
				const unsigned long long w[2] = { v[u].x, v[u].y };
#pragma unroll
				for (int k = 0; k < 2; k++)
					if (i + k < c) {
						const uint32_t idx = ((uint32_t)(w[k] >> 32) >> shift) & mask;
						if (!HAS_R || ((s_cr[idx >> 1] >> ((idx & 1u) * 16u)) & 0xFFFFu)) {
							atomicAdd(&s_cl[idx >> 1], 1u << ((idx & 1u) * 16u));
							atomicMin(&s_first[idx], (uint32_t)w[k]);
							adds++;
						}
					}
			}
		}
	}
	__syncthreads();

	/* emit: thread t owns the words [t * W, t * W + W) of halves = 2 W consecutive entries */
	const uint32_t W = rem > LW_EMIT_REM ? 1u << (rem - LW_EMIT_REM) : 1u, nwords = T / 2;
	uint32_t cl2[1u << (LW_MAX_REM - LW_EMIT_REM)], cr2[1u << (LW_MAX_REM - LW_EMIT_REM)];
	uint32_t mine = 0;
	unsigned long long sums = 0;	/* low half: right rows counted, high half: left rows counted */
#pragma unroll
	for (int k = 0; k < (int)(1u << (LW_MAX_REM - LW_EMIT_REM)); k++) {
		cl2[k] = 0u;
		cr2[k] = 0x00010001u;
		if ((uint32_t)k < W && threadIdx.x * W + (uint32_t)k < nwords) {
			cl2[k] = s_cl[threadIdx.x * W + (uint32_t)k];
			if (HAS_R)
				cr2[k] = s_cr[threadIdx.x * W + (uint32_t)k];
			mine += ((cl2[k] & 0xFFFFu) ? 1u : 0u) + ((cl2[k] >> 16) ? 1u : 0u);
			sums += ((unsigned long long)((cl2[k] & 0xFFFFu) + (cl2[k] >> 16)) << 32) | (HAS_R ? (cr2[k] & 0xFFFFu) + (cr2[k] >> 16) : 0u);
		}
	}
	const unsigned long long want = ((unsigned long long)adds << 32);
	const unsigned long long got = lw_block_sum(sums, s_red), asked = lw_block_sum(want, s_red);
	if (got != (asked | (HAS_R ? rows_r : 0u))) {	/* a 16-bit count overflowed: two levels and their hot-key path take over */
		if (threadIdx.x == 0)
			mdb_raise(a.status, 1024u);
		return;
	}
	uint32_t total;
	uint32_t pos = mdb_block_excl_scan(mine, s_scan, &total);
	if (!total)
		return;
	if (a.kbits) {
		if (threadIdx.x == 0) {
			const uint32_t nb = atomicAdd(a.rec_count, total);
			if (nb + total > a.rec_cap) {
				mdb_raise(a.status, 8u);
				s_base = 0xFFFFFFFFu;
			} else {
				s_base = nb;
				atomicAdd(a.rec_valid, total);
			}
		}
		__syncthreads();
		if (s_base == 0xFFFFFFFFu)
			return;
		pos += s_base;
	}
	unsigned long long joined = 0;
	uint32_t last_first = 0;	/* largest first row id of this digit's groups */
#pragma unroll
	for (int k = 0; k < (int)(1u << (LW_MAX_REM - LW_EMIT_REM)); k++) {
		if ((uint32_t)k >= W)
			continue;
#pragma unroll
		for (int e = 0; e < 2; e++) {
			const uint32_t cl = (cl2[k] >> (16 * e)) & 0xFFFFu;
			if (!cl)
				continue;
			if (HAS_R && cl > 1u)
				mdb_raise(a.status, GC_ST_LEFT_DUPS);
			const uint32_t s = 2u * (threadIdx.x * W + (uint32_t)k) + (uint32_t)e;
			const uint32_t first = s_first[s];
			const unsigned long long c = (unsigned long long)cl * ((cr2[k] >> (16 * e)) & 0xFFFFu);
			joined += c;
			last_first = first > last_first ? first : last_first;
			if (a.kbits && a.keyed_cbits) {
				if (c >> a.keyed_cbits)
					mdb_raise(a.status, 256u);	/* COUNT(*) does not fit a keyed record: redone with plain records */
				a.rec[pos++] = ((unsigned long long)first << (64 - a.kbits)) | ((unsigned long long)((leaf << rem) | s) << a.keyed_cbits) | c;
			} else if (a.kbits) {
				if (c >> (64 - a.kbits))
					mdb_raise(a.status, 4u);	/* COUNT(*) does not fit beside the row id */
				if (c >> (32 - (a.kbits < 32 ? a.kbits : 31)))
					mdb_raise(a.status, 16u);	/* ... nor in a 4-byte record */
				a.rec[pos++] = ((unsigned long long)first << (64 - a.kbits)) | c;
			} else {
				if (first < a.dense_n)
								a.dense_cnt[first] = (int64_t)c;
			}
		}
	}
	joined = lw_block_sum(joined, s_red);
	if (threadIdx.x == 0 && joined)
		atomicAdd(a.joined, joined);
	/* the largest first row id (status word 9): the groups of a plain GROUP BY over few values all begin in the first rows of the
	 * table - the ordering sort then sizes its regions for the row-id range that occurs */
#pragma unroll
	for (int o = 32; o; o >>= 1) {
		const uint32_t other = (uint32_t)__shfl_xor((int)last_first, o, MDB_WAVE);
		last_first = other > last_first ? other : last_first;
	}
	__syncthreads();	/* (s_red is free again) */
	if (mdb_lane() == 0)
		s_red[threadIdx.x >> 6] = last_first;
	__syncthreads();
	if (threadIdx.x == 0) {		/* ONE atomic per workgroup: 8192 of them on one address are 0.1 ms */
		uint32_t m = 0;
#pragma unroll
		for (int w = 0; w < LW_THREADS / 64; w++)
			m = (uint32_t)s_red[w] > m ? (uint32_t)s_red[w] : m;
		if (m)
			atomicMax(a.status + 9, m);
	}
}

/* ------------------------------------------------------------------ the same over the digits of ONE 4096-digit pass (round 4)
 *
 * Key windows of 2^24 ... 2^27 values (10^8 unique keys per table: the benchmark's variants U and S) took two 9-bit levels and
 * k_leaf_direct: the left table's 8-byte words written and read twice.  Here both tables go through ONE 4096-digit pass
 * (mdb_dev_shard.hip: k_shard_scatter_wide - 2-byte words for the right table, 4-byte ROW words for the left one: the row's place
 * in its tile | hash bits, one header per run naming the tile) and one workgroup joins a digit of up to 2^15 values from ONE table
 * of 4 bytes per value - right rows (5 bits) above the first left row (27 bits: min over the left rows, the count bits are final by
 * then) - plus 4 bits of left rows per value: 144 KiB of LDS.  More than 31 right or 15 left rows of one key are noticed (the sums of
 * the fields fall short of the rows counted) and reported - flag 1024: two levels and their hot-key path take over.
 *   A digit's words (at most 4 chunks of eight right words and 8 chunks of four left words per thread: the regions' capacity is
 * checked by the caller) are loaded once, all loads in flight together - and, the kernel being one 1024-thread workgroup per CU that
 * nothing else overlaps with, the NEXT digit's words are requested as soon as the current ones are counted, while its groups are
 * emitted (persistent grid; ablations at 10^8 x 10^8 unique keys: loads + clears alone 0.235 ms of a 0.85 ms two-pass version).
 *   A left word's row id needs the last header before it, which may lie in another lane's or another wave's words: a ballot inside
 * the wave, one LDS word per wave and one barrier per 4096 words across them (a region begins with a header: nothing is carried from
 * one region into the next by mistake).  Records and flags are k_leaf_wide's; the groups leave by wave-level append (coalesced). */
/* a barrier that waits for the wave's LDS operations only: the next digit's global loads stay in flight across it (__syncthreads() waits
 * for every outstanding memory operation) */
__device__ static inline void lw12_barrier(void)
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#define LW12_RB 4	/* 16-byte chunks per thread: right (8 words each) ... */
#define LW12_LB 8	/* ... and left (4 words each) */
#define LW12_NSUB 8	/* sub-regions per digit (SH_NSUB of mdb_dev_shard.hip) */
#define LW12_MAX_CR 31u
#define LW12_MAX_CL 15u

/* NX = 1: a further right table on the same key (A JOIN B ON a = b JOIN C ON a = c GROUP BY a: BASELINE configs[4]) - partitioned like the
 * first one; its rows are counted into the (still unused) 4-bit fields, the two counts multiplied into the 5-bit field before the left rows
 * come (a product beyond 31, or 16 rows of a key in the further table: flag 1024 like every count that does not fit). */
template <int NX, bool DN = false /* the bit-per-row form of the groups (gc_args.dn_bits) */>
__global__ __launch_bounds__(LW_THREADS) void k_leaf_wide12(gc_args a, uint32_t rem /* hash bits below the digit */, uint32_t nsub)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lw_lds[];
	__shared__ unsigned long long s_red[LW_THREADS / 64];
	__shared__ uint32_t s_red32[LW_THREADS / 64];
	__shared__ uint32_t s_base;
	/* [buffer][side][sub-region]: first 16-byte chunk (prefix) and words of a digit's sub-regions, side 0 left (4 words per chunk), 1 right (8) */
	__shared__ uint32_t s_chunk0[2][2 + NX][17], s_cnt[2][2 + NX][16];		/* (nsub <= 16; side 2: the further right table) */
	__shared__ uint32_t s_wlast[2][LW_THREADS / 64], s_carry[2];
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = mdb_lane();	/* (the wave: a scalar - what is derived from it is not kept in vector registers across the digit loop) */
	const uint32_t T = 1u << rem, mask = T - 1u;
	uint32_t *const s_fc = lw_lds;			/* right rows << 27 | first left row */
	uint32_t *const s_cl = lw_lds + T;		/* left rows per key (only of keys that have right rows), 4 bits each */
	uint32_t leaf = blockIdx.x, buf = 0;
	if (nsub != LW12_NSUB) {	/* (the caller's layout is not the one this kernel walks: say so, never an empty result) */
		if (threadIdx.x == 0)
			mdb_raise(a.status, 1024u);
		return;
	}
	/* (the bit-per-row form's pilot: the first dn_pilot digits only, nothing written but the counters - gc_args.dn_pilot) */
	const uint32_t run_leaves = (DN && a.dn_pilot && a.dn_pilot < a.nleaves) ? a.dn_pilot : a.nleaves;
	if (leaf >= run_leaves)
		return;
	auto seg_count = [&](uint32_t d) -> uint32_t {		/* threads 0 .. 16 * (2 + NX) - 1: (side, sub-region) */
		uint32_t tid = threadIdx.x;
		asm volatile("" : "+v"(tid));	/* (as in fetch_all below: no address kept across the digit loop) */
		const uint32_t side = tid >> 4, j = tid & 15u;
		if (tid >= 16u * (2u + NX) || j >= nsub)
			return 0u;
		const uint32_t c0 = (side == 0 ? a.cnt_l : side == 1 ? a.cnt_r : a.cnt_x[0])[j * a.nleaves + d];
		const uint32_t cap = side == 0 ? a.cap_l : side == 1 ? a.cap_r : a.cap_x[0];
		return c0 < cap ? c0 : cap;
	};
	auto seg_prefix = [&](uint32_t b) {		/* threads 0 .. 1 + NX */
		uint32_t run = 0, tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		for (uint32_t j = 0; j < nsub; j++) {
			s_chunk0[b][tid][j] = run;
			run += tid ? (s_cnt[b][tid][j] + 7u) >> 3 : (s_cnt[b][0][j] + 3u) >> 2;
		}
		s_chunk0[b][tid][nsub] = run;
	};
	uint4 vr[LW12_RB], vl[LW12_LB / 2];	/* (the left table's chunks 4 .. 7 take the right table's registers once its words are counted) */
	uint32_t nvr = 0, nvl = 0;	/* words of each chunk, 4 bits each */
	/* chunk q of a side lies in the last sub-region j whose first chunk c0[j] is <= q, at word (d * 8 + j) * cap + (q - c0[j]) * per of the
	 * table's buffer, and holds min(per, cnt[j] - (q - c0[j]) * per) words.  The three per-region terms of that - where the region's
	 * chunk 0 would sit minus c0[j] * per, and cnt[j] + c0[j] * per - are the same for every thread: read once per digit and side, kept in
	 * scalar registers, selected by seven compares (as a search loop over LDS per chunk, 12 chunks per thread and digit, and then as
	 * 135 instructions of 64-bit address arithmetic per chunk, this was the largest single part of the kernel) */
	auto fetch_all = [&](uint32_t d, uint32_t b, const int sides /* 1 left (chunks 0 .. 3), 8 left (chunks 4 .. 7, into the right table's registers), 2 right,
							       * 4 the further right table (into the right table's registers) */) {
#pragma unroll
		for (int side = 1 + NX; side >= 0; side--) {
			if (!(sides & (1 << side)) && !(side == 0 && (sides & 8)))
				continue;
			const int u_lo = (side == 0 && !(sides & 1)) ? LW12_LB / 2 : 0, u_hi = side ? LW12_RB : ((sides & 8) ? LW12_LB : LW12_LB / 2);
			const uint32_t per = side ? 8u : 4u, cap = side == 0 ? a.cap_l : side == 1 ? a.cap_r : a.cap_x[0];
			uint32_t c0[LW12_NSUB + 1], delta[LW12_NSUB], endq[LW12_NSUB];
#pragma unroll
			for (int j = 0; j <= LW12_NSUB; j++)
				c0[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk0[b][side][j]);
#pragma unroll
			for (int j = 0; j < LW12_NSUB; j++) {
				const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[b][side][j]);
				delta[j] = (d * LW12_NSUB + (uint32_t)j) * cap - c0[j] * per;	/* (word index: below 2^32, checked by the caller) */
				endq[j] = cnt + c0[j] * per;
			}
			uint32_t nvs = side ? 0u : nvl;
			/* (what a thread's chunks and the table's address come to is the same for every digit: left to itself the compiler keeps them in
			 * vector registers across the digit loop, spills them with a further right table, and reloads each one in front of its load
			 * behind s_waitcnt vmcnt(0) - the four loads of a side then go out one after the other's answer.  Recomputed per call instead.) */
			uint32_t tid = threadIdx.x;
			asm volatile("" : "+v"(tid));
			const void *tbl = side == 1 ? (const void *)a.hv_r : side == 2 ? (const void *)a.hv_x[0] : (const void *)a.hv_l;
			asm volatile("" : "+s"(tbl));
#pragma unroll
			for (int u = u_lo; u < u_hi; u++) {
				const uint32_t q = (uint32_t)u * LW_THREADS + tid, qp = q * per;
				uint4 v = make_uint4(0u, 0u, 0u, 0u);
				uint32_t nv = 0u;
				if (q < c0[LW12_NSUB]) {
					uint32_t dl = delta[0], en = endq[0];
#pragma unroll
					for (int j = 1; j < LW12_NSUB; j++) {
						const bool in = c0[j] <= q;
						dl = in ? delta[j] : dl;
						en = in ? endq[j] : en;
					}
					nv = en - qp < per ? en - qp : per;
					if (side)
						v = lw_load_nt(reinterpret_cast<const uint16_t *>(tbl) + (dl + qp));
					else
						v = lw_load_nt(reinterpret_cast<const uint32_t *>(tbl) + (dl + qp));
				}
				if (side)
					vr[u < LW12_RB ? u : 0] = v;
				else if (u < LW12_LB / 2)
					vl[u < LW12_LB / 2 ? u : 0] = v;
				else
					vr[u >= LW12_LB / 2 ? u - LW12_LB / 2 : 0] = v;
				nvs = (nvs & ~(15u << (4 * u))) | (nv << (4 * u));
			}
			if (side)
				nvr = nvs;
			else
				nvl = nvs;
		}
	};
	{
		const uint32_t c = seg_count(leaf);
		if (threadIdx.x < 16u * (2u + NX))
			s_cnt[0][threadIdx.x >> 4][threadIdx.x & 15u] = c;
		if (threadIdx.x == 0)
			s_carry[0] = 0u;
		__syncthreads();
		if (threadIdx.x < 2u + NX)
			seg_prefix(0);
		__syncthreads();
		fetch_all(leaf, 0, 2);
	}
	unsigned long long joined = 0;
	uint32_t cleared = 0;		/* (bit-per-row form) rows found not to be a group's first row */
	uint32_t last_first = 0;	/* largest first row id of this workgroup's groups */
	uint32_t it = 0;
	bool bad = false;		/* (uniform) a digit whose words do not fit the registers, or whose counts overflowed */
	for (; leaf < run_leaves; leaf += gridDim.x, buf ^= 1u) {
		const uint32_t next = leaf + gridDim.x;
		const uint32_t nch_l = s_chunk0[buf][0][nsub], nch_r = s_chunk0[buf][1][nsub];
		if (nch_l > LW12_LB * LW_THREADS || nch_r > LW12_RB * LW_THREADS || (NX && s_chunk0[buf][NX ? 2 : 1][nsub] > LW12_RB * LW_THREADS)) {	/* (the caller sized the regions so that this cannot happen) */
			bad = true;
			break;
		}
		/* the left words: on their way while the tables are cleared and the right words counted (the right words were requested while the
		 * previous digit's groups left: held across that phase they are 16 registers, with the left ones 48 - spills, whose reloads wait
		 * for everything in flight) */
		fetch_all(leaf, buf, 1);
		const uint32_t next_c = next < run_leaves ? seg_count(next) : 0u;	/* (on its way while this digit is counted) */
		{	/* (16 bytes a store: a quarter of the LDS instructions - the kernel is bound by their number; T >= 2^12) */
			uint32_t tid = threadIdx.x;
			asm volatile("" : "+v"(tid));	/* (no address kept - and spilled - across the digit loop) */
			const uint4 none = make_uint4(0x07FFFFFFu, 0x07FFFFFFu, 0x07FFFFFFu, 0x07FFFFFFu), zero = make_uint4(0u, 0u, 0u, 0u);
			for (uint32_t s = tid; s < T / 4u; s += LW_THREADS)
				reinterpret_cast<uint4 *>(s_fc)[s] = none;
			for (uint32_t s = tid; s < T / 32u; s += LW_THREADS)
				reinterpret_cast<uint4 *>(s_cl)[s] = zero;
		}
		lw12_barrier();
		uint32_t adds = 0, radds = 0;
#pragma unroll
		for (int u = 0; u < LW12_RB; u++) {
			const uint32_t w[4] = { vr[u].x, vr[u].y, vr[u].z, vr[u].w }, nv = (nvr >> (4 * u)) & 15u;
			if (nv == 8u) {		/* (all but a sub-region's last chunk; the words are below 2^rem as they were written) */
#pragma unroll
				for (uint32_t k = 0; k < 4u; k++) {
					atomicAdd(&s_fc[w[k] & 0xFFFFu], 1u << 27);
					atomicAdd(&s_fc[w[k] >> 16], 1u << 27);
				}
			} else {
#pragma unroll
				for (uint32_t k = 0; k < 8u; k++)
					if (k < nv)
						atomicAdd(&s_fc[(w[k >> 1] >> (16u * (k & 1u))) & mask], 1u << 27);
			}
			radds += nv;
		}
		if (threadIdx.x < 16u * (2u + NX))
			s_cnt[buf ^ 1u][threadIdx.x >> 4][threadIdx.x & 15u] = next_c;
		bool prod_bad = false;
		if (NX) {
			/* the further right table: its words into the registers the first one's have left, its rows into the 4-bit fields; then
			 * the 5-bit field becomes the product of the two counts and the 4-bit fields are the left table's again */
			fetch_all(leaf, buf, 4);
			uint32_t xadds = 0;
#pragma unroll
			for (int u = 0; u < LW12_RB; u++) {
				const uint32_t w[4] = { vr[u].x, vr[u].y, vr[u].z, vr[u].w }, nv = (nvr >> (4 * u)) & 15u;
#pragma unroll
				for (uint32_t k = 0; k < 8u; k++)
					if (k < nv) {
						const uint32_t idx = (w[k >> 1] >> (16u * (k & 1u))) & mask;
						atomicAdd(&s_cl[idx >> 3], 1u << ((idx & 7u) * 4u));
					}
				xadds += nv;
			}
			lw12_barrier();
			uint32_t sum_b = 0, sum_x = 0;
			for (uint32_t s0 = 0; s0 < T; s0 += LW_THREADS * 8u) {		/* (a thread: eight consecutive values = one word of 4-bit fields) */
				const uint32_t wi = (s0 >> 3) + threadIdx.x;
				if (wi >= T / 8u)
					continue;
				const uint32_t cw = s_cl[wi];
				uint32_t worst = 0;
#pragma unroll
				for (int hf = 0; hf < 2; hf++) {	/* (four values at a time: the kernel has no registers to spare) */
					const uint4 fq = *reinterpret_cast<const uint4 *>(s_fc + wi * 8u + 4u * hf);
					uint32_t f[4] = { fq.x, fq.y, fq.z, fq.w };
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const uint32_t cb = f[e] >> 27, cx = (cw >> (4 * (4 * hf + e))) & 15u, pr = cb * cx;
						sum_b += cb;
						sum_x += cx;
						worst = pr > worst ? pr : worst;
						f[e] = (pr << 27) | 0x07FFFFFFu;
					}
					*reinterpret_cast<uint4 *>(s_fc + wi * 8u + 4u * hf) = make_uint4(f[0], f[1], f[2], f[3]);
				}
				prod_bad = prod_bad || worst > LW12_MAX_CR;
				s_cl[wi] = 0u;
			}
			/* (counts that overflowed their fields: the fields sum to less than the rows counted) */
			unsigned long long d2 = (((unsigned long long)sum_x << 32) | sum_b) - (((unsigned long long)xadds << 32) | radds);
#pragma unroll
			for (int o = 32; o; o >>= 1)
				d2 += __shfl_down(d2, o, MDB_WAVE);
			const uint64_t pb = __ballot(prod_bad);
			lw12_barrier();
			if (lane == 0) {
				s_red[wave] = d2;
				s_red32[wave] = pb ? 1u : 0u;
			}
			lw12_barrier();
			unsigned long long t = 0ull;
			uint32_t anyp = 0u;
#pragma unroll
			for (int w = 0; w < LW_THREADS / 64; w++) {
				t += s_red[w];
				anyp |= s_red32[w];
			}
			prod_bad = anyp != 0u || t != 0ull;
			radds = 0;	/* (the 5-bit fields now hold products: their sum is not the rows counted - checked above) */
		}
		fetch_all(leaf, buf, 8);	/* (the right words are counted: their registers take the left table's chunks 4 .. 7) */
		lw12_barrier();
		if (threadIdx.x < 2u + NX)
			seg_prefix(buf ^ 1u);
#pragma unroll
		for (int u = 0; u < LW12_LB; u++) {
			if ((uint32_t)u * LW_THREADS >= nch_l)	/* (uniform) */
				continue;
			const uint4 vw = u < LW12_LB / 2 ? vl[u < LW12_LB / 2 ? u : 0] : vr[u >= LW12_LB / 2 ? u - LW12_LB / 2 : 0];
			const uint32_t w[4] = { vw.x, vw.y, vw.z, vw.w }, nv = (nvl >> (4 * u)) & 15u;
			/* the last header among this thread's words, the nearest earlier lane's, the nearest earlier wave's of this round, else what the
			 * previous round left */
			uint32_t my_last = 0u;
#pragma unroll
			for (uint32_t k = 0; k < 4u; k++)
				if (k < nv && (w[k] >> 31))
					my_last = w[k];
			const uint64_t bal = __ballot(my_last != 0u), before = bal & mdb_lanemask_lt();
			uint32_t cur = (uint32_t)__shfl((int)my_last, before ? 63 - __clzll((long long)before) : 0, MDB_WAVE);
			const uint32_t p = it & 1u;
			/* (a lane every lane of the wave names alike: read through a scalar register, not through the LDS crossbar - the kernel is bound by
			 * what it asks of the LDS pipeline, round 6) */
			const uint32_t wave_last = (uint32_t)__builtin_amdgcn_readlane((int)my_last, bal ? 63 - __clzll((long long)bal) : 0);
			if (lane == 0)
				s_wlast[p][wave] = bal ? wave_last : 0u;
			lw12_barrier();
			{
				uint32_t ln = lane;
				asm volatile("" : "+v"(ln));	/* (the slot's address: worked out here, not reloaded from scratch behind vmcnt(0)) */
				const uint32_t x = ln < LW_THREADS / 64 ? s_wlast[p][ln] : 0u;
				const uint64_t ball = __ballot(x != 0u), earlier = ball & ((1ull << wave) - 1ull);
				const uint32_t carried = s_carry[p];
				const uint32_t from_waves = (uint32_t)__builtin_amdgcn_readlane((int)x, earlier ? 63 - __clzll((long long)earlier) : 0);
				const uint32_t round_last = (uint32_t)__builtin_amdgcn_readlane((int)x, ball ? 63 - __clzll((long long)ball) : 0);
				if (!before)
					cur = earlier ? from_waves : carried;
				if (threadIdx.x == 0)
					s_carry[p ^ 1u] = ball ? round_last : carried;
			}
			it++;
			uint32_t fcv[4];
#pragma unroll
			for (uint32_t k = 0; k < 4u; k++)
				fcv[k] = s_fc[w[k] & mask];	/* (a header's or an absent word's slot is read and ignored) */
			uint32_t row2 = cur << 1;	/* the tile's first row (the header's top bit leaves) */
			if (DN) {
				/* the bit-per-row form: the table's answer - the smallest row so far - says which of two rows of a key is NOT its first
				 * (four requests go out, then the answers are looked at); a row without partner is no group at all */
				uint32_t prev[4], rowk[4];
#pragma unroll
				for (uint32_t k = 0; k < 4u; k++) {
					const bool hdr = (w[k] >> 31) != 0u;
					row2 = hdr ? w[k] << 1 : row2;
					const uint32_t idx = w[k] & mask, fc = fcv[k];
					rowk[k] = row2 + ((w[k] >> 15) & 0x7FFFu);
					prev[k] = 0xFFFFFFFEu;	/* (header / absent word) */
					if (k < nv && !hdr) {
						prev[k] = 0xFFFFFFFFu;	/* (no partner) */
						if (fc >> 27) {
							prev[k] = atomicMin(&s_fc[idx], (fc & 0xF8000000u) | rowk[k]) & 0x07FFFFFFu;
							atomicAdd(&s_cl[idx >> 3], 1u << ((idx & 7u) * 4u));
							adds++;
						}
					}
				}
#pragma unroll
				for (uint32_t k = 0; k < 4u; k++) {
					if (prev[k] == 0xFFFFFFFEu || prev[k] == 0x07FFFFFFu)
						continue;	/* (nothing there, or the key's first row so far) */
					const uint32_t loser = prev[k] == 0xFFFFFFFFu ? rowk[k] : (prev[k] > rowk[k] ? prev[k] : rowk[k]);
					if (a.dn_bits)
						atomicAnd(&a.dn_bits[loser >> 5], ~(1u << (loser & 31u)));
					cleared++;
				}
			} else {
#pragma unroll
			for (uint32_t k = 0; k < 4u; k++) {
				const bool hdr = (w[k] >> 31) != 0u;
				row2 = hdr ? w[k] << 1 : row2;
				const uint32_t idx = w[k] & mask, fc = fcv[k];
				if (k < nv && !hdr && (fc >> 27)) {	/* (the right rows are all counted: the top bits are final, the minimum is over the row id) */
					atomicMin(&s_fc[idx], (fc & 0xF8000000u) | (row2 + ((w[k] >> 15) & 0x7FFFu)));
					atomicAdd(&s_cl[idx >> 3], 1u << ((idx & 7u) * 4u));
					adds++;
				}
			}
			}
		}
		lw12_barrier();

		/* groups of the digit: the non-zero left counts.  Wave w owns the values [w * T / 16, (w + 1) * T / 16): their groups leave side by side,
		 * 64 values a step, written by consecutive lanes.  With one workgroup per CU the kernel is bound by the instructions it issues
		 * (ablations: this pass 0.16 of 0.41 ms at 50 instructions per value): one loop per record format, the rare conditions - a COUNT(*) that
		 * does not fit its field - tested once per digit on the largest COUNT seen, not per value. */
		const uint32_t per_wave = T / (LW_THREADS / 64), words_per_wave = per_wave / 8;
		uint32_t mine = 0, sum_cl = 0, sum_cr = 0;
		for (uint32_t wi = lane; wi < words_per_wave; wi += MDB_WAVE) {
			const uint32_t cw = s_cl[wave * words_per_wave + wi];
			mine += (uint32_t)__popc((cw | (cw >> 1) | (cw >> 2) | (cw >> 3)) & 0x11111111u);
			sum_cl += (((cw & 0x0F0F0F0Fu) + ((cw >> 4) & 0x0F0F0F0Fu)) * 0x01010101u) >> 24;	/* (eight nibbles of at most 15: their sum fits a byte) */
		}
		{
			uint32_t t = mine;
#pragma unroll
			for (int o = 32; o; o >>= 1)
				t += (uint32_t)__shfl_xor((int)t, o, MDB_WAVE);
			if (lane == 0)
				s_red32[wave] = t;
		}
		lw12_barrier();
		if (threadIdx.x == 0) {
			uint32_t total = 0;
#pragma unroll
			for (int w = 0; w < LW_THREADS / 64; w++)
				total += s_red32[w];
			uint32_t nb = 0xFFFFFFFFu;
			if (total && DN) {	/* (no list: the groups are the bits that stay set) */
				atomicAdd(a.rec_valid, total);
				nb = 0u;
			} else if (total) {
				nb = atomicAdd(a.rec_count, total);
				if (nb + total > a.rec_cap) {
					mdb_raise(a.status, 8u);
					nb = 0xFFFFFFFFu;
				} else {
					atomicAdd(a.rec_valid, total);
				}
			}
			s_base = nb;
		}
		/* registers are free: the next digit's right words, in flight while this one's groups are written */
		if (next < run_leaves)
			fetch_all(next, buf ^ 1u, 2);
		lw12_barrier();
		const uint32_t base = s_base;
		uint32_t run = 0;	/* groups of the waves before this one, then of this wave so far (uniform) */
		{
			uint32_t t = lane < wave ? s_red32[lane] : 0u;	/* (at most 16 waves) */
#pragma unroll
			for (int o = 32; o; o >>= 1)
				t += (uint32_t)__shfl_xor((int)t, o, MDB_WAVE);
			run = base + t;		/* (base = ~0: nothing is written) */
		}
		uint32_t el = lane;
		asm volatile("" : "+v"(el));	/* (addresses worked out now: a reload from scratch would wait for the words just asked for) */
		const uint32_t nib = (el & 7u) * 4u, s_begin = wave * per_wave + el;
		uint32_t cmax = 0, jsum = 0, clmax = 0;
		/* FMT 0: 8-byte records, 1: 4-byte records, 2: keyed 8-byte records */
#define LW12_EMIT(FMT)                                                                                                                  \
		for (uint32_t s0 = 0; s0 < per_wave; s0 += 4u * MDB_WAVE) {	/* (per_wave >= 256: a digit has 2^12 values at least) */   \
			uint32_t fc4[4], cw4[4];                                                                                       \
			_Pragma("unroll") for (int e = 0; e < 4; e++) {                                                                \
				const uint32_t sl = s_begin + s0 + (uint32_t)e * MDB_WAVE;                                             \
				fc4[e] = s_fc[sl];                                                                                     \
				cw4[e] = s_cl[sl >> 3];                                                                                \
			}                                                                                                              \
			_Pragma("unroll") for (int e = 0; e < 4; e++) {                                                                \
				const uint32_t fc = fc4[e], cl = (cw4[e] >> nib) & 15u, cr = fc >> 27;                                 \
				sum_cr += cr;                                                                                          \
				const uint64_t m = __ballot(cl != 0u);                                                                 \
				if (!m)                                                                                                \
					continue;                                                                                      \
				const uint32_t pos = run + (uint32_t)__popcll(m & mdb_lanemask_lt());                                  \
				run += (uint32_t)__popcll(m);                                                                          \
				if (cl && base != 0xFFFFFFFFu) {                                                                       \
					const uint32_t first = fc & 0x07FFFFFFu, c = cl * cr;                                          \
					jsum += c;                                                                                     \
					cmax = c > cmax ? c : cmax;                                                                    \
					clmax = cl > clmax ? cl : clmax;                                                               \
					last_first = first > last_first ? first : last_first;                                          \
					if (FMT == 1)                                                                           \
						reinterpret_cast<uint32_t *>(a.rec)[pos] = (first << (32 - a.kbits)) | c;              \
					else if (FMT == 0)                                                                             \
						a.rec[pos] = ((unsigned long long)first << (64 - a.kbits)) | c;                        \
					else                                                                                           \
						a.rec[pos] = ((unsigned long long)first << (64 - a.kbits)) |                           \
							     ((unsigned long long)((leaf << rem) | (s_begin + s0 + (uint32_t)e * MDB_WAVE)) << a.keyed_cbits) | c; \
				}                                                                                                      \
			}                                                                                                              \
		}
		if (DN) {
			/* no list to write, so no position to work out: a lane takes FOUR consecutive values - one 16-byte read of the table, one word of
			 * 4-bit fields - instead of one value in 64 (two LDS reads and a ballot a value, round 6: the pass 0.1 ms of the kernel's 0.35) */
			if (base != 0xFFFFFFFFu)
				for (uint32_t q = el; q < per_wave / 4u; q += MDB_WAVE) {
					const uint32_t sl0 = wave * per_wave + 4u * q;
					const uint4 fq = *reinterpret_cast<const uint4 *>(s_fc + sl0);
					const uint32_t cw = s_cl[sl0 >> 3] >> (16u * (q & 1u)), f[4] = { fq.x, fq.y, fq.z, fq.w };
					bool odd = false;
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const uint32_t cl = (cw >> (4 * e)) & 15u, cr = f[e] >> 27, c = cl * cr, first = cl ? f[e] & 0x07FFFFFFu : 0u;
						sum_cr += cr;
						jsum += c;
						cmax = c > cmax ? c : cmax;
						clmax = cl > clmax ? cl : clmax;
						last_first = first > last_first ? first : last_first;
						odd = odd || c > 1u;
					}
					if (!__ballot(odd))
						continue;
#pragma unroll
					for (int e = 0; e < 4; e++) {	/* (groups whose COUNT is not 1: the exceptions - rare in this form) */
						const uint32_t c = ((cw >> (4 * e)) & 15u) * (f[e] >> 27);
						const uint64_t me = __ballot(c > 1u);
						if (!me)
							continue;
						uint32_t eb = 0;
						if (lane == (uint32_t)__ffsll((long long)me) - 1u)
							eb = atomicAdd(&a.dn_cnt[1], (uint32_t)__popcll(me));
						eb = (uint32_t)__shfl((int)eb, __ffsll((long long)me) - 1, MDB_WAVE);
						if (c > 1u) {
							const uint32_t ep = eb + (uint32_t)__popcll(me & mdb_lanemask_lt());
							if (!a.dn_exc)
								;	/* (the pilot: counted only) */
							else if (ep < a.dn_exc_cap)
								a.dn_exc[ep] = ((unsigned long long)(f[e] & 0x07FFFFFFu) << 32) | c;
							else
								mdb_raise(a.status, 131072u);
						}
					}
				}
			else
				for (uint32_t q = el; q < per_wave / 4u; q += MDB_WAVE) {	/* (no group in the digit: the right rows are still checked) */
					const uint4 fq = *reinterpret_cast<const uint4 *>(s_fc + wave * per_wave + 4u * q);
					sum_cr += (fq.x >> 27) + (fq.y >> 27) + (fq.z >> 27) + (fq.w >> 27);
				}
		} else if (a.keyed_cbits) {
			LW12_EMIT(2)
		} else if (a.rec32) {
			LW12_EMIT(1)
		} else {
			LW12_EMIT(0)
		}
#undef LW12_EMIT
		joined += jsum;
		if (clmax > 1u)
			mdb_raise(a.status, GC_ST_LEFT_DUPS);
		if (cmax && !DN) {	/* (a COUNT(*) of at most 15 * 31) */
			if (a.keyed_cbits && (cmax >> a.keyed_cbits))
				mdb_raise(a.status, 256u);	/* COUNT(*) does not fit a keyed record: redone with plain records */
			if (!a.keyed_cbits && (cmax >> (32 - (a.kbits < 32 ? a.kbits : 31))))
				mdb_raise(a.status, 16u | (a.rec32 ? 512u : 0u));	/* ... a 4-byte record (written on a remembered verdict: redone with 8-byte ones) */
		}
		const unsigned long long sums = ((unsigned long long)sum_cl << 32) | (NX ? 0u : sum_cr);	/* low half: right rows counted, high half: left rows counted */
		/* a count field that overflowed carried into its neighbour (or out of the word): the fields then sum to less than was added */
		const unsigned long long want = ((unsigned long long)adds << 32) | radds;
		unsigned long long diff = sums - want;
#pragma unroll
		for (int o = 32; o; o >>= 1)
			diff += __shfl_down(diff, o, MDB_WAVE);
		lw12_barrier();	/* (s_red is free; every wave has read its part of the tables: they may be cleared) */
		if (lane == 0)
			s_red[wave] = diff;
		lw12_barrier();
		diff = 0ull;
#pragma unroll
		for (int w = 0; w < LW_THREADS / 64; w++)
			diff += s_red[w];
		if (diff != 0ull || prod_bad) {
			bad = true;
			break;
		}
	}
	if (bad && threadIdx.x == 0)
		mdb_raise(a.status, 1024u);
	joined = lw_block_sum(joined, s_red);
	if (threadIdx.x == 0 && joined)
		atomicAdd(a.joined, joined);
	if (DN) {
		const unsigned long long cl_all = lw_block_sum((unsigned long long)cleared, s_red);
		if (threadIdx.x == 0 && cl_all)
			atomicAdd(&a.dn_cnt[0], (uint32_t)cl_all);
	}
#pragma unroll
	for (int o = 32; o; o >>= 1) {
		const uint32_t other = (uint32_t)__shfl_xor((int)last_first, o, MDB_WAVE);
		last_first = other > last_first ? other : last_first;
	}
	__syncthreads();	/* (s_red is free again) */
	if (lane == 0)
		s_red[wave] = last_first;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t m = 0;
#pragma unroll
		for (int w = 0; w < LW_THREADS / 64; w++)
			m = (uint32_t)s_red[w] > m ? (uint32_t)s_red[w] : m;
		if (m)
			atomicMax(a.status + 9, m);
	}
}

/* ------------------------------------------------------------------ k_leaf_wide with k_leaf_wide12's table (round 4)
 *
 * ONE partition level (512 digits of up to 2^14 values), joins whose right table travels as 2-byte words: the table of k_leaf_wide12 -
 * 4 bytes per key value (right rows in 5 bits above the first left row in 27) + 4 bits of left rows - is 72 KiB where k_leaf_wide's
 * 8 bytes per value are 128: TWO workgroups per CU, one streaming while the other clears or emits, and all 512 digits of the benchmark's
 * variant D resident at once instead of in two rounds.  More than 31 right or 15 left rows of one key are noticed (checksums) and
 * reported - flag 4096: the caller launches k_leaf_wide on the same partitioned tables and remembers the columns. */
__global__ __launch_bounds__(LW_THREADS, 8 /* waves per SIMD: two workgroups per CU */) void k_leaf_wide4(gc_args a, uint32_t rem, uint32_t shift, uint32_t nsub)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lw_lds[];
	__shared__ unsigned long long s_red[LW_THREADS / 64];
	__shared__ uint32_t s_red32[LW_THREADS / 64];
	__shared__ uint32_t s_base;
	const uint32_t T = 1u << rem, mask = T - 1u, leaf = blockIdx.x, wave = threadIdx.x >> 6, lane = mdb_lane();
	uint32_t *const s_fc = lw_lds;			/* right rows << 27 | first left row */
	uint32_t *const s_cl = lw_lds + T;		/* left rows per key (only of keys that have right rows), 4 bits each */
	for (uint32_t s = threadIdx.x; s < T; s += LW_THREADS)
		s_fc[s] = 0x07FFFFFFu;
	for (uint32_t s = threadIdx.x; s < T / 8; s += LW_THREADS)
		s_cl[s] = 0u;
	uint32_t *const s_rg = s_cl + (T / 8 < 16u ? 16u : T / 8);	/* ranged emit (a.rg_rec): groups per range of first row ids, then their places */
	if (a.rg_rec)
		for (uint32_t r = threadIdx.x; r < a.rg_n; r += LW_THREADS)
			s_rg[r] = 0u;
	__syncthreads();
	uint32_t rows_r = 0;
	{
		const uint16_t *const hv_r16 = reinterpret_cast<const uint16_t *>(a.hv_r);
		for (uint32_t sub = 0; sub < nsub; sub++) {
			const uint32_t c0 = a.cnt_r[sub * a.nleaves + leaf], c = c0 < a.cap_r ? c0 : a.cap_r;
			const uint16_t *const src = hv_r16 + (size_t)(leaf * nsub + sub) * a.cap_r;
			rows_r += c;
			for (uint32_t j0 = 0; j0 < c; j0 += 8u * LW_THREADS * LW_UNROLL) {	/* uniform trip count */
				uint4 v[LW_UNROLL];
#pragma unroll
				for (int u = 0; u < LW_UNROLL; u++) {
					const uint32_t j = j0 + 8u * ((uint32_t)u * LW_THREADS + threadIdx.x);
					v[u] = make_uint4(0u, 0u, 0u, 0u);
					if (j < c)
						v[u] = lw_load_nt(src + j);
				}
#pragma unroll
				for (int u = 0; u < LW_UNROLL; u++) {
					const uint32_t j = j0 + 8u * ((uint32_t)u * LW_THREADS + threadIdx.x);
					const uint32_t w[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
					if (j + 8u <= c) {
#pragma unroll
						for (int k = 0; k < 8; k++)
							atomicAdd(&s_fc[(w[k >> 1] >> (16 * (k & 1))) & mask], 1u << 27);
					} else {
#pragma unroll
						for (int k = 0; k < 8; k++)
							if (j + k < c)
								atomicAdd(&s_fc[(w[k >> 1] >> (16 * (k & 1))) & mask], 1u << 27);
					}
				}
			}
		}
	}
	__syncthreads();
	uint32_t adds = 0;
	for (uint32_t sub = 0; sub < nsub; sub++) {
		const uint32_t c0 = a.cnt_l[sub * a.nleaves + leaf], c = c0 < a.cap_l ? c0 : a.cap_l;
		const uint64_t *const src = a.hv_l + (size_t)(leaf * nsub + sub) * a.cap_l;
		for (uint32_t i0 = 0; i0 < c; i0 += 2u * LW_THREADS * LW_UNROLL) {
			ulonglong2 v[LW_UNROLL];
#pragma unroll
			for (int u = 0; u < LW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * LW_THREADS + threadIdx.x);
				v[u] = make_ulonglong2(0ull, 0ull);
				if (i < c)
					v[u] = *reinterpret_cast<const ulonglong2 *>(src + i);
			}
#pragma unroll
			for (int u = 0; u < LW_UNROLL; u++) {
				const uint32_t i = i0 + 2u * ((uint32_t)u * LW_THREADS + threadIdx.x);
				const unsigned long long w[2] = { v[u].x, v[u].y };
#pragma unroll
				for (int k = 0; k < 2; k++)
					if (i + k < c) {
						const uint32_t idx = ((uint32_t)(w[k] >> 32) >> shift) & mask, fc = s_fc[idx];
						if (fc >> 27) {		/* (the right rows are all counted: the top bits are final, the minimum is over the row id) */
							atomicMin(&s_fc[idx], (fc & 0xF8000000u) | (uint32_t)w[k]);
							atomicAdd(&s_cl[idx >> 3], 1u << ((idx & 7u) * 4u));
							adds++;
						}
					}
			}
		}
	}
	__syncthreads();
	/* groups: wave w owns the values [w * T / 16, (w + 1) * T / 16) - see k_leaf_wide12 */
	const uint32_t per_wave = T / (LW_THREADS / 64), words_per_wave = per_wave / 8;
	uint32_t mine = 0, sum_cl = 0, sum_cr = 0;
	for (uint32_t wi = lane; wi < words_per_wave; wi += MDB_WAVE) {
		const uint32_t cw = s_cl[wave * words_per_wave + wi];
		mine += (uint32_t)__popc((cw | (cw >> 1) | (cw >> 2) | (cw >> 3)) & 0x11111111u);
		sum_cl += (((cw & 0x0F0F0F0Fu) + ((cw >> 4) & 0x0F0F0F0Fu)) * 0x01010101u) >> 24;
	}
	{
		uint32_t t = mine;
#pragma unroll
		for (int o = 32; o; o >>= 1)
			t += (uint32_t)__shfl_xor((int)t, o, MDB_WAVE);
		if (lane == 0)
			s_red32[wave] = t;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
#pragma unroll
		for (int w = 0; w < LW_THREADS / 64; w++)
			total += s_red32[w];
		uint32_t nb = 0xFFFFFFFFu;
		if (total && a.rg_rec) {	/* (no list: the records go to their ranges) */
			atomicAdd(a.rec_valid, total);
			nb = 0u;
		} else if (total) {
			nb = atomicAdd(a.rec_count, total);
			if (nb + total > a.rec_cap) {
				mdb_raise(a.status, 8u);
				nb = 0xFFFFFFFFu;
			} else {
				atomicAdd(a.rec_valid, total);
			}
		}
		s_base = nb;
	}
	__syncthreads();
	const uint32_t base = s_base;
	uint32_t run = 0;
	{
		uint32_t t = lane < wave ? s_red32[lane] : 0u;
#pragma unroll
		for (int o = 32; o; o >>= 1)
			t += (uint32_t)__shfl_xor((int)t, o, MDB_WAVE);
		run = base + t;
	}
	const uint32_t nib = (lane & 7u) * 4u, s_begin = wave * per_wave + lane;
	uint32_t cmax = 0, jsum = 0, last_first = 0, clmax = 0;
	if (a.rg_rec && base != 0xFFFFFFFFu) {
		/* the ordering kernel's ranges of 2^rg_shift first row ids are filled here: a digit's groups per range are counted, a place for
		 * them reserved with one global atomic per (digit, range), and every record written there - ~8 records side by side per run */
		for (uint32_t s0 = 0; s0 < per_wave; s0 += MDB_WAVE) {
			const uint32_t sl = s_begin + s0;
			if (s0 + lane < per_wave && ((s_cl[sl >> 3] >> nib) & 15u))
				atomicAdd(&s_rg[(s_fc[sl] & 0x07FFFFFFu) >> a.rg_shift], 1u);
		}
		__syncthreads();
		for (uint32_t r = threadIdx.x; r < a.rg_n; r += LW_THREADS) {
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
		for (uint32_t s0 = 0; s0 < per_wave; s0 += MDB_WAVE) {
			const uint32_t sl = s_begin + s0;
			const bool live = s0 + lane < per_wave;
			const uint32_t fc = live ? s_fc[sl] : 0u, cl = live ? (s_cl[sl >> 3] >> nib) & 15u : 0u, cr = fc >> 27;
			sum_cr += cr;
			if (!cl)
				continue;
			const uint32_t first = fc & 0x07FFFFFFu, c = cl * cr, r = first >> a.rg_shift;
			jsum += c;
			cmax = c > cmax ? c : cmax;
			clmax = cl > clmax ? cl : clmax;
			last_first = first > last_first ? first : last_first;
			const uint32_t at = atomicAdd(&s_rg[r], 1u);
			if (at < a.rg_cap)
				a.rg_rec[(size_t)r * a.rg_cap + at] = ((unsigned long long)first << (64 - a.kbits)) |
								       (a.keyed_cbits ? ((unsigned long long)((leaf << rem) | sl) << a.keyed_cbits) : 0ull) | c;
		}
	}
	for (uint32_t s0 = 0; s0 < ((a.rg_rec && base != 0xFFFFFFFFu) ? 0u : per_wave); s0 += MDB_WAVE) {	/* (per_wave >= 4: tables of 64 values at least; lanes beyond per_wave idle) */
		const uint32_t sl = s_begin + s0;
		const bool live = s0 + lane < per_wave;
		const uint32_t fc = live ? s_fc[sl] : 0u, cl = live ? (s_cl[sl >> 3] >> nib) & 15u : 0u, cr = fc >> 27;
		sum_cr += cr;
		const uint64_t m = __ballot(cl != 0u);
		if (!m)
			continue;
		const uint32_t pos = run + (uint32_t)__popcll(m & mdb_lanemask_lt());
		run += (uint32_t)__popcll(m);
		if (cl && base != 0xFFFFFFFFu) {
			const uint32_t first = fc & 0x07FFFFFFu, c = cl * cr;
			jsum += c;
			cmax = c > cmax ? c : cmax;
			clmax = cl > clmax ? cl : clmax;
			last_first = first > last_first ? first : last_first;
			if (a.keyed_cbits)
				a.rec[pos] = ((unsigned long long)first << (64 - a.kbits)) | ((unsigned long long)((leaf << rem) | sl) << a.keyed_cbits) | c;
			else
				a.rec[pos] = ((unsigned long long)first << (64 - a.kbits)) | c;
		}
	}
	if (clmax > 1u)
		mdb_raise(a.status, GC_ST_LEFT_DUPS);
	if (cmax) {
		if (a.keyed_cbits && (cmax >> a.keyed_cbits))
			mdb_raise(a.status, 256u);	/* COUNT(*) does not fit a keyed record: redone with plain records */
		if (!a.keyed_cbits && (cmax >> (32 - (a.kbits < 32 ? a.kbits : 31))))
			mdb_raise(a.status, 16u);	/* ... a 4-byte record */
	}
	/* a count field that overflowed carried into its neighbour (or out of the word): the fields then sum to less than was added */
	const unsigned long long sums = ((unsigned long long)sum_cl << 32) | sum_cr, want = (unsigned long long)adds << 32;
	const unsigned long long diff = lw_block_sum(sums - want, s_red);
	if (diff != (unsigned long long)rows_r) {
		if (threadIdx.x == 0)
			mdb_raise(a.status, 4096u);
		return;
	}
	const unsigned long long joined = lw_block_sum((unsigned long long)jsum, s_red);
	if (threadIdx.x == 0 && joined)
		atomicAdd(a.joined, joined);
#pragma unroll
	for (int o = 32; o; o >>= 1) {
		const uint32_t other = (uint32_t)__shfl_xor((int)last_first, o, MDB_WAVE);
		last_first = other > last_first ? other : last_first;
	}
	__syncthreads();	/* (s_red is free again) */
	if (lane == 0)
		s_red[wave] = last_first;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t m = 0;
#pragma unroll
		for (int w = 0; w < LW_THREADS / 64; w++)
			m = (uint32_t)s_red[w] > m ? (uint32_t)s_red[w] : m;
		if (m)
			atomicMax(a.status + 9, m);
	}
}

/* ------------------------------------------------------------------ launchers (mdb_dev_join.hip: gc_begin / gc_finish) */

int leaf_wide_launch(mdb_dev_ctx *ctx, const gc_args &a, uint32_t nleaves, uint32_t rem, uint32_t shift, uint32_t nsub, bool has_r, bool r16)
{
	const size_t lds = ((size_t)(has_r ? 8 : 6) << rem);
	if (has_r && r16) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_join_wide", (k_leaf_wide<true, true>), nleaves, LW_THREADS, lds, a, rem, shift, nsub);
	} else if (has_r) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_join_wide", (k_leaf_wide<true>), nleaves, LW_THREADS, lds, a, rem, shift, nsub);
	} else {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_group_wide", (k_leaf_wide<false>), nleaves, LW_THREADS, lds, a, rem, shift, nsub);
	}
	return MIDORIDB_OK;
}

int leaf_wide4_launch(mdb_dev_ctx *ctx, const gc_args &a, uint32_t nleaves, uint32_t rem, uint32_t shift, uint32_t nsub)
{
	const size_t lds4 = ((size_t)4 << rem) + (((size_t)1 << rem) / 2 < 64 ? 64 : ((size_t)1 << rem) / 2) + (a.rg_rec ? (size_t)a.rg_n * 4 : 0);
	MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
	MDB_LAUNCH_LDS(ctx, "leaf_join_wide4", k_leaf_wide4, nleaves, LW_THREADS, lds4, a, rem, shift, nsub);
	return MIDORIDB_OK;
}

/* a digit's words must fit k_leaf_wide12's registers: 8 sub-regions of at most LW12_LB (LW12_RB) sixteen-byte chunks per thread */
bool leaf_wide12_fits(const mdb_dev_ctx *ctx, uint64_t n_l, uint64_t n_r, uint64_t n_x /* a further right table's rows, 0: none */)
{
	return (uint64_t)mdb_scatter4096_cap(ctx, n_l, true) * LW12_NSUB <= (uint64_t)LW12_LB * LW_THREADS * 4u &&
	       (uint64_t)mdb_scatter4096_cap(ctx, n_r, false) * LW12_NSUB <= (uint64_t)LW12_RB * LW_THREADS * 8u &&
	       (!n_x || (uint64_t)mdb_scatter4096_cap(ctx, n_x, false) * LW12_NSUB <= (uint64_t)LW12_RB * LW_THREADS * 8u);
}

/* one 1024-thread workgroup per CU walks the digits (the next digit's words are loaded while the current one's groups leave) */
int leaf_wide12_launch(mdb_dev_ctx *ctx, const gc_args &a, uint32_t nleaves, uint32_t rem, uint32_t nsub, int nextra)
{
	const size_t lds = ((size_t)4 << rem) + ((size_t)1 << rem) / 2;
	uint32_t wgrid = nleaves < (uint32_t)ctx->num_cus ? nleaves : (uint32_t)ctx->num_cus;
	if (a.dn_pilot && a.dn_pilot < wgrid)
		wgrid = a.dn_pilot;
	if (nextra && (a.dn_bits || a.dn_pilot)) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide12<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_join_wide12_bits", (k_leaf_wide12<1, true>), wgrid, LW_THREADS, lds, a, rem, nsub);
	} else if (nextra) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide12<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_join_wide12", k_leaf_wide12<1>, wgrid, LW_THREADS, lds, a, rem, nsub);
	} else if (a.dn_bits || a.dn_pilot) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide12<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_join_wide12_bits", (k_leaf_wide12<0, true>), wgrid, LW_THREADS, lds, a, rem, nsub);
	} else {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leaf_wide12<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "leaf_join_wide12", k_leaf_wide12<0>, wgrid, LW_THREADS, lds, a, rem, nsub);
	}
	return MIDORIDB_OK;
}
