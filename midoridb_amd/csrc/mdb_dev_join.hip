/*
 * mdb_dev_join.hip - the fused INNER JOIN + GROUP BY join key + COUNT(*) operator (north-star query): per-leaf LDS kernels
 * (hashed tables, direct-address tables of the compact narrow form, hot keys; the kernels that join a whole first-level digit
 * per workgroup live in mdb_dev_leaf_wide.hip), its planning (key sample, key forms, pruning, which leaf form), retries, the
 * split form for the multi-GPU exchange and the N-way form, and the drivers of
 *
 *   mdb_dev_join_group_count[_multi | _begin / _finish | _i32]
 *   mdb_dev_group_count       GROUP BY key + COUNT(*) over one key column (fast paths: mdb_dev_groupby.hip)
 *
 * The ordering of the group records lives in mdb_dev_order.hip, the materialising join (mdb_dev_join_pairs) in
 * mdb_dev_pairs.hip, the 4096-digit pass in mdb_dev_shard.hip; what the files share is mdb_dev_join_internal.h.
 *
 * After mdb_partition_table() every leaf holds all rows (of both tables) whose hashed key shares
 * the same top bits (in unspecified order; orders are restored from the row ids).  Persistent
 * workgroups walk the leaves: each builds an open-addressing hash table in LDS keyed by the 64-bit
 * hashed key (fmix64 is a bijection, so equal hash <=> equal key; the value 0 = "empty slot", the
 * single key that hashes to 0 is kept in a dedicated side slot) from the side with fewer rows, then
 * streams the other side through it.  All counters are LDS atomics; global memory is read once,
 * coalesced, and written once.  Groups leave as 64-bit records (first row id, COUNT) that a radix
 * sort + k_order_leaf put into the reference's order; hot keys take the k_hot_* path; joins whose
 * right keys are unique emit one record per pair (k_leaf_pairs_unique) through the same ordering.
 *
 * Reference semantics reproduced (file:line = reference src/engine/executor_select.c):
 *   - a NULL join key matches nothing (:557-579)            -> NULL keys were dropped at level 0
 *   - N:M duplicates: every pair is a joined row (:1096-1141) -> COUNT = cntL * cntR per key
 *   - GROUP BY keeps the first row of each group, in order (:1542-1583)
 *                                                            -> groups ordered by first L position
 *   - join output is left-major / right-minor (:1096-1141)   -> pairs ordered by (pos_l, pos_r)
 */
#include "mdb_dev_join_internal.h"
#include "mdb_dev_rowjoin.h"

__global__ void k_gather_i32(const int32_t *__restrict__ keys, const uint32_t *__restrict__ sel, uint64_t n, int64_t *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		out[i] = (int64_t)keys[sel[i]];
}

/* ------------------------------------------------------------------ fused join + group count */


/*
 * Persistent form: gridDim.x workgroups (2 per CU) walk the leaves with stride gridDim.x.  The first
 * LEAF_BATCH keys per thread of BOTH sides of the next leaf are requested before the current leaf's
 * results are emitted and the tables re-initialised, so the HBM round trip overlaps LDS work instead of
 * being paid three times per leaf (offsets -> left keys -> right keys), which made the one-leaf-per-
 * workgroup version latency-bound (70 % of wave time in s_waitcnt, profiles/r01).
 */
struct gc_batch {
	uint64_t hv_l[LEAF_BATCH];
	uint32_t rid_l[LEAF_BATCH];
	uint64_t hv_r[LEAF_BATCH];
	uint32_t h32_r[LEAF_BATCH];	/* narrow form: the right side's 4-byte words */
};


/* the same in two steps, so that the words can be requested one whole leaf before they are needed */
__device__ static inline uint2 gc_leaf_raw(const uint32_t *off, const uint32_t *cnt, uint32_t cap, uint32_t leaf)
{
	/* The index is hidden from the uniformity analysis: a load the compiler knows to be wave-uniform is moved into
	 * scalar registers on arrival, i.e. waited for on the spot.  gc_leaf_decode() makes the words scalar again, one
	 * leaf later. */
	asm volatile("" : "+v"(leaf));
	uint2 w;
	if (cap) {
		w.x = cnt[leaf];
		w.y = 0;
	} else {
		w.x = off[leaf];
		w.y = off[leaf + 1];
	}
	return w;
}

__device__ static inline void gc_leaf_decode(uint2 w, uint32_t cap, uint32_t leaf, uint32_t *b, uint32_t *e)
{
	w.x = __builtin_amdgcn_readfirstlane(w.x);
	w.y = __builtin_amdgcn_readfirstlane(w.y);
	if (cap) {
		*b = leaf * cap;
		*e = *b + (w.x < cap ? w.x : cap);
	} else {
		*b = w.x;
		*e = w.y;
	}
}

/* NW: 0 = 64-bit hashes, 1 = narrow words, 2 = whatever gc_args.narrow says (the rarely taken hot-key kernels) */
template <int NW>
__device__ static inline bool gc_is_narrow(const gc_args &a)
{
	return NW == 2 ? a.narrow != 0 : NW == 1;
}

/* The batch keeps the words as loaded (a conversion at load time would make the prefetch wait for its own loads);
 * they are decoded where they are consumed. */
template <bool IS_L, int NW>
__device__ static inline uint64_t gc_batch_hv(const gc_args &a, const gc_batch &b, int u)
{
	if (!gc_is_narrow<NW>(a))
		return IS_L ? b.hv_l[u] : b.hv_r[u];
	if (IS_L)
		return gc_narrow_hv(b.hv_l[u]);
	return ((uint64_t)b.h32_r[u] << 32) | b.h32_r[u];	/* right side: 4-byte words */
}

template <int NW>
__device__ static inline uint32_t gc_batch_rid(const gc_args &a, const gc_batch &b, int u)
{
	return gc_is_narrow<NW>(a) ? (uint32_t)b.hv_l[u] : b.rid_l[u];
}

template <int NW>
__device__ static inline void gc_load_l(const gc_args &a, uint32_t base, uint32_t end, gc_batch &b)
{
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		const uint32_t i = base + (uint32_t)u * GC_THREADS + threadIdx.x;
		b.hv_l[u] = 0;
		b.rid_l[u] = 0;
		if (i < end) {
			b.hv_l[u] = a.hv_l[i];
			if (!gc_is_narrow<NW>(a))
				b.rid_l[u] = a.rid_l[i];
		}
	}
}

template <int NW>
__device__ static inline void gc_load_r(const gc_args &a, uint32_t base, uint32_t end, gc_batch &b)
{
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		const uint32_t j = base + (uint32_t)u * GC_THREADS + threadIdx.x;
		if (gc_is_narrow<NW>(a)) {
			b.h32_r[u] = 0;
			if (j < end)
				b.h32_r[u] = reinterpret_cast<const uint32_t *>(a.hv_r)[j];
		} else {
			b.hv_r[u] = j < end ? a.hv_r[j] : 0;
		}
	}
}

template <bool HAS_R, int NW>
__device__ static inline void gc_prefetch(const gc_args &a, uint32_t l0, uint32_t l1, uint32_t r0, uint32_t r1, gc_batch &b)
{
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		const uint32_t i = l0 + (uint32_t)u * GC_THREADS + threadIdx.x;
		b.hv_l[u] = 0;
		b.rid_l[u] = 0;
		if (i < l1) {
			b.hv_l[u] = a.hv_l[i];
			if (!gc_is_narrow<NW>(a))
				b.rid_l[u] = a.rid_l[i];
		}
		if (HAS_R) {
			const uint32_t j = r0 + (uint32_t)u * GC_THREADS + threadIdx.x;
			if (gc_is_narrow<NW>(a)) {
				b.h32_r[u] = 0;
				if (j < r1)
					b.h32_r[u] = reinterpret_cast<const uint32_t *>(a.hv_r)[j];
			} else {
				b.hv_r[u] = j < r1 ? a.hv_r[j] : 0;
			}
		}
	}
}

/* One side of a leaf goes INTO the table (insert-or-find, count, first left row id) ... */
__device__ static inline uint32_t gc_wave_min_u32(uint32_t v)
{
#pragma unroll
	for (int d = 1; d < MDB_WAVE; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_xor((int)v, d, MDB_WAVE);
		v = o < v ? o : v;
	}
	return v;
}

/* MERGE (the plain GROUP BY instance): lanes of a wave that hold the same value are merged before the table is touched -
 * one insert + one counter update for the group instead of one per row on the same LDS address (GROUP BY over 10^4 - 10^6
 * distinct values: leaves with a few values and thousands of rows).  Given up after the first round that finds no
 * duplicates of the leader, so columns of distinct values pay one ballot. */
template <bool IS_L, int NW, bool MERGE = false>
__device__ static inline void gc_build_side(const gc_args &a, unsigned long long *s_key, unsigned long long *s_cnt, uint32_t *s_first,
					    gc_batch &b, uint32_t x0, uint32_t x1, uint32_t own[LEAF_BATCH])
{
	for (uint32_t base = x0; base < x1; base += GC_THREADS * LEAF_BATCH) {
		if (base != x0) {
			if (IS_L)
				gc_load_l<NW>(a, base, x1, b);
			else
				gc_load_r<NW>(a, base, x1, b);
		}
		/* first probe of every key of the batch issued back to back (the CAS round trips overlap);
		 * only keys whose first slot is taken by another key walk on, one after the other */
		uint32_t s_[LEAF_BATCH], step_[LEAF_BATCH];
		unsigned long long old_[LEAF_BATCH];
		bool act_[LEAF_BATCH];
		uint32_t mult_[LEAF_BATCH], rid_[LEAF_BATCH];
		bool drop_[LEAF_BATCH];
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint32_t i = base + (uint32_t)u * GC_THREADS + threadIdx.x;
			const uint64_t hv = gc_batch_hv<IS_L, NW>(a, b, u);
			mult_[u] = 1;
			rid_[u] = IS_L ? gc_batch_rid<NW>(a, b, u) : 0u;
			drop_[u] = false;
			if (MERGE && (a.merge_all || x1 - x0 > GC_THREADS * LEAF_BATCH)) {	/* (uniform) a leaf beyond one register batch holds
												 * duplicates for sure; merge_all: the sample says most do */
				const bool in = i < x1;
				uint64_t pending = __ballot(in);
				for (int round = 0; round < 8 && pending; round++) {
					const int leader = __ffsll((long long)pending) - 1;
					const uint32_t llo = (uint32_t)__shfl((int)(uint32_t)hv, leader, MDB_WAVE);
					const uint32_t lhi = (uint32_t)__shfl((int)(uint32_t)(hv >> 32), leader, MDB_WAVE);
					const bool same = in && !drop_[u] && (uint32_t)hv == llo && (uint32_t)(hv >> 32) == lhi;
					const uint64_t grp = __ballot(same);
					const uint32_t cnt = (uint32_t)__popcll(grp);
					if (cnt < 8)
						break;		/* few lanes per value: the LDS atomics cope, and a round costs more than it saves */
					const uint32_t rmin = IS_L ? gc_wave_min_u32(same ? rid_[u] : 0xFFFFFFFFu) : 0u;
					if (same) {
						if ((int)mdb_lane() == leader) {
							mult_[u] = cnt;
							rid_[u] = rmin;
						} else {
							drop_[u] = true;
						}
					}
					pending &= ~grp;
				}
			}
			if (base == x0)
				own[u] = 0xFFFFFFFFu;
			act_[u] = i < x1 && hv != 0 && !drop_[u];
			s_[u] = leaf_slot(hv, GC_SLOTS);
			step_[u] = leaf_step(hv, GC_SLOTS);
			old_[u] = act_[u] ? atomicCAS(&s_key[s_[u]], 0ull, (unsigned long long)hv) : 0ull;
		}
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint32_t i = base + (uint32_t)u * GC_THREADS + threadIdx.x;
			const uint64_t hv = gc_batch_hv<IS_L, NW>(a, b, u);
			if (i >= x1 || drop_[u])
				continue;
			uint32_t s = GC_SLOTS;		/* the key whose hash is 0 has the side slot */
			bool created = false;
			if (act_[u]) {
				s = s_[u];
				unsigned long long old = old_[u];
				uint32_t probe = 1;
				while (old != 0ull && old != hv) {
					if (probe++ >= GC_SLOTS) {
						s = 0xFFFFFFFFu;
						break;
					}
					s += step_[u];
					if (s >= GC_SLOTS)
						s -= GC_SLOTS;
					old = atomicCAS(&s_key[s], 0ull, (unsigned long long)hv);
				}
				created = old == 0ull;
			}
			if (s == 0xFFFFFFFFu) {
				mdb_raise(a.status, 1u);
			} else {
				if (IS_L) {
					atomicAdd(&s_cnt[s], (unsigned long long)mult_[u]);
					atomicMin(&s_first[s], rid_[u]);
				} else {
					atomicAdd(&s_cnt[s], (unsigned long long)mult_[u] << 32);
				}
				if (created && base == x0)
					own[u] = s;	/* this thread emits (and clears) the group */
			}
		}
	}
}

/* ... and the other side only LOOKS UP: a key that is not in the table costs one LDS read */
template <bool IS_L, int NW>
__device__ static inline void gc_probe_side(const gc_args &a, const unsigned long long *s_key, unsigned long long *s_cnt, uint32_t *s_first,
					    gc_batch &b, uint32_t x0, uint32_t x1)
{
	for (uint32_t base = x0; base < x1; base += GC_THREADS * LEAF_BATCH) {
		if (base != x0) {
			if (IS_L)
				gc_load_l<NW>(a, base, x1, b);
			else
				gc_load_r<NW>(a, base, x1, b);
		}
		/* same shape as the build: the first slot of every key is read before any is examined */
		uint32_t ps_[LEAF_BATCH], pstep_[LEAF_BATCH];
		unsigned long long cur_[LEAF_BATCH];
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint64_t hv = gc_batch_hv<IS_L, NW>(a, b, u);
			ps_[u] = leaf_slot(hv, GC_SLOTS);
			pstep_[u] = leaf_step(hv, GC_SLOTS);
			cur_[u] = s_key[ps_[u]];
		}
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint32_t j = base + (uint32_t)u * GC_THREADS + threadIdx.x;
			const uint64_t hv = gc_batch_hv<IS_L, NW>(a, b, u);
			if (j >= x1)
				continue;
			uint32_t s = GC_SLOTS;
			if (hv != 0) {
				unsigned long long cur = cur_[u];
				uint32_t probe = 1;
				s = ps_[u];
				while (cur != hv) {
					if (cur == 0ull || probe++ >= GC_SLOTS) {
						s = 0xFFFFFFFFu;
						break;
					}
					s += pstep_[u];
					if (s >= GC_SLOTS)
						s -= GC_SLOTS;
					cur = s_key[s];
				}
			}
			if (s != 0xFFFFFFFFu) {
				if (IS_L) {
					atomicAdd(&s_cnt[s], 1ull);
					atomicMin(&s_first[s], gc_batch_rid<NW>(a, b, u));
				} else {
					atomicAdd(&s_cnt[s], 1ull << 32);
				}
			}
		}
	}
}

/* Hot keys.  A leaf that received tens of thousands of rows holds few distinct keys with very many duplicates
 * (more distinct keys than table slots is an error anyway), and one LDS atomic per row on the same slot serialises:
 * 2*10^7 rows of one key took 77 ms.  For such leaves (GC_HEAVY rows or more on the side at hand) the lanes of a wave
 * first merge their duplicates - leader's key, ballot of the lanes that hold the same key, one table operation for
 * the whole group - so the table sees one update per distinct key per wave instead of one per row. */
#define GC_HEAVY (8u * GC_THREADS * LEAF_BATCH)	/* floor of the hot threshold (gc_args.heavy_l / heavy_r) */

template <bool IS_L, bool INSERT, int NW = 2>
__device__ static inline void gc_side_heavy(const gc_args &a, unsigned long long *s_key, unsigned long long *s_cnt, uint32_t *s_first,
					    gc_batch &b, uint32_t x0, uint32_t x1)
{
	for (uint32_t base = x0; base < x1; base += GC_THREADS * LEAF_BATCH) {
		if (base != x0) {
			if (IS_L)
				gc_load_l<NW>(a, base, x1, b);
			else
				gc_load_r<NW>(a, base, x1, b);
		}
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint32_t i = base + (uint32_t)u * GC_THREADS + threadIdx.x;
			const uint64_t hv = gc_batch_hv<IS_L, NW>(a, b, u);
			const uint32_t rid = IS_L ? gc_batch_rid<NW>(a, b, u) : 0u;
			const bool active = i < x1;
			uint64_t pending = __ballot(active);
			while (pending) {	/* wave-uniform: one round per distinct key among the wave's rows */
				const int leader = __ffsll((long long)pending) - 1;
				const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)hv, leader, MDB_WAVE);
				const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(hv >> 32), leader, MDB_WAVE);
				const uint64_t lhv = ((uint64_t)hi << 32) | lo;
				const bool mine = active && hv == lhv;
				const uint64_t grp = __ballot(mine);
				const uint32_t rmin = IS_L ? gc_wave_min_u32(mine ? rid : 0xFFFFFFFFu) : 0u;
				if ((int)mdb_lane() == leader) {
					uint32_t s = GC_SLOTS;
					if (lhv != 0)
						s = INSERT ? leaf_insert(s_key, GC_SLOTS, lhv) : leaf_find(s_key, GC_SLOTS, lhv);
					if (s == 0xFFFFFFFFu) {
						if (INSERT)
							mdb_raise(a.status, 1u);
					} else if (IS_L) {
						atomicAdd(&s_cnt[s], (unsigned long long)__popcll(grp));
						atomicMin(&s_first[s], rmin);
					} else {
						atomicAdd(&s_cnt[s], (unsigned long long)__popcll(grp) << 32);
					}
				}
				pending &= ~grp;
			}
		}
	}
}

template <bool HAS_R, bool BUILD_R, bool HEAVY, bool NARROW = false>
__global__ __launch_bounds__(GC_THREADS, 8) void k_leaf_group_count(gc_args a)
{
	constexpr int NW = HEAVY ? 2 : (NARROW ? 1 : 0);
	/* The HEAVY instance (single-workgroup fallback of the hot-key path) only works when the plain one met hot leaves. */
	if (HEAVY) {
		const uint32_t st = *(volatile const uint32_t *)a.status;
		if ((st & 2u) || !(st & 64u))
			return;
	}

	/* slot = hashed key (0 = empty) + packed counters (low 32 bits: left rows, high 32 bits: right rows)
	 * + first left row id.  The tables are zeroed once; afterwards the emit pass clears exactly the slots
	 * it reads as occupied, which halves the LDS traffic of a clear-everything-per-leaf loop. */
	__shared__ unsigned long long s_key[GC_SLOTS];
	__shared__ unsigned long long s_cnt[GC_SLOTS + 1];	/* [GC_SLOTS] = the key whose hash is 0 */
	__shared__ uint32_t s_first[GC_SLOTS + 1];
	__shared__ unsigned long long s_sum;
	__shared__ uint32_t s_chunk[4];		/* record list chunk of this workgroup: [0] base [1] used [2] size [3] valid records */

	for (uint32_t s = threadIdx.x; s <= GC_SLOTS; s += GC_THREADS) {
		if (s < GC_SLOTS)
			s_key[s] = 0ull;
		s_cnt[s] = 0ull;
		s_first[s] = 0xFFFFFFFFu;
	}
	if (threadIdx.x == 0)
		s_sum = 0ull;
	if (threadIdx.x < 4)
		s_chunk[threadIdx.x] = 0;	/* size 0: the first live leaf reserves a chunk */
	unsigned long long mine = 0;
	uint32_t nvalid = 0;

	uint32_t leaf = blockIdx.x;
	uint32_t l0 = 0, l1 = 0, r0 = 0, r1 = 0;
	gc_batch b;
	/* Leaf ranges are requested TWO leaves ahead and decoded one iteration later: read where they are needed (or one
	 * leaf ahead but converted at once) each of the two loads was followed by s_waitcnt vmcnt(0) - two exposed L2 round
	 * trips per leaf on the critical path of a kernel that has ~4 us per leaf. */
	uint2 raw_l = make_uint2(0, 0), raw_r = make_uint2(0, 0);	/* of leaf + gridDim.x */
	if (leaf < a.nleaves) {
		gc_leaf_range(a.off_l, a.cnt_l, a.cap_l, leaf, &l0, &l1);
		if (HAS_R)
			gc_leaf_range(a.off_r, a.cnt_r, a.cap_r, leaf, &r0, &r1);
		gc_prefetch<HAS_R, NW>(a, l0, l1, r0, r1, b);
		if (leaf + gridDim.x < a.nleaves) {
			raw_l = gc_leaf_raw(a.off_l, a.cnt_l, a.cap_l, leaf + gridDim.x);
			if (HAS_R)
				raw_r = gc_leaf_raw(a.off_r, a.cnt_r, a.cap_r, leaf + gridDim.x);
		}
	}
	__syncthreads();
	while (leaf < a.nleaves) {
		const uint32_t next = leaf + gridDim.x, next2 = next + gridDim.x;
		uint32_t nl0 = 0, nl1 = 0, nr0 = 0, nr1 = 0;
		if (next < a.nleaves) {		/* requested during the previous leaf */
			gc_leaf_decode(raw_l, a.cap_l, next, &nl0, &nl1);
			if (HAS_R)
				gc_leaf_decode(raw_r, a.cap_r, next, &nr0, &nr1);
		}
		if (next2 < a.nleaves) {	/* in flight during this leaf, decoded at the top of the next */
			raw_l = gc_leaf_raw(a.off_l, a.cnt_l, a.cap_l, next2);
			if (HAS_R)
				raw_r = gc_leaf_raw(a.off_r, a.cnt_r, a.cap_r, next2);
		}
		const bool nonempty = l0 != l1 && (!HAS_R || r0 != r1);	/* otherwise no group can come out of this leaf */
		const bool heavy_l = l1 - l0 >= a.heavy_l, heavy_r = HAS_R && r1 - r0 >= a.heavy_r;	/* hot keys: see gc_side_heavy */
		const bool hot = nonempty && (heavy_l || heavy_r);
		const bool live = HEAVY ? hot : (nonempty && !hot);
		if (!HEAVY && hot && threadIdx.x == 0)
			mdb_raise(a.status, 64u);	/* left to the hot-key path */
		/* a leaf whose left side fits one register batch (the normal case) is emitted by the threads that
		 * created its table slots; only oversized (skewed) leaves scan the whole table */
		const uint32_t build_rows = (HAS_R && BUILD_R) ? r1 - r0 : l1 - l0;
		const bool by_owner = build_rows <= GC_THREADS * LEAF_BATCH;
		uint32_t own[LEAF_BATCH];
		if (live && a.kbits) {
			/* Record list space: this leaf emits at most one record per left row (and per table slot).
			 * When the workgroup's current chunk cannot hold that, the unused tail is zero-filled (the
			 * ordering sort skips zero records) and a new chunk is reserved with ONE global atomic - i.e.
			 * one atomic per ~10-170 leaves instead of one on the critical path of every leaf. */
			const uint32_t rows = (HAS_R && (r1 - r0) < (l1 - l0)) ? r1 - r0 : l1 - l0;	/* a group needs a row on both sides */
			const uint32_t need = (rows < GC_SLOTS ? rows : GC_SLOTS) + 1;
			const uint32_t base = s_chunk[0], used = s_chunk[1], size = s_chunk[2];
			if (used + need > size) {		/* uniform: every thread read the same words */
				for (uint32_t i = used + threadIdx.x; i < size; i += GC_THREADS)
					a.rec[base + i] = 0ull;
				__syncthreads();		/* everyone has read s_chunk */
				if (threadIdx.x == 0) {
					const uint32_t want = need > GC_REC_CHUNK ? need : GC_REC_CHUNK;
					const uint32_t nb = atomicAdd(a.rec_count, want);
					if (nb + want > a.rec_cap) {
						mdb_raise(a.status, 8u);	/* list capacity exhausted: sizing bug, reported as an error */
						s_chunk[0] = 0;
						s_chunk[2] = 0;
					} else {
						s_chunk[0] = nb;
						s_chunk[2] = want;
					}
					s_chunk[1] = 0;
				}
				/* visible to everyone after the barrier that ends the build phase */
			}
		}
		if (live) {
			/* The table is built from the side with fewer rows per leaf (the right one when it is not the
			 * larger table): with N:1 data (every left key unique, few of them matched) the left rows then
			 * cost one LDS read each instead of an insert, a count, a minimum and a clear. */
			if (HAS_R && BUILD_R) {
				if (HEAVY && heavy_r)
					gc_side_heavy<false, true>(a, s_key, s_cnt, s_first, b, r0, r1);
				else
					gc_build_side<false, NW>(a, s_key, s_cnt, s_first, b, r0, r1, own);
				__syncthreads();
				if (HEAVY && heavy_l)
					gc_side_heavy<true, false>(a, s_key, s_cnt, s_first, b, l0, l1);
				else
					gc_probe_side<true, NW>(a, s_key, s_cnt, s_first, b, l0, l1);
				__syncthreads();
			} else {
				if (HEAVY && heavy_l)
					gc_side_heavy<true, true>(a, s_key, s_cnt, s_first, b, l0, l1);
				else
					gc_build_side<true, NW, !HAS_R>(a, s_key, s_cnt, s_first, b, l0, l1, own);
				__syncthreads();
				if (HAS_R) {
					if (HEAVY && heavy_r)
						gc_side_heavy<false, false>(a, s_key, s_cnt, s_first, b, r0, r1);
					else
						gc_probe_side<false, NW>(a, s_key, s_cnt, s_first, b, r0, r1);
					__syncthreads();
				}
			}
		}

		/* request the next leaf's first batches now: they travel while this leaf is emitted */
		if (next < a.nleaves)
			gc_prefetch<HAS_R, NW>(a, nl0, nl1, nr0, nr1, b);

		if (live) {
			/* emit one COUNT(*) per group and clear the slot.  Record mode: the groups of this leaf are
			 * appended to one global list (a single atomic per leaf reserves the space) and ordered
			 * afterwards by a radix sort on the first row id; dense mode: COUNT(*) is scattered to the
			 * group's first row position (8-byte random writes, only kept as a fallback). */
			const uint32_t cbase = s_chunk[0], csize = s_chunk[2];
			/* owner mode: one round per register-batch element, plus one for the side slot of the hash-0 key in the
			 * single leaf that holds it (a wave that reads the flag after thread 0 cleared it has nothing to do there) */
			const int iters = by_owner ? (s_cnt[GC_SLOTS] != 0ull ? LEAF_BATCH + 1 : LEAF_BATCH) : GC_EMIT_ITERS;
#pragma unroll
			for (int it = 0; it < GC_EMIT_ITERS; it++) {
				if (it >= iters)
					break;
				unsigned long long recv = 0;
				uint32_t s = 0xFFFFFFFFu;
				if (by_owner) {
					/* slots created by this thread, plus (thread 0) the side slot of the hash-0 key */
					if (it < LEAF_BATCH)
						s = own[it];
					else if (it == LEAF_BATCH && threadIdx.x == 0)
						s = GC_SLOTS;
				} else {
					s = threadIdx.x + (uint32_t)it * GC_THREADS;
					if (s > GC_SLOTS)
						s = 0xFFFFFFFFu;
				}
				if (s != 0xFFFFFFFFu) {
					const unsigned long long c2 = s_cnt[s];
					const bool occupied = s < GC_SLOTS ? (by_owner || s_key[s] != 0ull) : c2 != 0ull;
					if (occupied) {
						const uint32_t cl = (uint32_t)c2, cr = (uint32_t)(c2 >> 32);
						const uint32_t first = s_first[s];
						if (s < GC_SLOTS)
							s_key[s] = 0ull;
						s_cnt[s] = 0ull;
						s_first[s] = 0xFFFFFFFFu;
						if (cl && (!HAS_R || cr)) {
							const unsigned long long c = HAS_R ? (unsigned long long)cl * cr : (unsigned long long)cl;
							mine += c;
							if (a.kbits) {
								if (c >> (64 - a.kbits))
									mdb_raise(a.status, 4u);	/* COUNT(*) does not fit beside the row id */
								if (c >> (32 - (a.kbits < 32 ? a.kbits : 31)))
									mdb_raise(a.status, 16u);	/* ... nor in a 4-byte record */
								recv = ((unsigned long long)first << (64 - a.kbits)) | c;
							} else {
								if (first < a.dense_n)
								a.dense_cnt[first] = (int64_t)c;
							}
						}
					}
				}
				if (a.kbits) {
					/* wave-level append: one LDS atomic per wave and iteration, no workgroup barrier */
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
							nvalid++;
						}
					}
				}
			}
			__syncthreads();	/* the next leaf builds into the cleared tables */
		}
		leaf = next;
		l0 = nl0;
		l1 = nl1;
		r0 = nr0;
		r1 = nr1;
	}
	if (a.kbits) {
		const uint32_t base = s_chunk[0], used = s_chunk[1], size = s_chunk[2];
		for (uint32_t i = used + threadIdx.x; i < size; i += GC_THREADS)
			a.rec[base + i] = 0ull;		/* unused tail of the last chunk */
		if (nvalid)
			atomicAdd(&s_chunk[3], nvalid);
	}
	if (mine)
		atomicAdd(&s_sum, mine);
	__syncthreads();
	if (threadIdx.x == 0 && s_sum)
		atomicAdd(a.joined, s_sum);
	if (threadIdx.x == 0 && a.kbits && s_chunk[3])
		atomicAdd(a.rec_valid, s_chunk[3]);
}

/* ------------------------------------------------------------------ direct-address leaves (compact narrow form)
 *
 * When the keys of both tables span fewer than 2^k values (k found from the key sample, verified for every key by the
 * first partition level), the 32-bit hash is mdb_mixk(key - base) in the top k bits of its field - a bijection of the
 * window - and the radix partition has consumed the top b1 + b2 of them: inside a leaf only rem = k - b1 - b2 bits tell
 * keys apart.  For rem <= LD_MAX_REM those bits INDEX the leaf's table: three plain arrays in LDS (right rows, left
 * rows, first left row per key) - no stored keys, no compare-and-swap, no probe chains, no "table full".  A right row
 * is ONE LDS add, a left row ONE LDS read (plus an add and a minimum when its key has right rows), and the emit pass
 * walks the arrays linearly.  The hashed kernel above spends ~120 wave instructions per 64 rows and is issue bound
 * (profiles/r01: 380 M vector + scalar instructions for 2 * 10^8 rows); this one is bound by the HBM stream of the
 * partitioned rows.  Hashing (instead of partitioning by key range) keeps the leaves evenly filled whatever part of the
 * window a table covers: the benchmark's right table holds the lowest sixteenth of the left table's key range.
 *
 * 512 threads, dynamic LDS of (12 << rem) + 32 bytes (24 KiB at rem = 11: four workgroups = 32 waves per CU).
 * Persistent: the first register round of both sides of the NEXT leaf is requested before the current leaf is emitted.
 */
#define LD_MAX_REM 12u
/* ROUND = rows of either side in the first (register) round of a leaf: 2048 (24 prefetch registers in the two sets, 64 in
 * all).  4096 for the leaves of 2^12 entries (2 x 3052 rows on average at 10^8 rows) needs 80 registers + 11 spilled and
 * measured 1-3 % slower than 2048 with the remaining rows loaded in the loop. */

template <int THREADS, int ROUND>
struct ld_regs {
	static constexpr int RB = ROUND / (4 * THREADS);	/* 16-byte loads per thread, right side: 4 words each */
	static constexpr int RA = ROUND / (2 * THREADS);	/* ... left side: 2 words each */
	uint4 b[RB];
	ulonglong2 a[RA];
	uint32_t l1, r1;		/* ends of the requested leaf's two row ranges (they start at leaf * cap) */
	uint2 next_l, next_r;		/* row counts of the leaf this register set will take NEXT, as loaded (gc_leaf_decode) */
};

/* Row counts of `leaf`, to be decoded one leaf's work later. */
template <bool HAS_R, int THREADS, int ROUND>
__device__ static inline void ld_request_counts(const gc_args &a, uint32_t leaf, ld_regs<THREADS, ROUND> &q)
{
	q.next_l = make_uint2(0, 0);
	q.next_r = make_uint2(0, 0);
	if (leaf < a.nleaves) {
		q.next_l = gc_leaf_raw(a.off_l, a.cnt_l, a.cap_l, leaf);
		if (HAS_R)
			q.next_r = gc_leaf_raw(a.off_r, a.cnt_r, a.cap_r, leaf);
	}
}

/* First-round loads of `leaf` (whose counts were requested into q before), untouched until they are consumed: a
 * conversion at load time would make the request wait for itself (see gc_batch).  A leaf is requested TWO leaves ahead of
 * its turn, so its loads have a whole leaf's work to land in; its counts are requested two leaves before that, so that
 * only the rows that exist are fetched (whole rounds regardless of the count were 30 % more bytes and 15 % slower). */
template <bool HAS_R, int THREADS, int ROUND>
__device__ static inline void ld_request(const gc_args &a, uint32_t leaf, ld_regs<THREADS, ROUND> &q)
{
	uint32_t l0, r0 = 0;
	gc_leaf_decode(q.next_l, a.cap_l, leaf, &l0, &q.l1);
	q.r1 = 1;
	if (HAS_R)
		gc_leaf_decode(q.next_r, a.cap_r, leaf, &r0, &q.r1);
	ld_request_counts<HAS_R, THREADS, ROUND>(a, leaf + 2 * gridDim.x, q);
#pragma unroll
	for (int u = 0; u < ld_regs<THREADS, ROUND>::RA; u++) {
		const uint32_t i = l0 + 2u * ((uint32_t)u * THREADS + threadIdx.x);
		q.a[u] = make_ulonglong2(0ull, 0ull);
		if (i < q.l1)
			q.a[u] = *reinterpret_cast<const ulonglong2 *>(a.hv_l + i);
	}
	if (HAS_R) {
#pragma unroll
		for (int u = 0; u < ld_regs<THREADS, ROUND>::RB; u++) {
			const uint32_t j = r0 + 4u * ((uint32_t)u * THREADS + threadIdx.x);
			q.b[u] = make_uint4(0u, 0u, 0u, 0u);
			if (j < q.r1)
				q.b[u] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(a.hv_r) + j);
		}
	}
}

/* one left row: narrow word = hash32 << 32 | row id */
template <bool HAS_R>
__device__ static inline void ld_left_row(unsigned long long w, uint32_t shift, uint32_t mask, const uint32_t *s_cr, uint32_t *s_cl,
					  uint32_t *s_first)
{
	const uint32_t idx = ((uint32_t)(w >> 32) >> shift) & mask;
	if (!HAS_R || s_cr[idx]) {
		atomicAdd(&s_cl[idx], 1u);
		atomicMin(&s_first[idx], (uint32_t)w);
	}
}

struct ld_state {
	uint32_t *s_cr, *s_cl, *s_first, *s_chunk;
	uint32_t *s_cx;		/* nextra arrays of T counters: the further right tables */
	uint32_t T, mask, shift, rem;
	unsigned long long mine;
	uint32_t nvalid;
};

/* one leaf: count the right rows, look the left rows up, request leaf + 2 * gridDim.x into the registers just consumed, emit */
template <bool HAS_R, int THREADS, int ROUND>
__device__ static inline void ld_leaf(const gc_args &a, ld_state &st, uint32_t leaf, ld_regs<THREADS, ROUND> &q)
{
	uint32_t *const s_cr = st.s_cr, *const s_cl = st.s_cl, *const s_first = st.s_first, *const s_chunk = st.s_chunk;
	const uint32_t T = st.T, mask = st.mask, shift = st.shift;
	const uint32_t l0 = leaf * a.cap_l, l1 = q.l1, r0 = HAS_R ? leaf * a.cap_r : 0u, r1 = q.r1;
	const bool nonempty = l0 != l1 && (!HAS_R || r0 != r1);
	const bool hot = nonempty && (l1 - l0 >= a.heavy_l || (HAS_R && r1 - r0 >= a.heavy_r));
	const bool live = nonempty && !hot;
	if (hot && threadIdx.x == 0)
		mdb_raise(a.status, 64u);	/* left to the hot-key path (hashed tables in global memory) */
	if (live && a.kbits) {
		/* record list space, reserved a chunk at a time: one global atomic per ~10-170 leaves (see k_leaf_group_count).
		 * (Reserving for the DISTINCT right keys instead of the rows - counted with returning adds - left fewer zero-filled
		 * gaps in the list but cost the kernel 25 %: profiles/micro/leaf_direct_exp.sh) */
		const uint32_t rows = (HAS_R && (r1 - r0) < (l1 - l0)) ? r1 - r0 : l1 - l0;	/* a group needs a row on both sides */
		const uint32_t need = (rows < T ? rows : T) + 1;
		const uint32_t base = s_chunk[0], used = s_chunk[1], size = s_chunk[2];
		if (used + need > size) {		/* uniform: every thread read the same words */
			for (uint32_t i = used + threadIdx.x; i < size; i += THREADS) {
				if (a.rec32)
					reinterpret_cast<uint32_t *>(a.rec)[base + i] = 0u;
				else
					a.rec[base + i] = 0ull;
			}
			__syncthreads();		/* everyone has read s_chunk */
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
			/* visible to everyone after the barrier that ends the count phase */
		}
	}
	if (live) {
		if (HAS_R) {
			/* right rows: one LDS add each (r0 is a multiple of 64: the rounds start on the leaf's first row) */
			const uint32_t *const hv_r32 = reinterpret_cast<const uint32_t *>(a.hv_r);
#pragma unroll
			for (int u = 0; u < ld_regs<THREADS, ROUND>::RB; u++) {
				const uint32_t j = r0 + 4u * ((uint32_t)u * THREADS + threadIdx.x);
				const uint32_t w[4] = { q.b[u].x, q.b[u].y, q.b[u].z, q.b[u].w };
#pragma unroll
				for (int k = 0; k < 4; k++)
					if (j + k < r1)
						atomicAdd(&s_cr[(w[k] >> shift) & mask], 1u);
			}
			for (uint32_t j = r0 + 4u * (ld_regs<THREADS, ROUND>::RB * THREADS + threadIdx.x); j < r1; j += 4u * THREADS) {
				const uint4 v = *reinterpret_cast<const uint4 *>(hv_r32 + j);
				const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int k = 0; k < 4; k++)
					if (j + k < r1)
						atomicAdd(&s_cr[(w[k] >> shift) & mask], 1u);
			}
			if (a.nextra) {		/* (uniform) further right tables on the same key: their rows, then the product of the counts */
				for (uint32_t x = 0; x < a.nextra; x++) {
					uint32_t *const s_c = st.s_cx + x * T;
					const uint32_t x0 = leaf * a.cap_x[x], xc = a.cnt_x[x][leaf], x1 = x0 + (xc < a.cap_x[x] ? xc : a.cap_x[x]);
					for (uint32_t j = x0 + 4u * threadIdx.x; j < x1; j += 4u * THREADS) {
						const uint4 v = *reinterpret_cast<const uint4 *>(a.hv_x[x] + j);
						const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
						for (int k = 0; k < 4; k++)
							if (j + k < x1)
								atomicAdd(&s_c[(w[k] >> shift) & mask], 1u);
					}
				}
				__syncthreads();
				for (uint32_t sl = threadIdx.x; sl < T; sl += THREADS) {
					unsigned long long c = s_cr[sl];
					for (uint32_t x = 0; x < a.nextra; x++) {
						c *= st.s_cx[x * T + sl];
						st.s_cx[x * T + sl] = 0u;
						if (c >> 32) {		/* the product of the right counts no longer fits: reported, the caller chains 2-table operators */
							mdb_raise(a.status, 2048u);
							c = 0;
						}
					}
					s_cr[sl] = (uint32_t)c;
				}
			}
			__syncthreads();
		}
		/* left rows: one LDS read each; rows whose key has right rows are counted and compete for "first" */
#pragma unroll
		for (int u = 0; u < ld_regs<THREADS, ROUND>::RA; u++) {
			const uint32_t i = l0 + 2u * ((uint32_t)u * THREADS + threadIdx.x);
			if (i < l1)
				ld_left_row<HAS_R>(q.a[u].x, shift, mask, s_cr, s_cl, s_first);
			if (i + 1 < l1)
				ld_left_row<HAS_R>(q.a[u].y, shift, mask, s_cr, s_cl, s_first);
		}
		for (uint32_t i = l0 + 2u * (ld_regs<THREADS, ROUND>::RA * THREADS + threadIdx.x); i < l1; i += 2u * THREADS) {
			const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.hv_l + i);
			ld_left_row<HAS_R>(v.x, shift, mask, s_cr, s_cl, s_first);
			if (i + 1 < l1)
				ld_left_row<HAS_R>(v.y, shift, mask, s_cr, s_cl, s_first);
		}
		__syncthreads();
	}

	/* these registers are free again: the leaf after next travels while this one is emitted and the next one processed */
	if (leaf + 2 * gridDim.x < a.nleaves)
		ld_request<HAS_R, THREADS, ROUND>(a, leaf + 2 * gridDim.x, q);

	if (live) {
		/* emit one record per key that has rows on both sides and clear what was touched (4-byte accesses, thread t takes
		 * entries t, t + THREADS, ...: four consecutive entries per thread with 16-byte LDS accesses measured 10 % slower
		 * when few entries are occupied and 4 % faster when all are - profiles/micro/leaf_direct_exp.sh) */
		const uint32_t cbase = s_chunk[0], csize = s_chunk[2];
		for (uint32_t s0 = 0; s0 < T; s0 += THREADS) {	/* uniform trip count */
			const uint32_t s = s0 + threadIdx.x;
			unsigned long long recv = 0;
			if (s < T) {
				const uint32_t cl = s_cl[s];
				uint32_t cr = 1u;
				if (HAS_R) {
					cr = s_cr[s];
					if (cr)
						s_cr[s] = 0u;
				}
				if (cl) {
					const uint32_t first = s_first[s];
					const unsigned long long c = (unsigned long long)cl * cr;
					s_cl[s] = 0u;
					s_first[s] = 0xFFFFFFFFu;
					st.mine += c;
					if (HAS_R && cl > 1u && cr)
						mdb_raise(a.status, GC_ST_LEFT_DUPS);	/* (a matched key with several left rows: mdb_dev_join_keys_ordered asks) */
					if (a.kbits && a.keyed_cbits) {
						if (c >> a.keyed_cbits)
							mdb_raise(a.status, 256u);	/* COUNT(*) does not fit a keyed record: redone with plain records */
						recv = ((unsigned long long)first << (64 - a.kbits)) |
						       ((unsigned long long)((leaf << st.rem) | s) << a.keyed_cbits) | c;
					} else if (a.kbits) {
						if (c >> (64 - a.kbits))
							mdb_raise(a.status, 4u);	/* COUNT(*) does not fit beside the row id */
						if (c >> (32 - (a.kbits < 32 ? a.kbits : 31)))
							mdb_raise(a.status, 16u);	/* ... nor in a 4-byte record (the ordering sort then moves 8-byte ones) */
						recv = a.rec32 ? (unsigned long long)(((uint32_t)first << (32 - a.kbits)) | (uint32_t)c)
							       : ((unsigned long long)first << (64 - a.kbits)) | c;
						if (a.rec32 && (c >> (32 - a.kbits)))
							mdb_raise(a.status, 512u);
					} else {
						if (first < a.dense_n)
								a.dense_cnt[first] = (int64_t)c;
					}
				}
			}
			if (a.kbits) {
				/* wave-level append: one LDS atomic per wave and iteration */
				const uint64_t m = __ballot(recv != 0ull);
				if (m) {
					const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
					uint32_t wbase = 0;
					if (mdb_lane() == leader)
						wbase = atomicAdd(&s_chunk[1], (uint32_t)__popcll(m));
					wbase = __shfl(wbase, (int)leader, MDB_WAVE);
					if (recv) {
						const uint32_t pos = wbase + (uint32_t)__popcll(m & mdb_lanemask_lt());
						if (pos < csize) {
							if (a.rec32)
								reinterpret_cast<uint32_t *>(a.rec)[cbase + pos] = (uint32_t)recv;
							else
								a.rec[cbase + pos] = recv;
						}
						st.nvalid++;
					}
				}
			}
		}
		__syncthreads();	/* the next leaf counts into the cleared arrays */
	}
}

template <bool HAS_R, int THREADS, int ROUND>
__global__ __launch_bounds__(THREADS, 4 * THREADS / 256) void k_leaf_direct	/* four workgroups per CU: at most 64 registers */(gc_args a, uint32_t rem, uint32_t shift)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t ld_lds[];
	ld_state st;
	st.T = 1u << rem;
	st.rem = rem;
	st.mask = st.T - 1u;
	st.shift = shift;
	st.s_cr = ld_lds;			/* right rows per key */
	st.s_cl = ld_lds + st.T;		/* left rows per key (joins: only of keys that have right rows) */
	st.s_first = ld_lds + 2 * st.T;		/* first left row per key */
	st.s_chunk = ld_lds + 3 * st.T;		/* record list chunk: [0] base [1] used [2] size [3] valid records; [4..7] see below */
	st.s_cx = ld_lds + 3 * st.T + 8;	/* (further right tables) */
	for (uint32_t s = threadIdx.x; s < a.nextra * st.T; s += THREADS)
		st.s_cx[s] = 0u;
	unsigned long long *const s_sum = reinterpret_cast<unsigned long long *>(ld_lds + 3 * st.T + 4);
	st.mine = 0;
	st.nvalid = 0;

	for (uint32_t s = threadIdx.x; s < st.T; s += THREADS) {
		st.s_cr[s] = 0u;
		st.s_cl[s] = 0u;
		st.s_first[s] = 0xFFFFFFFFu;
	}
	if (threadIdx.x < 8)
		st.s_chunk[threadIdx.x] = 0u;		/* [4..5] = s_sum, [6] = distinct right keys of the current leaf */

	/* two register sets: while leaf i is processed out of one, leaf i + 1 sits (or still travels) in the other */
	ld_regs<THREADS, ROUND> qa, qb;
	uint32_t leaf = blockIdx.x;
	ld_request_counts<HAS_R, THREADS, ROUND>(a, leaf, qa);
	ld_request_counts<HAS_R, THREADS, ROUND>(a, leaf + gridDim.x, qb);
	if (leaf < a.nleaves)
		ld_request<HAS_R, THREADS, ROUND>(a, leaf, qa);
	if (leaf + gridDim.x < a.nleaves)
		ld_request<HAS_R, THREADS, ROUND>(a, leaf + gridDim.x, qb);
	__syncthreads();
	while (leaf < a.nleaves) {
		ld_leaf<HAS_R, THREADS, ROUND>(a, st, leaf, qa);
		leaf += gridDim.x;
		if (leaf >= a.nleaves)
			break;
		ld_leaf<HAS_R, THREADS, ROUND>(a, st, leaf, qb);
		leaf += gridDim.x;
	}
	if (a.kbits) {
		const uint32_t base = st.s_chunk[0], used = st.s_chunk[1], size = st.s_chunk[2];
		for (uint32_t i = used + threadIdx.x; i < size; i += THREADS) {	/* unused tail of the last chunk */
			if (a.rec32)
				reinterpret_cast<uint32_t *>(a.rec)[base + i] = 0u;
			else
				a.rec[base + i] = 0ull;
		}
		if (st.nvalid)
			atomicAdd(&st.s_chunk[3], st.nvalid);
	}
	if (st.mine)
		atomicAdd(s_sum, st.mine);
	__syncthreads();
	if (threadIdx.x == 0 && *s_sum)
		atomicAdd(a.joined, *s_sum);
	if (threadIdx.x == 0 && a.kbits && st.s_chunk[3])
		atomicAdd(a.rec_valid, st.s_chunk[3]);
}

/* ------------------------------------------------------------------ semi-join filter (compact narrow form)
 *
 * Bitmap of the hashed key values the RIGHT table holds, one bit per 2^coarse adjacent values, built from its
 * partitioned form: the 2^rem values a leaf can hold are a contiguous slice of the bitmap, so ONE WAVE per leaf sets the
 * bits in LDS and writes the slice with coalesced stores (no global atomics); leaves are consecutive in the bitmap, so
 * the slices of a first-level digit are too - which is what the left table's second partition level loads. */
#define LB_WAVES 4u
__global__ __launch_bounds__(LB_WAVES * MDB_WAVE) void k_leaf_bitmap(const uint32_t *__restrict__ hv_r, const uint32_t *__restrict__ cnt_r, uint32_t cap_r,
								 uint32_t nleaves, uint32_t rem, uint32_t coarse, uint32_t shift,
								 uint32_t *__restrict__ bits)
{
	__shared__ uint32_t s_bits[LB_WAVES][(1u << LD_MAX_REM) / 32];
	const uint32_t wave = threadIdx.x / MDB_WAVE, lane = mdb_lane();
	const uint32_t words = 1u << (rem - coarse - 5u), mask = (1u << rem) - 1u;	/* rem - coarse >= 5: at least one word */
	const uint32_t rounds = (nleaves + gridDim.x * LB_WAVES - 1) / (gridDim.x * LB_WAVES);
	for (uint32_t k = 0; k < rounds; k++) {
		const uint32_t leaf = (k * gridDim.x + blockIdx.x) * LB_WAVES + wave;
		for (uint32_t w = lane; w < words; w += MDB_WAVE)
			s_bits[wave][w] = 0u;
		__syncthreads();
		if (leaf < nleaves) {
			const uint32_t c = cnt_r[leaf], r0 = leaf * cap_r, r1 = r0 + (c < cap_r ? c : cap_r);
			for (uint32_t j = r0 + 4u * lane; j < r1; j += 4u * MDB_WAVE) {
				const uint4 v = *reinterpret_cast<const uint4 *>(hv_r + j);	/* (regions start 16-byte aligned and are padded) */
				const uint32_t h[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int q = 0; q < 4; q++)
					if (j + q < r1) {
						const uint32_t idx = ((h[q] >> shift) & mask) >> coarse;
						atomicOr(&s_bits[wave][idx >> 5], 1u << (idx & 31u));
					}
			}
		}
		__syncthreads();
		if (leaf < nleaves)
			for (uint32_t w = lane; w < words; w += MDB_WAVE)
				bits[(size_t)leaf * words + w] = s_bits[wave][w];
		__syncthreads();
	}
}

/* ------------------------------------------------------------------ hot keys across the whole chip
 *
 * One workgroup streams a leaf at ~16 GB/s; a key with 10^7 duplicates would pin one workgroup for 10+ ms while the
 * other 511 idle.  Hot leaves (GC_HEAVY rows or more on a side; the plain kernel skips them and raises status bit 6)
 * are therefore cut into slices of HOT_SLICE rows that ALL resident workgroups share: every slice is merged into an
 * LDS table with gc_side_heavy (a wave's duplicates first collapse into one update), the table is flushed into the
 * leaf's table in global memory with one atomic triple per distinct key, and a last small kernel turns the global
 * tables into group records exactly like the emit phase of the plain kernel.  At most HOT_MAX hot leaves take this
 * path; beyond that the single-workgroup HEAVY instance of the leaf kernel does the work.
 */
#define HOT_MAX 64u
#define HOT_SLOTS 8191u		/* global table per hot leaf (prime; a leaf holds at most GC_SLOTS distinct keys) */
#define HOT_SLICE 65536u

struct hot_args {
	uint32_t *count;			/* number of hot leaves found */
	uint32_t *leaf;				/* [HOT_MAX] */
	unsigned long long *g_key;		/* [HOT_MAX][HOT_SLOTS] hashed key, 0 = empty */
	unsigned long long *g_cnt;		/* [HOT_MAX][HOT_SLOTS + 1] packed counts; [HOT_SLOTS] = the key whose hash is 0 */
	uint32_t *g_first;			/* [HOT_MAX][HOT_SLOTS + 1] */
};

template <bool HAS_R>
__global__ void k_hot_list(gc_args a, hot_args h)
{
	const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
	if (leaf >= a.nleaves)
		return;
	uint32_t l0, l1, r0 = 0, r1 = 1;
	gc_leaf_range(a.off_l, a.cnt_l, a.cap_l, leaf, &l0, &l1);
	if (HAS_R)
		gc_leaf_range(a.off_r, a.cnt_r, a.cap_r, leaf, &r0, &r1);
	if (l0 == l1 || r0 == r1)
		return;
	if (l1 - l0 >= a.heavy_l || (HAS_R && r1 - r0 >= a.heavy_r)) {
		const uint32_t i = atomicAdd(h.count, 1u);
		if (i < HOT_MAX)
			h.leaf[i] = leaf;
	}
}

__global__ void k_hot_clear(hot_args h)
{
	const uint32_t nh = *h.count < HOT_MAX ? *h.count : HOT_MAX;
	const uint64_t total = (uint64_t)nh * (HOT_SLOTS + 1);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t t = i / (HOT_SLOTS + 1), s = i % (HOT_SLOTS + 1);
		if (s < HOT_SLOTS)
			h.g_key[t * HOT_SLOTS + s] = 0ull;
		h.g_cnt[i] = 0ull;
		h.g_first[i] = 0xFFFFFFFFu;
	}
}

__device__ static inline uint32_t hot_insert(unsigned long long *keys, uint64_t hv)
{
	uint32_t s = leaf_slot(hv, HOT_SLOTS);
	const uint32_t step = leaf_step(hv, HOT_SLOTS);
	for (uint32_t probe = 0; probe < HOT_SLOTS; probe++) {
		const unsigned long long old = atomicCAS(&keys[s], 0ull, (unsigned long long)hv);
		if (old == 0ull || old == hv)
			return s;
		s += step;
		if (s >= HOT_SLOTS)
			s -= HOT_SLOTS;
	}
	return 0xFFFFFFFFu;
}

template <bool HAS_R>
__global__ __launch_bounds__(GC_THREADS, 8) void k_hot_slices(gc_args a, hot_args h)
{
	__shared__ unsigned long long s_key[GC_SLOTS];
	__shared__ unsigned long long s_cnt[GC_SLOTS + 1];
	__shared__ uint32_t s_first[GC_SLOTS + 1];
	for (uint32_t s = threadIdx.x; s <= GC_SLOTS; s += GC_THREADS) {
		if (s < GC_SLOTS)
			s_key[s] = 0ull;
		s_cnt[s] = 0ull;
		s_first[s] = 0xFFFFFFFFu;
	}
	__syncthreads();
	const uint32_t nh = *h.count;	/* <= HOT_MAX guaranteed by the host */
	uint32_t task = 0;
	gc_batch b;
	for (uint32_t t = 0; t < nh; t++) {
		const uint32_t leaf = h.leaf[t];
		uint32_t l0, l1, r0 = 0, r1 = 0;
		gc_leaf_range(a.off_l, a.cnt_l, a.cap_l, leaf, &l0, &l1);
		if (HAS_R)
			gc_leaf_range(a.off_r, a.cnt_r, a.cap_r, leaf, &r0, &r1);
		const uint32_t nsl = (l1 - l0 + HOT_SLICE - 1) / HOT_SLICE, nsr = (r1 - r0 + HOT_SLICE - 1) / HOT_SLICE;
		for (uint32_t k = 0; k < nsl + nsr; k++, task++) {
			if (task % gridDim.x != blockIdx.x)
				continue;	/* uniform */
			const bool is_l = k < nsl;
			const uint32_t x0 = is_l ? l0 + k * HOT_SLICE : r0 + (k - nsl) * HOT_SLICE;
			const uint32_t xe = is_l ? l1 : r1;
			const uint32_t x1 = x0 + HOT_SLICE < xe ? x0 + HOT_SLICE : xe;
			if (is_l) {
				gc_load_l<2>(a, x0, x1, b);
				gc_side_heavy<true, true>(a, s_key, s_cnt, s_first, b, x0, x1);
			} else {
				gc_load_r<2>(a, x0, x1, b);
				gc_side_heavy<false, true>(a, s_key, s_cnt, s_first, b, x0, x1);
			}
			__syncthreads();
			/* flush the slice's table into the leaf's global table and clear it */
			unsigned long long *gk = h.g_key + (uint64_t)t * HOT_SLOTS;
			unsigned long long *gc = h.g_cnt + (uint64_t)t * (HOT_SLOTS + 1);
			uint32_t *gf = h.g_first + (uint64_t)t * (HOT_SLOTS + 1);
			for (uint32_t s = threadIdx.x; s <= GC_SLOTS; s += GC_THREADS) {
				const unsigned long long c2 = s_cnt[s];
				if (!c2)
					continue;
				uint32_t g = HOT_SLOTS;
				if (s < GC_SLOTS) {
					g = hot_insert(gk, s_key[s]);
					s_key[s] = 0ull;
				}
				if (g == 0xFFFFFFFFu) {
					mdb_raise(a.status, 1u);
				} else {
					atomicAdd(&gc[g], c2);
					if (is_l)
						atomicMin(&gf[g], s_first[s]);
				}
				s_cnt[s] = 0ull;
				s_first[s] = 0xFFFFFFFFu;
			}
			__syncthreads();
		}
	}
}

/* one workgroup per hot leaf: global table -> group records (or dense counts), like the emit phase of the leaf kernel */
template <bool HAS_R>
__global__ __launch_bounds__(1024) void k_hot_finish(gc_args a, hot_args h)
{
	__shared__ unsigned long long s_sum;
	__shared__ uint32_t s_valid;
	const uint32_t t = blockIdx.x;
	if (t >= *h.count)
		return;
	if (threadIdx.x == 0) {
		s_sum = 0ull;
		s_valid = 0;
	}
	__syncthreads();
	const unsigned long long *gc = h.g_cnt + (uint64_t)t * (HOT_SLOTS + 1);
	const uint32_t *gf = h.g_first + (uint64_t)t * (HOT_SLOTS + 1);
	unsigned long long mine = 0;
	uint32_t nvalid = 0;
	for (uint32_t s = threadIdx.x; s <= HOT_SLOTS; s += blockDim.x) {
		const unsigned long long c2 = gc[s];
		const uint32_t cl = (uint32_t)c2, cr = (uint32_t)(c2 >> 32);
		if (!cl || (HAS_R && !cr))
			continue;
		const unsigned long long c = HAS_R ? (unsigned long long)cl * cr : (unsigned long long)cl;
		const uint32_t first = gf[s];
		mine += c;
		if (a.kbits) {
			if (c >> (64 - a.kbits))
				mdb_raise(a.status, 4u);
			if (c >> (32 - (a.kbits < 32 ? a.kbits : 31)))
				mdb_raise(a.status, 16u);
			const uint32_t pos = atomicAdd(a.rec_count, 1u);
			if (pos < a.rec_cap)
				a.rec[pos] = ((unsigned long long)first << (64 - a.kbits)) | c;
			else
				mdb_raise(a.status, 8u);
			nvalid++;
		} else {
			if (first < a.dense_n)
								a.dense_cnt[first] = (int64_t)c;
		}
	}
	if (mine)
		atomicAdd(&s_sum, mine);
	if (nvalid)
		atomicAdd(&s_valid, nvalid);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_sum)
			atomicAdd(a.joined, s_sum);
		if (s_valid)
			atomicAdd(a.rec_valid, s_valid);
	}
}

/* ------------------------------------------------------------------ compaction of the dense count array
 * (implemented in mdb_dev_filter.hip: predicate "count <> 0" -> ascending positions) */
size_t mdb_filter_arena_bytes(uint64_t n);
int mdb_filter_nonzero64(mdb_dev_ctx *ctx, const int64_t *vals, uint64_t n, uint32_t *out_sel, uint32_t **d_total);

/* ------------------------------------------------------------------ NULL-key group (plain GROUP BY only) */

__global__ __launch_bounds__(256) void k_null_stats(const uint64_t *__restrict__ nullbits, uint64_t n,
						    unsigned long long *cnt_first /* [0]=count [1]=first */)
{
	const uint64_t words = (n + 63) >> 6;
	unsigned long long c = 0, first = ~0ull;
	for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t m = nullbits[w];
		if (w == words - 1 && (n & 63))
			m &= (1ull << (n & 63)) - 1ull;
		if (m) {
			c += (unsigned long long)__popcll(m);
			const unsigned long long f = (w << 6) + (unsigned long long)(__ffsll((long long)m) - 1);
			if (f < first)
				first = f;
		}
	}
	if (c) {
		atomicAdd(&cnt_first[0], c);
		atomicMin(&cnt_first[1], first);
	}
}

__global__ void k_null_poke(const unsigned long long *cnt_first, int64_t *dense_cnt)
{
	if (threadIdx.x == 0 && blockIdx.x == 0 && cnt_first[0])
		dense_cnt[cnt_first[1]] = (int64_t)cnt_first[0];
}

__global__ void k_null_rec(const unsigned long long *cnt_first, unsigned long long *rec, uint32_t *rec_count, uint32_t *rec_valid,
			   uint32_t rec_cap, uint32_t kbits, uint32_t *status)
{
	if (threadIdx.x == 0 && blockIdx.x == 0 && cnt_first[0]) {
		if (cnt_first[0] >> (64 - kbits))
			atomicOr(status, 4u);
		if (cnt_first[0] >> (32 - (kbits < 32 ? kbits : 31)))
			atomicOr(status, 16u);
		const uint32_t pos = atomicAdd(rec_count, 1u);
		if (pos < rec_cap) {
			rec[pos] = (cnt_first[1] << (64 - kbits)) | cnt_first[0];
			atomicAdd(rec_valid, 1u);
		} else {
			atomicOr(status, 8u);
		}
	}
}

/* ------------------------------------------------------------------ group-count drivers */


/* MDB_DIRECT_LEAF=0 keeps the compact narrow form off (A/B measurements, soaks of the hashed leaf kernel) */
bool ld_disabled(void)
{
	static int v = -1;
	if (v < 0) {
		const char *e = mdb_knob("MDB_DIRECT_LEAF");
		v = (e && e[0] == '0') ? 1 : 0;
	}
	return v == 1;
}

/* slots of the group-record list: every group once, plus the zero-filled gaps of the chunked reservation
 * (at most one leaf's worth per chunk, one unfinished chunk per workgroup) */
uint64_t gc_rec_capacity(mdb_dev_ctx *ctx, uint64_t n_l)
{
	return n_l + n_l / 4 + (uint64_t)(4 * ctx->num_cus + 4) * GC_REC_CHUNK + 4096;	/* two kernel instances, one open chunk per workgroup */
}

/* The operator runs in two halves so that a multi-GPU pipeline can partition the left table while the
 * right table is still arriving over xGMI (mdb_dev_join_group_count_begin / _finish). */
struct gc_state {
	bool active;
	const int64_t *keys_l;
	const uint64_t *null_l;
	uint64_t n_l, n_r_cap;
	bool has_r, null_group, fast, want_records, no_build_r;
	bool defer_ok;		/* the caller runs gc_begin and gc_finish back to back: the left table may be partitioned in gc_finish */
	bool narrow;		/* 32-bit hashes; the left words carry the row ids (see mdb_partition_table) */
	bool keys32;		/* both key columns are int32 arrays (received over xGMI in the 4-byte wire format) */
	int64_t base;		/* narrow form: centre of the key window (mdb_partition_table) */
	uint32_t key_bits;	/* compact narrow form offered by the key sample: every key in [key_lo, key_lo + 2^key_bits) (0 = none) */
	int64_t key_lo;
	bool direct;		/* ... taken: the leaves are joined by k_leaf_direct (decided in gc_begin, where the leaf count is known) */
	bool fast1;		/* gc_window.fast1 */
	bool one_level;		/* ... by k_leaf_wide: ONE partition level of 9 bits, tables of 2^(key_bits - 9) entries */
	bool wide12;		/* ... by k_leaf_wide over the digits of ONE 4096-digit pass per table (mdb_scatter4096): key windows of 2^24 ... 2^27 values */
	bool selective;		/* hint of the key sample: most left rows will find no partner */
	bool by_span;		/* ... because the right table's keys cover a small part of the left table's range */
	bool prunable;		/* the right table's keys cover less than 7/8 of the left table's range (sample) */
	bool own_call;		/* begin and finish are the two halves of ONE operator call (not the split API): the first-level cursors and the ordering
				 * ranges' fills may live in the status block's counter area, cleared with it (MDB_ZERO_BLK_*) */
	bool r_based;		/* the compact window covers the right table's keys only: min-max pruning must run (gc_window.r_based) */
	bool defer_l64;		/* the same in the 64-bit form (keys that fit no 2^32 window): min-max pruning on the raw keys */
	bool defer_l;		/* the LEFT table is partitioned after the right one, in gc_finish (compact narrow form, unsplit call): the
				 * right table's first level records its exact key range, the left table's drops the rows outside */
	uint32_t semijoin;	/* != 0: the LEFT table is partitioned after the right one (gc_finish), its second level dropping the rows
				 * whose bit is clear in a bitmap of the right table's hashed keys; the value is log2 of the adjacent
				 * hashed values that share a bit, plus 1 */
	int b1, b2;
	mdb_part_result pl;
	/* further right tables on the same key (mdb_dev_join_group_count_multi) */
	int nextra;
	const int64_t *xkeys[GC_MAX_EXTRA];
	const uint64_t *xnull[GC_MAX_EXTRA];
	uint64_t xn[GC_MAX_EXTRA];
};

struct gc_extras {
	int n;
	const int64_t *keys[GC_MAX_EXTRA];
	const uint64_t *nulls[GC_MAX_EXTRA];
	uint64_t rows[GC_MAX_EXTRA];
};
static thread_local const gc_extras *gc_pending_extras = NULL;	/* set by mdb_dev_join_group_count_multi around its call of the operator */

/* a remembered "these columns overflow the digit-per-workgroup leaves' counts" (ctx->lw_bad_*) serves MDB_BAD_LEAF_USES calls */
static bool gc_lw_bad(mdb_dev_ctx *ctx, const gc_state *st)
{
	if (!(ctx->lw_bad_keys == st->keys_l && ctx->lw_bad_nl == st->n_l && ctx->lw_bad_nr == st->n_r_cap))
		return false;
	if (++ctx->lw_bad_uses > MDB_BAD_LEAF_USES) {
		ctx->lw_bad_keys = NULL;
		return false;
	}
	return true;
}

/* first half: size and claim the scratch arena, clear the status words, partition the left table */
static int gc_begin(mdb_dev_ctx *ctx, gc_state *st)
{
	mdb_choose_bits(st->n_l, GC_TARGET, &st->b1, &st->b2);
	const bool lw_bad = gc_lw_bad(ctx, st);
	if (st->r_based && !(st->has_r && st->defer_ok && !st->active && st->fast))
		st->key_bits = 0;	/* (a window of the right table's keys only needs the left table pruned: not in this call - plain narrow form) */
	/* key windows of 2^15 ... 2^23 values: one 9-bit level and k_leaf_wide (MDB_ONE_LEVEL=0 switches it off; tables of fewer
	 * than 2^21 rows in all keep the two-level form, whose fixed costs are smaller) */
	st->one_level = st->narrow && st->key_bits >= 9u + LW_MIN_REM && st->key_bits <= 9u + LW_MAX_REM && st->fast && st->want_records &&
			!ld_disabled() && !lw_bad &&
			st->n_l + (st->has_r ? st->n_r_cap : 0) >= (1ull << 21) &&
			st->n_l < 3000000000ull && st->n_r_cap < 3000000000ull &&
			!(mdb_knob("MDB_ONE_LEVEL") && mdb_knob("MDB_ONE_LEVEL")[0] == '0');
	if (st->nextra)
		st->one_level = false;	/* (further right tables: the two-level direct-address kernel counts them) */
	/* key windows of 2^24 ... 2^27 values (10^8 unique keys: variants U and S): what two 9-bit levels and k_leaf_direct did - the left
	 * table's 8-byte words written and read twice - is ONE 4096-digit pass per table (2-byte words for the right table, 4-byte row words
	 * for the left one) and k_leaf_wide12 over digits of up to 2^15 values.  Unsplit calls of a two-table join over int64 columns, at
	 * most 2^27 left rows (the leaf keeps a first row in 27 bits) and region capacities the leaf's registers hold; from 2^24 rows in all
	 * (measured equal to the two-level form at 8 * 10^6 rows per table, 15 % faster at 10^8: profiles/micro/one_pass_4096_sweep.py;
	 * MDB_WIDE12_MIN=<rows> moves the threshold, MDB_WIDE12=0 switches the form off) */
	{
		const char *e = mdb_knob("MDB_WIDE12"), *e2 = mdb_knob("MDB_WIDE12_MIN");
		const uint64_t min_rows = e2 && atoll(e2) > 0 ? (uint64_t)atoll(e2) : (1ull << 24);
		st->wide12 = !st->one_level && st->narrow && st->has_r && st->key_bits > 9u + LW_MAX_REM && st->key_bits <= 12u + LW_MAX_REM + 1u && st->fast &&
			     st->want_records && !ld_disabled() && st->defer_ok && !st->active && st->nextra <= 1 && !st->keys32 &&
			     !st->prunable /* (a right table that covers part of the left table's key range: min-max pruning drops most left rows first) */ &&
			     !lw_bad &&
			     st->n_l + st->n_r_cap >= min_rows && st->n_l <= (1ull << 27) /* (k_leaf_wide12 keeps a first row in 27 bits) */ &&
			     st->n_r_cap < 0xF0000000ull && !(e && e[0] == '0');
		/* ... and a digit's words must fit the leaf kernel's registers: 8 sub-regions of at most LW12_LB (LW12_RB) chunks per thread */
		if (st->wide12 && !leaf_wide12_fits(ctx, st->n_l, st->n_r_cap, st->nextra ? st->xn[0] : 0))
			st->wide12 = false;
	}
	if (st->wide12) {
		st->b1 = 12;
		st->b2 = 0;
	}
	if (st->one_level) {
		st->b1 = st->key_bits <= 15u ? 8 : 9;	/* (a 2^15-value window: 256 regions of 128 values - see gc_window.fast1) */
		st->b2 = 0;
	} else if (st->fast1) {
		st->fast = false;	/* (two fast levels: leaves of one or two values with thousands of rows each - regions would overflow) */
	}
	/* the narrow form of a join needs the right side's 4-byte layout (two fast levels); plain GROUP BY has no such limit */
	if (!st->one_level && !st->wide12 && st->narrow && st->has_r && !mdb_partition_w32_applies(st->n_r_cap, st->b1, st->b2, st->fast))
		st->narrow = false;
	/* compact narrow form + direct-address leaves: what the partition leaves of the key_bits-wide hash must index a table
	 * of at most 2^LD_MAX_REM entries; fewer than 2^4 would mean leaves of a handful of keys with thousands of rows each
	 * (same-address LDS atomics: the hashed kernel's wave-level merging handles those better) */
	st->direct = st->narrow && st->key_bits && st->fast && st->b2 > 0 && st->want_records && !ld_disabled() &&
		     st->key_bits >= (uint32_t)(st->b1 + st->b2) + 4u && st->key_bits <= (uint32_t)(st->b1 + st->b2) + LD_MAX_REM;
	if (st->one_level || st->wide12)
		st->direct = true;
	if (st->nextra && !(st->direct && st->has_r && st->fast))
		return GC_NOT_SERVED;
	if (st->direct && !st->one_level && !st->wide12) {
		/* the direct-address kernel has no table to overflow and pays a fixed price per leaf (three barriers, the emit scan):
		 * it prefers FEWER, larger leaves than the hashed kernel's 3833-slot table allows - tables of 2^LD_MAX_REM entries when
		 * the second level has the bits to give (10^8 x 10^8 rows: 2^15 leaves of 2 x 3052 rows instead of 2^16; second-level
		 * fan-out 128: -4 % on its scatter kernels as well).  MDB_LD_REM=<bits> overrides the target. */
		const char *e = mdb_knob("MDB_LD_REM");
		/* (further right tables take 4 more bytes of LDS per entry each: tables of 2^11 entries keep three workgroups on a CU -
		 * three tables of 10^8 rows: leaf kernel 0.72 -> 0.52 ms) */
		const uint32_t want = e && atoi(e) >= 4 && atoi(e) <= (int)LD_MAX_REM ? (uint32_t)atoi(e) : (st->nextra ? LD_MAX_REM - 1u : LD_MAX_REM);
		while (st->b2 > 1 && st->key_bits - (uint32_t)(st->b1 + st->b2) < want &&
		       (1u << (st->b1 + st->b2 - 1)) >= 16u * (uint32_t)ctx->num_cus)	/* ... while every workgroup still has leaves to walk */
			st->b2--;
	}
	/* two 9-bit levels give at most 2^18 leaves; a leaf's LDS hash table holds GC_SLOTS distinct keys.  The direct-address
	 * kernel has no such table (its leaves hold whatever rows share 2^rem key values): 10^9 x 10^9 rows of surrogate keys run
	 * on one GPU */
	if (!st->direct && (st->n_l >> (st->b1 + st->b2)) > (uint64_t)GC_SLOTS * 7 / 10)
		return mdb_set_err(ctx, -MIDORIDB_ERROR,
				   "%llu build rows exceed what one GPU shard groups in LDS (about %llu): partition the tables across GPUs "
				   "(mdb_dev_partition_by_dest)",
				   (unsigned long long)st->n_l, (unsigned long long)(((uint64_t)GC_SLOTS * 7 / 10) << (2 * MDB_MAX_RADIX_BITS)));
	/* Semi-join filter: when the key sample says that most left rows have no partner (a fact table joined with a
	 * dimension that covers part of its key range - the benchmark's variant D: 1 row in 16), the right table is
	 * partitioned FIRST, its hashed keys become a bitmap (k_leaf_bitmap: one contiguous slice per leaf), and the second
	 * partition level of the left table drops every row whose bit is clear before it is ranked and written: that level's
	 * writes and the leaf kernel's reads shrink to the rows that may have a partner.  The slice of a first-level digit
	 * (<= 32 KiB: one bit per 2^c adjacent hashed values when the window is wide) is staged in LDS per tile; the first level
	 * cannot filter - its rows are in table order, a lookup there costs a 128-byte line from L2 per row (measured: slower).
	 * A wrong hint costs time, never results.  MDB_SEMIJOIN=0 switches it off, MDB_SEMIJOIN_SLICE=<log2 bits> sizes the slice. */
	/* Min-max pruning, when the call is not split and the key sample says the right table's keys do not cover the left
	 * table's whole range (or a bitmap is wanted, which also needs the right table first): the right table goes first and its first partition level
	 * records the exact range of its keys (a pair of stores per tile, reduced by one small kernel); the left table's first level reads the two words
	 * from device memory and drops every row outside - the classic dimension-range pruning of a fact table, exact, at the
	 * price of one compare per row.  Where it removes most left rows (the right table's SPAN is small: by_span) the bitmap
	 * below would filter nothing more and is not built. */
	const char *prune_env = mdb_knob("MDB_MINMAX_PRUNE");		/* 0: never, 2: whatever the key sample says (tests) */
	st->defer_l = st->narrow && st->fast && (st->b2 > 0 || st->one_level) && st->has_r && st->defer_ok && !st->active &&
		      (st->prunable || (st->direct && st->selective) || (prune_env && prune_env[0] == '2')) && !(prune_env && prune_env[0] == '0');
	/* ... and in the 64-bit form (hashes, snowflake ids: keys beyond every 2^32 window) just the same - the range test does not
	 * care how wide the keys are: the right table's first level records the smallest and the largest KEY, the left table's drops
	 * the rows outside before they are hashed, ranked or written (12 bytes per row and level in this form) */
	st->defer_l64 = !st->narrow && st->fast && st->b2 > 0 && st->has_r && st->defer_ok && !st->active && !st->nextra && !st->keys32 &&
			(st->prunable || (prune_env && prune_env[0] == '2')) && !(prune_env && prune_env[0] == '0');
	st->semijoin = 0;
	if (st->defer_l && st->direct && !st->one_level && st->selective && !st->by_span) {
		const char *e = mdb_knob("MDB_SEMIJOIN"), *e2 = mdb_knob("MDB_SEMIJOIN_SLICE");
		const uint32_t slice_max = e2 && atoi(e2) >= 7 && atoi(e2) <= 18 ? (uint32_t)atoi(e2) : 17u;	/* log2 bits: 2^17 = 16 KiB */
		const uint32_t below0 = st->key_bits - (uint32_t)st->b1;		/* hash bits below the first-level digit */
		const uint32_t coarse = below0 > slice_max ? below0 - slice_max : 0u;
		const uint32_t rem = st->key_bits - (uint32_t)(st->b1 + st->b2);
		if (!(e && e[0] == '0') && coarse <= 3u && rem >= coarse + 5u && below0 - coarse >= 7u)
			st->semijoin = coarse + 1u;
	}
	size_t need = st->wide12 ? mdb_scatter4096_arena_bytes(ctx, st->n_l, true)
		      : st->one_level ? mdb_partition_level0_arena_bytes(st->n_l, st->b1, st->fast1) : mdb_partition_arena_bytes(st->n_l, st->b1, st->b2, true, st->fast);
	if (st->semijoin)
		need += mdb_align_up(((size_t)1 << (st->key_bits - (st->semijoin - 1u))) / 8) + 4096;
	if (st->one_level && st->has_r) {	/* (k_leaf_wide4's ranged emit: the ordering kernel's ranges, filled by the leaf kernel - tables of up to 1536 ranges) */
		const size_t ranges = (st->n_l >> ORDER_RANGE_BITS) + 2 < 1538 ? (size_t)(st->n_l >> ORDER_RANGE_BITS) + 2 : 1538;
		need += mdb_align_up(ranges * ORDER_RANGE_CAP * 8) + mdb_align_up(ranges * 4) + 512;
	}
	if (st->defer_l)
		need += mdb_align_up(mdb_part_minmax_words(st->n_r_cap) * 4) + 256;
	if (st->defer_l64)
		need += mdb_align_up(mdb_part_minmax_words(st->n_r_cap) * 8) + 256;
	if (st->has_r)
		need += st->wide12 ? mdb_scatter4096_arena_bytes(ctx, st->n_r_cap, false)
		      : st->one_level ? mdb_partition_level0_arena_bytes(st->n_r_cap, st->b1)
				      : mdb_partition_arena_bytes(st->n_r_cap, st->b1, st->b2, false, st->fast);
	for (int x = 0; x < st->nextra; x++)
		need += st->wide12 ? mdb_scatter4096_arena_bytes(ctx, st->xn[x], false) : mdb_partition_arena_bytes(st->xn[x], st->b1, st->b2, false, st->fast) + 512;
	{
		uint32_t kb = 0;
		int s1 = 0, s2 = 0;
		need += mdb_align_up(gc_rec_capacity(ctx, st->n_l) * 8) + mdb_align_up(st->n_l * 4) + 4096;
		need += mdb_align_up((size_t)HOT_MAX * HOT_SLOTS * 8) + mdb_align_up((size_t)HOT_MAX * (HOT_SLOTS + 1) * 8) +
			mdb_align_up((size_t)HOT_MAX * (HOT_SLOTS + 1) * 4) + 4096;	/* hot-key path (only touched when needed) */
		if (st->want_records && order_bits(st->n_l, &kb, &s1, &s2))
			need += mdb_partition_raw_arena_bytes(gc_rec_capacity(ctx, st->n_l), s1, s2, 1u << (kb - (uint32_t)(s1 + s2)), true,
							      order_digits0(st->n_l, kb, s1), (uint64_t)1 << (kb - (uint32_t)s1)) +
				mdb_partition_raw_arena_bytes(gc_rec_capacity(ctx, st->n_l), s1, s2, 1u << (kb - (uint32_t)(s1 + s2)), false, 0) +
				2 * (((size_t)1 << (s1 + s2)) + 4096) * 8;
		else
			need += mdb_filter_arena_bytes(st->n_l);
	}
	if (st->wide12 && st->nextra <= 1)	/* (the groups as one bit per left row + exceptions, mdb_dev_dense.hip) */
		need += mdb_dense_arena_bytes(st->n_l) + mdb_align_up((st->n_l / 8 + 4096) * 8);
	if (ctx->explain) {	/* (mdb_dev_explain_*: the plan is made - no arena, no launch) */
		ctx->explain->arena_mib = (uint32_t)((need + (1u << 20) - 1) >> 20);
		return GC_EXPLAINED;
	}
	int rc = mdb_arena_begin(ctx, need);
	if (rc)
		return rc;
	/* d_status u32 words: [0] flags (bit 0 leaf table overflow, bit 1 fast-layout region overflow, bit 2
	 * COUNT too large for a record), [1] record count, [2..3] joined rows (u64), [4..7] NULL-group stats */
	if (st->own_call && !st->wide12) {
		/* the status words and, behind them, the counters this call's kernels start from zero: one fill for all of them */
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, ((size_t)MDB_ZERO_BLK_OFF + MDB_ZERO_BLK_WORDS) * 4, ctx->stream));
	} else {
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	}
	if (!st->defer_l && !st->defer_l64 && !st->wide12) {
		mdb_part_filter lflt;
		memset(&lflt, 0, sizeof(lflt));
		lflt.level0_only = st->one_level;
		lflt.loose = st->one_level && st->fast1;
		if (st->own_call) {
			lflt.cursor0_ext = ctx->d_status + MDB_ZERO_BLK_OFF + MDB_ZERO_BLK_SLOT;
			lflt.cursor0_ext_words = MDB_ZERO_BLK_SLOT;
		}
		rc = mdb_partition_table(ctx, st->keys_l, st->null_l, st->n_l, st->b1, st->b2, !st->narrow, false, st->fast, &st->pl,
					 st->narrow ? 1 : 0, st->keys32, st->direct ? st->key_lo : st->base, st->direct ? st->key_bits : 0u,
					 st->one_level ? &lflt : NULL);
		if (rc)
			return rc;
	}
	st->active = true;
	return MIDORIDB_OK;
}

/* second half: partition the right table, join + count per leaf, order the groups, deliver */

/* ---- decisions that only a caller's statistics open (mdb_dev_call_stats), shared by the operator and by mdb_dev_explain_join_group_count */
/* one-level joins: the groups are few enough (a group needs a right key: at most as many as values in the right column's range) for the
 * ordering kernel's ranges - k_leaf_wide4 writes its records straight into them on the FIRST call */
static bool gc_ranged_by_stats(const mdb_dev_ctx *ctx, const int64_t *keys_l, const int64_t *keys_r, uint64_t n_l, uint64_t n_r, uint32_t kbits, uint32_t *rg_n)
{
	if (!(ctx->cs_on && !ctx->explain_as_sample && ctx->cs_kl == keys_l && ctx->cs_has_r && ctx->cs_kr == keys_r && ctx->cs_r.min <= ctx->cs_r.max) ||
	    (mdb_knob("MDB_ORDER_RANGES") && mdb_knob("MDB_ORDER_RANGES")[0] == '0'))
		return false;
	const uint64_t span_r = (uint64_t)ctx->cs_r.max - (uint64_t)ctx->cs_r.min + 1;
	uint64_t bound = span_r && span_r < n_r ? span_r : n_r;
	bound = bound < n_l ? bound : n_l;
	return order_ranges_apply(n_l, kbits, bound, rg_n);
}

/* the one-pass 4096-digit join of two tables: no key twice in the left column, none twice in the right one (MDB_COL_DISTINCT, measured at
 * ingest), and the right column holds EVERY value of its range, which covers the left column's - every left row is a group of COUNT 1,
 * whatever the statement before this one was: the groups leave as one bit per left row without a pilot launch and its host round trip */
static bool gc_bits_by_stats(const mdb_dev_ctx *ctx, int nextra, const int64_t *keys_l, const int64_t *keys_r, uint64_t n_r)
{
	return !nextra && ctx->dn_distrust == 0 && ctx->cs_on && !ctx->explain_as_sample && ctx->cs_kl == keys_l && ctx->cs_has_r && ctx->cs_kr == keys_r &&
	       (ctx->cs_l.flags & MDB_COL_DISTINCT) && (ctx->cs_r.flags & MDB_COL_DISTINCT) && !ctx->cs_r.nulls && ctx->cs_r.min <= ctx->cs_r.max &&
	       (uint64_t)ctx->cs_r.max - (uint64_t)ctx->cs_r.min + 1 == n_r && ctx->cs_l.min >= ctx->cs_r.min && ctx->cs_l.max <= ctx->cs_r.max;
}

/* whether the bit-per-row form of the groups may be asked at all (every left row must reach the leaf kernel for its bit to be looked at) */
static bool gc_bits_possible(const gc_state *st, bool records, bool has_r, uint64_t cap, uint64_t n_l)
{
	return st->wide12 && st->nextra <= 1 && records && has_r && cap && n_l >= ((uint64_t)1 << 22) && !st->null_l && !st->r_based &&
	       !(mdb_knob("MDB_JOIN_BITS") && mdb_knob("MDB_JOIN_BITS")[0] == '0');
}

/* mdb_dev_explain_join_group_count: the plan gc_begin has just made, as mdb_dev_last_plan would report it after the run */
static void gc_explain_fill(mdb_dev_ctx *ctx, const gc_state *st, const int64_t *keys_r, uint64_t n_r, uint64_t cap)
{
	struct mdb_dev_plan_info *o = ctx->explain;
	uint32_t kbits = 0, rg_n = 0;
	int sb1 = 0, sb2 = 0;
	const bool records = st->want_records && order_bits(st->n_l, &kbits, &sb1, &sb2);
	o->key_form = st->narrow ? (st->key_bits ? 2u : 1u) : 0u;
	o->key_bits = st->narrow ? st->key_bits : 0u;
	o->levels = (st->one_level || st->wide12) ? 1u : 2u;
	o->digits = st->wide12 ? 4096u : 512u;
	o->minmax_pruned = (st->defer_l || st->defer_l64) ? 1u : 0u;
	o->semijoin = st->semijoin;
	o->multi_one_pass = st->nextra ? 1u : 0u;
	o->from_stats = ctx->explain_as_sample ? 0u : ctx->pl_from_stats;
	o->samples = ctx->explain_as_sample ? 1u : 0u;
	const bool w16 = st->one_level && st->has_r && !(mdb_knob("MDB_WORDS16") && mdb_knob("MDB_WORDS16")[0] == '0');
	const bool leaf4 = st->one_level && st->has_r && w16 && records && !st->null_group && st->n_l <= (1ull << 27) && st->key_bits >= (uint32_t)st->b1 + 10u &&
			   !(mdb_knob("MDB_LEAF4") && mdb_knob("MDB_LEAF4")[0] == '0');
	o->ranged_order = leaf4 && gc_ranged_by_stats(ctx, st->keys_l, keys_r, st->n_l, n_r, kbits, &rg_n) ? 1u : 0u;
	if (gc_bits_possible(st, records, st->has_r, cap, st->n_l))
		o->groups_as_bits = gc_bits_by_stats(ctx, st->nextra, st->keys_l, keys_r, n_r) ? 3u : 1u /* (a pilot launch decides) */;
}

static int gc_finish(mdb_dev_ctx *ctx, gc_state *st, const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r,
		     int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap, uint64_t *out_groups,
		     uint64_t *out_joined)
{
	const int64_t *keys_l = st->keys_l;
	const uint64_t *null_l = st->null_l;
	const uint64_t n_l = st->n_l;
	const bool has_r = st->has_r, null_group = st->null_group, want_records = st->want_records;
	/* hash table on the side with fewer rows per leaf (the leaves are sized by the left table) */
	const bool build_r = has_r && !st->no_build_r && n_r <= n_l;
	mdb_part_result pl = st->pl, pr;
	int rc;

	st->active = false;
	memset(&pr, 0, sizeof(pr));
	if (st->wide12) {
		/* both tables through one 4096-digit pass each; a key outside the window is reported (the left table's, when the window was
		 * taken from the right table's keys alone, is dropped: it has no partner) */
		if (n_r > st->n_r_cap)
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "right table larger than announced at begin()");
		void *rb = NULL, *lb = NULL;
		uint32_t *rcur = NULL, *lcur = NULL;
		const uint32_t cap_r = mdb_scatter4096_cap(ctx, n_r, false), cap_l = mdb_scatter4096_cap(ctx, n_l, true);
		rc = mdb_scatter4096(ctx, keys_r, null_r, n_r, st->key_lo, st->key_bits, true, 0u, cap_r, false, "part_scatter_wide12_r", &rb, &rcur);
		if (rc)
			return rc;
		rc = mdb_scatter4096(ctx, keys_l, null_l, n_l, st->key_lo, st->key_bits, !st->r_based, (uint32_t)((1ull << st->key_bits) - 1ull), cap_l, true,
				     "part_scatter_wide12_l", &lb, &lcur);
		if (rc)
			return rc;
		memset(&pl, 0, sizeof(pl));
		pl.hv = (uint64_t *)lb;
		pl.leaf_cnt = lcur;
		pl.leaf_cap = cap_l;
		pr.hv = (uint64_t *)rb;
		pr.leaf_cnt = rcur;
		pr.leaf_cap = cap_r;
		pl.nleaves = pr.nleaves = 4096u;
		pl.bits_total = pr.bits_total = 12u;
		pl.nsub = pr.nsub = 8u;
		pl.w32 = pr.w32 = true;
		pr.w16 = true;
	}
	mdb_part_result px[GC_MAX_EXTRA];
	memset(px, 0, sizeof(px));
	if (st->wide12 && st->nextra) {
		/* the further right table like the first one; its keys outside the window are dropped, not reported: the window holds every key of
		 * the left table, so such a key joins nothing */
		void *xb = NULL;
		uint32_t *xcur = NULL;
		const uint32_t cap_x = mdb_scatter4096_cap(ctx, st->xn[0], false);
		rc = mdb_scatter4096(ctx, st->xkeys[0], st->xnull[0], st->xn[0], st->key_lo, st->key_bits, false, (uint32_t)((1ull << st->key_bits) - 1ull), cap_x, false,
				     "part_scatter_wide12_r", &xb, &xcur);
		if (rc)
			return rc;
		px[0].hv = (uint64_t *)xb;
		px[0].leaf_cnt = xcur;
		px[0].leaf_cap = cap_x;
	}
	if (has_r && !st->wide12) {
		if (n_r > st->n_r_cap)
			return mdb_set_err(ctx, -MIDORIDB_ERROR, "right table larger than announced at begin()");
		if (st->narrow && !st->one_level && !mdb_partition_w32_applies(n_r, st->b1, st->b2, st->fast))
			return GC_RETRY_WIDE;	/* split form: the left side was prepared narrow for a right table of another size */
		mdb_part_filter rflt;
		memset(&rflt, 0, sizeof(rflt));
		rflt.level0_only = st->one_level;
		rflt.out16 = st->one_level && !(mdb_knob("MDB_WORDS16") && mdb_knob("MDB_WORDS16")[0] == '0');
		if (st->own_call) {
			rflt.cursor0_ext = ctx->d_status + MDB_ZERO_BLK_OFF;
			rflt.cursor0_ext_words = MDB_ZERO_BLK_SLOT;
		}
		if (st->defer_l) {
			/* [16] smallest, [17] largest key - window base of the right table (min-max pruning) */
			rflt.minmax_out = ctx->d_status + GC_ST_MINMAX;
			rflt.minmax_tiles = (uint32_t *)mdb_arena_take(ctx, mdb_part_minmax_words(n_r) * 4);
			if (!rflt.minmax_tiles)
				return -MIDORIDB_INTERNAL;
		}
		if (st->defer_l64) {
			/* [GC_ST_MINMAX64 ..]: smallest, largest key of the right table as two signed 64-bit words */
			rflt.minmax64_out = reinterpret_cast<long long *>(ctx->d_status + GC_ST_MINMAX64);
			rflt.minmax64_tiles = (unsigned long long *)mdb_arena_take(ctx, mdb_part_minmax_words(n_r) * 8);
			if (!rflt.minmax64_tiles)
				return -MIDORIDB_INTERNAL;
		}
		rc = mdb_partition_table(ctx, keys_r, null_r, n_r, st->b1, st->b2, false, false, st->fast, &pr, st->narrow ? 2 : 0, st->keys32,
					 st->direct ? st->key_lo : st->base, st->direct ? st->key_bits : 0u,
					 (st->defer_l || st->one_level || st->defer_l64) ? &rflt : NULL);
		if (rc)
			return rc;
	}
	if (st->defer_l64) {
		mdb_part_filter flt;
		memset(&flt, 0, sizeof(flt));
		flt.range64_in = reinterpret_cast<const long long *>(ctx->d_status + GC_ST_MINMAX64);
		flt.expect_pruned = st->by_span;
		rc = mdb_partition_table(ctx, keys_l, null_l, n_l, st->b1, st->b2, true, false, st->fast, &st->pl, 0, st->keys32, st->base, 0u, &flt);
		if (rc)
			return rc;
		pl = st->pl;
	}
	if (st->nextra && !st->wide12) {
		/* the further right tables, partitioned exactly like the first: same window, same bits, 4-byte words.  Their keys
		 * outside the window are dropped, not reported: the window holds every key of the left table (or of the first right
		 * table, with the left one pruned to it), so such a key joins nothing */
		uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned) + 520;
		h[0] = 0u;
		h[1] = st->key_bits >= 32u ? 0xFFFFFFFFu : ((1u << st->key_bits) - 1u);
		MDB_HIP(ctx, hipMemcpyAsync(ctx->d_status + GC_ST_WINDOW, h, 8, hipMemcpyHostToDevice, ctx->stream));
		for (int x = 0; x < st->nextra; x++) {
			mdb_part_filter xf;
			memset(&xf, 0, sizeof(xf));
			xf.range_in = ctx->d_status + GC_ST_WINDOW;
			if (!mdb_partition_w32_applies(st->xn[x], st->b1, st->b2, st->fast))
				return GC_NOT_SERVED;
			rc = mdb_partition_table(ctx, st->xkeys[x], st->xnull[x], st->xn[x], st->b1, st->b2, false, false, st->fast, &px[x], 2, st->keys32,
						 st->key_lo, st->key_bits, &xf);
			if (rc)
				return rc;
			if (!px[x].w32 || !px[x].leaf_cap || !px[x].leaf_cnt || px[x].nleaves != (1u << (st->b1 + st->b2)))
				return GC_NOT_SERVED;
		}
	}
	if (st->defer_l && !st->semijoin) {
		mdb_part_filter flt;
		memset(&flt, 0, sizeof(flt));
		flt.range_in = ctx->d_status + GC_ST_MINMAX;
		flt.expect_pruned = st->by_span && !st->one_level;
		flt.level0_only = st->one_level;
		if (st->own_call) {
			flt.cursor0_ext = ctx->d_status + MDB_ZERO_BLK_OFF + MDB_ZERO_BLK_SLOT;
			flt.cursor0_ext_words = MDB_ZERO_BLK_SLOT;
		}
		rc = mdb_partition_table(ctx, keys_l, null_l, n_l, st->b1, st->b2, false, false, st->fast, &st->pl, 1, st->keys32,
					 st->direct ? st->key_lo : st->base, st->direct ? st->key_bits : 0u, &flt);
		if (rc)
			return rc;
		pl = st->pl;
	}
	if (st->semijoin) {
		/* the right table's hashed keys as a bitmap, then the left table through it */
		const uint32_t coarse = st->semijoin - 1u, rem = st->key_bits - (uint32_t)(st->b1 + st->b2);
		const size_t bytes = ((size_t)1 << (st->key_bits - coarse)) / 8;
		uint32_t *bits = (uint32_t *)mdb_arena_take(ctx, bytes);
		if (!bits)
			return -MIDORIDB_INTERNAL;
		if (!pr.leaf_cap || !pr.w32)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "semi-join filter: the right table is not in the 4-byte fixed-capacity layout");
		const uint32_t groups = (pr.nleaves + LB_WAVES - 1) / LB_WAVES, resident = 16u * (uint32_t)ctx->num_cus;
		MDB_LAUNCH(ctx, "leaf_bitmap", k_leaf_bitmap, groups < resident ? groups : resident, LB_WAVES * MDB_WAVE,
			   reinterpret_cast<const uint32_t *>(pr.hv), pr.leaf_cnt, pr.leaf_cap, pr.nleaves, rem, coarse, 32u - st->key_bits, bits);
		mdb_part_filter flt;
		memset(&flt, 0, sizeof(flt));
		flt.range_in = ctx->d_status + GC_ST_MINMAX;
		flt.expect_pruned = st->by_span;
		flt.bits = bits;
		flt.words = 1u << (st->key_bits - (uint32_t)st->b1 - coarse - 5u);
		flt.shift = 32u - st->key_bits + coarse;
		rc = mdb_partition_table(ctx, keys_l, null_l, n_l, st->b1, st->b2, false, false, st->fast, &st->pl, 1, st->keys32, st->key_lo, st->key_bits,
					 &flt);
		if (rc)
			return rc;
		pl = st->pl;
	}
	/* ---- result ordering: record mode (sort the groups by first row id) or dense mode (fallback) */
	uint32_t kbits = 0;
	int sb1 = 0, sb2 = 0;
	const bool records = want_records && order_bits(n_l, &kbits, &sb1, &sb2);
	const uint32_t ord_range = records ? (1u << (kbits - (uint32_t)(sb1 + sb2))) : 0;
	int64_t *dense = NULL;
	unsigned long long *rec = NULL;
	uint32_t *sel = (uint32_t *)mdb_arena_take(ctx, n_l * 4);
	if (records) {
		rec = (unsigned long long *)mdb_arena_take(ctx, gc_rec_capacity(ctx, n_l) * 8);
	} else {
		dense = (int64_t *)mdb_arena_take(ctx, n_l * 8);
	}
	if (!sel || (!rec && !dense))
		return -MIDORIDB_INTERNAL;
	/* d_status u32 words: [0] flags, [1] record-list length, [2..3] joined rows (u64), [4..7] NULL-group stats,
	 * [8] number of records (groups) */
	uint32_t *d_rec_count = ctx->d_status + 1;
	uint32_t *d_rec_valid = ctx->d_status + 8;
	unsigned long long *d_joined = (unsigned long long *)(ctx->d_status + 2);
	unsigned long long *d_nullst = (unsigned long long *)(ctx->d_status + 4);
	if (dense)
		MDB_HIP(ctx, hipMemsetAsync(dense, 0, n_l * 8, ctx->stream));

	gc_args a;
	a.hv_l = pl.hv;
	a.rid_l = pl.rid;
	a.off_l = pl.leaf_off;
	a.cnt_l = pl.leaf_cnt;
	a.cap_l = pl.leaf_cap;
	a.hv_r = pr.hv;
	a.off_r = pr.leaf_off;
	a.cnt_r = pr.leaf_cnt;
	a.cap_r = pr.leaf_cap;
	a.dense_cnt = dense;
	a.dense_n = (uint32_t)n_l;
	a.rec = rec;
	a.rec_count = d_rec_count;
	a.rec_valid = d_rec_valid;
	a.rec_cap = records ? (uint32_t)gc_rec_capacity(ctx, n_l) : 0;
	a.kbits = records ? kbits : 0;
	a.joined = d_joined;
	a.status = ctx->d_status;
	a.nleaves = pl.nleaves;
	a.nextra = (uint32_t)st->nextra;
	for (int x = 0; x < GC_MAX_EXTRA; x++) {
		a.hv_x[x] = x < st->nextra ? reinterpret_cast<const uint32_t *>(px[x].hv) : NULL;
		a.cnt_x[x] = x < st->nextra ? px[x].leaf_cnt : NULL;
		a.cap_x[x] = x < st->nextra ? px[x].leaf_cap : 0u;
	}
	a.narrow = st->narrow ? 1u : 0u;
	/* keyed records (see gc_args.keyed_cbits) when the join is selective - few groups, their first rows scattered over the left
	 * table - and a COUNT(*) of at least 4 bits fits beside the row id and the hashed key (10 bits at 10^8 rows, 27 key bits);
	 * a COUNT that does not fit is reported by the kernel and the operator redone with plain records (and remembered).
	 * MDB_KEYED_RECORDS=0 switches them off */
	uint32_t keyed_cbits = 0;
	if (st->direct && has_r && !st->nextra && st->selective && records && !ctx->keyed_distrust && pl.leaf_cap && pr.leaf_cap && kbits >= 13 &&
	    kbits + st->key_bits + 4u <= 64u && !(mdb_knob("MDB_KEYED_RECORDS") && mdb_knob("MDB_KEYED_RECORDS")[0] == '0'))
		keyed_cbits = 64u - kbits - st->key_bits;
	if (ctx->keyed_distrust > 0)
		ctx->keyed_distrust--;
	a.keyed_cbits = keyed_cbits;
	/* 4-byte records straight from the leaf kernel when the last run over these very columns saw every COUNT(*) fit beside the
	 * row id (variant U: 10^8 records - 0.4 GB less to write and 0.4 GB less for the ordering sort to read) */
	const bool r32_same = ctx->r32_ok && ctx->r32_kl == keys_l && ctx->r32_nl == n_l && ctx->r32_kr == keys_r && ctx->r32_nr == n_r;
	a.rec32 = (r32_same && st->direct && !st->one_level && has_r && records && !keyed_cbits && kbits < 32 && sb2 > 0 && pl.leaf_cap && pr.leaf_cap) ? 1u : 0u;
	/* 4096 sampled keys with fewer than 4050 distinct values among them: at most a few 10^5 distinct values in the column */
	a.merge_all = (!has_r && ctx->gh_keys == keys_l && ctx->gh_n == n_l && ctx->gh_distinct < 4050u) ? 1u : 0u;
	{
		/* hot = far above the side's average leaf: a much larger probe table spread evenly over the leaves is not skew */
		const uint64_t avg_l = n_l / (pl.nleaves ? pl.nleaves : 1), avg_r = has_r ? n_r / (pl.nleaves ? pl.nleaves : 1) : 0;
		const uint64_t hl = 8 * avg_l > GC_HEAVY ? 8 * avg_l : GC_HEAVY, hr = 8 * avg_r > GC_HEAVY ? 8 * avg_r : GC_HEAVY;
		a.heavy_l = hl > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)hl;
		a.heavy_r = hr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)hr;
	}
	/* one-level joins with 2-byte right words: k_leaf_wide4's 4 bytes per key value (two workgroups per CU) unless these columns are known to
	 * hold more than 31 right or 15 left rows of a key (MDB_LEAF4=0 switches it off) */
	const bool leaf4 = st->one_level && has_r && pr.w16 && records && !null_group && n_l <= (1ull << 27) && st->key_bits >= (uint32_t)pl.bits_total + 10u &&
			   !(ctx->l4_bad_keys == keys_l && ctx->l4_bad_nl == n_l && ctx->l4_bad_nr == n_r && ++ctx->l4_bad_uses <= MDB_BAD_LEAF_USES) &&
			   !(mdb_knob("MDB_LEAF4") && mdb_knob("MDB_LEAF4")[0] == '0');
	/* ... and when the last join over these very columns told how many groups to expect, and they are few enough for the ordering kernel's
	 * ranges of 2^16 row ids, k_leaf_wide4 writes its records straight into those ranges: no record list, none of the two scatter levels that
	 * would partition it by row id (MDB_ORDER_RANGES=0 switches it off) */
	uint32_t rg_n = 0;
	bool ranged = leaf4 && ctx->lg_valid && !ctx->lg_nextra && ctx->lg_kl == keys_l && ctx->lg_nl == n_l && ctx->lg_kr == keys_r && ctx->lg_nr == n_r &&
		      order_ranges_apply(n_l, kbits, ctx->lg_groups + ctx->lg_groups / 8, &rg_n) &&
		      !(mdb_knob("MDB_ORDER_RANGES") && mdb_knob("MDB_ORDER_RANGES")[0] == '0');
	/* ... or the caller's statistics say so before any join has run (mdb_dev_call_stats): a group needs a right key, and there are at
	 * most as many of those as values in the right column's range */
	if (!ranged && leaf4)
		ranged = gc_ranged_by_stats(ctx, keys_l, keys_r, n_l, n_r, kbits, &rg_n);
	a.rg_rec = NULL;
	a.rg_cnt = NULL;
	a.rg_cap = a.rg_shift = a.rg_n = 0;
	if (ranged) {
		a.rg_rec = (unsigned long long *)mdb_arena_take(ctx, (size_t)rg_n * ORDER_RANGE_CAP * 8);
		const bool rg_in_block = st->own_call && rg_n <= MDB_ZERO_BLK_WORDS - 2u * MDB_ZERO_BLK_SLOT;
		a.rg_cnt = rg_in_block ? ctx->d_status + MDB_ZERO_BLK_OFF + 2u * MDB_ZERO_BLK_SLOT : (uint32_t *)mdb_arena_take(ctx, (size_t)rg_n * 4);
		if (!a.rg_rec || !a.rg_cnt)
			return -MIDORIDB_INTERNAL;
		/* (the arena hands out whole 256-byte units: cleared as such - a length that is no multiple of 16 bytes costs the runtime a second fill
		 * kernel, 5 us of the step) */
		if (!rg_in_block)
			MDB_HIP(ctx, hipMemsetAsync(a.rg_cnt, 0, mdb_align_up((size_t)rg_n * 4), ctx->stream));
		a.rg_cap = ORDER_RANGE_CAP;
		a.rg_shift = ORDER_RANGE_BITS;
		a.rg_n = rg_n;
	}
	/* The last join over these columns made nearly every left row a group of COUNT 1 (a primary key joined with another, or with its
	 * foreign keys - BASELINE configs[2]'s variant U): k_leaf_wide12 then clears one bit per left row that is NO group's first row and
	 * lists the groups whose COUNT is not 1, instead of a record per group and the ordering sort of 10^8 records (MDB_JOIN_BITS=0: never) */
	unsigned long long *dn_bits = NULL;
	a.dn_bits = NULL;
	a.dn_exc = NULL;
	a.dn_exc_cap = 0;
	a.dn_pilot = 0;
	a.dn_cnt = ctx->d_status + 12;
	/* (every left row must reach the leaf kernel for its bit to be looked at: no NULL keys, no window that covers the right table's keys
	 * only - left rows outside it are dropped by the first level; rows dropped for another reason show as G + cleared != n_l below and
	 * send the call to the record form) */
	if (gc_bits_possible(st, records, has_r, cap, n_l)) {
		bool want_bits = false, by_pilot = false, by_stats = false;
		/* The catalog says (MDB_COL_DISTINCT, measured at ingest): no key twice in the left column, none twice in the right one, and the
		 * right column holds EVERY value of its range, which covers the left column's - every left row is a group of COUNT 1, whatever
		 * the statement before this one was: no pilot launch and its host round trip, nothing remembered by the columns' addresses.
		 * (Two tables only; a promise that does not hold shows below as it does for the other two ways to get here: G + cleared != n_l,
		 * or more exceptions than the list holds - the call is redone with records.) */
		if (gc_bits_by_stats(ctx, st->nextra, keys_l, keys_r, n_r)) {
			want_bits = true;
			by_stats = true;
		} else if (ctx->lg_valid && ctx->lg_nextra == (uint32_t)st->nextra && ctx->lg_kl == keys_l && ctx->lg_nl == n_l && ctx->lg_kr == keys_r && ctx->lg_nr == n_r) {
			/* (what the last join over these very columns delivered) */
			if (ctx->lg_groups >= n_l - n_l / 16 && ctx->lg_joined <= ctx->lg_groups + ctx->lg_groups / 16) {
				if (ctx->dn_distrust > 0)
					ctx->dn_distrust--;
				else
					want_bits = true;
			}
		} else {
			/* nothing remembered (a first statement): the pilot - the same kernel over the first 64 of the 4096 digits (all rows of a key are
			 * in one digit: a fair sample of the keys), nothing written but the counters: left rows that are no group's first row, groups
			 * whose COUNT is not 1.  One in 16 of the rows at most each: the bit-per-row form */
			a.dn_pilot = 64;
			if ((rc = leaf_wide12_launch(ctx, a, pl.nleaves, st->key_bits - 12u, pl.nsub, st->nextra)))
				return rc;
			a.dn_pilot = 0;
			uint64_t *hp = ctx->h_pinned;
			MDB_HIP(ctx, hipMemcpyAsync(&hp[1], ctx->d_status, 56, hipMemcpyDeviceToHost, ctx->stream));
			MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
			const uint32_t *pw = reinterpret_cast<const uint32_t *>(&hp[1]);
			const uint64_t p_groups = pw[8], p_cleared = pw[12], p_exc = pw[13], p_rows = p_groups + p_cleared;
			want_bits = p_rows && p_cleared * 16 <= p_rows && p_exc * 16 <= p_rows;
			by_pilot = true;
			if (mdb_knob("MDB_DEBUG_GROUP"))
				fprintf(stderr, "join + GROUP BY (pilot over 64 digits): %llu left rows, %llu no group's first row, %llu groups of COUNT != 1 -> %s\n",
					(unsigned long long)p_rows, (unsigned long long)p_cleared, (unsigned long long)p_exc, want_bits ? "a bit per left row" : "records");
			MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, 13 * sizeof(uint32_t), ctx->stream));	/* (the flags of word 0 stay: they are facts about the data) */
		}
		if (want_bits) {
			if ((rc = mdb_dense_bits_begin(ctx, n_l, &dn_bits)))
				return rc;
			a.dn_bits = reinterpret_cast<unsigned int *>(dn_bits);
			a.dn_exc_cap = (uint32_t)(n_l / 8 + 4096);
			a.dn_exc = (unsigned long long *)mdb_arena_take(ctx, (size_t)a.dn_exc_cap * 8);
			if (!a.dn_exc)
				return -MIDORIDB_INTERNAL;
			ctx->pl_bits = by_stats ? 3u : by_pilot ? 1u : 2u;
		}
	}
	{
		/* persistent grid: two 75 KiB workgroups fit one CU's 160 KiB of LDS */
		const uint32_t resident = 2u * (uint32_t)ctx->num_cus;
		const uint32_t grid = pl.nleaves < resident ? pl.nleaves : resident;
		/* (the direct kernel addresses leaf i at i * cap: should a table have fallen back to exact offsets - more than 2^32
		 * region words - the hashed kernel below joins the compact words just as well, they are injective too) */
		if (st->wide12) {
			if ((rc = leaf_wide12_launch(ctx, a, pl.nleaves, st->key_bits - 12u, pl.nsub, st->nextra)))
				return rc;
		} else if (st->one_level) {
			/* ... by ALL the hash bits below the first level's 9 (k_leaf_wide) */
			if (!pl.nsub || !pl.leaf_cap || (has_r && (!pr.nsub || !pr.w32 || pr.nsub != pl.nsub || pr.nleaves != pl.nleaves)))
				return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "one-level direct leaves: the tables are not in the first-level layout");
			const uint32_t rem = st->key_bits - pl.bits_total, shift = 32u - st->key_bits;
			if (has_r && pr.w16 && leaf4)
				rc = leaf_wide4_launch(ctx, a, pl.nleaves, rem, shift, pl.nsub);
			else
				rc = leaf_wide_launch(ctx, a, pl.nleaves, rem, shift, pl.nsub, has_r, has_r && pr.w16);
			if (rc)
				return rc;
		} else if (st->direct && pl.leaf_cap && (!has_r || pr.leaf_cap)) {
			/* compact narrow form: the leaf's table is indexed by the hash bits the partition left over */
			const uint32_t rem = st->key_bits - pl.bits_total, shift = 32u - st->key_bits;
			if (st->nextra && !(pl.leaf_cap && pr.leaf_cap))
				return GC_NOT_SERVED;
			const size_t lds = ((size_t)(12 + 4 * st->nextra) << rem) + 32;
			/* four 512-thread workgroups = the CU's 32 waves (three measured the same, 256-thread workgroups 10-30 % slower) */
			uint32_t per_cu = (uint32_t)((size_t)(160 * 1024) / lds);
			per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
			const uint32_t dgrid = pl.nleaves < per_cu * (uint32_t)ctx->num_cus ? pl.nleaves : per_cu * (uint32_t)ctx->num_cus;
			if (has_r) {
				MDB_LAUNCH_LDS(ctx, "leaf_join_direct", (k_leaf_direct<true, 512, 2048>), dgrid, 512, lds, a, rem, shift);
			} else {
				MDB_LAUNCH_LDS(ctx, "leaf_group_direct", (k_leaf_direct<false, 512, 2048>), dgrid, 512, lds, a, rem, shift);
			}
		} else if (st->nextra) {
			return GC_NOT_SERVED;
		} else if (has_r && build_r && st->narrow) {
			MDB_LAUNCH(ctx, "leaf_join_group_count", (k_leaf_group_count<true, true, false, true>), grid, GC_THREADS, a);
		} else if (has_r && build_r) {
			MDB_LAUNCH(ctx, "leaf_join_group_count", (k_leaf_group_count<true, true, false>), grid, GC_THREADS, a);
		} else if (has_r && st->narrow) {
			MDB_LAUNCH(ctx, "leaf_join_group_count", (k_leaf_group_count<true, false, false, true>), grid, GC_THREADS, a);
		} else if (has_r) {
			MDB_LAUNCH(ctx, "leaf_join_group_count", (k_leaf_group_count<true, false, false>), grid, GC_THREADS, a);
		} else if (st->narrow) {
			MDB_LAUNCH(ctx, "leaf_group_count", (k_leaf_group_count<false, false, false, true>), grid, GC_THREADS, a);
		} else {
			MDB_LAUNCH(ctx, "leaf_group_count", (k_leaf_group_count<false, false, false>), grid, GC_THREADS, a);
		}
	}
	if (null_group && null_l) {
		MDB_HIP(ctx, hipMemsetAsync(d_nullst + 1, 0xFF, 8, ctx->stream));
		MDB_LAUNCH(ctx, "null_stats", k_null_stats, 256, 256, null_l, n_l, d_nullst);
		if (records) {
			MDB_LAUNCH(ctx, "null_rec", k_null_rec, 1, 64, d_nullst, rec, d_rec_count, d_rec_valid, a.rec_cap, kbits, ctx->d_status);
		} else {
			MDB_LAUNCH(ctx, "null_poke", k_null_poke, 1, 64, d_nullst, dense);
		}
	}

	/* G, J and the status flags come back with one sync; the sort is sized by G */
	uint64_t *h = ctx->h_pinned;
	uint64_t G = 0, list_len = 0;
	uint32_t *first_out = out_first ? out_first : sel;
	/* ... except where the leaf kernel has filled the ordering kernel's ranges itself: that kernel needs no size, and is launched right behind it
	 * (its writes bounded by the caller's capacity; should the status words ask for another path, that path writes the columns again) - the
	 * step's only sync then comes after its last kernel (MDB_ORDER_EARLY=0: after the leaf kernel, as elsewhere) */
	bool ordered_early = false;
	if (ranged && cap && !(mdb_knob("MDB_ORDER_EARLY") && mdb_knob("MDB_ORDER_EARLY")[0] == '0')) {
		rc = order_presorted(ctx, a.rg_rec, a.rg_cnt, rg_n, kbits, out_first, out_count, keys_l, out_key, st->keys32, keyed_cbits, st->key_bits,
				     st->key_lo, cap);
		if (rc)
			return rc;
		ordered_early = true;
	}
	MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 56, hipMemcpyDeviceToHost, ctx->stream));	/* (words 12, 13: the bit-per-row form's counters) */
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const uint32_t dn_cleared = reinterpret_cast<const uint32_t *>(&h[1])[12], dn_exceptions = reinterpret_cast<const uint32_t *>(&h[1])[13];
	if (leaf4 && ((uint32_t)h[1] & (4096u | 8192u)) && !((uint32_t)h[1] & (2u | 128u))) {
		/* a key with more rows than k_leaf_wide4's count fields hold: the same partitioned tables through k_leaf_wide (16-bit counts), now
		 * and for these columns; a range of row ids with more groups than its region holds (more groups than last time, or bunched):
		 * k_leaf_wide4 again, into the record list */
		const bool counts_bad = ((uint32_t)h[1] & 4096u) != 0;
		if (counts_bad) {
			ctx->l4_bad_keys = keys_l;
			ctx->l4_bad_uses = 0;
			ctx->l4_bad_nl = n_l;
			ctx->l4_bad_nr = n_r;
		}
		ranged = false;
		a.rg_rec = NULL;
		ctx->lg_valid = false;
		uint32_t *hw = reinterpret_cast<uint32_t *>(ctx->h_pinned) + 528;
		hw[0] = (uint32_t)h[1] & ~(4096u | 8192u | 256u | 16u | 8u);
		MDB_HIP(ctx, hipMemcpyAsync(ctx->d_status, hw, 4, hipMemcpyHostToDevice, ctx->stream));
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, 12, ctx->stream));	/* record-list length, joined rows */
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 8, 0, 8, ctx->stream));	/* records, largest first row */
		if ((rc = counts_bad ? leaf_wide_launch(ctx, a, pl.nleaves, st->key_bits - pl.bits_total, 32u - st->key_bits, pl.nsub, true, true)
				     : leaf_wide4_launch(ctx, a, pl.nleaves, st->key_bits - pl.bits_total, 32u - st->key_bits, pl.nsub)))
			return rc;
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (!counts_bad && ((uint32_t)h[1] & 4096u) && !((uint32_t)h[1] & (2u | 128u))) {	/* (the counts overflow as well: seen only now) */
			ctx->l4_bad_keys = keys_l;
			ctx->l4_bad_uses = 0;
			ctx->l4_bad_nl = n_l;
			ctx->l4_bad_nr = n_r;
			hw[0] = (uint32_t)h[1] & ~(4096u | 8192u | 256u | 16u | 8u);
			MDB_HIP(ctx, hipMemcpyAsync(ctx->d_status, hw, 4, hipMemcpyHostToDevice, ctx->stream));
			MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, 12, ctx->stream));
			MDB_HIP(ctx, hipMemsetAsync(ctx->d_status + 8, 0, 8, ctx->stream));
			if ((rc = leaf_wide_launch(ctx, a, pl.nleaves, st->key_bits - pl.bits_total, 32u - st->key_bits, pl.nsub, true, true)))
				return rc;
			MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
			MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		}
	}
	if ((uint32_t)h[1] & 128u)
		return st->direct ? GC_RETRY_PLAIN : GC_RETRY_WIDE;	/* a key outside the window: the 32-bit hashes mean nothing */
	if (st->nextra && ((uint32_t)h[1] & (2u | 64u | 2048u)))
		return GC_NOT_SERVED;	/* skewed keys, hot leaves, a product of counts beyond 32 bits: the chain of two-table operators */
	if ((uint32_t)h[1] & 2u)
		return GC_RETRY_EXACT;	/* a leaf outgrew its fixed-capacity region (skewed keys) */
	if ((uint32_t)h[1] & 1024u) {
		ctx->lw_bad_uses = 0;
		ctx->lw_bad_keys = keys_l;	/* a key with 2^16 or more rows on one side: two levels and their hot-key path, now and for these columns */
		ctx->lw_bad_nl = n_l;
		ctx->lw_bad_nr = st->n_r_cap;
		return GC_RETRY_TWO_LEVEL;
	}
	if (((uint32_t)h[1] & 64u) && keyed_cbits) {
		ctx->keyed_distrust = 64;	/* hot leaves go through kernels that write plain records: redo with those everywhere */
		return GC_RETRY_UNKEYED;
	}
	if (a.rec32 && ((uint32_t)h[1] & (64u | 512u))) {
		ctx->r32_ok = false;		/* a COUNT(*) that no longer fits, or hot leaves (their kernels write 8-byte records) */
		return GC_RETRY_REC64;
	}
	if ((uint32_t)h[1] & 64u) {
		/* hot keys: the plain kernel left the leaves with GC_HEAVY or more rows on a side to this path */
		hot_args ha;
		ha.count = (uint32_t *)mdb_arena_take(ctx, 64);
		ha.leaf = (uint32_t *)mdb_arena_take(ctx, HOT_MAX * 4);
		ha.g_key = (unsigned long long *)mdb_arena_take(ctx, (size_t)HOT_MAX * HOT_SLOTS * 8);
		ha.g_cnt = (unsigned long long *)mdb_arena_take(ctx, (size_t)HOT_MAX * (HOT_SLOTS + 1) * 8);
		ha.g_first = (uint32_t *)mdb_arena_take(ctx, (size_t)HOT_MAX * (HOT_SLOTS + 1) * 4);
		if (!ha.count || !ha.leaf || !ha.g_key || !ha.g_cnt || !ha.g_first)
			return -MIDORIDB_INTERNAL;
		MDB_HIP(ctx, hipMemsetAsync(ha.count, 0, 4, ctx->stream));
		if (has_r) {
			MDB_LAUNCH(ctx, "hot_list", k_hot_list<true>, (pl.nleaves + 255) / 256, 256, a, ha);
		} else {
			MDB_LAUNCH(ctx, "hot_list", k_hot_list<false>, (pl.nleaves + 255) / 256, 256, a, ha);
		}
		MDB_HIP(ctx, hipMemcpyAsync(&h[9], ha.count, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		const uint32_t nhot = (uint32_t)h[9];
		const uint32_t resident = 2u * (uint32_t)ctx->num_cus;
		if (nhot <= HOT_MAX) {
			MDB_LAUNCH(ctx, "hot_clear", k_hot_clear, 256, 256, ha);
			if (has_r) {
				MDB_LAUNCH(ctx, "hot_slices", k_hot_slices<true>, resident, GC_THREADS, a, ha);
				MDB_LAUNCH(ctx, "hot_finish", k_hot_finish<true>, nhot, 1024, a, ha);
			} else {
				MDB_LAUNCH(ctx, "hot_slices", k_hot_slices<false>, resident, GC_THREADS, a, ha);
				MDB_LAUNCH(ctx, "hot_finish", k_hot_finish<false>, nhot, 1024, a, ha);
			}
		} else {
			/* very many hot leaves: each is streamed by one workgroup (the HEAVY instance of the leaf kernel) */
			const uint32_t grid = pl.nleaves < resident ? pl.nleaves : resident;
			if (has_r && build_r) {
				MDB_LAUNCH(ctx, "leaf_hot_keys", (k_leaf_group_count<true, true, true>), grid, GC_THREADS, a);
			} else if (has_r) {
				MDB_LAUNCH(ctx, "leaf_hot_keys", (k_leaf_group_count<true, false, true>), grid, GC_THREADS, a);
			} else {
				MDB_LAUNCH(ctx, "leaf_hot_keys", (k_leaf_group_count<false, false, true>), grid, GC_THREADS, a);
			}
		}
		MDB_HIP(ctx, hipMemcpyAsync(&h[1], ctx->d_status, 40, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	if (records) {
		list_len = h[1] >> 32;
		G = (uint32_t)h[5];
	} else {
		uint32_t *d_total = NULL;
		rc = mdb_filter_nonzero64(ctx, dense, n_l, sel, &d_total);
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemcpyAsync(&h[0], d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		G = (uint32_t)h[0];
	}
	const uint32_t status = (uint32_t)h[1];
	const uint64_t joined = h[2];
	if (status & 2u)
		return GC_RETRY_EXACT;	/* a leaf outgrew its fixed-capacity region (skewed keys) */
	if (status & 256u) {
		ctx->keyed_distrust = 64;	/* a COUNT(*) too large for a keyed record: plain records, now and for a while */
		return GC_RETRY_UNKEYED;
	}
	if (status & 4u)
		return GC_RETRY_DENSE;	/* a COUNT(*) too large to share a 64-bit record with its row id */
	if ((status & 1u) && build_r)
		return GC_RETRY_BUILD_L;
	if (status & 1u)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL,
				   "leaf hash table overflow (more than %u distinct keys in one leaf): unsupported key skew", GC_SLOTS);
	if (status & 8u)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "group record list exhausted (sizing bug)");
	if (G > cap)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups",
				   (unsigned long long)cap, (unsigned long long)G);
	if (G && records) {
		/* (one-level leaves report the largest first row id: the sort's first-level regions are sized for the digits below it) */
		const uint64_t last_first = (uint32_t)(h[5] >> 32);
		const uint64_t n_ord = (st->one_level && last_first && last_first < n_l) ? last_first + 1 : n_l;
		if (dn_bits) {
			if (mdb_knob("MDB_DEBUG_GROUP"))
				fprintf(stderr, "join + GROUP BY (bit per left row): %llu groups, %u rows cleared of %llu, %u exceptions, status %u\n",
					(unsigned long long)G, dn_cleared, (unsigned long long)n_l, dn_exceptions, status);
			if ((status & 131072u) || (uint64_t)G + dn_cleared != n_l) {	/* (more groups of COUNT != 1 than last time: the record form) */
				ctx->dn_distrust = 32;
				ctx->pl_bits = 0;
				return GC_RETRY_NODENSE;
			}
			/* every left row a group: the group keys ARE the left key column, in its order - a caller that said so (MDB_KEYS_MAY_ALIAS) reads
			 * them there and nothing is copied (0.8 GB read + 0.8 GB written at 10^8 rows); every COUNT 1 and MDB_COUNTS_OPTIONAL: no COUNT
			 * column either (mdb_dev_last_plan says which) */
			const bool alias = ctx->key_alias_ok && !st->keys32 && out_key && G == n_l;
			const bool ones = ctx->counts_optional && dn_exceptions == 0;
			ctx->pl_keys_left = alias ? 1u : 0u;
			ctx->pl_counts_one = ones ? 1u : 0u;
			rc = mdb_dense_emit(ctx, dn_bits, n_l, a.dn_exc, dn_exceptions, out_first, ones ? NULL : out_count, keys_l, st->keys32, alias ? NULL : out_key);
			if (!rc)
				MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		} else if (ranged && ordered_early)
			rc = MIDORIDB_OK;	/* (done, and waited for with the status words) */
		else if (ranged)
			rc = order_presorted(ctx, a.rg_rec, a.rg_cnt, rg_n, kbits, out_first, out_count, keys_l, out_key, st->keys32, keyed_cbits, st->key_bits,
					     st->key_lo);
		else
			rc = order_records(ctx, rec, list_len, n_ord, kbits, sb1, sb2, out_first, out_count, NULL, keys_l, out_key, st->keys32,
					   !(status & 16u), keyed_cbits, st->key_bits, st->key_lo, a.rec32 != 0, G);
		if (rc == GC_RETRY_REC64)
			ctx->r32_ok = false;
		if (rc)
			return rc;
		/* remember whether 4-byte records would do for these columns */
		ctx->r32_ok = st->direct && !st->one_level && has_r && !keyed_cbits && !(status & 16u);
		ctx->r32_kl = keys_l;
		ctx->r32_nl = n_l;
		ctx->r32_kr = keys_r;
		ctx->r32_nr = n_r;
	} else if (G) {
		rc = mdb_dev_gather64(ctx, dense, NULL, sel, G, out_count, NULL);
		if (rc)
			return rc;
		if (out_key && st->keys32) {
			MDB_LAUNCH(ctx, "gather_i32", k_gather_i32, (uint32_t)((G + 255) / 256), 256, reinterpret_cast<const int32_t *>(keys_l), sel, G,
				   out_key);
		} else if (out_key) {
			rc = mdb_dev_gather64(ctx, keys_l, NULL, sel, G, out_key, NULL);
			if (rc)
				return rc;
		}
		if (out_first)
			MDB_HIP(ctx, hipMemcpyAsync(out_first, sel, G * 4, hipMemcpyDeviceToDevice, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	*out_groups = G;
	if (out_joined)
		*out_joined = joined;
	if (has_r) {
		ctx->lg_nextra = (uint32_t)st->nextra;
		ctx->lg_kl = keys_l;
		ctx->lg_nl = n_l;
		ctx->lg_kr = keys_r;
		ctx->lg_nr = n_r;
		ctx->lg_groups = G;
		ctx->lg_joined = joined;
		ctx->lg_valid = true;
	}
	ctx->last_narrow = st->direct ? 2 : (st->narrow ? 1 : 0);
	ctx->last_left_dups_known = st->direct && has_r && !(status & 64u);	/* (the hot-key kernels and the hashed leaves do not say) */
	ctx->last_left_dups = (status & GC_ST_LEFT_DUPS) != 0;
	ctx->last_semijoin = (int)st->semijoin | ((st->defer_l || st->defer_l64) ? 0x100 : 0) | (st->one_level ? 0x200 : 0) | (st->nextra ? 0x400 : 0) |
			     (st->wide12 ? 0x1000 : 0) | (ranged ? 0x2000 : 0);
	return MIDORIDB_OK;
}

/* ---- narrow form: 32-bit hashes when every key of both tables lies inside the int32 range -------------------------
 *
 * The reference's integers are 32-bit in practice (literals go through atoi, compares through int: SURVEY D5), and a
 * key column of such values does not need 8-byte hashes plus 4-byte row ids on its way through the two partition
 * levels: the left side travels as ONE word (hash32 << 32 | row id), 8 instead of 12 bytes per row and pass.  Nothing
 * is assumed: the first partition level checks every key it reads and raises status bit 7 on the first one outside the
 * range, and the operator is then redone with 64-bit hashes (GC_RETRY_WIDE).  To keep that retry for adversarial
 * inputs only, 2 x 4096 evenly spaced keys are looked at first (one tiny kernel + one sync, paid only by tables large
 * enough for the bytes to matter).  mdb_dev_set_narrow_keys(): 0 never, 1 as described (default), 2 always try. */


__device__ static inline long long gc_wave_min_i64(long long v)
{
#pragma unroll
	for (int d = 1; d < MDB_WAVE; d <<= 1) {
		const long long o = __shfl_xor(v, d, MDB_WAVE);
		v = o < v ? o : v;
	}
	return v;
}

__device__ static inline long long gc_wave_max_i64(long long v)
{
#pragma unroll
	for (int d = 1; d < MDB_WAVE; d <<= 1) {
		const long long o = __shfl_xor(v, d, MDB_WAVE);
		v = o > v ? o : v;
	}
	return v;
}

/* mm[0] = smallest, mm[1] = largest of 2 x GC_NARROW_SAMPLE evenly spaced non-NULL keys */
template <typename K>	/* int64_t, or int32_t for key columns that crossed xGMI in the 4-byte wire format */
__global__ void k_key_sample(const K *__restrict__ kl, const uint64_t *__restrict__ nl_bits, uint64_t nl,
			     const K *__restrict__ kr, const uint64_t *__restrict__ nr_bits, uint64_t nr, long long *mm)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	long long lo = 0x7FFFFFFFFFFFFFFFll, hi = -0x7FFFFFFFFFFFFFFFll - 1;
	if (t < GC_NARROW_SAMPLE) {
		if (nl) {
			const uint64_t i = gc_sample_pos(t, nl);
			if (!(nl_bits && mdb_bit_is_set(nl_bits, i))) {
				lo = kl[i];
				hi = kl[i];
			}
		}
		if (kr && nr) {
			const uint64_t j = gc_sample_pos(t, nr);
			if (!(nr_bits && mdb_bit_is_set(nr_bits, j))) {
				const long long v = kr[j];
				lo = v < lo ? v : lo;
				hi = v > hi ? v : hi;
			}
		}
	}
	lo = gc_wave_min_i64(lo);
	hi = gc_wave_max_i64(hi);
	if (mdb_lane() == 0 && lo <= hi) {
		atomicMin(&mm[0], lo);
		atomicMax(&mm[1], hi);
	}
	/* the two tables' own ranges (mm[2..3] left, mm[4..5] right): how much of the left table's key range the right one covers */
	long long llo = 0x7FFFFFFFFFFFFFFFll, lhi = -0x7FFFFFFFFFFFFFFFll - 1, rlo = llo, rhi = lhi;
	if (t < GC_NARROW_SAMPLE) {
		if (nl) {
			const uint64_t i = gc_sample_pos(t, nl);
			if (!(nl_bits && mdb_bit_is_set(nl_bits, i)))
				llo = lhi = kl[i];
		}
		if (kr && nr) {
			const uint64_t j = gc_sample_pos(t, nr);
			if (!(nr_bits && mdb_bit_is_set(nr_bits, j)))
				rlo = rhi = kr[j];
		}
	}
	llo = gc_wave_min_i64(llo);
	lhi = gc_wave_max_i64(lhi);
	rlo = gc_wave_min_i64(rlo);
	rhi = gc_wave_max_i64(rhi);
	if (mdb_lane() == 0) {
		if (llo <= lhi) {
			atomicMin(&mm[2], llo);
			atomicMax(&mm[3], lhi);
		}
		if (rlo <= rhi) {
			atomicMin(&mm[4], rlo);
			atomicMax(&mm[5], rhi);
		}
	}
}

/* smallest / largest of 2 x GC_NARROW_SAMPLE evenly spaced non-NULL keys (lo > hi: nothing but NULLs); remembered by the
 * columns, so that the decisions that need it (narrow form, direct tables) share one kernel + sync, and a repeated query
 * pays none.  fresh: take the sample again (a remembered verdict just proved wrong). */
int gc_sample_range(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			   const uint64_t *null_r, uint64_t n_r, bool fresh, int64_t *lo, int64_t *hi, bool keys32)
{
	if (!keys_r)
		n_r = 0;
	if (ctx->cs_on && ctx->cs_kl == keys_l && (!keys_r || (ctx->cs_has_r && ctx->cs_kr == keys_r)) && !keys32) {
		/* the caller's statistics of exactly these columns (mdb_dev_call_stats): what a sample would estimate, known - no kernel, no
		 * synchronisation, nothing remembered by address */
		const struct mdb_dev_col_stats &l = ctx->cs_l, &r = ctx->cs_r;
		const bool has_l = n_l && l.min <= l.max, has_r = keys_r && n_r && r.min <= r.max;
		*lo = INT64_MAX;
		*hi = INT64_MIN;
		if (has_l) {
			*lo = l.min;
			*hi = l.max;
		}
		if (has_r) {
			*lo = r.min < *lo ? r.min : *lo;
			*hi = r.max > *hi ? r.max : *hi;
		}
		ctx->sr_span_l = has_l ? (uint64_t)l.max - (uint64_t)l.min + 1 : 0;
		ctx->sr_span_r = has_r ? (uint64_t)r.max - (uint64_t)r.min + 1 : 0;
		if (has_l && !ctx->sr_span_l)
			ctx->sr_span_l = ~0ull;		/* (the whole int64 range) */
		if (has_r && !ctx->sr_span_r)
			ctx->sr_span_r = ~0ull;
		ctx->sr_rlo = has_r ? r.min : INT64_MAX;
		ctx->sr_rhi = has_r ? r.max : INT64_MIN;
		ctx->sr_kl = keys_l;
		ctx->sr_nl = n_l;
		ctx->sr_kr = keys_r;
		ctx->sr_nr = n_r;
		ctx->sr_lo = *lo;
		ctx->sr_hi = *hi;
		ctx->sr_valid = 1;
		ctx->sr_uses = 0;
		ctx->pl_from_stats = 1;
		return MIDORIDB_OK;
	}
	if (!fresh && ctx->sr_valid && ctx->sr_kl == keys_l && ctx->sr_nl == n_l && ctx->sr_kr == keys_r && ctx->sr_nr == n_r &&
	    ++ctx->sr_uses < GC_HINT_USES) {
		*lo = ctx->sr_lo;
		*hi = ctx->sr_hi;
		return MIDORIDB_OK;
	}
	long long *mm = (long long *)(ctx->d_status + 10);
	int64_t *h = (int64_t *)ctx->h_pinned;
	for (int i = 0; i < 6; i += 2) {
		h[i] = INT64_MAX;
		h[i + 1] = INT64_MIN;
	}
	MDB_HIP(ctx, hipMemcpyAsync(mm, h, 48, hipMemcpyHostToDevice, ctx->stream));
	ctx->pl_samples++;
	if (keys32) {
		MDB_LAUNCH(ctx, "key_sample", k_key_sample<int32_t>, GC_NARROW_SAMPLE / 256, 256, reinterpret_cast<const int32_t *>(keys_l), null_l, n_l,
			   reinterpret_cast<const int32_t *>(keys_r), null_r, n_r, mm);
	} else {
		MDB_LAUNCH(ctx, "key_sample", k_key_sample<int64_t>, GC_NARROW_SAMPLE / 256, 256, keys_l, null_l, n_l, keys_r, null_r, n_r, mm);
	}
	MDB_HIP(ctx, hipMemcpyAsync(h, mm, 48, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*lo = h[0];
	*hi = h[1];
	/* spans of the two tables' sampled keys (0 = nothing sampled) */
	ctx->sr_span_l = h[2] <= h[3] ? (uint64_t)h[3] - (uint64_t)h[2] + 1 : 0;
	ctx->sr_span_r = h[4] <= h[5] ? (uint64_t)h[5] - (uint64_t)h[4] + 1 : 0;
	ctx->sr_rlo = h[4];
	ctx->sr_rhi = h[5];
	ctx->sr_kl = keys_l;
	ctx->sr_nl = n_l;
	ctx->sr_kr = keys_r;
	ctx->sr_nr = n_r;
	ctx->sr_lo = *lo;
	ctx->sr_hi = *hi;
	ctx->sr_valid = 1;
	ctx->sr_uses = 0;
	return MIDORIDB_OK;
}

void gc_narrow_note(mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const int64_t *keys_r, uint64_t n_r, bool narrow,
			   int64_t base, uint32_t key_bits, int64_t key_lo)
{
	ctx->nh_kbits = narrow ? key_bits : 0u;
	ctx->nh_lo = key_lo;
	ctx->nh_kl = keys_l;
	ctx->nh_nl = n_l;
	ctx->nh_kr = keys_r;
	ctx->nh_nr = keys_r ? n_r : 0;
	ctx->nh_result = narrow ? 1 : 0;
	ctx->nh_base = base;
	ctx->nh_uses = 0;
}

/* *narrow: try the narrow form; *base: centre of the 2^32-wide window of key values it will be tried with (0 = the plain
 * int32 range, whenever the sampled keys lie inside it) */
/* The compact window offered by a key sample [lo, hi]: the sampled span, padded by a sixteenth of itself + 4096 on either
 * side for the extremes the sample missed, rounded up to a power of two (the slack is split between the two ends).
 * *kbits = 0: none (span of 2^31 or more, or nothing but NULLs sampled). */
/* what the previous join over the same columns learned: few groups for many left rows (every use is exact whatever it says:
 * the bitmap filter only drops rows that can have no partner) */
static bool gc_learned_selective(const mdb_dev_ctx *ctx, const int64_t *keys_l, uint64_t n_l, const int64_t *keys_r, uint64_t n_r)
{
	return ctx->lg_valid && !ctx->lg_nextra && ctx->lg_kl == keys_l && ctx->lg_nl == n_l && ctx->lg_kr == keys_r && ctx->lg_nr == n_r && keys_r &&
	       ctx->lg_groups < n_l / 4;
}

/* exact: [lo, hi] is the catalog's range of the column (every key lies inside), not a sample's: no margin around it */
static void gc_compact_window(int64_t lo, int64_t hi, uint32_t *kbits, int64_t *wlo, bool exact = false)
{
	*kbits = 0;
	*wlo = 0;
	if (lo > hi)
		return;
	const uint64_t span = (uint64_t)hi - (uint64_t)lo;
	if (span >= (1ull << 31))
		return;
	const uint64_t pad = exact ? 0 : span / 16 + 4096, need = span + 2 * pad + 1;
	uint32_t k = 8;
	while ((1ull << k) < need)
		k++;
	if (k > 31)
		return;
	*kbits = k;
	*wlo = (int64_t)((uint64_t)lo - pad - (((1ull << k) - need) >> 1));
}


int gc_narrow_guess(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			   const uint64_t *null_r, uint64_t n_r, bool *narrow, int64_t *base, gc_window *win, bool keys32,
			   bool prune_ok /* unsplit call: min-max pruning can run */)
{
	*narrow = false;
	*base = 0;
	prune_ok = prune_ok && !(mdb_knob("MDB_MINMAX_PRUNE") && mdb_knob("MDB_MINMAX_PRUNE")[0] == '0');
	if (win) {
		win->kbits = 0;
		win->lo = 0;
		win->selective = false;
		win->by_span = false;
		win->prunable = false;
		win->r_based = false;
	}
	if (ctx->narrow_mode == 0 && n_l && keys_r && n_r && prune_ok && win && n_l + n_r >= GC_NARROW_MIN_ROWS) {
		/* the narrow forms are switched off, min-max pruning is not: what the key sample says about the two tables' ranges */
		int64_t lo = 0, hi = 0;
		const int src = gc_sample_range(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, false, &lo, &hi, keys32);
		if (src)
			return src;
		win->by_span = ctx->sr_span_l && ctx->sr_span_r && ctx->sr_span_r < ctx->sr_span_l / 4;
		win->prunable = ctx->sr_span_l && ctx->sr_span_r && ctx->sr_span_r / 7 < ctx->sr_span_l / 8;
	}
	if (ctx->narrow_mode == 0 || n_l == 0)
		return MIDORIDB_OK;
	if (ctx->narrow_mode == 2) {
		*narrow = true;
		return MIDORIDB_OK;
	}
	if (n_l + (keys_r ? n_r : 0) < GC_NARROW_MIN_ROWS)
		return MIDORIDB_OK;
	bool fresh = false;
	ctx->guess_remembered = false;
	const bool by_stats = ctx->cs_on && ctx->cs_kl == keys_l && (!keys_r || (ctx->cs_has_r && ctx->cs_kr == keys_r)) && !keys32;
	if (by_stats) {
		/* (decided afresh from the caller's statistics every time: it costs nothing and depends on nothing remembered) */
	} else if (ctx->nh_distrust > 0) {
		ctx->nh_distrust--;
		fresh = true;
	} else if (ctx->nh_result >= 0 && ctx->nh_kl == keys_l && ctx->nh_nl == n_l && ctx->nh_kr == keys_r && ctx->nh_nr == (keys_r ? n_r : 0) &&
		   !(ctx->nh_r_based && !prune_ok) &&	/* (a window of the right table's keys alone is no use to a call that cannot prune) */
		   ++ctx->nh_uses < GC_HINT_USES) {
		*narrow = ctx->nh_result == 1;	/* same columns as last time: what held then (gc_narrow_note) */
		ctx->guess_remembered = true;
		*base = ctx->nh_base;
		if (win && *narrow) {
			win->kbits = ctx->nh_kbits;
			win->lo = ctx->nh_lo;
			win->selective = ctx->nh_selective || gc_learned_selective(ctx, keys_l, n_l, keys_r, n_r);
			win->by_span = ctx->nh_by_span;
			win->prunable = ctx->nh_prunable;
			win->r_based = ctx->nh_r_based;
		} else if (win) {	/* (the 64-bit form prunes by the key range as well) */
			win->by_span = ctx->nh_by_span;
			win->prunable = ctx->nh_prunable;
		}
		return MIDORIDB_OK;
	}
	int64_t lo = 0, hi = 0;
	const int src = gc_sample_range(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, fresh, &lo, &hi, keys32);
	if (src)
		return src;
	if (lo > hi) {
		*narrow = true;		/* nothing but NULLs in the sample */
	} else if (lo >= -(1ll << 31) && hi < (1ll << 31)) {
		*narrow = true;
	} else if ((uint64_t)hi - (uint64_t)lo < (1ull << 31)) {
		/* a window anywhere in the int64 range (surrogate keys that start at 10^12, timestamps ...): centred on the
		 * sample, so that keys up to 2^30 beyond either end of what was sampled still fit */
		*narrow = true;
		*base = (int64_t)((uint64_t)lo + (((uint64_t)hi - (uint64_t)lo) >> 1));
	}
	uint32_t kb = 0;
	int64_t wlo = 0;
	if (*narrow)
		gc_compact_window(lo, hi, &kb, &wlo);
	const bool by_span = keys_r && n_r && ctx->sr_span_l && ctx->sr_span_r && ctx->sr_span_r < ctx->sr_span_l / 4;
	/* ... or the last join over these very columns found a partner for under a quarter of the left rows: a right table of few
	 * distinct keys spread over the left table's whole range (no sample of 4096 keys shows that) */
	const bool selective = keys_r && n_r && (by_span || n_r < n_l / 4 || gc_learned_selective(ctx, keys_l, n_l, keys_r, n_r));
	const bool prunable = keys_r && n_r && ctx->sr_span_l && ctx->sr_span_r && ctx->sr_span_r / 7 < ctx->sr_span_l / 8;
	/* the right table covers a small part of the left table's range and min-max pruning will drop the left rows outside it:
	 * the compact window need only cover the RIGHT table's keys (variant D: 23 key bits instead of 27 - 4096 leaves of
	 * 24 000 right rows instead of 32 768 leaves of 3 000: the leaf kernel's fixed price per leaf, 0.29 -> 0.13 ms) */
	bool r_based = false;
	if (*narrow && by_span && prune_ok && win) {
		uint32_t kb2 = 0;
		int64_t wlo2 = 0;
		gc_compact_window(ctx->sr_rlo, ctx->sr_rhi, &kb2, &wlo2);
		if (kb2 && (!kb || kb2 < kb)) {
			kb = kb2;
			wlo = wlo2;
			r_based = true;
		}
	}
	ctx->nh_r_based = r_based;
	if (win) {
		win->r_based = r_based;
		win->kbits = kb;
		win->lo = wlo;
		win->selective = selective;
		win->by_span = by_span;
		win->prunable = prunable;
	}
	ctx->nh_by_span = by_span;
	ctx->nh_prunable = prunable;
	gc_narrow_note(ctx, keys_l, n_l, keys_r, n_r, *narrow, *base, kb, wlo);
	ctx->nh_selective = selective;
	return MIDORIDB_OK;
}

static int group_count_run(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
			   const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, bool has_r, bool null_group, bool fast,
			   bool want_records, bool no_build_r, bool narrow, int64_t base, gc_window win, bool keys32, int64_t *out_key, int64_t *out_count,
			   uint32_t *out_first, uint64_t cap, uint64_t *out_groups, uint64_t *out_joined)
{
	*out_groups = 0;
	if (out_joined)
		*out_joined = 0;
	if (n_l == 0 || (has_r && n_r == 0))
		return MIDORIDB_OK;
	gc_state st;
	memset(&st, 0, sizeof(st));
	st.keys_l = keys_l;
	st.null_l = null_l;
	st.n_l = n_l;
	st.n_r_cap = n_r;
	st.has_r = has_r;
	st.null_group = null_group;
	st.fast = fast;
	st.want_records = want_records;
	st.no_build_r = no_build_r;
	st.narrow = narrow;
	st.base = base;
	st.key_bits = narrow ? win.kbits : 0u;
	st.key_lo = win.lo;
	st.selective = win.selective;
	st.fast1 = win.fast1;
	st.by_span = win.by_span;
	st.prunable = win.prunable;
	st.r_based = win.r_based;
	st.keys32 = keys32;
	st.defer_ok = true;
	st.own_call = !(mdb_knob("MDB_ZERO_BLOCK") && mdb_knob("MDB_ZERO_BLOCK")[0] == '0');
	if (gc_pending_extras && has_r) {
		st.nextra = gc_pending_extras->n;
		for (int x = 0; x < st.nextra; x++) {
			st.xkeys[x] = gc_pending_extras->keys[x];
			st.xnull[x] = gc_pending_extras->nulls[x];
			st.xn[x] = gc_pending_extras->rows[x];
		}
	}
	int rc = gc_begin(ctx, &st);
	if (rc == GC_EXPLAINED)		/* (mdb_dev_explain_*: gc_begin stopped in front of its arena) */
		gc_explain_fill(ctx, &st, keys_r, n_r, cap);
	if (rc)
		return rc;
	return gc_finish(ctx, &st, keys_r, null_r, n_r, out_key, out_count, out_first, cap, out_groups, out_joined);
}

static int group_count_common(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
			      const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, bool has_r, bool null_group,
			      int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap, uint64_t *out_groups,
			      uint64_t *out_joined, bool keys32 = false)
{
	/* first the histogram-free layout for the second partition level; the exact layout is the fallback
	 * when skewed keys overflow a leaf region (detected on the device, reported with the results) */
	bool fast = true, records = true, no_build_r = false, narrow = false;
	int64_t base = 0;
	gc_window win = { 0, 0, false, false, false, false };
	int rc = MIDORIDB_OK;
	/* plain GROUP BY whose key sample held duplicates (at most a few 10^5 distinct values): the leaves hold a few values with
	 * hundreds or thousands of rows each, their sizes vary by whole multiples, and the fixed-capacity layout would overflow
	 * and be redone anyway - start with the exact layout */
	/* ... unless there are still some 10^4 - 10^5 of them (3500 distinct among 4096 sampled: about 1.3 * 10^4 in the column) and
	 * the one-level form applies: its 256 or 512 first-level regions hold 50 values or more each; sized at 1.5 x the average
	 * (mdb_part_filter.loose) they take the variation of that number */
	bool fast1 = false;
	if (!has_r && ctx->gh_keys == keys_l && ctx->gh_n == n_l && ctx->gh_distinct < 4050u) {
		if (ctx->gh_distinct >= 3500u)
			fast1 = true;
		else
			fast = false;
	}
	rc = gc_narrow_guess(ctx, keys_l, null_l, n_l, has_r ? keys_r : NULL, null_r, n_r, &narrow, &base, &win, keys32, has_r);
	if (rc)
		return rc;
	if (keys32) {		/* int32 columns are inside the plain narrow form's window whatever the sample says; the sample offers the compact one */
		if (!narrow)
			win.kbits = 0;
		narrow = ctx->narrow_mode != 0;
		base = 0;
	}
	win.fast1 = fast1;
	/* the histogram-free layout overflowed on these very columns last time (skewed or heavily duplicated keys): exact at once */
	if (fast && ctx->ex_keys == keys_l && ctx->ex_nl == n_l && ctx->ex_nr == (has_r ? n_r : 0) && ++ctx->ex_uses < GC_HINT_USES) {
		fast = false;
		win.fast1 = false;
	}
	for (int attempt = 0; attempt < 6; attempt++) {
		ctx->pl_retries = (uint32_t)attempt;
		ctx->pl_key_bits = narrow ? win.kbits : 0u;
		rc = group_count_run(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, has_r, null_group, fast, records, no_build_r, narrow,
				     base, win, keys32, out_key, out_count, out_first, cap, out_groups, out_joined);
		if ((rc == GC_RETRY_PLAIN || rc == GC_RETRY_WIDE) && ctx->guess_remembered && ctx->narrow_mode == 1 && !keys32) {
			/* a REMEMBERED verdict proved wrong: the buffers hold other data than when it was made (a caller's allocator handed
			 * the same addresses out again).  Not a reason to give the narrow forms up: forget, look at the data itself, go on */
			ctx->nh_result = -1;
			ctx->sr_valid = 0;
			ctx->nh_distrust = 1;
			narrow = false;
			base = 0;
			rc = gc_narrow_guess(ctx, keys_l, null_l, n_l, has_r ? keys_r : NULL, null_r, n_r, &narrow, &base, &win, keys32, has_r);
			if (rc)
				return rc;
			win.fast1 = fast1 && fast;
		} else if (rc == GC_RETRY_PLAIN) {
			/* the sample missed the column's extremes: the plain narrow form (any 2^32-wide window) is tried next,
			 * and remembered for these columns */
			win.kbits = 0;
			if (ctx->narrow_mode == 1)
				gc_narrow_note(ctx, keys_l, n_l, has_r ? keys_r : NULL, n_r, true, base);
		} else if (rc == GC_RETRY_WIDE) {
			narrow = false;
			if (ctx->narrow_mode == 1 && !keys32)
				ctx->nh_distrust = 8;	/* whatever said "narrow" was wrong: look at the data itself the next few times */
			if (ctx->narrow_mode == 1 && !keys32)
				gc_narrow_note(ctx, keys_l, n_l, has_r ? keys_r : NULL, n_r, false);	/* the sample missed a wide key */
		} else if (rc == GC_RETRY_EXACT) {
			fast = false;
			win.fast1 = false;
			ctx->ex_keys = keys_l;
			ctx->ex_nl = n_l;
			ctx->ex_nr = has_r ? n_r : 0;
			ctx->ex_uses = 0;
		}	/* (gc_begin then also leaves the narrow form of a join: it is only built on the fast layout) */
		else if (rc == GC_RETRY_DENSE)
			records = false;
		else if (rc == GC_RETRY_UNKEYED || rc == GC_RETRY_REC64 || rc == GC_RETRY_TWO_LEVEL || rc == GC_RETRY_NODENSE)
			;		/* (gc_finish has set ctx->keyed_distrust / cleared ctx->r32_ok / noted the columns in ctx->lw_bad_*) */
		else if (rc == GC_RETRY_BUILD_L)
			no_build_r = true;
		else
			break;
	}
	return rc == GC_EXPLAINED ? MIDORIDB_OK : rc;
}

/* ------------------------------------------------------------------ tiny inputs: one kernel, one workgroup
 *
 * The reference's own tests and README work on a handful of rows; through the partitioned pipeline such a query is ~20
 * launches and two synchronisations (~170 us).  Up to GC_THREADS * LEAF_BATCH rows per table one workgroup does the whole
 * operator in LDS: table from the left rows (COUNT, first row), right rows counted into it, and - because a left row that
 * is the first of its group knows so - the groups leave in first-occurrence order by a prefix sum over the rows, no sort. */

struct tiny_args {
	const int64_t *keys_l;
	const uint64_t *null_l;
	uint32_t n_l;
	const int64_t *keys_r;
	const uint64_t *null_r;
	uint32_t n_r;
	uint32_t null_group;
	int64_t *out_key;
	int64_t *out_count;
	uint32_t *out_first;
	uint32_t cap;
	uint32_t *status;	/* [1] groups, [2..3] joined rows, [0] bit 12: more groups than cap */
};

template <bool HAS_R>
__global__ __launch_bounds__(GC_THREADS) void k_tiny_group_count(tiny_args a)
{
	__shared__ unsigned long long s_key[GC_SLOTS];
	__shared__ unsigned long long s_cnt[GC_SLOTS + 2];	/* [GC_SLOTS] the value whose hash is 0, [GC_SLOTS + 1] the NULL group */
	__shared__ uint32_t s_first[GC_SLOTS + 2];
	__shared__ uint32_t s_tmp[32];
	__shared__ unsigned long long s_sum;
	for (uint32_t s = threadIdx.x; s < GC_SLOTS + 2; s += GC_THREADS) {
		if (s < GC_SLOTS)
			s_key[s] = 0ull;
		s_cnt[s] = 0ull;
		s_first[s] = 0xFFFFFFFFu;
	}
	if (threadIdx.x == 0)
		s_sum = 0ull;
	__syncthreads();
	uint32_t slot[LEAF_BATCH];
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		const uint32_t i = threadIdx.x + (uint32_t)u * GC_THREADS;
		slot[u] = 0xFFFFFFFFu;
		if (i >= a.n_l)
			continue;
		if (a.null_l && mdb_bit_is_set(a.null_l, i)) {
			if (!HAS_R && a.null_group)
				slot[u] = GC_SLOTS + 1;
		} else {
			const uint64_t hv = mdb_fmix64((uint64_t)a.keys_l[i]);
			slot[u] = hv ? leaf_insert(s_key, GC_SLOTS, hv) : GC_SLOTS;	/* 2048 values at most: the table cannot fill */
		}
		if (slot[u] != 0xFFFFFFFFu) {
			atomicAdd(&s_cnt[slot[u]], 1ull);
			atomicMin(&s_first[slot[u]], i);
		}
	}
	__syncthreads();
	if (HAS_R) {
#pragma unroll
		for (int u = 0; u < LEAF_BATCH; u++) {
			const uint32_t j = threadIdx.x + (uint32_t)u * GC_THREADS;
			if (j >= a.n_r || (a.null_r && mdb_bit_is_set(a.null_r, j)))
				continue;
			const uint64_t hv = mdb_fmix64((uint64_t)a.keys_r[j]);
			const uint32_t s = hv ? leaf_find(s_key, GC_SLOTS, hv) : GC_SLOTS;
			if (s != 0xFFFFFFFFu && (uint32_t)s_cnt[s])
				atomicAdd(&s_cnt[s], 1ull << 32);
		}
		__syncthreads();
	}
	/* a left row that is the first of its group (and whose group survives the join) is a result row */
	bool head[LEAF_BATCH];
	unsigned long long cnt[LEAF_BATCH];
	unsigned long long joined = 0;
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		const uint32_t i = threadIdx.x + (uint32_t)u * GC_THREADS;
		head[u] = false;
		cnt[u] = 0;
		if (slot[u] != 0xFFFFFFFFu && s_first[slot[u]] == i) {
			const unsigned long long c2 = s_cnt[slot[u]];
			const unsigned long long cl = (uint32_t)c2, cr = c2 >> 32;
			cnt[u] = HAS_R ? cl * cr : cl;
			head[u] = cnt[u] != 0;
			joined += cnt[u];
		}
	}
	/* positions: rows are visited in index order across (u, thread): row = u * GC_THREADS + thread, so the prefix runs
	 * over u = 0 first, then u = 1 */
	uint32_t base = 0;
#pragma unroll
	for (int u = 0; u < LEAF_BATCH; u++) {
		uint32_t total;
		const uint32_t ex = mdb_block_excl_scan(head[u] ? 1u : 0u, s_tmp, &total);
		if (head[u]) {
			const uint32_t pos = base + ex;
			if (pos < a.cap) {
				const uint32_t i = threadIdx.x + (uint32_t)u * GC_THREADS;
				if (a.out_first)
					a.out_first[pos] = i;
				a.out_count[pos] = (int64_t)cnt[u];
				if (a.out_key)
					a.out_key[pos] = a.keys_l[i];
			}
		}
		base += total;
	}
	if (joined)
		atomicAdd(&s_sum, joined);
	__syncthreads();
	if (threadIdx.x == 0) {
		a.status[0] = base > a.cap ? 4096u : 0u;
		a.status[1] = base;
		*(unsigned long long *)(a.status + 2) = s_sum;
	}
}

/* 0 = done, 1 = not applicable, < 0 = error */
int tiny_group_count(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			    const uint64_t *null_r, uint64_t n_r, bool has_r, bool null_group, int64_t *out_key, int64_t *out_count,
			    uint32_t *out_first, uint64_t cap, uint64_t *out_groups, uint64_t *out_joined)
{
	if (n_l == 0 || n_l > TINY_ROWS || (has_r && (n_r == 0 || n_r > TINY_ROWS)) || (!out_count && !ctx->explain))
		return 1;
	if (ctx->explain) {
		ctx->explain->small_form = 1;
		return MIDORIDB_OK;
	}
	ctx->pl_small_form = 1;
	tiny_args a;
	memset(&a, 0, sizeof(a));
	a.keys_l = keys_l;
	a.null_l = null_l;
	a.n_l = (uint32_t)n_l;
	a.keys_r = keys_r;
	a.null_r = null_r;
	a.n_r = has_r ? (uint32_t)n_r : 0u;
	a.null_group = null_group ? 1u : 0u;
	a.out_key = out_key;
	a.out_count = out_count;
	a.out_first = out_first;
	a.cap = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
	a.status = ctx->d_status;
	if (has_r) {
		MDB_LAUNCH(ctx, "tiny_join_group_count", k_tiny_group_count<true>, 1, GC_THREADS, a);
	} else {
		MDB_LAUNCH(ctx, "tiny_group_count", k_tiny_group_count<false>, 1, GC_THREADS, a);
	}
	uint32_t *h = (uint32_t *)ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h, ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h[0] & 4096u)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups", (unsigned long long)cap,
				   (unsigned long long)h[1]);
	*out_groups = h[1];
	if (out_joined)
		*out_joined = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
	ctx->last_narrow = 0;
	ctx->last_semijoin = 0;
	return 0;
}


/* Groups in UNSPECIFIED order (the caller did not pass MDB_ORDER_FIRST and wants no first rows): nothing has to remember which
 * left row a group began with, so no row id travels through the partition levels and no ordering sort runs - the pipeline of the
 * sharded operator's receiver (mdb_dev_shard.hip) on this GPU's own first-level regions: 4-byte (or 2-byte) words of the k-bit
 * window hash for BOTH tables, counts per leaf, (key decoded from the table slot, COUNT) written where the leaf kernel finds it.
 * Window = the right table's sampled key range, padded like the compact form's; every right key is checked against it on the
 * device, left rows outside it join nothing.  10^8 x 10^8 unique keys: 1.5 ms of kernels where the ordered operator takes 2.56.
 * 0 = done, 1 = not served (window too wide, a key outside it, skew, a region overflow ...): the ordered operator answers. */
static int gc_unordered_try(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, const int64_t *keys_r,
			    const uint64_t *null_r, uint64_t n_r, int64_t *out_key, int64_t *out_count, uint64_t cap, uint64_t *out_groups,
			    uint64_t *out_joined, int n_extra = 0, const int64_t *const *keys_x = NULL, const uint64_t *const *null_x = NULL,
			    const uint64_t *n_x = NULL /* further right tables on the same key: their rows outside the window join nothing */)
{
	if (n_l + n_r < (1ull << 21) || ctx->narrow_mode == 0 || ld_disabled() || (mdb_knob("MDB_UNORDERED") && mdb_knob("MDB_UNORDERED")[0] == '0'))
		return 1;
	bool fresh = false;
	uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
	mdb_shard_plan plan;
again: {
	int64_t lo = 0, hi = 0;
	int rc = gc_sample_range(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, fresh || ctx->nh_distrust > 0, &lo, &hi);
	const bool remembered = ctx->sr_uses > 0;
	if (rc)
		return rc;
	if (!ctx->sr_span_r || !ctx->sr_span_l)
		return 1;
	uint32_t kb = 0;
	int64_t wlo = 0;
	gc_compact_window(ctx->sr_rlo, ctx->sr_rhi, &kb, &wlo);
	if (!kb || kb > 30u)
		return 1;
	uint64_t n_max[MDB_SHARD_MAX_TABS] = { n_l, n_r, 0, 0 };
	if (n_extra < 0 || 2 + n_extra > MDB_SHARD_MAX_TABS)
		return 1;
	for (int t = 0; t < n_extra; t++)
		n_max[2 + t] = n_x[t];
	if (mdb_shard_plan_make(1, 0, 2u + (uint32_t)n_extra, n_max, 0, (int64_t)(ctx->sr_span_l + ctx->sr_span_l / 8), wlo, wlo + (int64_t)((1ull << kb) - 1), &plan))
		return 1;
	rc = mdb_arena_begin(ctx, mdb_shard_arena_bytes(&plan));
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
	const int64_t *keys[MDB_SHARD_MAX_TABS] = { keys_l, keys_r, NULL, NULL };
	const uint64_t *nulls[MDB_SHARD_MAX_TABS] = { null_l, null_r, NULL, NULL };
	const void *regions[MDB_SHARD_MAX_TABS] = { NULL, NULL, NULL, NULL };
	const uint32_t *cursors[MDB_SHARD_MAX_TABS] = { NULL, NULL, NULL, NULL };
	for (int t = 0; t < n_extra; t++) {
		keys[2 + t] = keys_x[t];
		nulls[2 + t] = null_x ? null_x[t] : NULL;
	}
	for (int x = 1 + n_extra; x >= 0; x--) {		/* (the right table first, as in the ordered operator) */
		rc = mdb_shard_partition(ctx, &plan, x, keys[x], nulls[x], n_max[x], &regions[x], &cursors[x]);
		if (rc)
			return rc;
	}
	rc = mdb_shard_join(ctx, &plan, regions, cursors, out_key, ctx->unordered_no_counts ? NULL : out_count, cap);	/* (world 1: what was sent is what arrived) */
	if (rc)
		return rc;
	MDB_HIP(ctx, hipMemcpyAsync(h, ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h[0]) {
		if (mdb_knob("MDB_DEBUG_UNORDERED"))
			fprintf(stderr, "unordered form not served: flags %u (k %u, b2 %u, rem %u)\n", h[0], plan.kbits, plan.b2, plan.rem);
		if ((h[0] & 128u) && remembered && !fresh) {
			fresh = true;	/* the window came from a remembered sample and the column's contents have changed since */
			goto again;
		}
		if (h[0] & 128u)
			ctx->nh_distrust = 8;	/* a right key outside the sampled window: look at the data itself the next few times */
		return 1;
	}
	}
	*out_groups = h[1];
	if (out_joined)
		*out_joined = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
	ctx->last_narrow = 2;
	ctx->last_semijoin = 0x100 | 0x800 | (plan.b2 ? 0 : 0x200);
	return 0;
}

/* GROUP BY key + COUNT(*) of ONE column as (key, COUNT) pairs in unspecified order (include/mdb_dev.h): the any-order operator's
 * pipeline with the table in the right table's place and no left table - one first level of 2-byte words (two levels beyond
 * 2^27 values), every leaf slot with rows is a group.  1 = not served (NULL keys, keys beyond a 2^30-value window, skew that
 * overflows a region, small tables): the caller's ordered operator answers. */
/* MDB_KEYS_MAY_ALIAS / MDB_COUNTS_OPTIONAL of the OUTERMOST join + GROUP BY call hold for it and the operators it calls */
struct gc_alias_scope {
	mdb_dev_ctx *c;
	bool a, o;
	gc_alias_scope(mdb_dev_ctx *ctx, uint32_t f) : c(ctx), a(ctx->key_alias_ok), o(ctx->counts_optional)
	{
		if (ctx->pl_depth == 1) {
			ctx->key_alias_ok = (f & MDB_KEYS_MAY_ALIAS) != 0;
			ctx->counts_optional = (f & MDB_COUNTS_OPTIONAL) != 0;
		}
	}
	~gc_alias_scope()
	{
		c->key_alias_ok = a;
		c->counts_optional = o;
	}
};

extern "C" int mdb_dev_group_count_keys(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int64_t *out_key,
					int64_t *out_count, uint64_t cap, uint64_t *out_groups)
{
	mdb_plan_scope plan_scope(ctx);
	if (!ctx || !out_groups || !out_key || !out_count)
		return -MIDORIDB_ERROR;
	*out_groups = 0;
	if (nullbits || n < (1ull << 21) || ctx->narrow_mode == 0 || ld_disabled() || (mdb_knob("MDB_UNORDERED") && mdb_knob("MDB_UNORDERED")[0] == '0'))
		return 1;
	mdb_memo_switch(ctx, keys, n, NULL, 0);
	uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned);
	for (int attempt = 0; attempt < 2; attempt++) {
		int64_t lo = 0, hi = 0;
		int rc = gc_sample_range(ctx, keys, NULL, n, NULL, NULL, 0, attempt > 0 || ctx->nh_distrust > 0, &lo, &hi);
		if (rc)
			return rc;
		const bool remembered = ctx->sr_uses > 0;
		uint32_t kb = 0;
		int64_t wlo = 0;
		if (lo <= hi)
			gc_compact_window(lo, hi, &kb, &wlo);
		if (!kb || kb > 30u)
			return 1;
		mdb_shard_plan plan;
		const uint64_t n_max[2] = { 0, n };
		if (mdb_shard_plan_make(1, 0, 2, n_max, 0, 0, wlo, wlo + (int64_t)((1ull << kb) - 1), &plan))
			return 1;
		plan.right_only = true;
		rc = mdb_arena_begin(ctx, mdb_shard_arena_bytes(&plan));
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 16 * sizeof(uint32_t), ctx->stream));
		const void *regions[2] = { NULL, NULL };
		const uint32_t *cursors[2] = { NULL, NULL };
		rc = mdb_shard_partition(ctx, &plan, 1, keys, NULL, n, &regions[1], &cursors[1]);
		if (!rc)
			rc = mdb_shard_partition(ctx, &plan, 0, NULL, NULL, 0, &regions[0], &cursors[0]);	/* (no rows: all-zero counters) */
		if (!rc)
			rc = mdb_shard_join(ctx, &plan, regions, cursors, out_key, out_count, cap);
		if (rc)
			return rc;
		MDB_HIP(ctx, hipMemcpyAsync(h, ctx->d_status, 16, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (!h[0]) {
			*out_groups = h[1];
			return MIDORIDB_OK;
		}
		if (mdb_knob("MDB_DEBUG_UNORDERED"))
			fprintf(stderr, "group_count_keys not served: flags %u (k %u, b2 %u, rem %u)\n", h[0], plan.kbits, plan.b2, plan.rem);
		if (!((h[0] & 128u) && remembered && attempt == 0)) {	/* (a remembered sample of a column whose contents changed: taken again, once) */
			if (h[0] & 128u)
				ctx->nh_distrust = 8;
			return 1;
		}
	}
	return 1;
}

extern "C" int mdb_dev_join_group_count(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
					const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, uint32_t flags,
					int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap,
					uint64_t *out_groups, uint64_t *out_joined)
{
	mdb_plan_scope plan_scope(ctx);
	*out_groups = 0;
	if (out_joined)
		*out_joined = 0;
	gc_alias_scope alias_scope_(ctx, flags);
	mdb_memo_switch(ctx, keys_l, n_l, keys_r, n_r);	/* what was learned about THIS pair of columns */
	if (!(flags & MDB_ORDER_FIRST) && !out_first && out_key && n_l && n_r) {
		/* no order asked for: no row ids, no ordering sort (otherwise - and whenever this form is not served - the groups come
		 * out in first-occurrence order, which satisfies both modes) */
		const int urc = gc_unordered_try(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, out_key, out_count, cap, out_groups, out_joined);
		if (urc <= 0)
			return urc;
		*out_groups = 0;
		if (out_joined)
			*out_joined = 0;
	}
	{
		const int trc = tiny_group_count(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, true, false, out_key, out_count, out_first, cap,
						 out_groups, out_joined);
		if (trc <= 0)
			return trc;
	}
	if (n_l && n_r) {
		/* both key columns inside one window of at most 4096 values (joins on a handful of hot values): counted directly */
		uint32_t *tmp_first = NULL;
		if (!out_first && mdb_cached_alloc(ctx, (cap ? cap : 1) * 4, (void **)&tmp_first))
			tmp_first = NULL;
		int drc = 1;
		if (out_first || tmp_first)
			drc = group_direct_try(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, false, out_key, out_first ? out_first : tmp_first, out_count,
					       cap, out_groups, out_joined);
		if (tmp_first)
			(void)mdb_cached_free(ctx, tmp_first);
		if (drc <= 0)
			return drc;
		*out_groups = 0;
		if (out_joined)
			*out_joined = 0;
	}
	return group_count_common(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, true, false, out_key, out_count, out_first,
				  cap, out_groups, out_joined);
}

/* ---- split form for pipelines whose right table arrives later (multi-GPU exchange) ---- */

static gc_state *gc_pending(mdb_dev_ctx *ctx)
{
	if (!ctx->pending_op)
		ctx->pending_op = calloc(1, sizeof(gc_state));
	return (gc_state *)ctx->pending_op;
}

static int gc_split_begin(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, uint64_t n_r_max, bool keys32)
{
	gc_state *st = gc_pending(ctx);
	if (!st)
		return -MIDORIDB_NOMEM;
	memset(st, 0, sizeof(*st));
	st->keys_l = keys_l;
	st->null_l = null_l;
	st->n_l = n_l;
	st->n_r_cap = n_r_max;
	st->has_r = true;
	st->null_group = false;
	st->fast = true;
	st->want_records = true;
	st->keys32 = keys32;
	if (n_l == 0)
		return MIDORIDB_OK;	/* nothing to prepare; finish() returns the empty result */
	/* the window comes from the left table's sample alone: the right table is checked as it is partitioned (a key outside
	 * sends finish() to the unsplit operator, which samples both tables) */
	gc_window win = { 0, 0, false, false, false, false };
	int rc = gc_narrow_guess(ctx, keys_l, null_l, n_l, NULL, NULL, 0, &st->narrow, &st->base, &win, keys32);
	if (rc)
		return rc;
	if (keys32) {		/* int32 columns are inside the plain narrow form's window whatever the sample says */
		if (!st->narrow)
			win.kbits = 0;
		st->narrow = ctx->narrow_mode != 0;
		st->base = 0;
	}
	st->key_bits = st->narrow ? win.kbits : 0u;
	st->key_lo = win.lo;
	return gc_begin(ctx, st);
}

static int gc_split_finish(mdb_dev_ctx *ctx, const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r, int64_t *out_key,
			   int64_t *out_count, uint32_t *out_first, uint64_t cap, uint64_t *out_groups, uint64_t *out_joined, bool keys32)
{
	gc_state *st = gc_pending(ctx);
	if (!st || !st->keys_l)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "join_group_count_finish without begin");
	if (st->keys32 != keys32)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "join_group_count_finish: key width differs from begin()");
	const int64_t *keys_l = st->keys_l;
	const uint64_t *null_l = st->null_l;
	const uint64_t n_l = st->n_l;
	*out_groups = 0;
	if (out_joined)
		*out_joined = 0;
	int rc = MIDORIDB_OK;
	if (n_l == 0 || n_r == 0) {
		if (st->active)
			rc = mdb_dev_sync(ctx);		/* drain the prepared left partition */
		st->active = false;
		st->keys_l = NULL;
		return rc;
	}
	if (n_r > st->n_r_cap) {
		/* more right rows than announced (a skewed exchange sent this GPU more than its share): the scratch arena was
		 * sized for the announced table - drop the prepared left partition and run the whole operator on the real sizes */
		rc = mdb_dev_sync(ctx);
		st->active = false;
		st->keys_l = NULL;
		if (rc)
			return rc;
		return group_count_common(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, true, false, out_key, out_count, out_first, cap,
					  out_groups, out_joined, keys32);
	}
	rc = gc_finish(ctx, st, keys_r, null_r, n_r, out_key, out_count, out_first, cap, out_groups, out_joined);
	st->keys_l = NULL;
	if (rc == GC_RETRY_EXACT || rc == GC_RETRY_DENSE || rc == GC_RETRY_BUILD_L || rc == GC_RETRY_WIDE || rc == GC_RETRY_PLAIN || rc == GC_RETRY_UNKEYED || rc == GC_RETRY_REC64 ||
	    rc == GC_RETRY_TWO_LEVEL || rc == GC_RETRY_NODENSE)	/* skew / huge counts / wide keys: redo the whole operator */
		rc = group_count_common(ctx, keys_l, null_l, n_l, keys_r, null_r, n_r, true, false, out_key, out_count, out_first,
					cap, out_groups, out_joined, keys32);
	return rc;
}

/* ---- one left table, up to three right tables, ONE key: A JOIN B ON a = b JOIN C ON a = c ... GROUP BY a, COUNT(*)
 *
 * The reference runs this shape as a recursive join (executor_select.c:1151-1280) followed by the GROUP BY loop (:1526-1588).
 * Here every table is partitioned once with the same hash; the direct-address leaf kernel counts the rows of every right
 * table into an LDS array of its own, multiplies the counts per key, and the left rows then meet ONE right count as in the
 * two-table operator (COUNT(*) = left rows x the product).  No joined row of any join exists, and the groups are ordered
 * once - where chaining two-table operators (what query_execute() did before, and what this function still does when the
 * keys do not take the compact narrow form, are skewed, or a product of counts outgrows 32 bits) partitions the group keys
 * again and orders twice. */
extern "C" int mdb_dev_join_group_count_multi(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l, int n_right,
					      const int64_t *const *keys_r, const uint64_t *const *null_r, const uint64_t *n_r, uint32_t flags,
					      int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap, uint64_t *out_groups,
					      uint64_t *out_joined)
{
	mdb_plan_scope plan_scope(ctx);
	gc_alias_scope alias_scope_(ctx, flags);
	if (!out_groups || n_right < 1 || n_right > 1 + GC_MAX_EXTRA || !keys_r || !n_r)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "join_group_count_multi: one to %d right tables", 1 + GC_MAX_EXTRA);
	*out_groups = 0;
	if (out_joined)
		*out_joined = 0;
	if (n_right == 1)
		return mdb_dev_join_group_count(ctx, keys_l, null_l, n_l, keys_r[0], null_r ? null_r[0] : NULL, n_r[0], flags, out_key, out_count, out_first,
						cap, out_groups, out_joined);
	if (!n_l)
		return MIDORIDB_OK;
	uint64_t smallest = n_l;
	for (int t = 0; t < n_right; t++) {
		if (!n_r[t])
			return MIDORIDB_OK;
		smallest = n_r[t] < smallest ? n_r[t] : smallest;
	}
	int rc = GC_NOT_SERVED;
	if (!(flags & MDB_ORDER_FIRST) && !out_first && out_key && smallest >= (1u << 16)) {
		/* no order asked for: every table partitioned once or twice without row ids, the right tables' counts multiplied in
		 * the leaf tables, no ordering sort */
		mdb_memo_switch(ctx, keys_l, n_l, keys_r[0], n_r[0]);
		const int urc = gc_unordered_try(ctx, keys_l, null_l, n_l, keys_r[0], null_r ? null_r[0] : NULL, n_r[0], out_key, out_count, cap, out_groups,
						 out_joined, n_right - 1, keys_r + 1, null_r ? null_r + 1 : NULL, n_r + 1);
		if (urc <= 0) {
			if (urc == 0)
				ctx->last_semijoin |= 0x400;
			return urc;
		}
		*out_groups = 0;
		if (out_joined)
			*out_joined = 0;
	}
	if (smallest >= (1u << 16) && n_l >= (1u << 20)) {	/* (small tables: the chain's single-workgroup and one-level forms are quicker) */
		gc_extras ex;
		memset(&ex, 0, sizeof(ex));
		ex.n = n_right - 1;
		for (int t = 1; t < n_right; t++) {
			ex.keys[t - 1] = keys_r[t];
			ex.nulls[t - 1] = null_r ? null_r[t] : NULL;
			ex.rows[t - 1] = n_r[t];
		}
		mdb_memo_switch(ctx, keys_l, n_l, keys_r[0], n_r[0]);
		gc_pending_extras = &ex;
		rc = group_count_common(ctx, keys_l, null_l, n_l, keys_r[0], null_r ? null_r[0] : NULL, n_r[0], true, false, out_key, out_count, out_first, cap,
					out_groups, out_joined);
		gc_pending_extras = NULL;
	}
	if (rc != GC_NOT_SERVED)
		return rc;
	if (ctx->explain) {	/* (mdb_dev_explain_*: the chain of two-table operators - its first step's plan, multi_one_pass 0) */
		uint64_t g0 = 0, j0 = 0;
		rc = mdb_dev_join_group_count(ctx, keys_l, null_l, n_l, keys_r[0], null_r ? null_r[0] : NULL, n_r[0], flags, NULL, NULL, NULL, n_l ? n_l : 1, &g0, &j0);
		ctx->explain->multi_one_pass = 0;
		return rc;
	}
	/* ---- the chain: groups of (L, R0), then (those group keys, R1) ..., counts multiplied */
	*out_groups = 0;
	int64_t *key[2] = { NULL, NULL }, *cnt[3] = { NULL, NULL, NULL };
	uint32_t *first[2] = { NULL, NULL }, *idx = NULL;
	uint64_t G = 0, J = 0;
	const uint64_t room = n_l ? n_l : 1;
	rc = MIDORIDB_OK;
	for (int i = 0; i < 2 && !rc; i++) {
		rc = mdb_dev_alloc(ctx, room * 8, (void **)&key[i]);
		if (!rc)
			rc = mdb_dev_alloc(ctx, room * 4, (void **)&first[i]);
	}
	for (int i = 0; i < 3 && !rc; i++)
		rc = mdb_dev_alloc(ctx, room * 8, (void **)&cnt[i]);
	if (!rc)
		rc = mdb_dev_alloc(ctx, room * 4, (void **)&idx);
	int cur = 0;		/* key[cur], cnt[cur], first[cur]: the groups so far */
	if (!rc)
		rc = mdb_dev_join_group_count(ctx, keys_l, null_l, n_l, keys_r[0], null_r ? null_r[0] : NULL, n_r[0], flags, key[0], cnt[0], first[0], room, &G,
					      &J);
	for (int t = 1; t < n_right && !rc && G; t++) {
		uint64_t G2 = 0, J2 = 0;
		const int nxt = cur ^ 1;
		rc = mdb_dev_join_group_count(ctx, key[cur], NULL, G, keys_r[t], null_r ? null_r[t] : NULL, n_r[t], flags, key[nxt], cnt[2], idx, room, &G2,
					      &J2);
		if (!rc)
			rc = mdb_dev_combine_counts(ctx, cnt[cur], first[cur], idx, cnt[2], G2, cnt[nxt], first[nxt], &J);
		cur = nxt;
		G = G2;
	}
	if (!rc && G > cap)
		rc = mdb_set_err(ctx, -MIDORIDB_ERROR, "group output capacity %llu too small for %llu groups", (unsigned long long)cap, (unsigned long long)G);
	if (!rc && G) {
		hipError_t e = hipMemcpyAsync(out_count, cnt[cur], G * 8, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess && out_key)
			e = hipMemcpyAsync(out_key, key[cur], G * 8, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess && out_first)
			e = hipMemcpyAsync(out_first, first[cur], G * 4, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess)
			rc = mdb_set_err(ctx, -MIDORIDB_INTERNAL, "join_group_count_multi: %s", hipGetErrorString(e));
	}
	for (int i = 0; i < 2; i++) {
		(void)mdb_dev_free(ctx, key[i]);
		(void)mdb_dev_free(ctx, first[i]);
	}
	for (int i = 0; i < 3; i++)
		(void)mdb_dev_free(ctx, cnt[i]);
	(void)mdb_dev_free(ctx, idx);
	if (!rc) {
		*out_groups = G;
		if (out_joined)
			*out_joined = G ? J : 0;
	}
	return rc;
}

extern "C" int mdb_dev_join_group_count_begin(mdb_dev_ctx *ctx, const int64_t *keys_l, const uint64_t *null_l, uint64_t n_l,
					      uint64_t n_r_max)
{
	mdb_plan_scope plan_scope(ctx);
	return gc_split_begin(ctx, keys_l, null_l, n_l, n_r_max, false);
}

extern "C" int mdb_dev_join_group_count_finish(mdb_dev_ctx *ctx, const int64_t *keys_r, const uint64_t *null_r, uint64_t n_r,
					       uint32_t flags, int64_t *out_key, int64_t *out_count, uint32_t *out_first,
					       uint64_t cap, uint64_t *out_groups, uint64_t *out_joined)
{
	mdb_plan_scope plan_scope(ctx);
	(void)flags;
	return gc_split_finish(ctx, keys_r, null_r, n_r, out_key, out_count, out_first, cap, out_groups, out_joined, false);
}

/* ---- int32 key columns (what arrives over xGMI in the 4-byte wire format): same operator, no widening pass ---- */

extern "C" int mdb_dev_join_group_count_i32(mdb_dev_ctx *ctx, const int32_t *keys_l, uint64_t n_l, const int32_t *keys_r, uint64_t n_r,
					    uint32_t flags, int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap,
					    uint64_t *out_groups, uint64_t *out_joined)
{
	mdb_plan_scope plan_scope(ctx);
	(void)flags;
	mdb_memo_switch(ctx, keys_l, n_l, keys_r, n_r);
	return group_count_common(ctx, reinterpret_cast<const int64_t *>(keys_l), NULL, n_l, reinterpret_cast<const int64_t *>(keys_r), NULL,
				  n_r, true, false, out_key, out_count, out_first, cap, out_groups, out_joined, true);
}

extern "C" int mdb_dev_join_group_count_begin_i32(mdb_dev_ctx *ctx, const int32_t *keys_l, uint64_t n_l, uint64_t n_r_max)
{
	mdb_plan_scope plan_scope(ctx);
	return gc_split_begin(ctx, reinterpret_cast<const int64_t *>(keys_l), NULL, n_l, n_r_max, true);
}

extern "C" int mdb_dev_join_group_count_finish_i32(mdb_dev_ctx *ctx, const int32_t *keys_r, uint64_t n_r, uint32_t flags,
						   int64_t *out_key, int64_t *out_count, uint32_t *out_first, uint64_t cap,
						   uint64_t *out_groups, uint64_t *out_joined)
{
	mdb_plan_scope plan_scope(ctx);
	(void)flags;
	return gc_split_finish(ctx, reinterpret_cast<const int64_t *>(keys_r), NULL, n_r, out_key, out_count, out_first, cap, out_groups,
			       out_joined, true);
}

extern "C" int mdb_dev_group_count(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, uint32_t flags,
				   uint32_t *out_first, int64_t *out_count, uint64_t cap, uint64_t *out_groups)
{
	mdb_plan_scope plan_scope(ctx);
	(void)flags;
	*out_groups = 0;
	/* The catalog says no value occurs twice in this NULL-free column (MDB_COL_DISTINCT: measured at ingest over all rows, followed through
	 * every append, dropped with an UPDATE): every row is the first and only row of its group - the reference's loop
	 * (/root/reference/src/engine/executor_select.c:1526-1588) would find no second row for any key.  Written without looking at a key:
	 * 12 bytes per row.  (The one form that TRUSTS a statistic: verifying it is the scan that measured it.) */
	if (ctx->cs_on && !ctx->explain_as_sample && ctx->cs_kl == keys && !ctx->cs_has_r && (ctx->cs_l.flags & MDB_COL_DISTINCT) && !nullbits && n && n <= cap &&
	    n < 0xFFFFFFFFull && !(mdb_knob("MDB_GROUP_IDENTITY") && mdb_knob("MDB_GROUP_IDENTITY")[0] == '0')) {
		if (ctx->explain) {
			ctx->explain->group_form = 3;
			ctx->explain->from_stats = 1;
			return MIDORIDB_OK;
		}
		int rc = mdb_group_identity(ctx, n, out_first, out_count);
		if (rc)
			return rc;
		ctx->pl_from_stats = 1;
		ctx->pl_group_form = 3;
		*out_groups = n;
		return MIDORIDB_OK;
	}
	mdb_memo_switch(ctx, keys, n, NULL, 0);
	{
		const int trc = tiny_group_count(ctx, keys, nullbits, n, NULL, NULL, 0, false, true, NULL, out_count, out_first, cap, out_groups, NULL);
		if (trc <= 0)
			return trc;
	}
	const int drc = group_direct_try(ctx, keys, nullbits, n, NULL, NULL, 0, true, NULL, out_first, out_count, cap, out_groups, NULL);
	if (drc <= 0)
		return drc;
	*out_groups = 0;
	const int hrc = group_hashed_try(ctx, keys, nullbits, n, true, out_first, out_count, cap, out_groups);
	if (hrc <= 0)
		return hrc;
	*out_groups = 0;
	/* round 5: a NULL-free key column inside a compact window of 2^18 ... 2^25 values goes through the band sort (mdb_dev_bandgroup.hip):
	 * one pass of 4-byte row words instead of 8-byte ones (and, MDB_GROUP_TILED=1, the tile sort of mdb_dev_rowjoin.hip: windows of 2^13 ... 2^27) */
	if (!nullbits && n >= ((uint64_t)1 << 21) && ctx->narrow_mode != 0 && !ld_disabled()) {
		for (int attempt = 0; attempt < 2; attempt++) {
			int64_t lo = 0, hi = 0;
			int rc = gc_sample_range(ctx, keys, NULL, n, NULL, NULL, 0, attempt > 0 || ctx->nh_distrust > 0, &lo, &hi);
			if (rc)
				return rc;
			const bool remembered = ctx->sr_uses > 0;
			uint32_t kb = 0;
			int64_t wlo = 0;
			if (lo <= hi)
				gc_compact_window(lo, hi, &kb, &wlo, ctx->pl_from_stats != 0);
			if (!kb)
				break;
			bool outside = false;
			rc = mdb_group_count_banded(ctx, keys, n, wlo, kb, out_first, out_count, cap, out_groups, &outside);
			if (rc <= 0)
				return rc;
			*out_groups = 0;
			if (!outside) {
				rc = mdb_group_count_tiled(ctx, keys, n, wlo, kb, out_first, out_count, cap, out_groups, &outside);
				if (rc <= 0)
					return rc;
				*out_groups = 0;
			}
			if (!(outside && remembered && attempt == 0)) {	/* (a remembered sample of a column whose contents changed: taken again, once) */
				if (outside)
					ctx->nh_distrust = 8;
				break;
			}
		}
	}
	return group_count_common(ctx, keys, nullbits, n, NULL, NULL, 0, false, true, NULL, out_count, out_first, cap, out_groups,
				  NULL);
}

/* ------------------------------------------------------------------ plans as data (include/mdb_dev.h: mdb_dev_explain_*)
 *
 * A context without a device - only what the decision code reads (CU count, key-form mode, an empty memo) - gets the caller's statistics
 * under two made-up column addresses, ctx->explain points at the answer, and the operator's ENTRY POINT is called: every path it takes
 * stops where it would launch (tiny_group_count, group_direct_try, group_count_run behind gc_begin). */
#define EXPLAIN_KL ((const int64_t *)0x1000)
#define EXPLAIN_KR ((const int64_t *)0x2000)
#define EXPLAIN_KX ((const int64_t *)0x3000)

static mdb_dev_ctx *explain_ctx(const struct mdb_dev_explain_request *rq, struct mdb_dev_plan_info *out, bool has_r)
{
	mdb_dev_ctx *ctx = new (std::nothrow) mdb_dev_ctx();
	if (!ctx)
		return NULL;
	memset(static_cast<mdb_col_memo *>(ctx), 0, sizeof(mdb_col_memo));
	ctx->nh_result = -1;
	ctx->device = -1;
	ctx->num_cus = rq->num_cus ? (int)rq->num_cus : 256;
	ctx->narrow_mode = 1;
	ctx->err[0] = 0;
	memset(out, 0, sizeof(*out));
	ctx->explain = out;
	ctx->explain_as_sample = rq->as_sample != 0;
	ctx->cs_on = true;
	ctx->cs_kl = EXPLAIN_KL;
	ctx->cs_l = rq->left;
	ctx->cs_has_r = has_r;
	if (has_r) {
		ctx->cs_kr = EXPLAIN_KR;
		ctx->cs_r = rq->right;
	}
	return ctx;
}

extern "C" int mdb_dev_explain_join_group_count(const struct mdb_dev_explain_request *rq, struct mdb_dev_plan_info *out)
{
	if (!rq || !out || rq->further_tables > 2)
		return -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = explain_ctx(rq, out, true);
	if (!ctx)
		return -MIDORIDB_NOMEM;
	const uint64_t n_l = rq->left.rows, n_r = rq->right.rows;
	const uint64_t *null_l = rq->left_nulls_bitmap ? (const uint64_t *)0x4000 : NULL;
	uint64_t groups = 0, joined = 0;
	int rc;
	if (rq->further_tables) {
		const int64_t *rk[3] = { EXPLAIN_KR, EXPLAIN_KX, EXPLAIN_KX + 0x1000 };
		const uint64_t *rn[3] = { NULL, NULL, NULL };
		const uint64_t rows[3] = { n_r, rq->further_rows[0], rq->further_rows[1] };
		rc = mdb_dev_join_group_count_multi(ctx, EXPLAIN_KL, null_l, n_l, 1 + (int)rq->further_tables, rk, rn, rows, MDB_ORDER_FIRST, NULL, NULL, NULL,
						    n_l, &groups, &joined);
	} else {
		rc = mdb_dev_join_group_count(ctx, EXPLAIN_KL, null_l, n_l, EXPLAIN_KR, NULL, n_r, MDB_ORDER_FIRST, NULL, NULL, NULL, n_l, &groups, &joined);
	}
	if (rc && mdb_knob("MDB_DEBUG_EXPLAIN"))
		fprintf(stderr, "mdb_dev_explain_join_group_count: %d (%s)\n", rc, ctx->err);
	delete ctx;
	return rc;
}

extern "C" int mdb_dev_explain_join_payload(const struct mdb_dev_explain_request *rq, int cells, struct mdb_dev_plan_info *out)
{
	if (!rq || !out || cells < 1 || cells > 2)
		return -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = explain_ctx(rq, out, true);
	if (!ctx)
		return -MIDORIDB_NOMEM;
	const void *pay[2] = { (const void *)0x5000, (const void *)0x6000 };
	void *dst[2] = { (void *)0x7000, (void *)0x8000 };
	if (rq->further_tables) {	/* several right tables on the one key: mdb_dev_join_payload_multi, the window = the statistics' ranges */
		struct mdb_dev_payload_right rt[3];
		memset(rt, 0, sizeof(rt));
		int64_t lo = rq->left.min < rq->right.min ? rq->left.min : rq->right.min, hi = rq->left.max > rq->right.max ? rq->left.max : rq->right.max;
		const uint32_t nrt = 1u + (rq->further_tables > 2u ? 2u : rq->further_tables);
		for (uint32_t t = 0; t < nrt; t++) {
			rt[t].keys = EXPLAIN_KR;
			rt[t].rows = t ? rq->further_rows[t - 1] : rq->right.rows;
			rt[t].npay = cells;
			for (int c = 0; c < cells; c++) {
				rt[t].pay_in[c] = pay[c];
				rt[t].out[c] = dst[c];
			}
		}
		int mrc = rq->left_nulls_bitmap ? 1 : mdb_dev_join_payload_multi(ctx, EXPLAIN_KL, NULL, rq->left.rows, rt, (int)nrt, lo, hi);
		if (mrc == 1)
			mrc = MIDORIDB_OK;	/* (not served: payload_form 0 - the caller joins table by table) */
		delete ctx;
		return mrc;
	}
	int rc = mdb_dev_join_payload(ctx, EXPLAIN_KL, rq->left_nulls_bitmap ? (const uint64_t *)0x4000 : NULL, rq->left.rows, EXPLAIN_KR, NULL, rq->right.rows, pay,
				      cells, dst);
	if (rc == 1)
		rc = MIDORIDB_OK;	/* (not served: payload_form 0 - mdb_dev_join_pairs and a gather answer) */
	delete ctx;
	return rc;
}

extern "C" int mdb_dev_explain_group_count(const struct mdb_dev_explain_request *rq, struct mdb_dev_plan_info *out)
{
	if (!rq || !out)
		return -MIDORIDB_ERROR;
	mdb_dev_ctx *ctx = explain_ctx(rq, out, false);
	if (!ctx)
		return -MIDORIDB_NOMEM;
	uint64_t groups = 0;
	const int rc = mdb_dev_group_count(ctx, EXPLAIN_KL, rq->left_nulls_bitmap ? (const uint64_t *)0x4000 : NULL, rq->left.rows, MDB_ORDER_FIRST, NULL, NULL,
					   rq->left.rows, &groups);
	delete ctx;
	return rc;
}
