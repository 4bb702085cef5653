/*
 * mdb_dev_partition.hip - LDS-staged MSD radix partitioning of a key column by the top
 * bits of fmix64(key).  This is the bandwidth-dominant stage of the join / GROUP BY pipeline.
 *
 * Why it exists: the reference joins by comparing every pair of rows (reference
 * src/engine/executor_select.c:1096-1141, O(nA*nB)) and groups by comparing every pair of
 * surviving rows (:1542-1583, O(n^2)).  Here both tables are split into 2^bits "leaves" whose
 * build side fits one CU's LDS; equal keys always land in the same leaf, so each leaf is joined /
 * grouped independently by one workgroup (mdb_dev_join.hip).
 *
 * The default ("FAST") form has NO histogram pass: every child owns a fixed-capacity region and each (tile, digit)
 * run reserves its place with one global atomic on the region's cursor (first level: 8 sub-regions per digit, one
 * per XCD, cursors laid out so that a tile's 256 atomics fall into 8 lines private to its XCD); an overflowing
 * region (skew) is flagged on the device and the caller redoes the operator with the exact form below, which is
 * also what the stable variant (materialising N:M join) and the destination partition (multi-GPU) use.
 *
 * Exact form, one level = histogram -> exclusive scan -> scatter, all tile based (MDB_TILE = 4096 keys):
 *
 *   k_part_hist     per tile: LDS histogram of the level's digit, written digit-major
 *   (scan)          one exclusive scan over [segment][digit][tile] gives every tile its output
 *                   offset per digit and, as a by-product, the exact start of every child segment
 *   k_part_scatter  per tile: one LDS atomic per key gives its rank inside its digit, keys are
 *                   staged sorted-by-digit in LDS, then written out as contiguous runs per digit
 *                   (coalesced; TILE/R keys per run)
 *
 * Level 0 reads the raw int64 keys (+ NULL bits, NULL rows are dropped - a NULL key never joins,
 * executor_select.c:557-579), hashes them, and attaches the row id; later levels move (hash, rid).
 * Order inside a leaf is unspecified; the consumers restore the reference's orders from the row ids.
 *
 * Memory traffic per level and key: hist reads 8 B, scatter reads 8(+4) B and writes 8(+4) B.
 * Blocks are mapped to tiles XCD-contiguously (blockIdx % 8 selects the XCD) so that adjacent
 * tiles - which extend each other's output runs - meet in the same 4 MiB L2.
 */
#include <stdlib.h>
#include <type_traits>
#include "mdb_dev_internal.h"

#define PART_THREADS 512
#define PART_WAVES (PART_THREADS / MDB_WAVE)
#define PART_ITEMS (MDB_TILE / PART_THREADS)		/* 8 keys per thread */
#define PART_WAVE_SPAN (MDB_TILE / PART_WAVES)		/* 512 consecutive keys per wave */
#define PART_MAX_R (1u << MDB_MAX_RADIX_BITS)
#define PART_FEW_DIGITS 4u	/* up to this many digits (destination GPUs) a wave ranks its keys with ballots, not one atomic per key */
#define PART_INVALID 0xFFFFFFFFu

struct mdb_level_args {
	/* level 0 input */
	const int64_t *keys;
	const uint64_t *nullbits;
	uint64_t n;
	/* level >= 1 input */
	const uint64_t *hv_in;
	const uint32_t *rid_in;
	const mdb_tile_desc *tiles;	/* NULL for level 0 (computed arithmetically) */
	/* output */
	uint64_t *hv_out;
	uint32_t *rid_out;		/* NULL = row ids not wanted */
	uint32_t *hist;			/* counts in, scanned offsets out */
	uint32_t ntiles;		/* tiles to cover (upper bound for level >= 1) */
	uint32_t R;			/* digits at this level */
	uint32_t shift;			/* RADIX mode: digit = (hv >> shift) & (R - 1) */
	uint32_t mbits;			/* bits needed to tell digits apart (ballot rounds of the STABLE form) */
	uint32_t mode;			/* enum mdb_digit_mode */
	uint32_t inverse_out;		/* write fmix64^-1(hv) (= the original key) instead of hv; 2 = as int32 (4-byte wire format) */
	/* FAST (histogram-free) form: child (seg, digit) owns the fixed-capacity region [child*cap, child*cap+cap) */
	uint32_t *cursor;		/* per child: elements placed so far (zeroed before the launch) */
	uint32_t cap;
	uint32_t skip_zero;		/* raw sort keys: a zero word is a gap in the input list, not a key */
	uint32_t nsub;			/* > 0: first level, child = digit * nsub + (block % nsub) sub-region; 0: child = seg * R + digit */
	uint32_t *status;		/* bit 1 set when a child overflowed its region */
	/* level 0, narrow form: every key must lie within 2^31 of narrow_base (otherwise bit 7 of *status is raised and the caller
	 * redoes the operator in the wide form); the word written is fmix32(low 32 bits of the key) << 32, plus the row id in
	 * the low half when narrow == 1 - hash and row id then travel as ONE 8-byte word and no row-id array exists - or
	 * the hash once more when narrow == 2 */
	uint32_t narrow;
	int64_t narrow_base;		/* narrow form: centre of the 2^32-wide window of key values that the 32-bit words can tell apart
					 * (compact form, narrow_kbits != 0: the window's LOW end) */
	uint32_t narrow_kbits;		/* compact narrow form: every key must lie in [narrow_base, narrow_base + 2^narrow_kbits); the 32-bit
					 * hash is mixk(key - narrow_base) in the TOP narrow_kbits bits of the field, zeros below */
	uint32_t keys32;		/* level 0: `keys` is an array of int32 (keys that crossed xGMI in the 4-byte wire format) */
	/* semi-join filter (second level, left side of a join in the compact narrow form; struct mdb_part_filter): the slice of
	 * first-level digit td.seg is words [seg * filter_words, ...); a row whose bit (hash32 >> filter_shift, masked to the
	 * slice) is clear can join nothing and is dropped before it is ranked */
	const uint32_t *filter;
	uint32_t filter_words, filter_shift;
	/* min-max pruning (first level, compact narrow form): the RIGHT table's launch leaves the smallest and the largest
	 * key - window base it saw in minmax_out[0..1] (atomics; the caller has stored 0xFFFFFFFF, 0 there), the LEFT table's
	 * launch - later on the same stream - reads them through range_in and drops every row outside: such a row can join
	 * nothing.  Exact, and free: one compare per row, no extra pass, no host round trip */
	uint32_t *minmax_out;
	uint32_t *minmax_final;	/* with minmax_out: the two words the reduce kernel's workgroups meet in - preset here (no key: lo > hi) */
	const uint32_t *range_in;
	/* the same for the 64-bit form (keys that fit no 2^32 window: hashes, snowflake ids - R64 instances): per-tile pairs of the
	 * smallest / largest key as order-preserving unsigned images (key ^ 2^63; the largest stored inverted) / the other table's
	 * range [lo, hi] as two signed keys in device memory */
	unsigned long long *minmax64_out;
	const long long *range64_in;
	/* PAY instance (first level of a join that carries payload cells through its one partition level): up to two 8-byte columns of
	 * the table, read at the row's own index and written at the row's place in the regions, beside its word */
	const uint64_t *pay_in[2];
	uint64_t *pay_out[2];
	uint32_t npay;
	/* first level: rows whose key lies outside [keep_lo, keep_hi] are dropped (keep_on; partition by destination: the other
	 * table's global key range is known before the exchange - nothing outside it can join on any GPU) */
	uint32_t keep_on;
	int64_t keep_lo, keep_hi;
	uint32_t own_on;		/* every key of this table was promised to lie in [own_lo, own_hi]: one that does not raises status bit 10 */
	int64_t own_lo, own_hi;
	uint32_t out16_shift;		/* OUT16 instance (first level only, compact narrow form, 4-byte words): what is WRITTEN is the 16-bit word
					 * (uint16_t)(hash32 >> out16_shift) - the hash bits below the digit, all the one-level leaf kernel needs */
	uint32_t fold64;		/* raw 4-byte words: the input is still the list of 8-byte records, folded on the fly (record >> 32 | low
					 * half: the caller knows that the two parts do not overlap) */
};

/* level-0 word of one key */
/* CF: the compact narrow form of an int64 column, known at COMPILE time (RIDW: the word's low half is the row id, else the
 * hash once more).  The generic body below picks the form per row from run-time fields - uniform branches and selects that
 * made the first-level kernels issue-bound (75 vector + 40 scalar instructions per row, profiles/r02). */
template <bool CF = false, bool RIDW = false>
__device__ static inline uint64_t part_hash_key(const mdb_level_args &a, uint64_t key, uint32_t rid, bool *bad, uint64_t *rel = nullptr)
{
	if (CF) {
		key -= (uint64_t)a.narrow_base;
		if (rel)
			*rel = key;
		*bad = *bad || (key >> a.narrow_kbits);
		const uint32_t h = mdb_mixk((uint32_t)key, a.narrow_kbits) << (32u - a.narrow_kbits);
		return ((uint64_t)h << 32) | (RIDW ? rid : h);
	}
	if (!a.narrow)
		return mdb_fmix64(key);
	key -= (uint64_t)a.narrow_base;		/* (wraps: the tests below are on the 64-bit difference) */
	if (rel)
		*rel = a.narrow_kbits ? key : key + 0x80000000ull;	/* a valid key's value is below 2^32 in both forms */
	uint32_t h;
	if (a.narrow_kbits) {
		*bad = *bad || (key >> a.narrow_kbits);
		h = mdb_mixk((uint32_t)key, a.narrow_kbits) << (32u - a.narrow_kbits);
	} else {
		*bad = *bad || (key + 0x80000000ull) >> 32;
		h = mdb_fmix32((uint32_t)key);
	}
	return ((uint64_t)h << 32) | (a.narrow == 1 ? rid : h);
}

template <bool CF = false, typename W>
__device__ static inline uint32_t part_digit(const mdb_level_args &a, W hv)
{
	if (CF || a.mode == MDB_DIGIT_RADIX)
		return (uint32_t)(hv >> a.shift) & (a.R - 1);	/* a.shift counts from the width of W */
	return (a.R & (a.R - 1)) == 0 ? (uint32_t)hv & (a.R - 1) : (uint32_t)hv % a.R;	/* 2, 4, 8 GPUs: no integer division */
}

/* XCD-contiguous block -> tile map: blocks b and b+8 share an XCD (round-robin dispatch), so give
 * each XCD one contiguous run of tiles.  The grid is rounded up to a multiple of 8. */
__device__ static inline uint32_t part_tile_of_block(void)
{
	const uint32_t per = gridDim.x >> 3;
	return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}

template <uint32_t TILE = MDB_TILE>
__device__ static inline mdb_tile_desc part_get_tile(const mdb_level_args &a, uint32_t t)
{
	mdb_tile_desc d;
	if (t >= a.ntiles) {
		d.start = d.len = d.hbase = d.nt = d.seg = 0;
		return d;
	}
	if (a.tiles)
		return a.tiles[t];
	const uint64_t start = (uint64_t)t * TILE;
	d.start = (uint32_t)start;
	d.len = start < a.n ? (uint32_t)((a.n - start) < TILE ? (a.n - start) : TILE) : 0;
	d.hbase = t;
	d.nt = a.ntiles;
	d.seg = 0;
	return d;
}

/* Load element i of the tile: returns false for a NULL key (level 0 only). */
template <bool LEVEL0>
__device__ static inline bool part_load(const mdb_level_args &a, const mdb_tile_desc &td, uint32_t i, uint64_t *hv,
					uint32_t *rid)
{
	const uint64_t g = (uint64_t)td.start + i;
	if (LEVEL0) {
		if (a.nullbits && mdb_bit_is_set(a.nullbits, g))
			return false;
		bool bad = false;
		const int64_t key = a.keys32 ? (int64_t)reinterpret_cast<const int32_t *>(a.keys)[g] : a.keys[g];
		*hv = part_hash_key(a, (uint64_t)key, (uint32_t)g, &bad);
		*rid = (uint32_t)g;
	} else {
		*hv = a.hv_in[g];
		*rid = a.rid_in ? a.rid_in[g] : 0;
	}
	return true;
}

/* Load the two adjacent elements 2p, 2p+1 (relative to the even-aligned base of the tile) with one 16-byte
 * access when both belong to the tile (8-byte accesses reach only ~0.6x of the 16-byte rate,
 * MI355X_MICROARCH.md).  lead = 1 when the tile starts on an odd element.  valid[k] = element exists and is
 * not NULL. */
template <bool LEVEL0, bool HAS_RID, bool RAW = false, bool INV = false, bool KEEP = true /* compile the key-range tests (by-destination kernels) */,
	  bool CF = false, bool RIDW = false /* see part_hash_key */>
__device__ static inline void part_load2(const mdb_level_args &a, const mdb_tile_desc &td, uint32_t p, uint64_t hv[2],
					 uint32_t rid[2], bool valid[2], uint64_t *rel = nullptr /* [2]: key - narrow_base (narrow forms) */,
					 const ulonglong2 *pre = nullptr /* the pair, already loaded (full, 16-byte aligned tiles: the caller
									  * issues a tile's loads together - see part_preload2) */,
					 int64_t *raw_out = nullptr /* [2]: the keys themselves (level 0; the 64-bit form's key-range tests) */)
{
	const uint32_t lead = td.start & 1u;
	const uint64_t base2 = (uint64_t)(td.start - lead);
	const uint32_t e0 = 2 * p, e1 = 2 * p + 1;
	const bool in0 = pre || (e0 >= lead && e0 < lead + td.len);
	const bool in1 = pre || e1 < lead + td.len;	/* e1 >= 1 >= lead always */
	const uint64_t g0 = base2 + e0;
	bool bad[2] = { false, false };
	int64_t raw_key[2] = { 0, 0 };
	hv[0] = hv[1] = 0;
	rid[0] = rid[1] = 0;
	valid[0] = in0;
	valid[1] = in1;
	if (in0 && in1) {
		if (LEVEL0) {
			ulonglong2 k;
			if (pre) {
				k = *pre;
			} else if (!CF && a.keys32) {
				const int2 q = *reinterpret_cast<const int2 *>(reinterpret_cast<const int32_t *>(a.keys) + g0);
				k.x = (uint64_t)(int64_t)q.x;
				k.y = (uint64_t)(int64_t)q.y;
			} else {
				/* (a non-temporal load here made the first-level kernels 3-8 % faster and the kernel that reads their output
				 * 10-20 % slower: +-1 % per query, same-box A/B) */
				k = *reinterpret_cast<const ulonglong2 *>(a.keys + g0);
			}
			raw_key[0] = (int64_t)k.x;
			raw_key[1] = (int64_t)k.y;
			hv[0] = INV ? k.x : part_hash_key<CF, RIDW>(a, k.x, (uint32_t)g0, &bad[0], rel ? &rel[0] : nullptr);	/* INV: the caller keeps the key itself */
			hv[1] = INV ? k.y : part_hash_key<CF, RIDW>(a, k.y, (uint32_t)g0 + 1, &bad[1], rel ? &rel[1] : nullptr);
			rid[0] = (uint32_t)g0;
			rid[1] = (uint32_t)g0 + 1;
		} else {
			const ulonglong2 k = pre ? *pre : *reinterpret_cast<const ulonglong2 *>(a.hv_in + g0);
			hv[0] = k.x;
			hv[1] = k.y;
			if (HAS_RID) {
				const uint2 q = *reinterpret_cast<const uint2 *>(a.rid_in + g0);
				rid[0] = q.x;
				rid[1] = q.y;
			}
		}
	} else if (in0 || in1) {
		const int k = in0 ? 0 : 1;
		const uint64_t g = g0 + (uint64_t)k;
		if (LEVEL0) {
			const int64_t key = (!CF && a.keys32) ? (int64_t)reinterpret_cast<const int32_t *>(a.keys)[g] : a.keys[g];
			raw_key[k] = key;
			hv[k] = INV ? (uint64_t)key : part_hash_key<CF, RIDW>(a, (uint64_t)key, (uint32_t)g, &bad[k], rel ? &rel[k] : nullptr);
			rid[k] = (uint32_t)g;
		} else {
			hv[k] = a.hv_in[g];
			if (HAS_RID)
				rid[k] = a.rid_in[g];
		}
	}
	if (LEVEL0 && raw_out) {
		raw_out[0] = raw_key[0];
		raw_out[1] = raw_key[1];
	}
	if (LEVEL0 && a.nullbits) {
		if (valid[0] && mdb_bit_is_set(a.nullbits, g0))
			valid[0] = false;
		if (valid[1] && mdb_bit_is_set(a.nullbits, g0 + 1))
			valid[1] = false;
	}
	if (LEVEL0 && KEEP && a.own_on && ((valid[0] && (raw_key[0] < a.own_lo || raw_key[0] > a.own_hi)) ||
				   (valid[1] && (raw_key[1] < a.own_lo || raw_key[1] > a.own_hi))))
		mdb_raise(a.status, 1024u);
	if (LEVEL0 && KEEP && a.keep_on) {
		valid[0] = valid[0] && raw_key[0] >= a.keep_lo && raw_key[0] <= a.keep_hi;
		valid[1] = valid[1] && raw_key[1] >= a.keep_lo && raw_key[1] <= a.keep_hi;
	}
	/* (with min-max pruning - range_in - a key outside the window is outside the right table's range, which lies inside the
	 * window: the caller drops the row, nothing to report) */
	if (LEVEL0 && !a.range_in && ((bad[0] && valid[0]) || (bad[1] && valid[1])))
		mdb_raise(a.status, 128u);	/* a key outside the int32 range: the narrow form does not apply */
	if (RAW && a.skip_zero) {
		valid[0] = valid[0] && hv[0] != 0;
		valid[1] = valid[1] && hv[1] != 0;
	}
}

/* A full tile that starts on an even element: every thread's pairs exist, and their loads can be issued TOGETHER, ahead of
 * the hashing and ranking.  Inside part_load2's range tests each load sits in its own branch, the compiler waits for it
 * (s_waitcnt vmcnt(0)) before the next one is issued, and a thread has ONE 16-byte load in flight: the first-level kernels
 * ran at 3.6 TB/s where the copy rate is 5. */
template <bool LEVEL0, int PAIRS>
__device__ static inline void part_preload2(const mdb_level_args &a, const mdb_tile_desc &td, ulonglong2 pre[PAIRS])
{
	const uint64_t *const src = (LEVEL0 ? reinterpret_cast<const uint64_t *>(a.keys) : a.hv_in) + td.start;
#pragma unroll
	for (int r = 0; r < PAIRS; r++) {
		if (LEVEL0) {	/* a key column is read once: non-temporal, so that it does not push the regions' half-written lines out of the L2s */
			typedef unsigned long long ull2_nt __attribute__((ext_vector_type(2)));
			const ull2_nt v = __builtin_nontemporal_load(reinterpret_cast<const ull2_nt *>(src + 2u * ((uint32_t)r * PART_THREADS + threadIdx.x)));
			pre[r] = make_ulonglong2(v.x, v.y);
		} else {
			pre[r] = *reinterpret_cast<const ulonglong2 *>(src + 2u * ((uint32_t)r * PART_THREADS + threadIdx.x));
		}
	}
}

/* 4-byte words (the right side of the narrow form beyond level 0): the four adjacent elements 4p .. 4p+3 relative to
 * the 4-aligned base of the tile, one 16-byte access when all belong to the tile */
__device__ static inline void part_load4_w32(const mdb_level_args &a, const mdb_tile_desc &td, uint32_t p, uint32_t hv[4], bool valid[4],
					      const uint4 *pre = nullptr /* the four words, already loaded (full aligned tiles) */)
{
	const uint32_t lead = td.start & 3u;
	const uint32_t *src = reinterpret_cast<const uint32_t *>(a.hv_in) + (uint64_t)(td.start - lead) + 4 * (uint64_t)p;
	const uint32_t e0 = 4 * p, end = lead + td.len;
	if (pre || (e0 >= lead && e0 + 3 < end)) {
		const uint4 q = pre ? *pre : *reinterpret_cast<const uint4 *>(src);
		hv[0] = q.x;
		hv[1] = q.y;
		hv[2] = q.z;
		hv[3] = q.w;
		valid[0] = valid[1] = valid[2] = valid[3] = true;
	} else {
#pragma unroll
		for (int k = 0; k < 4; k++) {
			valid[k] = e0 + k >= lead && e0 + k < end;
			hv[k] = valid[k] ? src[k] : 0u;
		}
	}
	if (a.skip_zero) {
#pragma unroll
		for (int k = 0; k < 4; k++)
			valid[k] = valid[k] && hv[k] != 0u;
	}
}

template <bool LEVEL0, bool RAW = false>
__global__ __launch_bounds__(PART_THREADS) void k_part_hist(mdb_level_args a)
{
	__shared__ uint32_t s_h[PART_MAX_R];
	const mdb_tile_desc td = part_get_tile(a, part_tile_of_block());
	if (td.len == 0)
		return;
	for (uint32_t d = threadIdx.x; d < a.R; d += PART_THREADS)
		s_h[d] = 0;
	__syncthreads();
	const bool full = td.len == MDB_TILE && !(td.start & 1u) && !(LEVEL0 && a.keys32);	/* (uniform) the tile's loads are issued together */
	ulonglong2 pre[PART_ITEMS / 2];
	if (full)
		part_preload2<LEVEL0, PART_ITEMS / 2>(a, td, pre);
#pragma unroll
	for (int r = 0; r < PART_ITEMS / 2; r++) {
		uint64_t hv[2];
		uint32_t rid[2];
		bool valid[2];
		if (full)
			part_load2<LEVEL0, false, RAW>(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, hv, rid, valid, nullptr, &pre[r]);
		else
			part_load2<LEVEL0, false, RAW>(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, hv, rid, valid);
		if (LEVEL0 && a.R <= PART_FEW_DIGITS) {	/* few destinations: one add per wave and digit (see k_part_scatter) */
			for (int k = 0; k < 2; k++) {
				const uint32_t dg = valid[k] ? part_digit(a, hv[k]) : PART_INVALID;
				for (uint32_t d = 0; d < a.R; d++) {
					const uint64_t m = __ballot(dg == d);
					if (m && mdb_lane() == (uint32_t)__ffsll((long long)m) - 1u)
						atomicAdd(&s_h[d], (uint32_t)__popcll(m));
				}
			}
			continue;
		}
		if (valid[0])
			atomicAdd(&s_h[part_digit(a, hv[0])], 1u);
		if (valid[1])
			atomicAdd(&s_h[part_digit(a, hv[1])], 1u);
	}
	__syncthreads();
	for (uint32_t d = threadIdx.x; d < a.R; d += PART_THREADS)
		a.hist[(uint64_t)td.hbase + (uint64_t)d * td.nt] = s_h[d];
}

/*
 * Scatter: the tile is ranked with LDS atomics (one ds_add_rtn per key returns the key's rank inside
 * its digit), staged in LDS sorted by digit, and written out as one contiguous run per digit.
 * Order inside a digit is arrival order, i.e. unspecified: no consumer depends on it (group order is
 * restored from row ids with atomicMin).  STABLE = true instead ranks with wave-level ballot
 * matching and keeps input order inside every digit; rocprofv3 showed that form issue-bound (SIMD
 * ~100 % busy at 2.3 TB/s vs 3.7 TB/s, profiles/r01/), so it is used only where order matters: the
 * right side of the materialising join, whose per-key row-id lists must come out ascending.
 *
 * LDS: hv 32 KiB (+ rid 16 KiB when row ids travel) + 3 KiB of per-digit words => 4 (3) workgroups/CU
 * (+16 KiB for STABLE).
 */
/* The rows of one thread in a FULL tile of the compact narrow form, no NULL bitmap: straight-line code, no per-row branch
 * (the generic loop below spends ~25 scalar branches per pair of rows on conditions that are the same for the whole launch).
 * MODE 1 (the one in use): min-max pruning, the left table - rows outside [range_lo, range_hi] are dropped (a key outside the
 * window is outside the range: nothing to report); 0: plain; 2: the right table of a pruned join - the smallest / largest
 * key - window base is recorded.  -> a key outside the window was seen (modes 0 and 2). */
template <bool W32v, int MODE, typename W, int ITEMS = PART_ITEMS>
__device__ static inline bool part_cf_rows(const mdb_level_args &a, const ulonglong2 *pre, uint32_t row0, W *hv, uint32_t *dig, uint64_t range_lo,
					   uint64_t range_hi, uint32_t &seen_min, uint32_t &seen_max)
{
	bool any_bad = false;
	const uint32_t dshift = a.shift - (W32v ? 0u : 32u);	/* the digit's place inside the 32-bit hash */
#pragma unroll
	for (int r = 0; r < ITEMS / 2; r++) {
#pragma unroll
		for (int e = 0; e < 2; e++) {
			const uint64_t key = (e ? pre[r].y : pre[r].x) - (uint64_t)a.narrow_base;
			bool ok = true;
			if (MODE == 1)
				ok = key >= range_lo && key <= range_hi;
			else
				any_bad = any_bad || (key >> a.narrow_kbits);
			if (MODE == 2) {
				seen_min = min(seen_min, (uint32_t)key);
				seen_max = max(seen_max, (uint32_t)key);
			}
			const uint32_t h = mdb_mixk((uint32_t)key, a.narrow_kbits) << (32u - a.narrow_kbits);
			const uint32_t rid = row0 + 2u * (uint32_t)r * PART_THREADS + (uint32_t)e;
			hv[2 * r + e] = W32v ? (W)h : (W)(((uint64_t)h << 32) | rid);
			dig[2 * r + e] = ok ? ((h >> dshift) & (a.R - 1u)) : PART_INVALID;
		}
	}
	return any_bad;
}

/* One instance of the scatter kernel = one of these structs: what the instance knows at compile time.
 *   LEVEL0  reads the key column (hashes, drops NULLs); otherwise the previous level's words through tile descriptors
 *   HAS_RID row ids travel in an array of their own (the 64-bit form's left side); STABLE keeps the input order inside a digit
 *   FAST    histogram-free: fixed-capacity regions + cursors (otherwise exact offsets from a histogram pass)
 *   RAW     the input words are ready-made sort keys (ordering of group records); W32: 4-byte words; OUT16: 2-byte words out
 *   INV     partition by destination GPU: what is written is the KEY (and the key-range tests are compiled in)
 *   FILT    second level with the semi-join bitmap; CF: first level of the compact narrow form over an int64 column
 * The struct's name is what a profiler shows as the kernel's template argument. */
struct pf_base {
	static constexpr bool LEVEL0 = false, HAS_RID = false, STABLE = false, FAST = false, RAW = false, W32 = false, INV = false, FILT = false,
			      OUT16 = false, CF = false, R64 = false, MM64 = false, PAY = false;
	static constexpr uint32_t TMUL = 1;	/* tile = TMUL x MDB_TILE rows (a histogram-free pass that cuts its own tiles: large tables) */
};
struct pf_word_hist : pf_base {  };
struct pf_word_hist_raw : pf_base { static constexpr bool RAW = true; };
struct pf_word : pf_base { static constexpr bool FAST = true; };
struct pf_word_semi : pf_base { static constexpr bool FAST = true; static constexpr bool FILT = true; };
struct pf_word_w32 : pf_base { static constexpr bool FAST = true; static constexpr bool W32 = true; };
struct pf_word_w32_out16 : pf_base { static constexpr bool FAST = true; static constexpr bool W32 = true; static constexpr bool OUT16 = true; };
struct pf_word_raw : pf_base { static constexpr bool FAST = true; static constexpr bool RAW = true; };
struct pf_word_raw_w32 : pf_base { static constexpr bool FAST = true; static constexpr bool RAW = true; static constexpr bool W32 = true; };
struct pf_word_rid_hist : pf_base { static constexpr bool HAS_RID = true; };
struct pf_word_rid : pf_base { static constexpr bool HAS_RID = true; static constexpr bool FAST = true; };
struct pf_word_rid_stable_hist : pf_base { static constexpr bool HAS_RID = true; static constexpr bool STABLE = true; };
struct pf_key_hist : pf_base { static constexpr bool LEVEL0 = true; };
struct pf_key_hist_dest : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool INV = true; };
struct pf_key : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; };
struct pf_key_cf : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool CF = true; };
struct pf_key_w32 : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool W32 = true; };
struct pf_key_w32_cf : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool W32 = true; static constexpr bool CF = true; };
struct pf_key_w32_out16 : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool W32 = true; static constexpr bool OUT16 = true; };
struct pf_key_w32_out16_cf : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool W32 = true; static constexpr bool OUT16 = true; static constexpr bool CF = true; };
struct pf_word_pay : pf_base { static constexpr bool FAST = true; static constexpr bool PAY = true; };	/* second level of a join that carries payload cells */
struct pf_key_cf_pay : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool CF = true; static constexpr bool PAY = true; };
struct pf_key_rid_hist : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool HAS_RID = true; };
struct pf_key_rid_hist_dest : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool HAS_RID = true; static constexpr bool INV = true; };
struct pf_key_rid : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool HAS_RID = true; static constexpr bool FAST = true; };
struct pf_key_rid_stable_hist : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool HAS_RID = true; static constexpr bool STABLE = true; };
/* the same with tiles of 2 x MDB_TILE rows (tables of 2^25 rows and more: half the cursor atomics, runs twice as long - whole 32-byte sectors of
 * 2-byte words; 10^8 rows: the right table's pass 0.241 -> 0.222 ms, the ordering sort's first level 0.174 -> 0.155, same-box A/B.  Smaller tables
 * keep MDB_TILE: fewer tiles than workgroup slots leave CUs idle) */
#ifndef PART_TMUL
#define PART_TMUL 2u
#endif
struct pf_key_cf_t2 : pf_key_cf { static constexpr uint32_t TMUL = PART_TMUL; };
struct pf_key_w32_out16_cf_t2 : pf_key_w32_out16_cf { static constexpr uint32_t TMUL = PART_TMUL; };
struct pf_word_raw_w32_t2 : pf_word_raw_w32 { static constexpr uint32_t TMUL = PART_TMUL; };
/* min-max pruning in the 64-bit form: the right table's first level records its key range (mm64), the left table's drops the rows outside (r64) */
struct pf_key_mm64 : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool FAST = true; static constexpr bool MM64 = true; };
struct pf_key_rid_r64 : pf_base { static constexpr bool LEVEL0 = true; static constexpr bool HAS_RID = true; static constexpr bool FAST = true; static constexpr bool R64 = true; };

template <typename F /* one of the pf_* structs above */>
__global__ __launch_bounds__(PART_THREADS) void k_part_scatter(mdb_level_args a)
{
	constexpr bool LEVEL0 = F::LEVEL0, HAS_RID = F::HAS_RID, STABLE = F::STABLE, FAST = F::FAST, RAW = F::RAW, W32 = F::W32, INV = F::INV, FILT = F::FILT,
		       OUT16 = F::OUT16, CF = F::CF, R64 = F::R64, MM64 = F::MM64, PAY = F::PAY;
	static_assert(!PAY || (FAST && !W32 && !HAS_RID && !RAW && !INV && !FILT && (CF || !LEVEL0)),
		      "payload cells travel with the compact narrow form's 8-byte hash | row id words: its first level, or a second level over them");
	static_assert(!(R64 || MM64) || (LEVEL0 && FAST && !RAW && !INV && !W32 && !CF && !STABLE), "64-bit key range: first level of the 64-bit form only");
	static_assert(!CF || (LEVEL0 && FAST && !RAW && !INV && !HAS_RID && !STABLE), "compact-form instance: first level of the narrow join forms only");
	static_assert(!OUT16 || (W32 && !RAW), "2-byte words out: the 4-byte form only");
	static_assert(!FILT || (!LEVEL0 && !HAS_RID && !STABLE && FAST && !RAW && !W32 && !INV), "semi-join filter: second level of the narrow left side only");
	/* INV (destination partition for the exchange): what is staged and written is the KEY, not its hash - the digit is
	 * taken from the hash once, at load time, and found again at write-out from the staged position (the tile-local
	 * digit starts are in LDS anyway), instead of hashing back and forth */
	static_assert(!INV || (LEVEL0 && !STABLE && !FAST && !RAW && !W32), "INV: unordered exact level-0 form only");
	/* W32: the words staged and written are 4-byte hashes (narrow form, right side: level 0 reads 8-byte keys and
	 * writes 4-byte words, level 1 reads and writes 4-byte words); never with row ids */
	static_assert(!W32 || (!HAS_RID && !STABLE && FAST && !(LEVEL0 && RAW)), "4-byte words: unordered FAST form without row ids only");
	constexpr uint32_t TILE = MDB_TILE * F::TMUL;
	constexpr int ITEMS = (int)(PART_ITEMS * F::TMUL);
	static_assert(F::TMUL == 1 || (FAST && !STABLE && !HAS_RID && !INV && !FILT && !PAY && !R64 && !MM64), "larger tiles: histogram-free passes that cut their own tiles only");
	typedef typename std::conditional<W32, uint32_t, uint64_t>::type W;
	/* Row ids are staged through the SAME LDS as the hashes, after the hashes have been written out (unordered
	 * form only): 35 KiB instead of 51 KiB per workgroup = 4 instead of 3 workgroups per CU, worth ~25 % of the
	 * kernel's time (occupancy is what hides the HBM latency of the scattered runs) */
	constexpr bool RID_SHARES_LDS = HAS_RID && !STABLE;
	__shared__ W s_hv[TILE];
	__shared__ uint32_t s_rid_own[(HAS_RID && !RID_SHARES_LDS) ? TILE : 1];
	uint32_t *const s_rid = RID_SHARES_LDS ? reinterpret_cast<uint32_t *>(s_hv) : s_rid_own;
	__shared__ uint32_t s_cnt[PART_MAX_R];		/* per-digit counters, then tile-local digit starts */
	__shared__ int32_t s_delta[PART_MAX_R];	/* global start of the digit's run minus its tile-local start */
	__shared__ uint32_t s_wcnt[STABLE ? PART_WAVES * PART_MAX_R : 1];	/* STABLE: per-wave digit counts, then bases */
	__shared__ uint8_t s_ok[FAST ? PART_MAX_R : 1];			/* FAST: the digit's run fits its region */
	__shared__ uint32_t s_tmp[32];

	if (LEVEL0 && FAST && !RAW && !INV && !HAS_RID && a.minmax_final && blockIdx.x == 0 && threadIdx.x == 0) {
		a.minmax_final[0] = 0xFFFFFFFFu;
		a.minmax_final[1] = 0u;
	}
	const mdb_tile_desc td = part_get_tile<TILE>(a, part_tile_of_block());
	if (td.len == 0)
		return;
	const uint32_t R = a.R;
	const uint32_t wave = threadIdx.x >> 6, lane = mdb_lane();

	for (uint32_t d = threadIdx.x; d < R; d += PART_THREADS)
		s_cnt[d] = 0;
	if (STABLE)
		for (uint32_t i = threadIdx.x; i < PART_WAVES * PART_MAX_R; i += PART_THREADS)
			s_wcnt[i] = 0;
	/* min-max pruning: the other table's key range (left side) / what this launch sees (right side) */
	uint64_t range_lo = 0, range_hi = ~0ull;
	uint32_t seen_min = 0xFFFFFFFFu, seen_max = 0u;
	__shared__ uint32_t s_mm[2 * PART_WAVES];
	if (LEVEL0 && a.range_in) {
		range_lo = a.range_in[0];
		range_hi = a.range_in[1];
	}
	/* 64-bit form: the keys' order-preserving unsigned images (key ^ 2^63) */
	uint64_t seen_min64 = ~0ull, seen_max64 = 0ull, r64_lo = 0ull, r64_hi = ~0ull;
	uint64_t *const s_mm64 = reinterpret_cast<uint64_t *>(s_delta);	/* (2 x PART_WAVES words of a buffer that step 3 fills later) */
	if (R64 && a.range64_in) {	/* the other table's key range, left there by its own first level earlier on this stream */
		r64_lo = (uint64_t)a.range64_in[0] ^ 0x8000000000000000ull;
		r64_hi = (uint64_t)a.range64_in[1] ^ 0x8000000000000000ull;
	}
	/* the filter slice of this tile's first-level digit lives in the staging buffer until the rows are ranked (the
	 * barrier after the load separates its last read from the first staged word) */
	uint32_t *const s_flt = reinterpret_cast<uint32_t *>(s_hv);
	if (FILT) {
		const uint4 *src = reinterpret_cast<const uint4 *>(a.filter + (size_t)td.seg * a.filter_words);
		for (uint32_t w = threadIdx.x; w < a.filter_words / 4; w += PART_THREADS)
			reinterpret_cast<uint4 *>(s_flt)[w] = src[w];
		__syncthreads();
	}

	/* 1. load (coalesced).  STABLE: wave w owns the 512 consecutive keys [w*512, w*512+512), so that
	 *    (wave, round, lane) order is input order; otherwise consecutive threads, consecutive keys. */
	W hv[ITEMS];
	uint32_t rid[ITEMS];
	uint32_t dig[ITEMS];
	uint32_t rank[ITEMS];
	if (STABLE) {
#pragma unroll
		for (int r = 0; r < ITEMS; r++) {
			const uint32_t i = wave * PART_WAVE_SPAN + (uint32_t)r * MDB_WAVE + lane;
			bool valid = i < td.len;
			uint64_t h = 0;
			rid[r] = 0;
			if (valid)
				valid = part_load<LEVEL0>(a, td, i, &h, &rid[r]);
			hv[r] = (W)h;
			dig[r] = valid ? part_digit(a, hv[r]) : PART_INVALID;
		}
	} else if (W32 && RAW && a.fold64) {
		/* group records on their way into the 4-byte ordering sort: 8-byte words in, folded */
		const bool full = td.len == TILE && !(td.start & 1u);	/* (uniform) */
		ulonglong2 pre[ITEMS / 2];
		if (full)
			part_preload2<false, ITEMS / 2>(a, td, pre);
#pragma unroll
		for (int r = 0; r < ITEMS / 2; r++) {
			bool valid[2];
			uint64_t h2[2];
			if (full)
				part_load2<false, false, true>(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, h2, &rid[2 * r], valid, nullptr, &pre[r]);
			else
				part_load2<false, false, true>(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, h2, &rid[2 * r], valid);
#pragma unroll
			for (int k = 0; k < 2; k++) {
				hv[2 * r + k] = (W)((uint32_t)(h2[k] >> 32) | (uint32_t)h2[k]);
				dig[2 * r + k] = valid[k] ? part_digit(a, hv[2 * r + k]) : PART_INVALID;
			}
		}
	} else if (W32 && !LEVEL0) {
		const bool full = td.len == TILE && !(td.start & 3u);	/* (uniform) */
		uint4 pre[ITEMS / 4];
		if (full) {
			const uint32_t *const src = reinterpret_cast<const uint32_t *>(a.hv_in) + td.start;
#pragma unroll
			for (int r = 0; r < ITEMS / 4; r++)
				pre[r] = *reinterpret_cast<const uint4 *>(src + 4u * ((uint32_t)r * PART_THREADS + threadIdx.x));
		}
#pragma unroll
		for (int r = 0; r < ITEMS / 4; r++) {
			uint32_t h4[4];
			bool valid[4];
			if (full)
				part_load4_w32(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, h4, valid, &pre[r]);
			else
				part_load4_w32(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, h4, valid);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				hv[4 * r + k] = (W)h4[k];
				rid[4 * r + k] = 0;
				dig[4 * r + k] = valid[k] ? part_digit(a, hv[4 * r + k]) : PART_INVALID;
			}
		}
	} else {
		constexpr bool PRE_OK = LEVEL0 || !HAS_RID;	/* (row-id arrays beyond the first level: the wide form keeps part_load2's own loads) */
		const bool full = PRE_OK && td.len == TILE && !(td.start & 1u) && !(LEVEL0 && !CF && a.keys32);	/* (uniform) */
		ulonglong2 pre[ITEMS / 2];
		if (full)
			part_preload2<LEVEL0, ITEMS / 2>(a, td, pre);
		/* (uniform) the pruned left table only: for the right table - with or without its key range recorded - the same
		 * straight-line code measured equal (0.247 ms) or slower (0.235 -> 0.258 ms) than the loop below */
		const bool straight = CF && !W32 && !PAY && full && !a.nullbits && a.range_in;
		if (straight)
			(void)part_cf_rows<W32, 1, W, ITEMS>(a, pre, td.start + 2u * threadIdx.x, hv, dig, range_lo, range_hi, seen_min, seen_max);
#pragma unroll
		for (int r = 0; r < ITEMS / 2 && !straight; r++) {
			bool valid[2];
			uint64_t h2[2];
			constexpr bool RANGE = LEVEL0 && FAST && !RAW && !INV && !HAS_RID;	/* (the narrow forms' first level) */
			uint64_t rel2[2] = { 0, 0 };
			/* (the key-range tests of the by-destination partition are compiled into its own instance only: as run-time
			 * branches they cost the join's first-level kernels 0.05 ms per 10^8 rows) */
			int64_t raw2[2] = { 0, 0 };
			if (full)
				part_load2<LEVEL0, HAS_RID, RAW, INV, INV, CF, !W32>(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, h2, &rid[2 * r], valid,
										     RANGE ? rel2 : nullptr, &pre[r], (R64 || MM64) ? raw2 : nullptr);
			else
				part_load2<LEVEL0, HAS_RID, RAW, INV, INV, CF, !W32>(a, td, (uint32_t)r * PART_THREADS + threadIdx.x, h2, &rid[2 * r], valid,
										     RANGE ? rel2 : nullptr, nullptr, (R64 || MM64) ? raw2 : nullptr);
			if (R64 || MM64) {	/* (order-preserving unsigned images) */
				rel2[0] = (uint64_t)raw2[0] ^ 0x8000000000000000ull;
				rel2[1] = (uint64_t)raw2[1] ^ 0x8000000000000000ull;
			}
			if (R64) {
				valid[0] = valid[0] && rel2[0] >= r64_lo && rel2[0] <= r64_hi;
				valid[1] = valid[1] && rel2[1] >= r64_lo && rel2[1] <= r64_hi;
			}
			if (MM64) {
#pragma unroll
				for (int k = 0; k < 2; k++)
					if (valid[k]) {
						seen_min64 = rel2[k] < seen_min64 ? rel2[k] : seen_min64;
						seen_max64 = rel2[k] > seen_max64 ? rel2[k] : seen_max64;
					}
			}
			if (RANGE && a.range_in) {		/* (uniform) */
				valid[0] = valid[0] && rel2[0] >= range_lo && rel2[0] <= range_hi;
				valid[1] = valid[1] && rel2[1] >= range_lo && rel2[1] <= range_hi;
			}
			if (RANGE && a.minmax_out) {
#pragma unroll
				for (int k = 0; k < 2; k++)
					if (valid[k]) {		/* (a valid row of a narrow form: rel < 2^32) */
						seen_min = rel2[k] < seen_min ? (uint32_t)rel2[k] : seen_min;
						seen_max = rel2[k] > seen_max ? (uint32_t)rel2[k] : seen_max;
					}
			}
			if (FILT) {		/* (narrow words: hash32 in the upper half) */
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const uint32_t bit = ((uint32_t)(h2[k] >> 32) >> a.filter_shift) & (a.filter_words * 32u - 1u);
					valid[k] = valid[k] && ((s_flt[bit >> 5] >> (bit & 31u)) & 1u);
				}
			}
			hv[2 * r] = (W)h2[0];		/* W32 at level 0: the narrow word holds the hash in both halves */
			hv[2 * r + 1] = (W)h2[1];
			dig[2 * r] = valid[0] ? part_digit<CF>(a, INV ? (W)mdb_fmix64(h2[0]) : hv[2 * r]) : PART_INVALID;
			dig[2 * r + 1] = valid[1] ? part_digit<CF>(a, INV ? (W)mdb_fmix64(h2[1]) : hv[2 * r + 1]) : PART_INVALID;
		}
	}
	if (LEVEL0 && FAST && !RAW && !INV && !HAS_RID && a.minmax_out) {
#pragma unroll
		for (int o = MDB_WAVE / 2; o > 0; o >>= 1) {
			const uint32_t omin = (uint32_t)__shfl_xor((int)seen_min, o, MDB_WAVE), omax = (uint32_t)__shfl_xor((int)seen_max, o, MDB_WAVE);
			seen_min = omin < seen_min ? omin : seen_min;
			seen_max = omax > seen_max ? omax : seen_max;
		}
		if (lane == 0) {
			s_mm[2 * wave] = seen_min;
			s_mm[2 * wave + 1] = seen_max;
		}
	}
	if (MM64) {
#pragma unroll
		for (int o = MDB_WAVE / 2; o > 0; o >>= 1) {
			const uint64_t omin = ((uint64_t)(uint32_t)__shfl_xor((int)(seen_min64 >> 32), o, MDB_WAVE) << 32) | (uint32_t)__shfl_xor((int)seen_min64, o, MDB_WAVE);
			const uint64_t omax = ((uint64_t)(uint32_t)__shfl_xor((int)(seen_max64 >> 32), o, MDB_WAVE) << 32) | (uint32_t)__shfl_xor((int)seen_max64, o, MDB_WAVE);
			seen_min64 = omin < seen_min64 ? omin : seen_min64;
			seen_max64 = omax > seen_max64 ? omax : seen_max64;
		}
		if (lane == 0) {
			s_mm64[2 * wave] = seen_min64;
			s_mm64[2 * wave + 1] = seen_max64;
		}
	}
	__syncthreads();
	if (MM64 && threadIdx.x == 0) {
		uint64_t mn = ~0ull, mx = 0ull;
#pragma unroll
		for (int w = 0; w < PART_WAVES; w++) {
			mn = s_mm64[2 * w] < mn ? s_mm64[2 * w] : mn;
			mx = s_mm64[2 * w + 1] > mx ? s_mm64[2 * w + 1] : mx;
		}
		const uint32_t t = part_tile_of_block();
		a.minmax64_out[2 * t] = mn;
		a.minmax64_out[2 * t + 1] = ~mx;	/* (stored inverted) */
	}
	if (LEVEL0 && FAST && !RAW && !INV && !HAS_RID && a.minmax_out && threadIdx.x == 0) {
		/* one pair of plain stores per tile, reduced by k_part_minmax_reduce: atomics on two words from every wave of every
		 * tile (2 x 10^5 of them on the same address) cost 4 ms, coherent loads of "the best so far" 0.2 ms */
		uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
		for (int w = 0; w < PART_WAVES; w++) {
			mn = s_mm[2 * w] < mn ? s_mm[2 * w] : mn;
			mx = s_mm[2 * w + 1] > mx ? s_mm[2 * w + 1] : mx;
		}
		const uint32_t t = part_tile_of_block();
		a.minmax_out[2 * t] = mn;
		a.minmax_out[2 * t + 1] = ~mx;		/* (stored inverted) */
	}

	/* 2. rank inside the digit */
	if (STABLE) {
		/* lanes with the same digit find each other with `mbits` ballots; the lowest such lane bumps
		 * the wave-private LDS counter: rank = keys of the digit earlier in this wave's span */
		volatile uint32_t *wc = &s_wcnt[wave * PART_MAX_R];
		const uint64_t lt = mdb_lanemask_lt();
#pragma unroll
		for (int r = 0; r < ITEMS; r++) {
			const bool valid = dig[r] != PART_INVALID;
			const uint32_t d = dig[r];
			uint64_t peers = __ballot(valid);
			for (uint32_t b = 0; b < a.mbits; b++) {
				const bool bit = (d >> b) & 1u;
				const uint64_t m = __ballot(valid && bit);
				peers &= bit ? m : ~m;
			}
			const uint32_t before = (uint32_t)__popcll(peers & lt);
			const uint32_t cnt = (uint32_t)__popcll(peers);
			uint32_t prev = 0;
			if (valid)
				prev = wc[d];
			if (valid && before == 0)
				wc[d] = prev + cnt;
			rank[r] = prev + before;
			__builtin_amdgcn_wave_barrier();
		}
		__syncthreads();
		/* per digit: per-wave counts -> per-wave bases, digit total */
		if (threadIdx.x < R) {
			uint32_t run = 0;
#pragma unroll
			for (int w = 0; w < PART_WAVES; w++) {
				const uint32_t t = s_wcnt[w * PART_MAX_R + threadIdx.x];
				s_wcnt[w * PART_MAX_R + threadIdx.x] = run;
				run += t;
			}
			s_cnt[threadIdx.x] = run;
		}
	} else if (INV && R <= PART_FEW_DIGITS) {
		/* 1, 2, 4 destination GPUs: a wave's 64 keys would meet on one or two LDS words (32- to 64-way serialisation of the
		 * returning atomics: the two by-destination kernels took 1.7 ms per 2 x 10^8 keys at one destination).  The lanes of
		 * a digit find each other with one ballot per digit; the lowest of them takes the wave's share of the counter */
		const uint64_t lt = mdb_lanemask_lt();
#pragma unroll
		for (int r = 0; r < ITEMS; r++) {
			rank[r] = 0u;
			for (uint32_t d = 0; d < R; d++) {
				const uint64_t m = __ballot(dig[r] == d);
				if (!m)
					continue;
				const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
				uint32_t base = 0;
				if (lane == leader)
					base = atomicAdd(&s_cnt[d], (uint32_t)__popcll(m));
				base = __shfl(base, (int)leader, MDB_WAVE);
				if (dig[r] == d)
					rank[r] = base + (uint32_t)__popcll(m & lt);
			}
		}
	} else {
		/* one returning LDS atomic per key; order inside a digit = arrival order (unspecified) */
#pragma unroll
		for (int r = 0; r < ITEMS; r++)
			rank[r] = dig[r] != PART_INVALID ? atomicAdd(&s_cnt[dig[r]], 1u) : 0u;
	}
	__syncthreads();

	/* 3. digit totals -> tile-local starts; remember where each digit's run begins globally */
	const uint32_t total_d = threadIdx.x < R ? s_cnt[threadIdx.x] : 0u;
	uint32_t tile_total;
	const uint32_t off_d = mdb_block_excl_scan(total_d, s_tmp, &tile_total);
	uint32_t fast_base = 0, fast_child = 0;
	if (threadIdx.x < R) {
		s_cnt[threadIdx.x] = off_d;
		if (FAST) {
			/* no histogram pass: reserve the run's place in the child's fixed-capacity region with one
			 * global atomic per (tile, digit).  The returned base is only needed for the write-out, so
			 * the atomic's round trip overlaps the LDS staging below. */
			/* first level: 8 sub-regions per digit, picked by blockIdx % 8 (= the XCD under round-robin
			 * dispatch, so an XCD's partial lines meet in its own L2) keep the contention per cursor low */
			fast_child = a.nsub ? threadIdx.x * a.nsub + (blockIdx.x % a.nsub) : td.seg * R + threadIdx.x;
			/* first-level cursors are laid out sub-major (cursor[sub * R + digit]): the 256 atomics of one tile
			 * fall into 8 consecutive 128-byte lines that only this XCD's tiles touch, instead of 64 lines
			 * shared by every XCD (atomics on one line serialise) */
			const uint32_t cidx = a.nsub ? (blockIdx.x % a.nsub) * R + threadIdx.x : fast_child;
			fast_base = total_d ? atomicAdd(&a.cursor[cidx], total_d) : 0u;
		} else {
			s_delta[threadIdx.x] = (int32_t)(a.hist[(uint64_t)td.hbase + (uint64_t)threadIdx.x * td.nt] - off_d);
		}
	}
	__syncthreads();

	/* 4. stage sorted by digit */
#pragma unroll
	for (int r = 0; r < ITEMS; r++) {
		if (dig[r] != PART_INVALID) {
			uint32_t pos = s_cnt[dig[r]] + rank[r];
			if (STABLE)
				pos += s_wcnt[wave * PART_MAX_R + dig[r]];
			rank[r] = pos;		/* staged position (reused for the row ids below) */
			s_hv[pos] = hv[r];
			if (HAS_RID && !RID_SHARES_LDS)
				s_rid[pos] = rid[r];
		}
	}
	if (FAST && threadIdx.x < R) {
		const bool ok = fast_base + total_d <= a.cap;
		if (!ok)
			mdb_raise(a.status, 2u);
		s_ok[threadIdx.x] = ok;
		s_delta[threadIdx.x] = (int32_t)(fast_child * a.cap + fast_base - off_d);
	}
	__syncthreads();

	/* 5. write out: consecutive threads write consecutive addresses inside each digit's run */
	uint32_t gpos[ITEMS];
#pragma unroll
	for (int k = 0; k < ITEMS; k++) {
		const uint32_t i = threadIdx.x + (uint32_t)k * PART_THREADS;
		gpos[k] = PART_INVALID;
		if (i >= tile_total)
			continue;
		const W h = s_hv[i];
		uint32_t d;
		if (INV) {	/* the last digit whose tile-local start is <= i (s_cnt[0] = 0) */
			uint32_t lo = 0, hi = R;
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_cnt[mid] <= i)
					lo = mid;
				else
					hi = mid;
			}
			d = lo;
		} else {
			d = part_digit<CF>(a, h);
		}
		if (FAST && !s_ok[d])
			continue;	/* overflowed child: the whole operator is re-run on the exact path */
		const uint32_t g = (uint32_t)((int32_t)i + s_delta[d]);
		gpos[k] = g;
		if (W32 && OUT16)
			reinterpret_cast<uint16_t *>(a.hv_out)[g] = (uint16_t)((uint32_t)h >> a.out16_shift);
		else if (W32)
			reinterpret_cast<uint32_t *>(a.hv_out)[g] = (uint32_t)h;
		else if (INV && a.inverse_out == 2) {
			if (((uint64_t)h + 0x80000000ull) >> 32)
				mdb_raise(a.status, 128u);	/* the caller's promise (column statistics) does not hold: reported, not truncated */
			reinterpret_cast<int32_t *>(a.hv_out)[g] = (int32_t)(int64_t)h;	/* 4-byte wire format */
		}
		else if (INV)
			a.hv_out[g] = h;
		else if (a.inverse_out == 2)
			reinterpret_cast<int32_t *>(a.hv_out)[g] = (int32_t)(int64_t)mdb_fmix64_inv(h);
		else
			a.hv_out[g] = a.inverse_out ? mdb_fmix64_inv(h) : h;
		if (HAS_RID && !RID_SHARES_LDS)
			a.rid_out[g] = s_rid[i];
	}
	if (PAY) {
		/* the payload cells take the hashes' way through the same LDS buffer, one column after the other: read at the row's own
		 * index (consecutive threads, consecutive pairs of rows), staged at the row's rank, written beside its word */
		uint64_t *const s_pay = reinterpret_cast<uint64_t *>(s_hv);
		for (uint32_t c = 0; c < a.npay; c++) {	/* (uniform) */
			/* (element r of this thread = pair r / 2 of the tile, relative to its even-aligned base - part_load2's addressing) */
			const uint64_t base2 = (uint64_t)td.start - (td.start & 1u);
			uint64_t pv[ITEMS];
#pragma unroll
			for (int r = 0; r < ITEMS; r++)
				pv[r] = dig[r] != PART_INVALID ? a.pay_in[c][base2 + 2u * ((uint32_t)(r >> 1) * PART_THREADS + threadIdx.x) + (uint32_t)(r & 1)] : 0ull;
			__syncthreads();	/* every word of the previous round has been read */
#pragma unroll
			for (int r = 0; r < ITEMS; r++)
				if (dig[r] != PART_INVALID)
					s_pay[rank[r]] = pv[r];
			__syncthreads();
#pragma unroll
			for (int k = 0; k < ITEMS; k++)
				if (gpos[k] != PART_INVALID)
					a.pay_out[c][gpos[k]] = s_pay[threadIdx.x + (uint32_t)k * PART_THREADS];
		}
	}
	if (RID_SHARES_LDS) {
		__syncthreads();	/* every hash has been read: the buffer now takes the row ids */
#pragma unroll
		for (int r = 0; r < ITEMS; r++)
			if (dig[r] != PART_INVALID)
				s_rid[rank[r]] = rid[r];
		__syncthreads();
#pragma unroll
		for (int k = 0; k < ITEMS; k++)
			if (gpos[k] != PART_INVALID)
				a.rid_out[gpos[k]] = s_rid[threadIdx.x + (uint32_t)k * PART_THREADS];
	}
}

/* ---- segment bookkeeping between levels -------------------------------------------------------
 *
 * After a level's scan, child segment q = (parent p, digit d) starts at scanned[tb[p]*R + d*nt_p].
 * k_part_children writes child starts (S*R + 1 entries, last = total) and each child's tile count
 * (S*R + 1 entries, last = 0; exclusive-scanned afterwards into the child tile bases).
 */
__global__ void k_part_children(const uint32_t *__restrict__ scanned, const uint32_t *__restrict__ tb, uint32_t S, uint32_t R,
				uint32_t *__restrict__ child_start, uint32_t *__restrict__ child_ntiles)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t nq = S * R;
	if (q > nq)
		return;
	auto start_of = [&](uint32_t qq) -> uint32_t {
		if (qq >= nq)
			return scanned[(uint64_t)tb[S] * R];	/* == grand total */
		const uint32_t p = qq / R, d = qq % R;
		const uint32_t ntp = tb[p + 1] - tb[p];
		return scanned[(uint64_t)tb[p] * R + (uint64_t)d * ntp];
	};
	const uint32_t s0 = start_of(q);
	child_start[q] = s0;
	if (q == nq) {
		child_ntiles[q] = 0;
	} else {
		const uint32_t s1 = start_of(q + 1);
		/* tiles are cut at multiples of MDB_TILE from the segment's even-aligned base, so every tile but
		 * the first starts on an even element (16-byte loads) */
		child_ntiles[q] = s1 > s0 ? (s1 - (s0 & ~1u) + MDB_TILE - 1) / MDB_TILE : 0;
	}
}

/* tile descriptors of a level >= 1 from its segments (seg_start, tile bases tb), one thread per tile */
__global__ void k_part_build_tiles(const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ tb, uint32_t S, uint32_t R,
				   mdb_tile_desc *__restrict__ tiles, uint32_t max_tiles)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= max_tiles)
		return;
	mdb_tile_desc d;
	d.start = d.len = d.hbase = d.nt = d.seg = 0;
	if (t < tb[S]) {
		/* largest p with tb[p] <= t  (tb is non-decreasing, tb[S] > t) */
		uint32_t lo = 0, hi = S;
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (tb[mid] <= t)
				lo = mid;
			else
				hi = mid;
		}
		const uint32_t p = lo, tl = t - tb[p];
		const uint32_t seg0 = seg_start[p], s1 = seg_start[p + 1];
		const uint32_t lo_abs = (seg0 & ~1u) + tl * MDB_TILE;
		const uint32_t s0 = lo_abs > seg0 ? lo_abs : seg0;
		const uint32_t e0 = (lo_abs + MDB_TILE) < s1 ? (lo_abs + MDB_TILE) : s1;
		d.start = s0;
		d.len = e0 - s0;
		d.nt = tb[p + 1] - tb[p];
		d.hbase = tb[p] * R + tl;
		d.seg = p;
	}
	tiles[t] = d;
}

/* ---- histogram-free FIRST level: regions (digit, sub) of fixed capacity with atomic cursors -------------
 * region r = digit * nsub + sub lives at [r * cap, r * cap + min(cursor[r], cap)) and belongs to parent
 * partition r / nsub.  The next level's tiles are cut from the regions. */
/* FAST first level: tiles per region (a region = the filled part of a fixed-capacity sub-region) and their
 * exclusive scan, in one single-workgroup launch (there are only R * PART_NSUB + 1 entries). */
/* cursor of region r = digit * nsub + sub (sub-major cursor layout, see k_part_scatter) */
__device__ static inline uint32_t part_region_cursor(const uint32_t *cursor, uint32_t r, uint32_t nreg, uint32_t nsub)
{
	return cursor[(r % nsub) * (nreg / nsub) + r / nsub];
}

__global__ __launch_bounds__(1024) void k_part_region_tiles_scan(const uint32_t *__restrict__ cursor, uint32_t nreg, uint32_t cap,
								 uint32_t nsub, uint32_t *__restrict__ tb)
{
	__shared__ uint32_t s_tmp[32];
	uint32_t carry = 0;
	for (uint32_t base = 0; base <= nreg; base += 1024) {
		const uint32_t r = base + threadIdx.x;
		uint32_t c = 0;
		if (r < nreg) {
			c = part_region_cursor(cursor, r, nreg, nsub);
			c = c < cap ? c : cap;
		}
		const uint32_t nt = (c + MDB_TILE - 1) / MDB_TILE;
		uint32_t total;
		const uint32_t ex = mdb_block_excl_scan(nt, s_tmp, &total);
		if (r <= nreg)
			tb[r] = carry + ex;
		carry += total;
	}
}

__global__ void k_part_build_tiles_regions(const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ tb, uint32_t nreg,
					   uint32_t cap, uint32_t nsub, mdb_tile_desc *__restrict__ tiles, uint32_t max_tiles)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= max_tiles)
		return;
	mdb_tile_desc d;
	d.start = d.len = d.hbase = d.nt = d.seg = 0;
	if (t < tb[nreg]) {
		uint32_t lo = 0, hi = nreg;
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (tb[mid] <= t)
				lo = mid;
			else
				hi = mid;
		}
		const uint32_t r = lo, tl = t - tb[r];
		uint32_t c = part_region_cursor(cursor, r, nreg, nsub);
		c = c < cap ? c : cap;
		d.start = r * cap + tl * MDB_TILE;		/* cap is a multiple of 64: every tile starts 16-byte aligned */
		d.len = (c - tl * MDB_TILE) < MDB_TILE ? (c - tl * MDB_TILE) : MDB_TILE;
		d.seg = r / nsub;
	}
	tiles[t] = d;
}

__global__ void k_part_seg0(uint32_t *seg_start, uint32_t *tb, uint32_t n, uint32_t ntiles)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		seg_start[0] = 0;
		seg_start[1] = n;
		tb[0] = 0;
		tb[1] = ntiles;
	}
}

/* the same for the 64-bit form: per-tile pairs of order-preserving images -> [lo, hi] as signed keys (an empty table: lo > hi) */
__global__ __launch_bounds__(1024) void k_part_minmax64_reduce(const unsigned long long *__restrict__ tile_mm, uint32_t ntiles, long long *__restrict__ out)
{
	__shared__ unsigned long long s_mn[16], s_mx[16];
	unsigned long long mn = ~0ull, mx = 0ull;
	for (uint32_t t = threadIdx.x; t < ntiles; t += 1024) {
		const unsigned long long a = tile_mm[2 * t], b = ~tile_mm[2 * t + 1];
		mn = a < mn ? a : mn;
		mx = b > mx ? b : mx;
	}
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned long long omn = ((unsigned long long)(uint32_t)__shfl_xor((int)(mn >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)mn, o, 64);
		const unsigned long long omx = ((unsigned long long)(uint32_t)__shfl_xor((int)(mx >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)mx, o, 64);
		mn = omn < mn ? omn : mn;
		mx = omx > mx ? omx : mx;
	}
	if ((threadIdx.x & 63u) == 0) {
		s_mn[threadIdx.x >> 6] = mn;
		s_mx[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 16; w++) {
			mn = s_mn[w] < mn ? s_mn[w] : mn;
			mx = s_mx[w] > mx ? s_mx[w] : mx;
		}
		out[0] = (long long)(mn ^ 0x8000000000000000ull);	/* (no key at all: lo = INT64_MAX > hi = INT64_MIN - every left row is dropped) */
		out[1] = (long long)(mx ^ 0x8000000000000000ull);
	}
}

/* per-tile (min, max) pairs of the right table's first level -> the two words the left table's first level reads.  One workgroup takes ~10 us
 * for the 195 KB of a 10^8-row table (what one CU can pull), so MM_GROUPS of them share the pairs and meet in the two words with one atomic pair per
 * wave; the words are preset by the scatter kernel's first workgroup (mdb_level_args.minmax_final) */
#define MM_GROUPS 64u
#define MM_THREADS 256u
__global__ __launch_bounds__(MM_THREADS) void k_part_minmax_reduce(const uint32_t *__restrict__ tile_mm, uint32_t ntiles, uint32_t *__restrict__ out)
{
	uint32_t mn = 0xFFFFFFFFu, mx = 0u;
	/* two pairs per 16-byte load, a workgroup's loads issued together where they are few */
	const uint4 *const mm2 = reinterpret_cast<const uint4 *>(tile_mm);
	const uint32_t n2 = ntiles / 2u;
	for (uint32_t t0 = blockIdx.x * MM_THREADS; t0 < n2; t0 += 4u * MM_GROUPS * MM_THREADS) {
		uint4 q[4];
#pragma unroll
		for (uint32_t u = 0; u < 4u; u++) {
			const uint32_t t = t0 + u * MM_GROUPS * MM_THREADS + threadIdx.x;
			q[u] = t < n2 ? mm2[t] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
		}
#pragma unroll
		for (uint32_t u = 0; u < 4u; u++) {
			mn = min(mn, min(q[u].x, q[u].z));
			mx = max(mx, max(~q[u].y, ~q[u].w));
		}
	}
	if ((ntiles & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
		mn = min(mn, tile_mm[2u * (ntiles - 1u)]);
		mx = max(mx, ~tile_mm[2u * (ntiles - 1u) + 1u]);
	}
#pragma unroll
	for (int o = MDB_WAVE / 2; o > 0; o >>= 1) {
		const uint32_t omin = (uint32_t)__shfl_xor((int)mn, o, MDB_WAVE), omax = (uint32_t)__shfl_xor((int)mx, o, MDB_WAVE);
		mn = omin < mn ? omin : mn;
		mx = omax > mx ? omax : mx;
	}
	if (mdb_lane() == 0 && mn <= mx) {	/* (a wave that saw no key has nothing to add) */
		atomicMin(&out[0], mn);
		atomicMax(&out[1], mx);
	}
}

/* ---- host orchestration ----------------------------------------------------------------------- */

struct part_carver {
	mdb_dev_ctx *ctx;
	bool dry;
	size_t bytes;
	bool failed;
	void *take(size_t b)
	{
		bytes += mdb_align_up(b ? b : 1);
		if (dry)
			return NULL;
		void *p = mdb_arena_take(ctx, b);
		if (!p)
			failed = true;
		return p;
	}
};

void mdb_choose_bits(uint64_t n, uint32_t target, int *bits1, int *bits2)
{
	uint64_t leaves = (n + target - 1) / target;
	int bits = 1;
	while (bits < 2 * MDB_MAX_RADIX_BITS && (1ull << bits) < leaves)
		bits++;
	if (bits <= 8) {
		*bits1 = bits;
		*bits2 = 0;
	} else if (bits <= 16) {
		*bits1 = 8;
		*bits2 = bits - 8;
	} else {
		*bits1 = 9;
		*bits2 = bits - 9;
	}
}

static inline uint32_t grid8(uint32_t tiles) { return ((tiles + 7u) / 8u) * 8u; }

#define PART_F_STABLE 1u	/* keep input order inside every leaf (ballot ranking) */
#define PART_F_FAST 2u		/* no histogram passes: fixed-capacity regions + atomic cursors (two-level partitions only) */
#define PART_F_NARROW 4u	/* narrow form: words = fmix32(key) in both halves (mdb_level_args.narrow = 2) */
#define PART_F_NARROW_RID 8u	/* narrow form with the row id in the low half of the word (narrow = 1; no row-id arrays) */
#define PART_F_KEYS32 16u	/* the key column is int32 */
#define PART_F_NO_GAPS 32u	/* raw words: a zero word is a word like any other, not a gap of the input list */
#define PART_F_IN32 128u	/* with PART_F_FOLD32: the raw list already holds 4-byte words (nothing to fold) */
#define PART_F_FOLD32 64u	/* raw 8-byte records are folded into 4-byte words by the first level (two-level fast layout only) */
#define PART_F_LOOSE 1024u	/* first-level regions of 1.5 x (instead of 1.25 x) the average: few values per region, many rows per value */
#define PART_F_OUT16 512u	/* with PART_F_STOP0 and 4-byte words: the first level writes 2-byte words (hash bits below the digit) */
#define PART_F_STOP0 256u	/* first level only (histogram-free layout), bits2 = 0: the consumer walks the digits' sub-regions */
#define PART_NSUB 8u		/* sub-regions per first-level digit in the FAST form */

/* capacity of one leaf region of the FAST form: 1.5 x the average leaf + 1024, rounded up to 64 */
static inline uint32_t part_fast_cap(uint64_t n, uint32_t nleaves)
{
	const uint64_t avg = (n + nleaves - 1) / nleaves;
	return (uint32_t)(((avg + avg / 2 + 1024) + 63) & ~63ull);
}

static int partition_impl(part_carver &cv, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int bits1, int bits2,
			  bool want_rid, uint32_t flags, uint32_t mode, uint32_t n_dest, bool inverse_out, uint64_t *final_hv_out,
			  const uint64_t *raw_hv, uint32_t cap_override, mdb_part_result *out, uint32_t *final_rid_out = NULL,
			  uint32_t digits0_used = 0, bool keys32_out = false, int64_t narrow_base = 0, uint32_t narrow_kbits = 0,
			  const mdb_part_filter *flt = NULL, uint64_t raw_digit_rows = 0)
{
	mdb_dev_ctx *ctx = cv.ctx;
	const bool dry = cv.dry;
	const bool stable = flags & PART_F_STABLE;
	const int nlevels = bits2 > 0 ? 2 : 1;
	const bool stop0 = (flags & PART_F_STOP0) && nlevels == 1;
	const bool out16 = stop0 && (flags & PART_F_OUT16) && (flags & PART_F_NARROW) && narrow_kbits && narrow_kbits - (uint32_t)bits1 <= 16u;
	const uint32_t Rl[2] = { mode == MDB_DIGIT_MOD ? n_dest : (1u << bits1), 1u << bits2 };
	const uint32_t nt0 = n ? (uint32_t)((n + MDB_TILE - 1) / MDB_TILE) : 1u;
	/* FAST applies to the second level only, and only while leaf * cap stays a 32-bit index */
	const uint32_t nleaves_total = nlevels == 2 ? Rl[0] * Rl[1] : Rl[0];
	const uint32_t fast_cap = cap_override ? cap_override : part_fast_cap(n, nleaves_total);
	const bool fast = (flags & PART_F_FAST) && !stable && (nlevels == 2 || stop0) && !final_hv_out &&
			  (uint64_t)nleaves_total * fast_cap < 0xFFFFFFFFull;
	/* ... and, with it, to the first level: PART_NSUB fixed-capacity sub-regions per digit */
	const uint32_t nreg0 = Rl[0] * PART_NSUB;
	/* raw sort keys need not cover the whole first digit range (row ids below n < 2^kbits): the regions are
	 * sized for the digits that can occur */
	const uint32_t nreg0_used = (digits0_used && digits0_used < Rl[0] ? digits0_used : Rl[0]) * PART_NSUB;
	uint64_t avg0 = (n + nreg0_used - 1) / nreg0_used;
	/* raw sort keys that are ROW IDS (the ordering of group records by first row): a first-level digit spans raw_digit_rows ids and every id
	 * occurs once at most, so min(n, raw_digit_rows) words is the most a digit can receive - and receives, where the first rows bunch
	 * (single-table GROUP BY over random duplicates: most groups' first rows lie in the first tenth of the table).  Regions sized for
	 * that never send the sort to its exact layout; the list's order has nothing to do with the row ids, so a digit's words spread
	 * evenly over its sub-regions */
	if (raw_digit_rows) {
		const uint64_t most = n < raw_digit_rows ? n : raw_digit_rows, per_sub = (most + PART_NSUB - 1) / PART_NSUB;
		if (per_sub > avg0)
			avg0 = per_sub;
	}
	/* (flt->region_cap: the caller fixes the capacity of a first-level region itself - the sharded operator, whose ranks must
	 * agree on it: the regions ARE the transfer blocks, mdb_dev_shard.hip) */
	const uint32_t cap0 = (flt && flt->region_cap) ? flt->region_cap
						       : (uint32_t)((((flags & PART_F_LOOSE) ? avg0 * 3 / 2 : avg0 * 5 / 4) + 1024 + 63) & ~63ull);
	const bool fast0 = fast && mode == MDB_DIGIT_RADIX && (uint64_t)nreg0_used * cap0 < 0xFFFFFFFFull;
	/* narrow form without row ids (right side of a join): 4-byte words from the first level's output on; only built for
	 * the histogram-free layout of both levels (callers ask mdb_partition_w32_applies() first) */
	const bool fold32 = (flags & PART_F_FOLD32) && raw_hv && fast0 && fast;
	const bool w32 = ((flags & PART_F_NARROW) && fast0 && fast) || fold32;
	if ((flags & (PART_F_NARROW | PART_F_FOLD32)) && !w32 && !dry)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "4-byte partition words need the two-level fast layout");
	if ((flags & PART_F_STOP0) && !(stop0 && fast0) && !dry)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "first-level-only partition: histogram-free layout, one level");

	uint64_t *hv_buf[2] = { NULL, NULL };
	uint32_t *rid_buf[2] = { NULL, NULL };
	for (int l = 0; l < nlevels; l++) {
		const uint64_t elems = (fast && l == 1) ? (uint64_t)nleaves_total * fast_cap
				       : ((fast0 && l == 0) ? (uint64_t)nreg0_used * cap0 : (n ? n : 1));	/* regions of digits that cannot occur are not allocated */
		if (l == nlevels - 1 && final_hv_out)
			hv_buf[l] = final_hv_out;	/* last level writes straight into the caller's buffer */
		else
			hv_buf[l] = (uint64_t *)cv.take(elems * (w32 ? 4 : 8));
		if (want_rid)
			rid_buf[l] = (l == nlevels - 1 && final_rid_out) ? final_rid_out : (uint32_t *)cv.take(elems * 4);
	}
	/* payload cells beside the words (mdb_part_filter.npay): one buffer per level, laid out like that level's words */
	uint64_t *pay_buf[2] = { NULL, NULL }, *pay_buf1[2] = { NULL, NULL };
	const int npay = (flt && fast0 && !w32 && (stop0 || (fast && nlevels == 2))) ? flt->npay : 0;
	for (int c = 0; c < npay && c < 2; c++) {
		pay_buf[c] = (uint64_t *)cv.take((uint64_t)nreg0_used * cap0 * 8);
		if (!stop0)
			pay_buf1[c] = (uint64_t *)cv.take((uint64_t)nleaves_total * fast_cap * 8);
	}

	/* segments of the current level */
	uint32_t S = 1;
	uint32_t *seg_start = (uint32_t *)cv.take(2 * 4);
	uint32_t *tb = (uint32_t *)cv.take(2 * 4);
	if (cv.failed)
		return -MIDORIDB_INTERNAL;
	if (!dry && !fast0)	/* the FAST first level has no parent segments to describe */
		MDB_LAUNCH(ctx, "part_seg0", k_part_seg0, 1, 64, seg_start, tb, (uint32_t)n, nt0);

	uint32_t ntiles = nt0;
	const mdb_tile_desc *tiles = NULL;
	uint32_t *leaf_cnt = NULL;
	int used_bits = 0;
	for (int l = 0; l < nlevels; l++) {
		const uint32_t R = Rl[l];
		const uint32_t nchild = S * R;
		const bool fast_level = fast && l == 1;
		mdb_level_args a;
		memset(&a, 0, sizeof(a));
		a.keys = keys;
		a.nullbits = nullbits;
		a.n = n;
		a.hv_in = l ? hv_buf[l - 1] : raw_hv;	/* raw_hv: level 0 reads ready-made 64-bit sort keys */
		a.skip_zero = (l == 0 && raw_hv && !(flags & PART_F_NO_GAPS)) ? 1u : 0u;
		a.fold64 = (fold32 && l == 0 && !(flags & PART_F_IN32)) ? 1u : 0u;
		a.rid_in = l ? rid_buf[l - 1] : NULL;
		a.tiles = tiles;
		a.hv_out = hv_buf[l];
		a.rid_out = rid_buf[l];
		a.ntiles = ntiles;
		a.R = R;
		a.mode = l == 0 ? mode : MDB_DIGIT_RADIX;
		a.inverse_out = (inverse_out && l == nlevels - 1) ? (keys32_out ? 2u : 1u) : 0u;
		a.narrow = l == 0 ? ((flags & PART_F_NARROW_RID) ? 1u : ((flags & PART_F_NARROW) ? 2u : 0u)) : 0u;
		a.status = ctx ? ctx->d_status : NULL;
		a.keys32 = (flags & PART_F_KEYS32) ? 1u : 0u;
		a.narrow_base = narrow_base;
		a.narrow_kbits = a.narrow ? narrow_kbits : 0u;
		a.own_on = (flt && l == 0 && flt->own_on) ? 1u : 0u;
		a.own_lo = flt ? flt->own_lo : 0;
		a.own_hi = flt ? flt->own_hi : 0;
		a.keep_on = (flt && l == 0 && flt->keep_on) ? 1u : 0u;
		a.keep_lo = flt ? flt->keep_lo : 0;
		a.keep_hi = flt ? flt->keep_hi : 0;
		a.minmax_out = (flt && l == 0 && flt->minmax_out) ? flt->minmax_tiles : NULL;	/* per-tile pairs, reduced after the launch */
		a.minmax_final = a.minmax_out ? flt->minmax_out : NULL;
		a.range_in = (flt && l == 0) ? flt->range_in : NULL;
		a.minmax64_out = (flt && l == 0 && flt->minmax64_out) ? flt->minmax64_tiles : NULL;
		a.range64_in = (flt && l == 0) ? flt->range64_in : NULL;
		a.filter = (flt && l == 1) ? flt->bits : NULL;
		a.filter_words = flt ? flt->words : 0u;
		a.filter_shift = flt ? flt->shift : 0u;
		if (a.mode == MDB_DIGIT_RADIX) {
			const int b = l == 0 ? bits1 : bits2;
			a.shift = (uint32_t)((w32 ? 32 : 64) - used_bits - b);
			a.mbits = (uint32_t)b;
		} else {
			uint32_t mb = 1;
			while ((1u << mb) < R)
				mb++;
			a.shift = 0;
			a.mbits = mb;
		}

		if (fast0 && l == 0) {
			/* histogram-free first level */
			const bool cursor_ext = !dry && flt && flt->cursor0_ext && nreg0 <= flt->cursor0_ext_words;
			uint32_t *cursor0 = cursor_ext ? flt->cursor0_ext : (uint32_t *)cv.take((size_t)nreg0 * 4);
			uint32_t *reg_nt = (uint32_t *)cv.take(((size_t)nreg0 + 1) * 4);
			const uint32_t next_tiles = (uint32_t)(n / MDB_TILE) + nreg0 + 1;
			mdb_tile_desc *next_desc = stop0 ? NULL : (mdb_tile_desc *)cv.take((size_t)next_tiles * sizeof(mdb_tile_desc));
			if (cv.failed)
				return -MIDORIDB_INTERNAL;
			if (!dry) {
				/* compact narrow form over an int64 column: the instances that know so at compile time */
				const bool cf = a.narrow && narrow_kbits && !(flags & PART_F_KEYS32) && mode == MDB_DIGIT_RADIX && !raw_hv && !want_rid;
				a.cursor = cursor0;
				a.cap = cap0;
				a.nsub = PART_NSUB;
				a.status = ctx->d_status;
				if (!cursor_ext)
					MDB_HIP(ctx, hipMemsetAsync(cursor0, 0, (size_t)nreg0 * 4, ctx->stream));
				/* tables of 2^25 rows and more: tiles of 2 x MDB_TILE rows in the instances that have them (MDB_TILE2=0: never) */
				const char *t2min = mdb_knob("MDB_TILE2_MIN");	/* (tests: the form on small tables) */
				const bool t2 = n >= (t2min && atoll(t2min) > 0 ? (uint64_t)atoll(t2min) : (1ull << 25)) &&
						!(mdb_knob("MDB_TILE2") && mdb_knob("MDB_TILE2")[0] == '0');
				const uint32_t ntiles2 = (uint32_t)((n + (uint64_t)PART_TMUL * MDB_TILE - 1) / ((uint64_t)PART_TMUL * MDB_TILE));
				if (raw_hv && fold32 && t2) {
					a.ntiles = ntiles2;
					MDB_LAUNCH(ctx, "sort_scatter_l0_w32", (k_part_scatter<pf_word_raw_w32_t2>), grid8(ntiles2), PART_THREADS, a);
				} else if (raw_hv && fold32) {
					MDB_LAUNCH(ctx, "sort_scatter_l0_w32", (k_part_scatter<pf_word_raw_w32>), grid8(ntiles), PART_THREADS, a);
				} else if (raw_hv) {
					MDB_LAUNCH(ctx, "sort_scatter_l0", (k_part_scatter<pf_word_raw>), grid8(ntiles), PART_THREADS, a);
				} else if (w32) {
					if (out16) {
						a.out16_shift = 32u - narrow_kbits;
						if (cf && t2) {
							a.ntiles = ntiles2;
							MDB_LAUNCH(ctx, "part_scatter_l0_w32", (k_part_scatter<pf_key_w32_out16_cf_t2>),
								   grid8(ntiles2), PART_THREADS, a);
						} else if (cf) {
							MDB_LAUNCH(ctx, "part_scatter_l0_w32", (k_part_scatter<pf_key_w32_out16_cf>),
								   grid8(ntiles), PART_THREADS, a);
						} else {
							MDB_LAUNCH(ctx, "part_scatter_l0_w32", (k_part_scatter<pf_key_w32_out16>),
								   grid8(ntiles), PART_THREADS, a);
						}
					} else if (cf) {
						MDB_LAUNCH(ctx, "part_scatter_l0_w32", (k_part_scatter<pf_key_w32_cf>),
							   grid8(ntiles), PART_THREADS, a);
					} else {
						MDB_LAUNCH(ctx, "part_scatter_l0_w32", (k_part_scatter<pf_key_w32>), grid8(ntiles), PART_THREADS, a);
					}
					if (a.minmax_out)
						MDB_LAUNCH(ctx, "part_minmax", k_part_minmax_reduce, MM_GROUPS, MM_THREADS, (const uint32_t *)a.minmax_out, a.ntiles, flt->minmax_out);	/* (every tile of the launch has left its pair) */
				} else if (npay) {
					if (!cf || a.narrow != 1u)
						return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "payload cells travel with the compact narrow form's hash | row id words only");
					a.npay = (uint32_t)npay;
					for (int c = 0; c < npay; c++) {
						a.pay_in[c] = reinterpret_cast<const uint64_t *>(flt->pay_in[c]);
						a.pay_out[c] = pay_buf[c];
					}
					MDB_LAUNCH(ctx, "part_scatter_l0_pay", (k_part_scatter<pf_key_cf_pay>), grid8(ntiles), PART_THREADS, a);
				} else if (want_rid && a.range64_in) {	/* 64-bit form, left table, min-max pruning */
					MDB_LAUNCH(ctx, "part_scatter_l0_rid_pruned", (k_part_scatter<pf_key_rid_r64>), grid8(ntiles), PART_THREADS, a);
				} else if (want_rid) {
					MDB_LAUNCH(ctx, "part_scatter_l0_rid", (k_part_scatter<pf_key_rid>), grid8(ntiles), PART_THREADS, a);
				} else if (a.minmax64_out) {		/* 64-bit form, right table: its key range recorded */
					MDB_LAUNCH(ctx, "part_scatter_l0", (k_part_scatter<pf_key_mm64>), grid8(ntiles), PART_THREADS, a);
					MDB_LAUNCH(ctx, "part_minmax", k_part_minmax64_reduce, 1, 1024, (const unsigned long long *)a.minmax64_out, ntiles,
						   flt->minmax64_out);
				} else if (a.range_in && cf && t2) {
					a.ntiles = ntiles2;
					MDB_LAUNCH(ctx, "part_scatter_l0_pruned", (k_part_scatter<pf_key_cf_t2>), grid8(ntiles2), PART_THREADS, a);
				} else if (a.range_in && cf) {	/* (the same instance under another name: its bytes differ - most rows are read, not written) */
					MDB_LAUNCH(ctx, "part_scatter_l0_pruned", (k_part_scatter<pf_key_cf>),
						   grid8(ntiles), PART_THREADS, a);
				} else if (a.range_in) {
					MDB_LAUNCH(ctx, "part_scatter_l0_pruned", (k_part_scatter<pf_key>), grid8(ntiles), PART_THREADS, a);
				} else if (cf && t2) {
					a.ntiles = ntiles2;
					MDB_LAUNCH(ctx, "part_scatter_l0", (k_part_scatter<pf_key_cf_t2>), grid8(ntiles2), PART_THREADS, a);
				} else if (cf) {
					MDB_LAUNCH(ctx, "part_scatter_l0", (k_part_scatter<pf_key_cf>), grid8(ntiles),
						   PART_THREADS, a);
				} else {
					MDB_LAUNCH(ctx, "part_scatter_l0", (k_part_scatter<pf_key>), grid8(ntiles), PART_THREADS, a);
				}
				if (!stop0) {
					MDB_LAUNCH(ctx, "part_region_tiles", k_part_region_tiles_scan, 1, 1024, cursor0, nreg0, cap0, PART_NSUB, reg_nt);
					MDB_LAUNCH(ctx, "part_build_tiles", k_part_build_tiles_regions, (next_tiles + 255) / 256, 256, cursor0, reg_nt, nreg0,
						   cap0, PART_NSUB, next_desc, next_tiles);
				}
			}
			if (stop0) {
				if (out) {
					out->hv = hv_buf[0];
					out->rid = NULL;
					out->leaf_off = NULL;
					out->leaf_cnt = cursor0;
					out->leaf_cap = cap0;
					out->nleaves = R;
					out->bits_total = (uint32_t)bits1;
					out->w32 = w32;
					out->w16 = out16 && w32;
					out->nsub = PART_NSUB;
					out->pay[0] = pay_buf[0];
					out->pay[1] = pay_buf[1];
				}
				return MIDORIDB_OK;
			}
			uint32_t real_tiles = next_tiles;
			if (!dry && flt && (flt->range_in || flt->range64_in) && flt->expect_pruned) {
				/* min-max pruning may have dropped most rows: the second level is launched for the tiles that exist (one
				 * 4-byte read-back and a synchronisation, ~15 us) - 26 000 workgroups that find an empty descriptor and leave
				 * cost 0.09 ms at 10^8 rows */
				uint64_t *h = ctx->h_pinned;
				MDB_HIP(ctx, hipMemcpyAsync(&h[15], reg_nt + nreg0, 4, hipMemcpyDeviceToHost, ctx->stream));
				MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
				const uint32_t got = (uint32_t)h[15];
				real_tiles = got < next_tiles ? (got ? got : 1u) : next_tiles;
			}
			used_bits += bits1;
			seg_start = NULL;
			tb = NULL;
			S = R;
			tiles = next_desc;
			ntiles = real_tiles;
			continue;
		}
		if (fast_level) {
			/* histogram-free last level: one cursor per leaf, runs placed with global atomics */
			leaf_cnt = (uint32_t *)cv.take((size_t)nchild * 4);
			if (cv.failed)
				return -MIDORIDB_INTERNAL;
			if (!dry) {
				a.cursor = leaf_cnt;
				a.cap = fast_cap;
				a.status = ctx->d_status;
				MDB_HIP(ctx, hipMemsetAsync(leaf_cnt, 0, (size_t)nchild * 4, ctx->stream));
				if (npay) {
					a.npay = (uint32_t)npay;
					for (int c = 0; c < npay; c++) {
						a.pay_in[c] = pay_buf[c];
						a.pay_out[c] = pay_buf1[c];
					}
					MDB_LAUNCH(ctx, "part_scatter_l1_pay", (k_part_scatter<pf_word_pay>), grid8(ntiles), PART_THREADS, a);
				} else if (want_rid) {
					MDB_LAUNCH(ctx, "part_scatter_l1_rid", (k_part_scatter<pf_word_rid>), grid8(ntiles),
						   PART_THREADS, a);
				} else if (raw_hv && fold32) {
					MDB_LAUNCH(ctx, "sort_scatter_l1_w32", (k_part_scatter<pf_word_raw_w32>), grid8(ntiles),
						   PART_THREADS, a);
				} else if (raw_hv) {
					MDB_LAUNCH(ctx, "sort_scatter_l1", (k_part_scatter<pf_word_raw>), grid8(ntiles),
						   PART_THREADS, a);
				} else if (w32) {
					MDB_LAUNCH(ctx, "part_scatter_l1_w32", (k_part_scatter<pf_word_w32>), grid8(ntiles),
						   PART_THREADS, a);
				} else if (a.filter) {
					MDB_LAUNCH(ctx, "part_scatter_l1_semi", (k_part_scatter<pf_word_semi>),
						   grid8(ntiles), PART_THREADS, a);
				} else {
					MDB_LAUNCH(ctx, "part_scatter_l1", (k_part_scatter<pf_word>), grid8(ntiles),
						   PART_THREADS, a);
				}
			}
			used_bits += bits2;
			S = nchild;
			seg_start = NULL;
			break;
		}

		const uint64_t hlen = (uint64_t)ntiles * R + 1;
		uint32_t *hist = (uint32_t *)cv.take(hlen * 4);
		uint32_t *scan_tmp = (uint32_t *)cv.take(mdb_scan_scratch_words(hlen) * 4);
		uint32_t *child_start = (uint32_t *)cv.take(((size_t)nchild + 1) * 4);
		uint32_t *child_nt = (uint32_t *)cv.take(((size_t)nchild + 1) * 4);
		uint32_t *child_scan_tmp = (uint32_t *)cv.take(mdb_scan_scratch_words((uint64_t)nchild + 1) * 4);
		/* tile descriptors of the NEXT level (upper bound: every child adds at most two partial tiles) */
		const uint32_t next_tiles = (uint32_t)(n / MDB_TILE) + 2 * nchild + 1;
		mdb_tile_desc *next_desc = NULL;
		if (l + 1 < nlevels)
			next_desc = (mdb_tile_desc *)cv.take((size_t)next_tiles * sizeof(mdb_tile_desc));
		if (cv.failed)
			return -MIDORIDB_INTERNAL;

		if (!dry) {
			a.hist = hist;
			MDB_HIP(ctx, hipMemsetAsync(hist, 0, hlen * 4, ctx->stream));
			if (l == 0 && raw_hv) {
				MDB_LAUNCH(ctx, "sort_hist_l0", (k_part_hist<false, true>), grid8(ntiles), PART_THREADS, a);
			} else if (l == 0) {
				MDB_LAUNCH(ctx, mode == MDB_DIGIT_MOD ? "dest_hist" : "part_hist_l0", k_part_hist<true>, grid8(ntiles), PART_THREADS, a);
			} else {
				MDB_LAUNCH(ctx, "part_hist_l1", k_part_hist<false>, grid8(ntiles), PART_THREADS, a);
			}
			int rc = mdb_scan_u32_inplace(ctx, hist, hlen, scan_tmp);
			if (rc)
				return rc;
			if (l == 0 && raw_hv) {
				MDB_LAUNCH(ctx, "sort_scatter_l0", (k_part_scatter<pf_word_hist_raw>), grid8(ntiles), PART_THREADS, a);
			} else if (stable && l == 0) {
				MDB_LAUNCH(ctx, "part_scatter_l0_stable", (k_part_scatter<pf_key_rid_stable_hist>), grid8(ntiles), PART_THREADS, a);
			} else if (stable) {
				MDB_LAUNCH(ctx, "part_scatter_l1_stable", (k_part_scatter<pf_word_rid_stable_hist>), grid8(ntiles), PART_THREADS, a);
			} else if (l == 0 && a.inverse_out && want_rid) {
				MDB_LAUNCH(ctx, "dest_scatter", (k_part_scatter<pf_key_rid_hist_dest>), grid8(ntiles), PART_THREADS, a);
			} else if (l == 0 && a.inverse_out) {
				MDB_LAUNCH(ctx, "dest_scatter", (k_part_scatter<pf_key_hist_dest>), grid8(ntiles), PART_THREADS, a);
			} else if (l == 0 && want_rid) {
				MDB_LAUNCH(ctx, "part_scatter_l0", (k_part_scatter<pf_key_rid_hist>), grid8(ntiles), PART_THREADS, a);
			} else if (l == 0) {
				MDB_LAUNCH(ctx, "part_scatter_l0", (k_part_scatter<pf_key_hist>), grid8(ntiles), PART_THREADS, a);
			} else if (want_rid) {
				MDB_LAUNCH(ctx, "part_scatter_l1", (k_part_scatter<pf_word_rid_hist>), grid8(ntiles), PART_THREADS, a);
			} else {
				MDB_LAUNCH(ctx, "part_scatter_l1", (k_part_scatter<pf_word_hist>), grid8(ntiles), PART_THREADS, a);
			}
			MDB_LAUNCH(ctx, "part_children", k_part_children, (nchild + 1 + 255) / 256, 256, hist, tb, S, R, child_start,
				   child_nt);
			if (l + 1 < nlevels) {
				rc = mdb_scan_u32_inplace(ctx, child_nt, (uint64_t)nchild + 1, child_scan_tmp);
				if (rc)
					return rc;
				MDB_LAUNCH(ctx, "part_build_tiles", k_part_build_tiles, (next_tiles + 255) / 256, 256, child_start,
					   child_nt, nchild, Rl[l + 1], next_desc, next_tiles);
			}
		}
		used_bits += (l == 0 ? bits1 : bits2);
		seg_start = child_start;
		tb = child_nt;
		S = nchild;
		tiles = next_desc;
		ntiles = next_tiles;
	}
	if (out) {
		out->hv = hv_buf[nlevels - 1];
		out->rid = rid_buf[nlevels - 1];
		out->leaf_off = seg_start;
		out->leaf_cnt = leaf_cnt;
		out->leaf_cap = fast ? fast_cap : 0;
		out->nleaves = S;
		out->bits_total = (uint32_t)used_bits;
		out->w32 = w32;
		out->w16 = false;
		out->nsub = 0;
		out->pay[0] = nlevels == 2 ? pay_buf1[0] : NULL;
		out->pay[1] = nlevels == 2 ? pay_buf1[1] : NULL;
	}
	return MIDORIDB_OK;
}

size_t mdb_partition_arena_bytes(uint64_t n, int bits1, int bits2, bool want_rid, bool fast, int npay)
{
	part_carver cv = { NULL, true, 0, false };
	mdb_part_filter flt;
	memset(&flt, 0, sizeof(flt));
	flt.npay = npay;
	(void)partition_impl(cv, NULL, NULL, n, bits1, bits2, want_rid, fast ? PART_F_FAST : 0u, MDB_DIGIT_RADIX, 0, false, NULL, NULL, 0, NULL, NULL, 0, false,
			     0, 0u, npay ? &flt : NULL);
	return cv.bytes + 4096 + (size_t)npay * 1024;
}

size_t mdb_partition_level0_arena_bytes(uint64_t n, int bits1, bool loose, int npay)
{
	part_carver cv = { NULL, true, 0, false };
	mdb_part_filter flt;
	memset(&flt, 0, sizeof(flt));
	flt.level0_only = true;
	flt.npay = npay;
	(void)partition_impl(cv, NULL, NULL, n, bits1, 0, false, PART_F_FAST | PART_F_STOP0 | (loose ? PART_F_LOOSE : 0u), MDB_DIGIT_RADIX, 0, false, NULL,
			     NULL, 0, NULL, NULL, 0, false, 0, 0u, npay ? &flt : NULL);
	return cv.bytes + 4096 + (size_t)npay * 512;
}

bool mdb_partition_w32_applies(uint64_t n, int bits1, int bits2, bool fast)
{
	part_carver cv = { NULL, true, 0, false };
	mdb_part_result res;
	memset(&res, 0, sizeof(res));
	(void)partition_impl(cv, NULL, NULL, n, bits1, bits2, false, (fast ? PART_F_FAST : 0u) | PART_F_NARROW, MDB_DIGIT_RADIX, 0, false, NULL,
			     NULL, 0, &res);
	return res.w32;
}

int mdb_partition_table(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n, int bits1, int bits2,
			bool want_rid, bool stable, bool fast, mdb_part_result *out, int narrow, bool keys32, int64_t narrow_base,
			uint32_t narrow_kbits, const mdb_part_filter *flt)
{
	const bool stop0 = flt && flt->level0_only;
	if (stop0 && (!narrow || !narrow_kbits || want_rid || stable || !fast || bits2 != 0 || flt->bits))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "first-level-only partition: compact narrow form, bits2 = 0, no bitmap");
	if (flt && flt->bits && (narrow != 1 || !narrow_kbits || want_rid || stable || !fast || bits2 <= 0 || flt->words < 4 || flt->words > MDB_TILE * 2 ||
				 (flt->words & (flt->words - 1))))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "semi-join filter: left side of the compact narrow form, two fast levels, slices of 4 ... 8192 words");
	if (flt && (flt->minmax_out || flt->range_in) && (!narrow || want_rid || stable || !fast || (bits2 <= 0 && !stop0) ||
							  (flt->minmax_out && (narrow != 2 || !flt->minmax_tiles))))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "min-max pruning: narrow forms, two fast levels");
	if (flt && (flt->minmax64_out || flt->range64_in) && (narrow || stable || !fast || bits2 <= 0 || (flt->minmax64_out && (want_rid || !flt->minmax64_tiles)) ||
							      (flt->range64_in && !want_rid)))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "min-max pruning, 64-bit form: two fast levels, the right table without and the left with row ids");
	if (flt && flt->npay && (narrow != 1 || !narrow_kbits || want_rid || stable || !fast || flt->npay < 0 || flt->npay > 2 || (bits2 <= 0 && !stop0)))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "payload cells: compact narrow form with hash | row id words, histogram-free layout, at most two columns");
	if (narrow_kbits && (!narrow || narrow_kbits < 8 || narrow_kbits > 32 || (uint32_t)(bits1 + bits2) > narrow_kbits))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "compact narrow form: bad window width");
	if (narrow && (stable || want_rid))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "narrow partitioning carries row ids inside the word");
	if (n >= 0xFFFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "table of %llu rows exceeds the 32-bit row-id limit of one GPU shard",
				   (unsigned long long)n);
	part_carver cv = { ctx, false, 0, false };
	if (stable && !want_rid)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "stable partitioning is only built with row ids");
	if ((uintptr_t)keys & (keys32 ? 7 : 15))
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "key columns must be 16-byte aligned on the device (8-byte for int32 keys)");
	return partition_impl(cv, keys, nullbits, n, bits1, bits2, want_rid,
			      (stable ? PART_F_STABLE : 0u) | (fast ? PART_F_FAST : 0u) | (narrow == 1 ? PART_F_NARROW_RID : 0u) |
				      (narrow == 2 ? PART_F_NARROW : 0u) | (keys32 ? PART_F_KEYS32 : 0u) | (stop0 ? PART_F_STOP0 : 0u) |
					      (stop0 && flt->out16 ? PART_F_OUT16 : 0u) | (stop0 && flt->loose ? PART_F_LOOSE : 0u),
			      MDB_DIGIT_RADIX, 0, false, NULL, NULL, 0, out, NULL, 0, false, narrow ? narrow_base : 0, narrow ? narrow_kbits : 0u, flt);
}

/* MSD radix partition of ready-made 64-bit sort keys (no hashing, no NULLs) by their top bits1 + bits2
 * bits; with two levels the last one uses the fixed-capacity layout with `leaf_cap` rows per leaf (the
 * caller guarantees no leaf can hold more).  Used to order GROUP BY results by first row id. */
size_t mdb_partition_raw_arena_bytes(uint64_t n, int bits1, int bits2, uint32_t leaf_cap, bool fast, uint32_t digits0_used, uint64_t digit0_rows)
{
	part_carver cv = { NULL, true, 0, false };
	(void)partition_impl(cv, NULL, NULL, n, bits1, bits2, false, fast ? PART_F_FAST : 0u, MDB_DIGIT_RADIX, 0, false, NULL,
			     (const uint64_t *)16, leaf_cap, NULL, NULL, digits0_used, false, 0, 0, NULL, fast ? digit0_rows : 0);
	return cv.bytes + 4096;
}

int mdb_partition_raw(mdb_dev_ctx *ctx, const uint64_t *hv, uint64_t n, int bits1, int bits2, uint32_t leaf_cap, bool fast,
		      uint32_t digits0_used, mdb_part_result *out, bool zero_is_gap, int fold32, uint64_t digit0_rows)
{
	part_carver cv = { ctx, false, 0, false };
	if (fold32 && (!fast || bits2 <= 0 || !zero_is_gap))
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "4-byte raw words need the two-level fast layout");
	return partition_impl(cv, NULL, NULL, n, bits1, bits2, false,
			      (fast ? PART_F_FAST : 0u) | (zero_is_gap ? 0u : PART_F_NO_GAPS) | (fold32 ? PART_F_FOLD32 : 0u) | (fold32 == 2 ? PART_F_IN32 : 0u),
			      MDB_DIGIT_RADIX, 0,
			      false, NULL, hv, leaf_cap, out, NULL, digits0_used, false, 0, 0, NULL, fast ? digit0_rows : 0);
}

/* ---- one histogram-free radix level over 4-byte words in caller-described tiles (the receiver's second level of the
 * sharded operator, mdb_dev_shard.hip: its input regions came from several ranks) ------------------------------------- */
int mdb_partition_words_level(mdb_dev_ctx *ctx, const uint32_t *words_in, const mdb_tile_desc *tiles, uint32_t ntiles, int bits, uint32_t shift,
			      uint32_t *words_out, uint32_t *cursor, uint32_t nchild, uint32_t cap, uint32_t out16_shift)
{
	if (bits < 1 || bits > MDB_MAX_RADIX_BITS || !ntiles)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "words level: bad digit width");
	mdb_level_args a;
	memset(&a, 0, sizeof(a));
	a.hv_in = reinterpret_cast<const uint64_t *>(words_in);
	a.tiles = tiles;
	a.hv_out = reinterpret_cast<uint64_t *>(words_out);
	a.ntiles = ntiles;
	a.R = 1u << bits;
	a.mode = MDB_DIGIT_RADIX;
	a.shift = shift;
	a.mbits = (uint32_t)bits;
	a.cursor = cursor;
	a.cap = cap;
	a.nsub = 0;
	a.status = ctx->d_status;
	MDB_HIP(ctx, hipMemsetAsync(cursor, 0, (size_t)nchild * 4, ctx->stream));
	if (out16_shift) {	/* the children are leaves whose key bits fit 16: 2-byte words out, (uint16_t)(word >> out16_shift) */
		a.out16_shift = out16_shift;
		MDB_LAUNCH(ctx, "part_scatter_l1_w32", (k_part_scatter<pf_word_w32_out16>), grid8(ntiles), PART_THREADS, a);
	} else {
		MDB_LAUNCH(ctx, "part_scatter_l1_w32", (k_part_scatter<pf_word_w32>), grid8(ntiles), PART_THREADS, a);
	}
	return MIDORIDB_OK;
}

/* ---- one stable LSD radix pass (ORDER BY) --------------------------------------------------------
 *
 * (key_in, rid_in) -> (key_out, rid_out), stably ordered by the `bits`-wide digit at `shift`: the flat-tiled
 * histogram / scan / ballot-ranked scatter of the exact partition path (STABLE keeps the input order inside
 * every digit, which is what makes least-significant-digit-first sorting and multi-column ORDER BY work). */
size_t mdb_sort_pass_hist_words(uint64_t n)
{
	const uint64_t nt = (n + MDB_TILE - 1) / MDB_TILE;
	return (size_t)(nt ? nt : 1) * 256u + 1u;
}

int mdb_sort_pass(mdb_dev_ctx *ctx, const uint64_t *key_in, const uint32_t *rid_in, uint64_t n, uint32_t shift, uint32_t bits,
		  uint64_t *key_out, uint32_t *rid_out, uint32_t *hist, uint32_t *scan_tmp)
{
	if (n == 0)
		return MIDORIDB_OK;
	if (bits < 1 || bits > 8 || n >= 0xFFFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "sort pass: bad digit width or too many rows");
	mdb_level_args a;
	memset(&a, 0, sizeof(a));
	a.n = n;
	a.hv_in = key_in;
	a.rid_in = rid_in;
	a.hv_out = key_out;
	a.rid_out = rid_out;
	a.ntiles = (uint32_t)((n + MDB_TILE - 1) / MDB_TILE);
	a.R = 1u << bits;
	a.mode = MDB_DIGIT_RADIX;
	a.shift = shift;
	a.mbits = bits;
	a.hist = hist;
	const uint64_t hlen = (uint64_t)a.ntiles * a.R;
	MDB_LAUNCH(ctx, "orderby_hist", k_part_hist<false>, grid8(a.ntiles), PART_THREADS, a);
	int rc = mdb_scan_u32_inplace(ctx, hist, hlen, scan_tmp);
	if (rc)
		return rc;
	MDB_LAUNCH(ctx, "orderby_scatter", (k_part_scatter<pf_word_rid_stable_hist>), grid8(a.ntiles), PART_THREADS, a);
	return MIDORIDB_OK;
}

/* ---- multi-GPU destination partition ------------------------------------------------------------ */

extern "C" int mdb_dev_partition_by_dest(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n,
					 uint32_t n_dest, int keys32, void *out_keys, uint32_t *out_rid, uint64_t *out_counts)
{
	return mdb_dev_partition_by_dest_pruned(ctx, keys, nullbits, n, n_dest, keys32, INT64_MIN, INT64_MAX, INT64_MIN, INT64_MAX, out_keys, out_rid,
						out_counts);
}

extern "C" int mdb_dev_partition_by_dest_pruned(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nullbits, uint64_t n,
						uint32_t n_dest, int keys32, int64_t keep_lo, int64_t keep_hi, int64_t own_lo, int64_t own_hi,
						void *out_keys, uint32_t *out_rid, uint64_t *out_counts)
{
	mdb_part_filter flt;
	memset(&flt, 0, sizeof(flt));
	flt.keep_on = keep_lo != INT64_MIN || keep_hi != INT64_MAX;
	flt.keep_lo = keep_lo;
	flt.keep_hi = keep_hi;
	flt.own_on = own_lo != INT64_MIN || own_hi != INT64_MAX;
	flt.own_lo = own_lo;
	flt.own_hi = own_hi;
	if (n_dest == 0 || n_dest > PART_MAX_R)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "partition_by_dest: n_dest must be in [1, %u]", PART_MAX_R);
	if (n >= 0xFFFFFFFFull)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "partition_by_dest: too many rows");
	if ((uintptr_t)keys & 15)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "key columns must be 16-byte aligned on the device");
	for (uint32_t d = 0; d < n_dest; d++)
		out_counts[d] = 0;
	if (n == 0)
		return MIDORIDB_OK;
	/* dry run for the arena size, then the real pass (one level, digit = low32(hash) mod n_dest,
	 * original keys written back through the inverse hash - as 4-byte integers when the caller knows from the
	 * column's statistics that every key fits - source row ids beside them when asked for) */
	const bool want_rid = out_rid != NULL;
	part_carver dry = { NULL, true, 0, false };
	(void)partition_impl(dry, NULL, NULL, n, 1, 0, want_rid, 0u, MDB_DIGIT_MOD, n_dest, true, (uint64_t *)out_keys, NULL, 0, NULL, out_rid, 0,
			     keys32 != 0, 0, 0, &flt);
	int rc = mdb_arena_begin(ctx, dry.bytes + 4096);
	if (rc)
		return rc;
	part_carver cv = { ctx, false, 0, false };
	mdb_part_result res;
	MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
	rc = partition_impl(cv, keys, nullbits, n, 1, 0, want_rid, 0u, MDB_DIGIT_MOD, n_dest, true, (uint64_t *)out_keys, NULL, 0, &res, out_rid, 0,
			    keys32 != 0, 0, 0, &flt);
	if (rc)
		return rc;
	uint32_t *h_off = (uint32_t *)ctx->h_pinned;
	MDB_HIP(ctx, hipMemcpyAsync(h_off, res.leaf_off, ((size_t)n_dest + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipMemcpyAsync(h_off + n_dest + 1, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
	MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h_off[n_dest + 1] & 1024u)
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "partition_by_dest: a key lies outside the range [%lld, %lld] promised for its column",
				   (long long)own_lo, (long long)own_hi);
	if (keys32 && (h_off[n_dest + 1] & 128u))
		return mdb_set_err(ctx, -MIDORIDB_ERROR, "partition_by_dest: a key does not fit the 4-byte wire format (keys32 needs every key in "
							 "the int32 range: check mdb_dev_key_range)");
	for (uint32_t d = 0; d < n_dest; d++)
		out_counts[d] = (uint64_t)h_off[d + 1] - h_off[d];
	return MIDORIDB_OK;
}
