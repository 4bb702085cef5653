/*
 * mdb_dev_order.hip - the group records of the fused operators into the reference's first-occurrence order (executor_select.c:1542-1583)
 * (split off mdb_dev_join.hip; what the files share: mdb_dev_join_internal.h).  Hand-written HIP for gfx950, HBM-bound
 * integer work: no MFMA.
 */
#include "mdb_dev_join_internal.h"

/* ------------------------------------------------------------------ ordering the groups by first row id
 *
 * The group records (first row id in the top kbits, COUNT(*) below) were radix-partitioned on the top
 * bits of the row id, so leaf i holds exactly the records whose row id lies in [i * range, (i+1) * range),
 * range <= ORD_RANGE (4096).  Row ids are distinct, so dropping each record at LDS slot (row id - i * range) and
 * compacting the slots in order sorts the leaf; leaves are already in order.  This is what reproduces the
 * reference's "survivors keep table order" (executor_select.c:1542-1583) without 8-byte random writes
 * into a table-sized array.
 */
#define ORD_THREADS 512
#define ORD_PER_THREAD 8
#define ORD_RANGE (ORD_THREADS * ORD_PER_THREAD)	/* 4096 row ids per ordering leaf (32 KiB of LDS slots: 4 workgroups per CU; 8192 ids x 1024 threads measured 15 % slower, 2048 x 256 no faster) */
#define ORD_RANGE_BITS 12


struct ord_args {
	const unsigned long long *rec;
	const uint32_t *off;		/* exact layout: leaf offsets = output positions */
	const uint32_t *cnt;		/* fast layout: records per leaf ... */
	const uint32_t *out_base;	/* ... and their exclusive prefix = output positions */
	uint32_t cap;
	uint32_t kbits, leaf_bits;
	uint32_t *out_first;
	int64_t *out_count;		/* the record's payload as int64 (COUNT(*)), or ... */
	uint32_t *out_val32;		/* ... payload - 1 as uint32 (right row id of a join pair) */
	const int64_t *keys;		/* optional: key column to gather the group keys from ... */
	int64_t *out_key;		/* ... into here (keys[first]) */
	uint32_t keys32;		/* `keys` is an int32 column */
	uint32_t rec32;			/* the records are 4-byte words: (row id << (32 - kbits)) | payload */
	uint32_t keyed_cbits;		/* != 0: keyed records (gc_args.keyed_cbits): payload = hashed key << keyed_cbits | COUNT(*); the group key
					 * is key_lo + mdb_unmixk(hashed key, key_bits), nothing is gathered */
	uint32_t key_bits;
	int64_t key_lo;
	uint32_t *status;		/* k_order_leaf_sparse: bit 1 when a leaf holds more records than it can rank (NULL: such a leaf is skipped in silence -
					 * the caller learns of it from the kernel that filled the leaves) */
	uint32_t out_cap;		/* k_order_leaf_sparse, != 0: rows of the output columns; groups beyond are not written (launched before the group
					 * count is known to the host) */
};

__global__ __launch_bounds__(ORD_THREADS) void k_order_leaf(ord_args a)
{
	__shared__ unsigned long long s_slot[ORD_RANGE];
	__shared__ uint32_t s_scan[32];
	const uint32_t leaf = blockIdx.x;
	uint32_t b, e, base;
	if (a.cap) {
		const uint32_t c = a.cnt[leaf];
		b = leaf * a.cap;
		e = b + (c < a.cap ? c : a.cap);
		base = a.out_base[leaf];
	} else {
		b = a.off[leaf];
		e = a.off[leaf + 1];
		base = b;
	}
	if (b == e)
		return;
	const uint32_t range_bits = a.kbits - a.leaf_bits;
	const uint32_t range = 1u << range_bits;
	const unsigned long long cmask = (1ull << (64 - a.kbits)) - 1ull;
	/* thread t owns the ORD_PER_THREAD consecutive slots [t * ORD_PER_THREAD, ...) */
#pragma unroll
	for (int k = 0; k < ORD_PER_THREAD; k++)
		s_slot[threadIdx.x + (uint32_t)k * ORD_THREADS] = 0ull;
	__syncthreads();
	if (a.rec32) {
		const uint32_t *const rec = reinterpret_cast<const uint32_t *>(a.rec);
		const uint32_t cm32 = (1u << (32 - a.kbits)) - 1u;
		for (uint32_t i = b + threadIdx.x; i < e; i += ORD_THREADS) {
			const uint32_t r = rec[i];
			s_slot[(r >> (32 - a.kbits)) & (range - 1)] = r & cm32;
		}
	} else {
		for (uint32_t i = b + threadIdx.x; i < e; i += ORD_THREADS) {
			const unsigned long long r = a.rec[i];
			s_slot[(uint32_t)(r >> (64 - a.kbits)) & (range - 1)] = r & cmask;	/* COUNT(*) >= 1 marks the slot */
		}
	}
	__syncthreads();
	unsigned long long c[ORD_PER_THREAD];
	uint32_t mine = 0;
#pragma unroll
	for (int k = 0; k < ORD_PER_THREAD; k++) {
		c[k] = s_slot[threadIdx.x * ORD_PER_THREAD + k];
		mine += c[k] != 0;
	}
	uint32_t total;
	uint32_t pos = mdb_block_excl_scan(mine, s_scan, &total);	/* (syncs: every slot has been read) */
	/* compact in LDS - (slot index, COUNT) packed in one word: COUNT < 2^(64-kbits) <= 2^51 - so that the
	 * global writes below are coalesced (thread-contiguous slots would scatter them 64 B apart) */
#pragma unroll
	for (int k = 0; k < ORD_PER_THREAD; k++)
		if (c[k])
			s_slot[pos++] = ((unsigned long long)(threadIdx.x * ORD_PER_THREAD + k) << 51) | c[k];
	__syncthreads();
	const uint32_t first_base = leaf << range_bits;
	for (uint32_t i = threadIdx.x; i < total; i += ORD_THREADS) {
		const unsigned long long v = s_slot[i];
		const uint32_t first = first_base + (uint32_t)(v >> 51);
		if (a.out_first)		/* (a caller that only wants keys and counts: 4 bytes per group less to write) */
			a.out_first[base + i] = first;
		if (a.keyed_cbits) {
			const unsigned long long pay = v & ((1ull << 51) - 1ull);
			a.out_count[base + i] = (int64_t)(pay & ((1ull << a.keyed_cbits) - 1ull));
			if (a.out_key)
				a.out_key[base + i] = a.key_lo + (int64_t)mdb_unmixk((uint32_t)(pay >> a.keyed_cbits), a.key_bits);
			continue;
		}
		if (a.out_val32)
			a.out_val32[base + i] = (uint32_t)(v & ((1ull << 51) - 1ull)) - 1u;
		else
			a.out_count[base + i] = (int64_t)(v & ((1ull << 51) - 1ull));
		if (a.out_key)
			a.out_key[base + i] = a.keys32 ? (int64_t)reinterpret_cast<const int32_t *>(a.keys)[first] : a.keys[first];
	}
}

/* The same for FEW records (selective joins: 6.25 * 10^6 groups among 10^8 left rows are 256 records per 4096-id leaf - 24 414
 * workgroups that mostly clear and scan empty LDS slots).  Leaves of 2^16 row ids instead: a record's place among its leaf's
 * records is the number of records with a smaller row id, i.e. the number of set bits below its own in a BITMAP of the leaf's
 * row ids (8 KiB of LDS) - records in registers, one LDS atomic each to set the bit, a block scan over the word popcounts,
 * one popcount each to rank; the records are then staged in LDS at their ranks, so that the three output columns are written
 * with consecutive threads on consecutive rows (written straight from the registers - 64 scattered rows per store
 * instruction - the kernel took 0.20 ms for 6.25 * 10^6 records, staged 0.054; k_order_leaf takes 0.10). */
#define OS_THREADS 1024
#define OS_RANGE_BITS 16u
#define OS_PER_THREAD 8
#define OS_MAX_REC (OS_THREADS * OS_PER_THREAD)

__global__ __launch_bounds__(OS_THREADS) void k_order_leaf_sparse(ord_args a)
{
	__shared__ uint32_t s_bits[1u << (OS_RANGE_BITS - 5)];
	__shared__ uint32_t s_pfx[1u << (OS_RANGE_BITS - 5)];
	__shared__ unsigned long long s_stage[OS_MAX_REC];	/* the records at their ranks */
	__shared__ uint32_t s_scan[32];
	const uint32_t leaf = blockIdx.x;
	const uint32_t c = a.cnt[leaf], b = leaf * a.cap, e = b + c;
	if (!c)
		return;
	uint32_t base;
	if (a.out_base) {
		base = a.out_base[leaf];
	} else {		/* (uniform) no scanned counts: the leaves before this one are few enough (<= 1536) to be added up here */
		uint32_t mine = 0;
		for (uint32_t i = threadIdx.x; i < leaf; i += OS_THREADS)
			mine += a.cnt[i];
		(void)mdb_block_excl_scan(mine, s_scan, &base);
		__syncthreads();	/* (s_scan is used again below) */
	}
	if (c > a.cap || c > OS_MAX_REC) {	/* the scatter's region overflowed, or more records than the registers of a workgroup hold
						 * (row ids bunched): the general path takes over */
		if (threadIdx.x == 0 && a.status)
			mdb_raise(a.status, 2u);
		return;
	}
	const uint32_t range_bits = a.kbits - a.leaf_bits, range = 1u << range_bits;
	const uint32_t words = range_bits > 5 ? 1u << (range_bits - 5) : 1u;
	for (uint32_t w = threadIdx.x; w < words; w += OS_THREADS)
		s_bits[w] = 0u;
	__syncthreads();
	unsigned long long r[OS_PER_THREAD];
#pragma unroll
	for (int k = 0; k < OS_PER_THREAD; k++) {
		const uint32_t i = b + threadIdx.x + (uint32_t)k * OS_THREADS;
		r[k] = a.rec[i < e ? i : b];	/* (unconditional loads - issued together; past the end: the first record, dropped) */
	}
#pragma unroll
	for (int k = 0; k < OS_PER_THREAD; k++) {
		const uint32_t i = b + threadIdx.x + (uint32_t)k * OS_THREADS;
		r[k] = i < e ? r[k] : 0ull;
		if (r[k]) {
			const uint32_t idx = (uint32_t)(r[k] >> (64 - a.kbits)) & (range - 1);
			atomicOr(&s_bits[idx >> 5], 1u << (idx & 31u));
		}
	}
	__syncthreads();
	/* exclusive prefix of the words' popcounts: thread t owns words [t * per, t * per + per) */
	const uint32_t per = (words + OS_THREADS - 1) / OS_THREADS;
	uint32_t mine = 0;
	for (uint32_t q = 0; q < per; q++) {
		const uint32_t w = threadIdx.x * per + q;
		if (w < words)
			mine += (uint32_t)__popc(s_bits[w]);
	}
	uint32_t total;
	uint32_t run = mdb_block_excl_scan(mine, s_scan, &total);
	for (uint32_t q = 0; q < per; q++) {
		const uint32_t w = threadIdx.x * per + q;
		if (w < words) {
			s_pfx[w] = run;
			run += (uint32_t)__popc(s_bits[w]);
		}
	}
	__syncthreads();
	const unsigned long long cmask = (1ull << (64 - a.kbits)) - 1ull;
#pragma unroll
	for (int k = 0; k < OS_PER_THREAD; k++) {
		if (!r[k])
			continue;
		const uint32_t idx = (uint32_t)(r[k] >> (64 - a.kbits)) & (range - 1);
		s_stage[s_pfx[idx >> 5] + (uint32_t)__popc(s_bits[idx >> 5] & ((1u << (idx & 31u)) - 1u))] = r[k];
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += OS_THREADS) {
		const unsigned long long rec = s_stage[i];
		const uint32_t idx = (uint32_t)(rec >> (64 - a.kbits)) & (range - 1);
		const uint32_t pos = base + i;
		if (a.out_cap && pos >= a.out_cap)
			break;
		const uint32_t first = (leaf << range_bits) + idx;
		const unsigned long long pay = rec & cmask;
		if (a.out_first)
			a.out_first[pos] = first;
		if (a.keyed_cbits) {
			a.out_count[pos] = (int64_t)(pay & ((1ull << a.keyed_cbits) - 1ull));
			if (a.out_key)
				a.out_key[pos] = a.key_lo + (int64_t)mdb_unmixk((uint32_t)(pay >> a.keyed_cbits), a.key_bits);
		} else {
			a.out_count[pos] = (int64_t)pay;
			if (a.out_key)
				a.out_key[pos] = a.keys32 ? (int64_t)reinterpret_cast<const int32_t *>(a.keys)[first] : a.keys[first];
		}
	}
}

/* bits of the ordering sort: leaves of at most ORD_RANGE row ids, at most 9 bits per level */
bool order_bits(uint64_t n_l, uint32_t *kbits, int *sb1, int *sb2)
{
	uint32_t k = 1;
	while (k < 32 && (1ull << k) < n_l)
		k++;
	int b = (int)k - ORD_RANGE_BITS;
	if (b < 1)
		b = 1;
	if (b > 2 * MDB_MAX_RADIX_BITS)
		return false;
	*kbits = k;
	if (b <= MDB_MAX_RADIX_BITS) {
		*sb1 = b;
		*sb2 = 0;
	} else {
		*sb1 = (b + 1) / 2;
		*sb2 = b - *sb1;
	}
	return true;
}

/* first-level digits of the ordering sort that can occur: row ids are < n_l, not < 2^kbits */
uint32_t order_digits0(uint64_t n_l, uint32_t kbits, int sb1)
{
	const uint32_t shift = kbits - (uint32_t)sb1;
	return (uint32_t)(((n_l ? n_l - 1 : 0) >> shift) + 1);
}

/* k_order_leaf_sparse is tried for row ids of kbits bits when its leaves (2^OS_RANGE_BITS ids) take one or two scatter levels */
static bool order_sparse_bits(uint32_t kbits)
{
	return kbits >= OS_RANGE_BITS + 2 && kbits - OS_RANGE_BITS <= 2 * MDB_MAX_RADIX_BITS;
}

/* most records the attempt is made for (a leaf holds OS_MAX_REC: the average leaves half of the slack of the scatter's regions),
 * and most list slots - zero-filled gaps included - its arena is reserved for */
static uint64_t order_sparse_most_records(uint32_t kbits)
{
	return (uint64_t)(OS_MAX_REC - 1024) * 2 / 3 << (kbits - OS_RANGE_BITS);
}

static uint64_t order_sparse_most_slots(uint32_t kbits)
{
	return 4 * order_sparse_most_records(kbits);
}

/* Order a record list ((row id << (64 - kbits)) | payload, zero words = gaps) by row id and deliver it:
 * histogram-free regions first; if one overflows (the gaps of the list can bunch the records of one XCD's tile
 * range) the exact layout redoes the sort.  Synchronises. */
/* rec32: every payload is below 2^(32 - kbits) - the sort's first level then folds the records into 4-byte words and
 * everything after it moves half the bytes (10^8 groups of one row each: 1.3 -> 0.9 ms for the ordering) */
int order_records(mdb_dev_ctx *ctx, const unsigned long long *rec, uint64_t list_len, uint64_t n_l, uint32_t kbits, int sb1,
			 int sb2, uint32_t *out_first, int64_t *out_count, uint32_t *out_val32, const int64_t *keys, int64_t *out_key,
			 bool keys32, bool rec32, uint32_t keyed_cbits, uint32_t key_bits, int64_t key_lo,
			 bool in32 /* the list already holds 4-byte records */, uint64_t n_rec /* records in the list (0: unknown) */)
{
	rec32 = (rec32 || in32) && sb2 > 0 && kbits < 32 && !keyed_cbits;
	if (in32 && !rec32)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "4-byte group records need the two-level ordering sort");
	const uint32_t ord_range = 1u << (kbits - (uint32_t)(sb1 + sb2));
	uint64_t *h = ctx->h_pinned;
	int rc;
	/* few 8-byte records of a join (out_count wanted): leaves of 2^16 row ids ranked through a bitmap (k_order_leaf_sparse) */
	if (!rec32 && !out_val32 && out_count && order_sparse_bits(kbits) &&
	    !(mdb_knob("MDB_ORDER_SPARSE") && mdb_knob("MDB_ORDER_SPARSE")[0] == '0')) {
		const uint32_t lb = kbits - OS_RANGE_BITS;
		const int s1 = (int)((lb + 1) / 2), s2 = (int)lb - s1;
		/* (the list has zero-filled gaps - chunk tails -, the scatter skips them: what counts is the number of records) */
		/* leaves that can hold records: those below n_l (a caller that knows where the largest row id lies passes that) */
		const uint64_t used_leaves = ((n_l ? n_l - 1 : 0) >> OS_RANGE_BITS) + 1;
		if ((n_rec ? n_rec : list_len) <= (uint64_t)(OS_MAX_REC - 1024) * 2 / 3 * used_leaves &&
		    (n_rec ? n_rec : list_len) <= order_sparse_most_records(kbits) && list_len <= order_sparse_most_slots(kbits)) {
			mdb_part_result ps;
			rc = mdb_partition_raw(ctx, (const uint64_t *)rec, list_len, s1, s2, 0, true, order_digits0(n_l, kbits, s1), &ps, true, 0);
			if (rc)
				return rc;
			if (ps.leaf_cap && !ps.w32) {
				uint32_t *obase = (uint32_t *)mdb_arena_take(ctx, ((size_t)ps.nleaves + 1) * 4);
				uint32_t *otmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words((uint64_t)ps.nleaves + 1) * 4);
				if (!obase || !otmp)
					return -MIDORIDB_INTERNAL;
				if (ps.nleaves <= MDB_SCAN_FROM_MAX) {
					rc = mdb_scan_u32_small_from(ctx, ps.leaf_cnt, ps.nleaves, obase);
				} else {
					MDB_HIP(ctx, hipMemcpyAsync(obase, ps.leaf_cnt, (size_t)ps.nleaves * 4, hipMemcpyDeviceToDevice, ctx->stream));
					MDB_HIP(ctx, hipMemsetAsync(obase + ps.nleaves, 0, 4, ctx->stream));
					rc = mdb_scan_u32_inplace(ctx, obase, (uint64_t)ps.nleaves + 1, otmp);
				}
				if (rc)
					return rc;
				ord_args oa;
				memset(&oa, 0, sizeof(oa));
				oa.rec = (const unsigned long long *)ps.hv;
				oa.cnt = ps.leaf_cnt;
				oa.cap = ps.leaf_cap;
				oa.out_base = obase;
				oa.kbits = kbits;
				oa.leaf_bits = lb;
				oa.out_first = out_first;
				oa.out_count = out_count;
				oa.keys = keys;
				oa.out_key = out_key;
				oa.keys32 = keys32 ? 1u : 0u;
				oa.keyed_cbits = keyed_cbits;
				oa.key_bits = key_bits;
				oa.key_lo = key_lo;
				oa.status = ctx->d_status;
				MDB_LAUNCH(ctx, "order_leaf_sparse", k_order_leaf_sparse, ps.nleaves, OS_THREADS, oa);
				MDB_HIP(ctx, hipMemcpyAsync(&h[8], ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
				MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
				if (!((uint32_t)h[8] & 2u))
					return MIDORIDB_OK;
				MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));	/* a region overflowed (row ids bunched): the general path */
			}
		}
	}
	for (int sort_fast = 1; sort_fast >= 0; sort_fast--) {
		mdb_part_result ps;
		rc = mdb_partition_raw(ctx, (const uint64_t *)rec, list_len, sb1, sb2, ord_range, sort_fast != 0, order_digits0(n_l, kbits, sb1),
				       &ps, true, (rec32 && sort_fast != 0) ? (in32 ? 2 : 1) : 0, (uint64_t)1 << (kbits - (uint32_t)sb1));
		if (rc)
			return rc;
		ord_args oa;
		memset(&oa, 0, sizeof(oa));
		oa.rec = (const unsigned long long *)ps.hv;
		oa.off = ps.leaf_off;
		oa.cnt = ps.leaf_cnt;
		oa.cap = ps.leaf_cap;
		oa.out_base = NULL;
		oa.kbits = kbits;
		oa.leaf_bits = (uint32_t)(sb1 + sb2);
		oa.out_first = out_first;
		oa.out_count = out_count;
		oa.out_val32 = out_val32;
		oa.keys = keys;
		oa.out_key = out_key;
		oa.keys32 = keys32 ? 1u : 0u;
		oa.rec32 = ps.w32 ? 1u : 0u;
		oa.keyed_cbits = keyed_cbits;
		oa.key_bits = key_bits;
		oa.key_lo = key_lo;
		if (ps.leaf_cap) {
			/* fast layout: output position of a leaf = exclusive prefix of the leaf sizes */
			uint32_t *obase = (uint32_t *)mdb_arena_take(ctx, ((size_t)ps.nleaves + 1) * 4);
			uint32_t *otmp = (uint32_t *)mdb_arena_take(ctx, mdb_scan_scratch_words((uint64_t)ps.nleaves + 1) * 4);
			if (!obase || !otmp)
				return -MIDORIDB_INTERNAL;
			if (ps.nleaves <= MDB_SCAN_FROM_MAX) {
				rc = mdb_scan_u32_small_from(ctx, ps.leaf_cnt, ps.nleaves, obase);
			} else {
				MDB_HIP(ctx, hipMemcpyAsync(obase, ps.leaf_cnt, (size_t)ps.nleaves * 4, hipMemcpyDeviceToDevice, ctx->stream));
				MDB_HIP(ctx, hipMemsetAsync(obase + ps.nleaves, 0, 4, ctx->stream));
				rc = mdb_scan_u32_inplace(ctx, obase, (uint64_t)ps.nleaves + 1, otmp);
			}
			if (rc)
				return rc;
			oa.out_base = obase;
		}
		MDB_LAUNCH(ctx, "order_leaf", k_order_leaf, ps.nleaves, ORD_THREADS, oa);
		MDB_HIP(ctx, hipMemcpyAsync(&h[8], ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (!sort_fast || !((uint32_t)h[8] & 2u))
			break;
		if (in32)
			return GC_RETRY_REC64;	/* (a region of the ordering sort overflowed: its exact layout reads 8-byte records) */
		MDB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));	/* the other flag bits were checked before */
	}
	return MIDORIDB_OK;
}

/* Group records that their leaf kernel wrote straight into the ranges k_order_leaf_sparse ranks (k_leaf_wide4's ranged emit): the two
 * scatter levels, their tile kernels and the record list are not there at all - variant D: 0.067 of the ordering's 0.115 ms. */
bool order_ranges_apply(uint64_t n_l, uint32_t kbits, uint64_t groups, uint32_t *nranges)
{
	static_assert(ORDER_RANGE_BITS == OS_RANGE_BITS && ORDER_RANGE_CAP == OS_MAX_REC, "the ranged emit fills k_order_leaf_sparse's leaves");
	if (!n_l || !order_sparse_bits(kbits))
		return false;
	const uint64_t used = ((n_l - 1) >> OS_RANGE_BITS) + 1;
	*nranges = (uint32_t)used;
	/* (the same head-room as order_records leaves its sparse attempt: an average range two thirds full at most) */
	return used <= 1536 && groups && groups <= (uint64_t)(OS_MAX_REC - 1024) * 2 / 3 * used;
}

int order_presorted(mdb_dev_ctx *ctx, const unsigned long long *regions, const uint32_t *counts, uint32_t nranges, uint32_t kbits, uint32_t *out_first,
		    int64_t *out_count, const int64_t *keys, int64_t *out_key, bool keys32, uint32_t keyed_cbits, uint32_t key_bits, int64_t key_lo,
		    uint64_t early_cap)
{
	/* (no scan of the counts: at most 1536 ranges - every workgroup adds up the counts before its own) */
	ord_args oa;
	memset(&oa, 0, sizeof(oa));
	oa.rec = regions;
	oa.cnt = counts;
	oa.cap = ORDER_RANGE_CAP;
	oa.out_base = NULL;
	oa.kbits = kbits;
	oa.leaf_bits = kbits - OS_RANGE_BITS;
	oa.out_first = out_first;
	oa.out_count = out_count;
	oa.keys = keys;
	oa.out_key = out_key;
	oa.keys32 = keys32 ? 1u : 0u;
	oa.keyed_cbits = keyed_cbits;
	oa.key_bits = key_bits;
	oa.key_lo = key_lo;
	/* early_cap != 0: launched behind the kernel that fills the ranges, before its status words have been looked at - no status of its own, no
	 * write beyond the caller's columns, no sync: the caller's one sync covers both kernels */
	oa.status = early_cap ? NULL : ctx->d_status;
	oa.out_cap = early_cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)early_cap;
	MDB_LAUNCH(ctx, "order_leaf_sparse", k_order_leaf_sparse, nranges, OS_THREADS, oa);
	if (!early_cap)
		MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MIDORIDB_OK;
}

/* exported for mdb_dev_sort.hip (multi-column GROUP BY): same list format, bits chosen here */
int mdb_order_records_by_rowid(mdb_dev_ctx *ctx, const unsigned long long *rec, uint64_t list_len, uint64_t n_rows, uint32_t kbits,
			       uint32_t *out_first, int64_t *out_count)
{
	uint32_t kb = 0;
	int sb1 = 0, sb2 = 0;
	if (!order_bits(n_rows, &kb, &sb1, &sb2) || kb != kbits)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "record ordering: unsupported row-id width");
	return order_records(ctx, rec, list_len, n_rows, kbits, sb1, sb2, out_first, out_count, NULL, NULL, NULL);
}

/* arena bytes of order_records() for a list of at most `cap` slots */
size_t order_records_arena_bytes(uint64_t cap, uint64_t n_l, uint32_t kbits, int sb1, int sb2)
{
	const uint32_t ord_range = 1u << (kbits - (uint32_t)(sb1 + sb2));
	size_t sparse = 0;	/* the attempt with 2^16-id leaves (k_order_leaf_sparse) comes first and may be followed by the general path */
	if (order_sparse_bits(kbits)) {
		const uint32_t lb = kbits - OS_RANGE_BITS;
		const int s1 = (int)((lb + 1) / 2), s2 = (int)lb - s1;
		const uint64_t most = order_sparse_most_slots(kbits);
		sparse = mdb_partition_raw_arena_bytes(cap < most ? cap : most, s1, s2, 0, true, order_digits0(n_l, kbits, s1)) +
			 2 * (((size_t)1 << lb) + 4096) * 8;
	}
	return sparse + mdb_partition_raw_arena_bytes(cap, sb1, sb2, ord_range, true, order_digits0(n_l, kbits, sb1), (uint64_t)1 << (kbits - (uint32_t)sb1)) +
	       mdb_partition_raw_arena_bytes(cap, sb1, sb2, ord_range, false, 0) + 2 * (((size_t)1 << (sb1 + sb2)) + 4096) * 8;
}

size_t mdb_order_records_arena_bytes(uint64_t cap, uint64_t n_rows, uint32_t *kbits_out)
{
	uint32_t kb = 0;
	int sb1 = 0, sb2 = 0;
	if (!order_bits(n_rows, &kb, &sb1, &sb2))
		return 0;
	*kbits_out = kb;
	return order_records_arena_bytes(cap, n_rows, kb, sb1, sb2);
}
