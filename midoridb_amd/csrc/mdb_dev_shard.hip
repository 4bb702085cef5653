/*
 * mdb_dev_shard.hip - device side of the sharded join + GROUP BY key + COUNT(*) with FIRST-LEVEL REGIONS ON THE WIRE
 * (include/mdb_dist.h: mdb_dist_join_group_count; the reference's phases it stands for are the join and
 * GROUP BY loops of src/engine/executor_select.c:1076-1149, 1526-1588).
 *
 * Before: partition by destination (histogram + scatter, a pass of its own), counts to the host, all-to-all of keys, and
 * then the local operator read what arrived and partitioned it again - 3.1 x the local operator before a byte crossed xGMI.
 * Now the sender's ONE pass is the join's first partition level itself: the compact narrow form's 512-digit histogram-free
 * scatter (mdb_dev_partition.hip), whose digit's top log2(world) bits are the destination rank.  A destination's regions
 * are one contiguous block of a size every rank can compute (fixed-capacity regions: the average + 1/16 + 1024 words), so the all-to-all is posted without
 * any count reaching a host; the region counters travel as 16 KiB beside it and are only ever read by kernels.  The
 * receiver builds region descriptors (source rank x digit x sub-region), and either joins a digit straight from its
 * regions (the hash bits below the digit index an LDS table: k_shard_leaf, one level) or runs one partition level of its
 * own over them first.  What travels is the k-bit hash of key - window base (4 bytes, or 2 when k - 9 <= 16): no row ids,
 * and the groups come out as (key, COUNT) pairs in leaf order - across ranks SQL leaves the order open, so the ordering
 * sort of the single-GPU operator (40 % of it for unique keys) does not run.
 *
 * HBM-bound byte work like the rest of the path: no MFMA.  Rooflines: sender pass = 8 n read + wbytes * n written per
 * table; receiver = the received words read once (+ one read-write level when b2 > 0) + 16 G written.
 */
#include "mdb_dev_internal.h"
#include "mdb_dev_scatter4096.h"

#define SH_NSUB 8u		/* = PART_NSUB of mdb_dev_partition.hip: sub-regions per first-level digit */
#define SH_D_BITS 9
#define SH_ONE_LEVEL_MAX_REM 14u	/* 2^14 entries x 8 bytes = 128 KiB of LDS */
#define SH_LEAF_REM 13u		/* two levels: leaves of 2^13 key values (64 KiB of counters, 1024 threads: 0.46 ms for 2 x 10^8 rows where
				 * 2^12-value leaves - twice as many workgroups, each with its fixed costs - take 0.82 and 2^14 0.55) */
#define SHW_MAX_REM 15u		/* ... and at most 15 key bits below them: 2-byte words with a spare bit, 2 x 2^15 16-bit counters = 128 KiB of LDS */
#define SH_RANGE_WORD 24	/* words of ctx->d_status that hold the left table's pruning range */

static uint32_t sh_ceil_log2(uint64_t v)
{
	uint32_t b = 0;
	while (b < 63 && (1ull << b) < v)
		b++;
	return b;
}

static uint32_t sh_round64(uint64_t v) { return (uint32_t)((v + 63) & ~63ull); }

int mdb_shard_plan_make(uint32_t world, uint32_t rank, uint32_t ntab, const uint64_t *n_max, int64_t l_lo, int64_t l_hi, int64_t r_lo, int64_t r_hi,
			mdb_shard_plan *p)
{
	memset(p, 0, sizeof(*p));
	if (world < 1 || world > 8 || (world & (world - 1)) || rank >= world || ntab < 2 || ntab > MDB_SHARD_MAX_TABS)
		return 1;
	const uint64_t n_l_max = n_max[0];
	if (r_lo > r_hi || l_lo > l_hi)
		return 1;		/* (a table without any key: the caller's ordinary path answers "no groups") */
	const uint64_t rspan = (uint64_t)r_hi - (uint64_t)r_lo + 1;
	if (rspan == 0 || rspan > (1ull << 30))
		return 1;
	uint32_t k = sh_ceil_log2(rspan);
	if (k < SH_D_BITS + 4u)
		k = SH_D_BITS + 4u;	/* (tiny windows: still 16 values per digit) */
	for (uint32_t x = 0; x < ntab; x++)
		if (n_max[x] >= 0xF0000000ull)
			return 1;
	/* windows of 2^24 .. 2^27 values, two tables: 4096 first-level digits (k_shard_scatter_wide) leave at most 15 key bits below the
	 * digit - one level, no second pass over either table (variant U: 2^27 values, 1.42 -> 1.0 ms) */
	const bool wide = ntab == 2 && k > SH_D_BITS + SH_ONE_LEVEL_MAX_REM && k <= SHW_D_BITS + SHW_MAX_REM &&
			  !(mdb_knob("MDB_SHARD_WIDE") && mdb_knob("MDB_SHARD_WIDE")[0] == '0');
	p->dbits = wide ? SHW_D_BITS : SH_D_BITS;
	p->world = world;
	p->ntab = ntab;
	p->rank = rank;
	p->D = 1u << p->dbits;
	p->Dp = p->D / world;
	p->nsub = SH_NSUB;
	p->kbits = k;
	p->key_lo = r_lo;
	p->l_rel_hi = rspan - 1;
	const uint32_t below = k - p->dbits;
	/* (4 bytes of LDS per table and key value of a leaf: three or four tables take leaves of half the values) */
	const uint32_t one_level_rem = ntab > 2 ? SH_ONE_LEVEL_MAX_REM - 1u : SH_ONE_LEVEL_MAX_REM;
	if (wide || (below <= one_level_rem && p->Dp >= 128u)) {
		p->b2 = 0;
		p->rem = below;
	} else {
		uint32_t leaf_rem = ntab > 2 ? SH_LEAF_REM - 1u : SH_LEAF_REM;
		{
			const char *e = mdb_knob("MDB_SHARD_REM");	/* (measurements) */
			if (e && atoi(e) >= 8 && atoi(e) <= (int)SH_ONE_LEVEL_MAX_REM)
				leaf_rem = (uint32_t)atoi(e);
		}
		int b2 = below > leaf_rem ? (int)(below - leaf_rem) : 2;
		if (b2 > MDB_MAX_RADIX_BITS)
			return 1;	/* (windows beyond 2^30 values) */
		if ((uint32_t)b2 > below - 2u)
			b2 = (int)below - 2;
		if (b2 < 1)
			return 1;
		p->b2 = b2;
		p->rem = below - (uint32_t)b2;
	}
	p->wbytes = (p->b2 == 0 && below <= 16u) ? 2u : 4u;
	/* regions: 1.25 x the average + 1024 words, like the single-GPU first level; the left table's by the rows expected to
	 * survive the pruning (its keys taken as evenly spread over its range; an underestimate overflows a region, which is
	 * reported and answered by the exact path) */
	const uint64_t lspan = (uint64_t)l_hi - (uint64_t)l_lo + 1;
	long double frac = lspan ? (long double)rspan / (long double)lspan : 1.0L;
	if (frac > 1.0L || lspan == 0)
		frac = 1.0L;
	const uint64_t nl_est = (uint64_t)((long double)n_l_max * frac * 1.1L) + 1;
	const uint64_t regions = (uint64_t)p->D * p->nsub;
	/* (a region's fill is the sum of its tiles' shares of a bijective hash: for 10^8 rows 24 414 +- 160 rows - the 1024 words
	 * cover that; the factor covers the XCDs' uneven tile counts.  Every word of slack crosses xGMI: 1/16, not the single-GPU
	 * first level's 1/4) */
	/* (4096 digits: a region holds an eighth of that - 3052 +- 55 rows -, and 1024 words of slack each would be a third of the buffer) */
	const uint64_t slack = wide ? 320 : 1024;
	p->cap[0] = sh_round64((nl_est < n_l_max ? nl_est : n_l_max) * 17 / 16 / regions + slack);
	for (uint32_t x = 1; x < ntab; x++)
		p->cap[x] = sh_round64(n_max[x] * 17 / 16 / regions + slack);	/* (further right tables: sized for all their rows, though only those in the window travel) */
	for (uint32_t x = 0; x < ntab; x++) {
		p->block_words[x] = (uint64_t)p->Dp * p->nsub * p->cap[x];
		if (p->block_words[x] * world >= 0xFFFFFFFFull)
			return 1;
	}
	if (p->b2) {
		/* the receiver's leaves: 1.5 x the average + 1024 (its rows are known only as a bound: what the regions can hold) */
		const uint64_t nleaves = (uint64_t)p->Dp << p->b2;
		for (uint32_t x = 0; x < ntab; x++) {
			const uint64_t bound = p->block_words[x] * world;	/* (what the regions can hold at most) */
			p->leaf_cap[x] = sh_round64(bound * 3 / 2 / nleaves + 1024);
			if (nleaves * p->leaf_cap[x] >= 0xFFFFFFFFull)
				return 1;
		}
	}
	return 0;
}

static size_t sh_recv_side_bytes(const mdb_shard_plan *p, int x)
{
	const uint64_t nreg = (uint64_t)p->D * p->nsub, max_tiles = p->block_words[x] * p->world / MDB_TILE + nreg + 1;
	size_t b = 2 * mdb_align_up(nreg * 4) + mdb_align_up((nreg + 1) * 4) + 4096;		/* region start / count, tile bases */
	if (p->b2)
		b += mdb_align_up(max_tiles * sizeof(mdb_tile_desc)) + mdb_align_up(((size_t)p->Dp << p->b2) * p->leaf_cap[x] * 4) +
		     mdb_align_up(((size_t)p->Dp << p->b2) * 4);
	return b;
}

size_t mdb_shard_arena_bytes(const mdb_shard_plan *p)
{
	/* sender: the region buffers + their cursors (the first-level partition's own carving); receiver: per table */
	size_t b = 65536;
	for (uint32_t x = 0; x < p->ntab; x++)
		b += mdb_align_up((size_t)p->D * p->nsub * p->cap[x] * 4) + 4 * mdb_align_up((size_t)p->D * p->nsub * 4 + 8) + 4096 + sh_recv_side_bytes(p, (int)x);
	return b;
}


/* ------------------------------------------------------------------ the wide fan-out form: 4096 first-level digits
 *
 * One pass per table that leaves at most 15 key bits below the digit for windows of up to 2^27 values, where the 512-digit
 * level needs a second pass over both tables.  What makes 4096 digits affordable is the tile: 32 768 (or 16 384) rows staged in
 * LDS as 2-byte words, so that a digit's run in a tile is still ~8 (4) words, written by consecutive lanes, and a tile still
 * costs one global atomic per digit - 12.5 M per 10^8 rows, as many as 512 digits x 4096-row tiles.  The staged word is all
 * that is kept per position: its spare 16th bit marks the first word of a digit's run, and the position's run - hence its
 * place in the region buffer - is the number of marks up to it (one ballot per 64 positions on top of per-chunk counts).
 * Region and cursor layout are the 512-digit level's (digit-major regions of `cap` words with nsub sub-regions per digit,
 * sub-major cursors): the receiver's descriptors do not care which kernel filled them. */
/* region capacity (words) of a table of at most n rows partitioned by mdb_scatter4096: the average + 1/16 + 320 words (a region's fill:
 * 3052 +- 55 rows at 10^8), plus - row words - one header per tile whose workgroup writes to the region's sub-region */
uint32_t mdb_scatter4096_cap(const mdb_dev_ctx *ctx, uint64_t n, bool row_words)
{
	const uint64_t regions = (uint64_t)(1u << SHW_D_BITS) * SH_NSUB, tile = 32768;
	uint64_t cap = n * 17 / 16 / regions + 320;
	if (row_words) {
		const uint64_t ntiles = (n + tile - 1) / tile, grid = ntiles < (uint64_t)ctx->num_cus ? (ntiles ? ntiles : 1) : (uint64_t)ctx->num_cus;
		const uint64_t rows_per_wg = (n + grid - 1) / grid + 1;
		cap += ((grid + SH_NSUB - 1) / SH_NSUB) * ((rows_per_wg + tile - 1) / tile) + 8;
	}
	return sh_round64(cap);
}

size_t mdb_scatter4096_arena_bytes(const mdb_dev_ctx *ctx, uint64_t n, bool row_words)
{
	const size_t nreg = (size_t)(1u << SHW_D_BITS) * SH_NSUB;
	return mdb_align_up(nreg * 4) + mdb_align_up(nreg * mdb_scatter4096_cap(ctx, n, row_words) * (row_words ? 4 : 2)) + 4096;
}

/* One 4096-digit pass over a key column (compact narrow form: every key in [key_lo, key_lo + 2^kbits), 12 < kbits <= 27): regions of `cap`
 * words - 2-byte hash bits below the digit, or 4-byte row words (see k_shard_scatter_wide) - and their cursors [nsub = 8][4096], both from the
 * arena.  report: a key outside the window raises status bit 7; otherwise rows with key - key_lo > rel_hi are dropped. */
int mdb_scatter4096(mdb_dev_ctx *ctx, const int64_t *keys, const uint64_t *nulls, uint64_t n, int64_t key_lo, uint32_t kbits, bool report, uint32_t rel_hi,
		    uint32_t cap, bool row_words, const char *name, void **regions, uint32_t **cursors)
{
	const size_t nreg = (size_t)(1u << SHW_D_BITS) * SH_NSUB;
	if (kbits <= SHW_D_BITS || kbits > SHW_D_BITS + SHW_MAX_REM || n >= 0xF0000000ull)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "4096-digit pass: %u key bits, %llu rows", kbits, (unsigned long long)n);
	uint32_t *cur = (uint32_t *)mdb_arena_take(ctx, nreg * 4);
	void *buf = mdb_arena_take(ctx, nreg * cap * (row_words ? 4 : 2));
	if (!cur || !buf)
		return -MIDORIDB_INTERNAL;
	MDB_HIP(ctx, hipMemsetAsync(cur, 0, nreg * 4, ctx->stream));
	*regions = buf;
	*cursors = cur;
	if (n == 0)
		return MIDORIDB_OK;
	shw_scatter_args a;
	memset(&a, 0, sizeof(a));
	a.keys = reinterpret_cast<const long long *>(keys);
	a.nullbits = reinterpret_cast<const unsigned long long *>(nulls);
	a.n = (uint32_t)n;
	a.key_lo = key_lo;
	a.kbits = kbits;
	a.rem = kbits - SHW_D_BITS;
	a.report = report ? 1u : 0u;
	a.rel_hi = rel_hi;
	a.out = buf;
	a.cursor = cur;
	a.cap = cap;
	a.nsub = SH_NSUB;
	a.status = ctx->d_status;
	const uint32_t tile = 32768u, ntiles = (uint32_t)((n + tile - 1) / tile);
	const uint32_t grid = ntiles < (uint32_t)ctx->num_cus ? ntiles : (uint32_t)ctx->num_cus;	/* one workgroup per CU, a contiguous range of rows each */
	a.rows_per_wg = (uint32_t)(((n + grid - 1) / grid + 1) & ~1ull);
	if (row_words) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_shard_scatter_wide<1024, 32, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
						 (int)shw_scatter_lds(tile, 4)));
		MDB_LAUNCH_LDS(ctx, name, (k_shard_scatter_wide<1024, 32, true>), grid, 1024, shw_scatter_lds(tile, 4), a);
	} else {
		/* hash words: the streaming form (mdb_dev_scatter4096.h) - 8 x 16 bytes of keys in flight per thread across all phases, as non-temporal
		 * loads (read once: they should not push the regions' half-written lines out of the L2s); same words, regions and cursors.  Same
		 * box, 10^8 unique keys in 2^27 values: 0.316 ms against 0.352 (profiles/r06/scatter4096_ab.txt) */
		const size_t lds = shs_stream_lds(tile, 2);
		if (nulls) {
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter4096_stream<1024, 32, false, true, 8, 0, 2>),
							 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			MDB_LAUNCH_LDS(ctx, name, (k_scatter4096_stream<1024, 32, false, true, 8, 0, 2>), grid, 1024, lds, a);
		} else {
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter4096_stream<1024, 32, false, false, 8, 0, 2>),
							 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			MDB_LAUNCH_LDS(ctx, name, (k_scatter4096_stream<1024, 32, false, false, 8, 0, 2>), grid, 1024, lds, a);
		}
	}
	return MIDORIDB_OK;
}

int mdb_shard_partition(mdb_dev_ctx *ctx, const mdb_shard_plan *p, int side, const int64_t *keys, const uint64_t *nulls, uint64_t n,
			const void **regions, const uint32_t **cursors)
{
	mdb_part_filter flt;
	mdb_part_result res;
	memset(&flt, 0, sizeof(flt));
	memset(&res, 0, sizeof(res));
	flt.level0_only = true;
	flt.out16 = p->wbytes == 2;
	flt.region_cap = p->cap[side];
	if (side != 1) {
		/* the left table (and any further right table) keeps the rows inside the right table's (global) key range = the
		 * window: what lies outside joins nothing on any rank */
		uint32_t *h = reinterpret_cast<uint32_t *>(ctx->h_pinned) + 512;
		h[0] = 0u;
		h[1] = (uint32_t)p->l_rel_hi;
		MDB_HIP(ctx, hipMemcpyAsync(ctx->d_status + SH_RANGE_WORD, h, 8, hipMemcpyHostToDevice, ctx->stream));
		flt.range_in = ctx->d_status + SH_RANGE_WORD;
	}
	if (n == 0) {
		/* nothing to partition: an all-zero cursor array and a region buffer nobody reads */
		uint32_t *cur = (uint32_t *)mdb_arena_take(ctx, (size_t)p->D * p->nsub * 4);
		void *buf = mdb_arena_take(ctx, (size_t)p->D * p->nsub * p->cap[side] * p->wbytes);
		if (!cur || !buf)
			return -MIDORIDB_INTERNAL;
		MDB_HIP(ctx, hipMemsetAsync(cur, 0, (size_t)p->D * p->nsub * 4, ctx->stream));
		*regions = buf;
		*cursors = cur;
		return MIDORIDB_OK;
	}
	if (p->dbits == SHW_D_BITS) {
		void *buf = NULL;
		uint32_t *cur = NULL;
		int rc4 = mdb_scatter4096(ctx, keys, nulls, n, p->key_lo, p->kbits, side == 1, (uint32_t)p->l_rel_hi, p->cap[side], false,
					  side ? "shard_scatter_wide_r" : "shard_scatter_wide_l", &buf, &cur);
		if (rc4)
			return rc4;
		*regions = buf;
		*cursors = cur;
		return MIDORIDB_OK;
	}
	int rc = mdb_partition_table(ctx, keys, nulls, n, SH_D_BITS, 0, false, false, true, &res, 2, false, p->key_lo, p->kbits, &flt);
	if (rc)
		return rc;
	if (!res.nsub || res.nsub != p->nsub || res.leaf_cap != p->cap[side] || !res.w32 || (p->wbytes == 2) != res.w16)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "sharded operator: the first level did not deliver the agreed region layout");
	*regions = res.hv;
	*cursors = res.leaf_cnt;
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ receiver */

/* region r = dl * (world * nsub) + q * nsub + s: the rows rank q sent for this rank's digit dl from its sub-region s */
__global__ void k_shard_regions(const uint32_t *__restrict__ cnt, uint32_t world, uint32_t D, uint32_t Dp, uint32_t nsub, uint32_t d0, uint32_t cap,
				uint32_t block_words, uint32_t *__restrict__ reg_start, uint32_t *__restrict__ reg_cnt, uint32_t *status)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, nreg = Dp * world * nsub;
	if (r >= nreg)
		return;
	const uint32_t dl = r / (world * nsub), q = (r / nsub) % world, s = r % nsub;
	uint32_t c = cnt[(size_t)q * D * nsub + (size_t)s * D + d0 + dl];	/* (the sender's cursors are laid out sub-major) */
	if (c == 0xFFFFFFFFu) {
		mdb_raise(status, 16384u);	/* rank q's first level failed: it sent counters that say so instead of counts (mdb_dist.hip) */
		c = 0;
	} else if (c > cap) {
		mdb_raise(status, 2u);		/* the sender's region overflowed (it said so on its own rank as well) */
		c = cap;
	}
	reg_start[r] = q * block_words + (dl * nsub + s) * cap;
	reg_cnt[r] = c;
}

/* tiles of MDB_TILE words per region: tb[r] = first tile of region r, tb[nreg] = their number (one workgroup) */
__global__ __launch_bounds__(1024) void k_shard_tiles_scan(const uint32_t *__restrict__ reg_cnt, uint32_t nreg, uint32_t *__restrict__ tb)
{
	__shared__ uint32_t s_tmp[32];
	uint32_t carry = 0;
	for (uint32_t base = 0; base <= nreg; base += 1024) {
		const uint32_t r = base + threadIdx.x;
		const uint32_t nt = r < nreg ? (reg_cnt[r] + MDB_TILE - 1) / MDB_TILE : 0u;
		uint32_t total;
		const uint32_t ex = mdb_block_excl_scan(nt, s_tmp, &total);
		if (r <= nreg)
			tb[r] = carry + ex;
		carry += total;
	}
}

__global__ void k_shard_tiles_build(const uint32_t *__restrict__ reg_start, const uint32_t *__restrict__ reg_cnt, const uint32_t *__restrict__ tb,
				    uint32_t nreg, uint32_t regs_per_seg, mdb_tile_desc *__restrict__ tiles, uint32_t max_tiles)
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
		const uint32_t r = lo, tl = t - tb[r], c = reg_cnt[r];
		d.start = reg_start[r] + tl * MDB_TILE;		/* (regions start at multiples of 64 words) */
		d.len = (c - tl * MDB_TILE) < MDB_TILE ? (c - tl * MDB_TILE) : MDB_TILE;
		d.seg = r / regs_per_seg;
	}
	tiles[t] = d;
}

struct sh_leaf_args {
	uint32_t ntab;			/* [0] left, [1 .. ntab) right tables */
	const void *words[MDB_SHARD_MAX_TABS];		/* 4- or 2-byte words */
	const uint32_t *seg_start[MDB_SHARD_MAX_TABS];	/* nseg per leaf; NULL: ONE segment per leaf at leaf * cap (the receiver's own level) */
	const uint32_t *seg_cnt[MDB_SHARD_MAX_TABS];	/* ... its counters (clamped to cap when seg_start is NULL) */
	uint32_t nseg, cap[MDB_SHARD_MAX_TABS];
	uint32_t rem, shift;		/* table index = (word >> shift) & (2^rem - 1) */
	uint32_t kbits, hash_base;	/* k-bit hash of slot s of leaf i = hash_base + (i << rem) + s */
	long long key_lo;
	long long *out_key, *out_count;
	uint32_t out_cap;
	uint32_t *status;		/* [0] flags, [1] groups so far, [2..3] joined rows (u64) */
	uint32_t right_only;		/* no left table: every slot the (one) right table has rows in is a group, COUNT(*) = those rows */
};

/* One workgroup per leaf: the right rows count into cr[] (several right tables on the same key: each into an array of its
 * own, multiplied per slot afterwards), the left rows into cl[] (only where the right side has rows), every slot with both
 * is a group: key = key_lo + unmixk(hash), COUNT(*) = cl * cr.  Plain 32-bit LDS counters; nothing is probed or compared -
 * the k-bit hash is a bijection of the window. */
template <int THREADS, typename WT>
__global__ __launch_bounds__(THREADS) void k_shard_leaf(sh_leaf_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t sh_lds[];
	__shared__ unsigned long long s_red[THREADS / 64];
	__shared__ uint32_t s_base, s_total, s_off;
	const uint32_t T = 1u << a.rem, mask = T - 1u, leaf = blockIdx.x;
	uint32_t *const s_cr = sh_lds, *const s_cl = sh_lds + T;	/* (right tables beyond the first: sh_lds + x * T, x = 2 ...) */
	for (uint32_t s = threadIdx.x; s < a.ntab * T; s += THREADS)
		sh_lds[s] = 0u;
	if (threadIdx.x == 0)
		s_total = 0;
	uint32_t groups = 0;		/* slots whose first left row this thread saw */
	__syncthreads();
	constexpr uint32_t PER = 16u / sizeof(WT);	/* words per 16-byte load */
	for (int pass = 0; pass < (int)a.ntab; pass++) {
		/* the right tables first (1, 2, ...), the left table last */
		const int side = pass + 1 < (int)a.ntab ? pass + 1 : 0;
		uint32_t *const s_cx = sh_lds + (side >= 2 ? (uint32_t)side * T : 0u);
		const WT *const base = reinterpret_cast<const WT *>(a.words[side]);
		for (uint32_t j = 0; j < a.nseg; j++) {
			uint32_t c, start;
			if (a.seg_start[side]) {
				start = a.seg_start[side][leaf * a.nseg + j];
				c = a.seg_cnt[side][leaf * a.nseg + j];
			} else {
				start = leaf * a.cap[side];
				c = a.seg_cnt[side][leaf];
				c = c < a.cap[side] ? c : a.cap[side];
			}
			const WT *const src = base + start;
			for (uint32_t i0 = 0; i0 < c; i0 += PER * THREADS * 4u) {	/* uniform trip count; four 16-byte loads in flight */
				uint4 v[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const uint32_t i = i0 + PER * ((uint32_t)u * THREADS + threadIdx.x);
					v[u] = make_uint4(0u, 0u, 0u, 0u);
					if (i < c)
						v[u] = *reinterpret_cast<const uint4 *>(src + i);
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const uint32_t i = i0 + PER * ((uint32_t)u * THREADS + threadIdx.x);
					const uint32_t w4[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
					for (uint32_t e = 0; e < PER; e++) {
						if (i + e >= c)
							continue;
						const uint32_t w = sizeof(WT) == 4 ? w4[e] : (w4[e >> 1] >> (16u * (e & 1u))) & 0xFFFFu;
						const uint32_t idx = (sizeof(WT) == 4 ? (w >> a.shift) : w) & mask;
						if (side >= 1) {
							if (atomicAdd(&s_cx[idx], 1u) == 0u && a.right_only)
								groups++;
						} else if (s_cr[idx] && atomicAdd(&s_cl[idx], 1u) == 0u)
							groups++;
					}
				}
			}
		}
		__syncthreads();
		if (a.ntab > 2 && pass + 2 == (int)a.ntab) {
			/* every right table is counted: one right count per slot = the product (a product beyond 32 bits is reported) */
			for (uint32_t sl = threadIdx.x; sl < T; sl += THREADS) {
				unsigned long long c = s_cr[sl];
				for (uint32_t x = 2; x < a.ntab; x++)
					c *= sh_lds[x * T + sl];
				if (c >> 32) {
					mdb_raise(a.status, 2048u);
					c = 0;
				}
				s_cr[sl] = (uint32_t)c;
			}
			__syncthreads();
		}
	}
	/* emit.  The groups of the leaf were counted while the left rows came in (a slot's first left row); one global atomic
	 * reserves their places.  Slots are walked THREADS at a time - consecutive threads, consecutive slots: conflict-free LDS
	 * reads - and the groups of a wave's 64 slots are written side by side (ballot + popcount, the wave's share of the
	 * workgroup's range from one LDS atomic): coalesced stores.  Any order will do: across ranks SQL leaves it open. */
	{
		uint32_t g = groups;
#pragma unroll
		for (int o = 32; o; o >>= 1)
			g += __shfl_down(g, o, MDB_WAVE);
		if (mdb_lane() == 0 && g)
			atomicAdd(&s_total, g);
	}
	__syncthreads();
	const uint32_t total = s_total;
	if (!total)
		return;
	if (threadIdx.x == 0) {
		const uint32_t nb = atomicAdd(a.status + 1, total);
		if ((uint64_t)nb + total > a.out_cap) {
			mdb_raise(a.status, 8u);
			s_base = 0xFFFFFFFFu;
		} else {
			s_base = nb;
		}
		s_off = 0;
	}
	__syncthreads();
	if (s_base == 0xFFFFFFFFu)
		return;
	unsigned long long joined = 0;
	for (uint32_t s0 = 0; s0 < T; s0 += THREADS) {
		const uint32_t s = s0 + threadIdx.x;
		const uint32_t cl = s < T ? (a.right_only ? s_cr[s] : s_cl[s]) : 0u;
		const uint64_t m = __ballot(cl != 0u);
		if (!m)
			continue;
		uint32_t wbase = 0;
		if (mdb_lane() == 0)
			wbase = atomicAdd(&s_off, (uint32_t)__popcll(m));
		wbase = __shfl(wbase, 0, MDB_WAVE);
		if (cl) {
			const uint32_t pos = s_base + wbase + (uint32_t)__popcll(m & mdb_lanemask_lt());
			const unsigned long long c = a.right_only ? (unsigned long long)cl : (unsigned long long)cl * s_cr[s];
			const uint32_t h = a.hash_base + (leaf << a.rem) + s;
			a.out_key[pos] = a.key_lo + (long long)mdb_unmixk(h, a.kbits);
			if (a.out_count)	/* (NULL: the caller wants the keys alone and learns from J == G that every COUNT is 1) */
				a.out_count[pos] = (long long)c;
			joined += c;
		}
	}
#pragma unroll
	for (int o = 32; o; o >>= 1)
		joined += __shfl_down(joined, o, MDB_WAVE);
	if (mdb_lane() == 0)
		s_red[threadIdx.x >> 6] = joined;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
#pragma unroll
		for (int w = 0; w < THREADS / 64; w++)
			t += s_red[w];
		if (t)
			atomicAdd(reinterpret_cast<unsigned long long *>(a.status + 2), t);
	}
}

/* The leaf kernel of the wide fan-out form: up to 2^15 key values per leaf, two tables, 16-bit counters (two per LDS word; a
 * count beyond 65 535 is reported - flag 2048 - and the operator takes another path), 2-byte words.  A leaf's rows arrive as
 * world x nsub short segments (a few thousand words each): they are read as ONE list of 16-byte chunks, so that every lane has
 * a load in flight whatever the segments' lengths. */
#define SHW_MAX_SEG 64u
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_shard_leaf_wide(sh_leaf_args a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t sh_lds[];
	__shared__ unsigned long long s_red[THREADS / 64];
	__shared__ uint32_t s_base, s_total, s_off;
	/* [side][segment]: first 16-byte chunk (prefix), start and length of the segment */
	__shared__ uint32_t s_seg_chunk0[2][SHW_MAX_SEG + 1], s_seg_start[2][SHW_MAX_SEG], s_seg_cnt[2][SHW_MAX_SEG];
	const uint32_t T = 1u << a.rem, HW = T >> 1, mask = T - 1u, leaf = blockIdx.x;
	uint32_t *const s_cr = sh_lds, *const s_cl = sh_lds + HW;
	/* both tables' segment tables at once (one round trip), while the counters are cleared */
	if (threadIdx.x < 2u * SHW_MAX_SEG) {
		const uint32_t side = threadIdx.x / SHW_MAX_SEG, j = threadIdx.x % SHW_MAX_SEG;
		if (j < a.nseg) {
			s_seg_start[side][j] = a.seg_start[side][leaf * a.nseg + j];
			s_seg_cnt[side][j] = a.seg_cnt[side][leaf * a.nseg + j];
		}
	}
	for (uint32_t s = threadIdx.x; s < 2u * HW; s += THREADS)
		sh_lds[s] = 0u;
	if (threadIdx.x == 0)
		s_total = 0;
	__syncthreads();
	if (threadIdx.x < 2u) {
		uint32_t run = 0;
		for (uint32_t j = 0; j < a.nseg; j++) {
			s_seg_chunk0[threadIdx.x][j] = run;
			run += (s_seg_cnt[threadIdx.x][j] + 7u) >> 3;
		}
		s_seg_chunk0[threadIdx.x][a.nseg] = run;
	}
	__syncthreads();
	/* chunk q of a table = eight 2-byte words of one of its segments (regions start at multiples of 64 words) */
	auto fetch = [&](int side, uint32_t q, uint32_t nchunks, uint4 &v, uint32_t &nv) {
		v = make_uint4(0u, 0u, 0u, 0u);
		nv = 0u;
		if (q >= nchunks)
			return;
		uint32_t lo = 0, hi = a.nseg;	/* the last segment whose first chunk is <= q */
		while (hi - lo > 1u) {
			const uint32_t mid = (lo + hi) >> 1;
			if (s_seg_chunk0[side][mid] <= q)
				lo = mid;
			else
				hi = mid;
		}
		const uint32_t off = (q - s_seg_chunk0[side][lo]) << 3, c = s_seg_cnt[side][lo];
		nv = c - off < 8u ? c - off : 8u;
		v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.words[side]) + s_seg_start[side][lo] + off);
	};
	/* (round 6) the counters' atomics do not come back: the groups are counted from the finished tables where they leave, and a 16-bit
	 * counter that overflowed shows as fields that sum to less than the rows added (a carry out of the low field takes 65 535 off the sum,
	 * out of the high one 65 536) - flag 2048 as before */
	uint32_t added[2] = { 0u, 0u };		/* rows this thread counted: [0] left (those whose key the right table holds), [1] right */
	auto count = [&](int side, const uint4 &v, uint32_t nv) {
		const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (uint32_t e = 0; e < 8u; e++) {
			if (e >= nv)
				continue;
			const uint32_t idx = ((w4[e >> 1] >> (16u * (e & 1u))) & 0xFFFFu) & mask, sh = (idx & 1u) << 4;
			if (side == 1) {
				atomicAdd(&s_cr[idx >> 1], 1u << sh);
				added[1]++;
			} else if ((s_cr[idx >> 1] >> sh) & 0xFFFFu) {
				atomicAdd(&s_cl[idx >> 1], 1u << sh);
				added[0]++;
			}
		}
	};
	const uint32_t nch_r = s_seg_chunk0[1][a.nseg], nch_l = s_seg_chunk0[0][a.nseg];
	/* the right table's first four chunks per thread AND the left table's are requested before anything is counted: the left
	 * rows are on their way while the right ones go through the LDS atomics */
	uint4 vr[4], vl[4];
	uint32_t nr[4], nl[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
		fetch(1, (uint32_t)u * THREADS + threadIdx.x, nch_r, vr[u], nr[u]);
#pragma unroll
	for (int u = 0; u < 4; u++)
		fetch(0, (uint32_t)u * THREADS + threadIdx.x, nch_l, vl[u], nl[u]);
#pragma unroll
	for (int u = 0; u < 4; u++)
		count(1, vr[u], nr[u]);
	for (uint32_t q0 = THREADS * 4u; q0 < nch_r; q0 += THREADS * 4u) {	/* (uniform trip count) */
#pragma unroll
		for (int u = 0; u < 4; u++)
			fetch(1, q0 + (uint32_t)u * THREADS + threadIdx.x, nch_r, vr[u], nr[u]);
#pragma unroll
		for (int u = 0; u < 4; u++)
			count(1, vr[u], nr[u]);
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 4; u++)
		count(0, vl[u], nl[u]);
	for (uint32_t q0 = THREADS * 4u; q0 < nch_l; q0 += THREADS * 4u) {
#pragma unroll
		for (int u = 0; u < 4; u++)
			fetch(0, q0 + (uint32_t)u * THREADS + threadIdx.x, nch_l, vl[u], nl[u]);
#pragma unroll
		for (int u = 0; u < 4; u++)
			count(0, vl[u], nl[u]);
	}
	__syncthreads();
	/* the groups leave (round 6): wave w owns the packed counter words [w * HW / waves, (w + 1) * HW / waves) - two key values each.  It reads
	 * its words ONCE into registers, counts its groups and takes its place among the workgroup's with ONE LDS atomic; the workgroup then takes
	 * its place in the output, and every wave writes the even values' groups and the odd values' of every 64 words side by side: ballots and
	 * popcounts only.  (Before: every wave iteration read two counters per lane, bumped a shared cursor and broadcast it through the crossbar -
	 * 32 times per wave: 0.15 ms of the kernel's 0.46 at 10^8 x 10^8 unique keys, profiles/r06/README.md section 10.) */
	unsigned long long joined = 0;
	/* (a lane takes PAIRS of packed words - four key values - with one 8-byte LDS read per table) */
	constexpr uint32_t NW = THREADS / 64, MAXIT = (1u << 15) / 4u / NW / 64u > 0u ? (1u << 15) / 4u / NW / 64u : 1u;	/* word pairs per lane at 2^15 values */
	const uint32_t lane = mdb_lane(), wave = threadIdx.x >> 6;
	const uint32_t HP = HW >> 1;		/* word pairs of a table (tables of 4 values and more; below: pair 0 holds the one word and a zero) */
	const uint32_t ppw = HP / NW;		/* pairs per wave (HP >= NW: tables of 2^7 values and more; else wave 0 takes all) */
	const uint32_t p_begin = ppw ? wave * ppw : 0u, p_cnt = ppw ? ppw : (wave == 0 ? (HP ? HP : 1u) : 0u);
	const uint32_t *const s_g = a.right_only ? s_cr : s_cl;
	auto pair_of = [&](const uint32_t *tab, uint32_t pi) -> uint2 {
		if (HW < 2u)	/* (a table of two key values: one word) */
			return make_uint2(tab[0], 0u);
		return *reinterpret_cast<const uint2 *>(tab + 2u * pi);
	};
	uint2 gw[MAXIT];
	uint32_t mine = 0, fsum_g = 0;
#pragma unroll
	for (uint32_t it = 0; it < MAXIT; it++) {
		const uint32_t pi = it * 64u + lane;
		gw[it] = pi < p_cnt ? pair_of(s_g, p_begin + pi) : make_uint2(0u, 0u);
		mine += ((gw[it].x & 0xFFFFu) ? 1u : 0u) + ((gw[it].x >> 16) ? 1u : 0u) + ((gw[it].y & 0xFFFFu) ? 1u : 0u) + ((gw[it].y >> 16) ? 1u : 0u);
		fsum_g += (gw[it].x & 0xFFFFu) + (gw[it].x >> 16) + (gw[it].y & 0xFFFFu) + (gw[it].y >> 16);
	}
#pragma unroll
	for (int o = 32; o; o >>= 1)
		mine += (uint32_t)__shfl_xor((int)mine, o, MDB_WAVE);
	uint32_t run = 0;
	if (lane == 0 && mine)
		run = atomicAdd(&s_total, mine);	/* (this wave's place among the workgroup's groups) */
	run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
	__syncthreads();
	const uint32_t total = s_total;
	if (threadIdx.x == 0) {
		uint32_t nb = 0;
		if (total) {
			nb = atomicAdd(a.status + 1, total);
			if ((uint64_t)nb + total > a.out_cap) {
				mdb_raise(a.status, 8u);
				nb = 0xFFFFFFFFu;
			}
		}
		s_base = nb;
	}
	__syncthreads();
	const bool writing = total && s_base != 0xFFFFFFFFu;
	run += s_base;
	const uint64_t below = mdb_lanemask_lt();
	uint32_t fsum_r = 0;
#pragma unroll
	for (uint32_t it = 0; it < MAXIT; it++) {
		const uint32_t pi = it * 64u + lane;
		if (it * 64u >= p_cnt)	/* (uniform) */
			continue;
		/* (the right table's counters: the COUNTs' other factor, and - summed - the check that none of them overflowed) */
		const uint2 r2 = pi < p_cnt ? pair_of(s_cr, p_begin + pi) : make_uint2(0u, 0u);
		fsum_r += (r2.x & 0xFFFFu) + (r2.x >> 16) + (r2.y & 0xFFFFu) + (r2.y >> 16);
		if (!writing || !__ballot((gw[it].x | gw[it].y) != 0u))	/* (uniform) */
			continue;
#pragma unroll
		for (uint32_t e = 0; e < 4u; e++) {
			const uint32_t gword = e < 2u ? gw[it].x : gw[it].y, rword = e < 2u ? r2.x : r2.y;
			const uint32_t cl = (gword >> (16u * (e & 1u))) & 0xFFFFu;
			const uint64_t m = __ballot(cl != 0u);
			if (cl) {
				const uint32_t pos = run + (uint32_t)__popcll(m & below), sl = 4u * (p_begin + pi) + e;
				const unsigned long long c = a.right_only ? (unsigned long long)cl : (unsigned long long)cl * ((rword >> (16u * (e & 1u))) & 0xFFFFu);
				const uint32_t h = a.hash_base + (leaf << a.rem) + sl;
				a.out_key[pos] = a.key_lo + (long long)mdb_unmixk(h, a.kbits);
				if (a.out_count)	/* (NULL: the caller wants the keys alone and learns from J == G that every COUNT is 1) */
					a.out_count[pos] = (long long)c;
				joined += c;
			}
			run += (uint32_t)__popcll(m);
		}
	}
	{
		/* counters that overflowed: the fields' sums against the rows added (left fields / rows in the high half, right in the low one) */
		const unsigned long long have = ((unsigned long long)(a.right_only ? 0u : fsum_g) << 32) | fsum_r;
		const unsigned long long want = ((unsigned long long)added[0] << 32) | added[1];
		unsigned long long diff = have - want;
#pragma unroll
		for (int o = 32; o; o >>= 1)
			diff += __shfl_down(diff, o, MDB_WAVE);
		__syncthreads();	/* (s_red: free) */
		if (lane == 0)
			s_red[wave] = diff;
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned long long t = 0;
#pragma unroll
			for (int w = 0; w < THREADS / 64; w++)
				t += s_red[w];
			if (t)
				mdb_raise(a.status, 2048u);
		}
		__syncthreads();
	}
#pragma unroll
	for (int o = 32; o; o >>= 1)
		joined += __shfl_down(joined, o, MDB_WAVE);
	if (mdb_lane() == 0)
		s_red[threadIdx.x >> 6] = joined;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
#pragma unroll
		for (int w = 0; w < THREADS / 64; w++)
			t += s_red[w];
		if (t)
			atomicAdd(reinterpret_cast<unsigned long long *>(a.status + 2), t);
	}
}

int mdb_shard_join(mdb_dev_ctx *ctx, const mdb_shard_plan *p, const void *const *recv, const uint32_t *const *cnt, int64_t *out_key,
		   int64_t *out_count, uint64_t cap, void *const *arrived)
{
	const uint32_t nreg = p->D * p->nsub, regs_per_digit = p->world * p->nsub, d0 = p->rank * p->Dp;
	uint32_t *reg_start[MDB_SHARD_MAX_TABS], *reg_cnt[MDB_SHARD_MAX_TABS];
	sh_leaf_args a;
	memset(&a, 0, sizeof(a));
	a.ntab = p->ntab;
	for (uint32_t x = 0; x < p->ntab; x++) {
		reg_start[x] = (uint32_t *)mdb_arena_take(ctx, (size_t)nreg * 4);
		reg_cnt[x] = (uint32_t *)mdb_arena_take(ctx, (size_t)nreg * 4);
		if (!reg_start[x] || !reg_cnt[x])
			return -MIDORIDB_INTERNAL;
	}
	/* one level: the leaf kernel needs every table; two levels: table x's own level only needs table x */
	for (uint32_t x = 0; x < p->ntab && p->b2 == 0; x++)
		if (arrived && arrived[x])
			MDB_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)arrived[x], 0));
	for (uint32_t x = 0; x < p->ntab && p->b2 == 0; x++) {
		MDB_LAUNCH(ctx, "shard_regions", k_shard_regions, (nreg + 255) / 256, 256, cnt[x], p->world, p->D, p->Dp, p->nsub, d0, p->cap[x],
			   (uint32_t)p->block_words[x], reg_start[x], reg_cnt[x], ctx->d_status);
	}
	a.rem = p->rem;
	a.shift = 32u - p->kbits;
	a.kbits = p->kbits;
	a.hash_base = d0 << (p->kbits - p->dbits);
	a.key_lo = p->key_lo;
	a.out_key = reinterpret_cast<long long *>(out_key);
	a.out_count = reinterpret_cast<long long *>(out_count);
	a.out_cap = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
	a.status = ctx->d_status;
	a.right_only = p->right_only ? 1u : 0u;
	const size_t lds = p->dbits == SHW_D_BITS ? (size_t)4 << p->rem : (size_t)(4 * p->ntab) << p->rem;
	if (lds > 150 * 1024)
		return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "sharded join: leaf tables of %zu bytes", lds);
	if (p->b2 == 0) {
		/* one level: a digit is joined straight from the regions every rank sent for it */
		a.nseg = regs_per_digit;
		for (uint32_t x = 0; x < p->ntab; x++) {
			a.words[x] = recv[x];
			a.seg_start[x] = reg_start[x];
			a.seg_cnt[x] = reg_cnt[x];
			a.cap[x] = p->cap[x];
		}
		/* (one level means k - 9 <= 14 key bits below the digit: they always fit the 2-byte words) */
		if (p->wbytes != 2)
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "sharded join: one-level plan without 2-byte words");
		if (p->dbits == SHW_D_BITS && (regs_per_digit > SHW_MAX_SEG || p->ntab != 2))
			return mdb_set_err(ctx, -MIDORIDB_INTERNAL, "sharded join: wide fan-out plan with %u segments per leaf, %u tables", regs_per_digit, p->ntab);
		/* two tables: 16-bit counters (half the LDS: two workgroups per CU at 2^14 values per digit; a count beyond 65 535 is
		 * reported and answered by another path) */
		if (p->dbits == SHW_D_BITS || (p->ntab == 2 && regs_per_digit <= SHW_MAX_SEG && p->rem >= 1u &&
					       !(mdb_knob("MDB_SHARD_LEAF_U16") && mdb_knob("MDB_SHARD_LEAF_U16")[0] == '0'))) {
			const size_t lds16 = (size_t)4 << p->rem;
			MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_shard_leaf_wide<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
			MDB_LAUNCH_LDS(ctx, "shard_leaf_wide", (k_shard_leaf_wide<1024>), p->Dp, 1024, lds16, a);
			return MIDORIDB_OK;
		}
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_shard_leaf<1024, uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "shard_leaf", (k_shard_leaf<1024, uint16_t>), p->Dp, 1024, lds, a);
		return MIDORIDB_OK;
	}
	/* two levels: the receiver's own level over the regions of all ranks, then its leaves */
	const uint32_t nleaves = p->Dp << p->b2;
	const bool leaf16 = p->rem <= 16u && !(mdb_knob("MDB_SHARD_LEAF16") && mdb_knob("MDB_SHARD_LEAF16")[0] == '0');
	for (uint32_t x = 0; x < p->ntab; x++) {
		if (arrived && arrived[x])
			MDB_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)arrived[x], 0));
		MDB_LAUNCH(ctx, "shard_regions", k_shard_regions, (nreg + 255) / 256, 256, cnt[x], p->world, p->D, p->Dp, p->nsub, d0, p->cap[x],
			   (uint32_t)p->block_words[x], reg_start[x], reg_cnt[x], ctx->d_status);
		const uint32_t max_tiles = (uint32_t)(p->block_words[x] * p->world / MDB_TILE) + nreg + 1;
		uint32_t *tb = (uint32_t *)mdb_arena_take(ctx, ((size_t)nreg + 1) * 4);
		mdb_tile_desc *tiles = (mdb_tile_desc *)mdb_arena_take(ctx, (size_t)max_tiles * sizeof(mdb_tile_desc));
		uint32_t *leaves = (uint32_t *)mdb_arena_take(ctx, (size_t)nleaves * p->leaf_cap[x] * 4);
		uint32_t *cursor = (uint32_t *)mdb_arena_take(ctx, (size_t)nleaves * 4);
		if (!tb || !tiles || !leaves || !cursor)
			return -MIDORIDB_INTERNAL;
		MDB_LAUNCH(ctx, "shard_tiles_scan", k_shard_tiles_scan, 1, 1024, reg_cnt[x], nreg, tb);
		MDB_LAUNCH(ctx, "shard_tiles_build", k_shard_tiles_build, (max_tiles + 255) / 256, 256, reg_start[x], reg_cnt[x], tb, nreg, regs_per_digit,
			   tiles, max_tiles);
		/* (a leaf's key bits fit 16: its words are written - and read by the leaf kernel - as 2 bytes) */
		int rc = mdb_partition_words_level(ctx, reinterpret_cast<const uint32_t *>(recv[x]), tiles, max_tiles, p->b2,
						   32u - p->dbits - (uint32_t)p->b2, leaves, cursor, nleaves, p->leaf_cap[x], leaf16 ? 32u - p->kbits : 0u);
		if (rc)
			return rc;
		a.words[x] = leaves;
		a.seg_start[x] = NULL;
		a.seg_cnt[x] = cursor;
		a.cap[x] = p->leaf_cap[x];
	}
	a.nseg = 1;
	if (leaf16 && p->rem > 10u) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_shard_leaf<1024, uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "shard_leaf", (k_shard_leaf<1024, uint16_t>), nleaves, 1024, lds, a);
	} else if (leaf16) {
		MDB_LAUNCH_LDS(ctx, "shard_leaf", (k_shard_leaf<512, uint16_t>), nleaves, 512, lds, a);
	} else if (p->rem > 10u) {
		MDB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_shard_leaf<1024, uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		MDB_LAUNCH_LDS(ctx, "shard_leaf", (k_shard_leaf<1024, uint32_t>), nleaves, 1024, lds, a);
	} else {
		MDB_LAUNCH_LDS(ctx, "shard_leaf", (k_shard_leaf<512, uint32_t>), nleaves, 512, lds, a);
	}
	return MIDORIDB_OK;
}
