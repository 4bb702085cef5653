// Micro-benchmark (round 5): what a vector memory instruction costs when its 64 lanes read P separate pieces of 64 / P consecutive lanes -
// the access shape of the row-order join's leaf (mdb_dev_rowjoin.hip: a tile's piece of a key digit is a few words long).  Every wave-
// instruction reads 64 x BYTES bytes; piece starts are pseudo-random 16-byte aligned places in a table that fits the L2s (16 MiB) or does
// not (2 GiB).  UNROLL loads are in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -o piece_loads piece_loads.hip && ./piece_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LPP /* lanes per piece */, int WORDS /* 4-byte words a lane loads: 1, 2, 4 */, int UNROLL>
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ tab, uint64_t tab_words, uint32_t iters, unsigned long long *out)
{
	const uint32_t lane = threadIdx.x & 63, piece = lane / LPP, in_piece = lane % LPP;
	uint64_t x = ((uint64_t)blockIdx.x * 1024 + (threadIdx.x & ~63u)) * 0x9E3779B97F4A7C15ull + piece * 0xC2B2AE3D27D4EB4Full;
	unsigned long long acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		uint32_t v[UNROLL][WORDS];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			x = x * 6364136223846793005ull + 1442695040888963407ull;
			const uint64_t start = ((x >> 20) % (tab_words / 4 - 64)) * 4;	// 16-byte aligned piece start
			const uint32_t *p = tab + start + (uint64_t)in_piece * WORDS;
			if (WORDS == 1)
				v[u][0] = *p;
			else if (WORDS == 2)
				*reinterpret_cast<uint2 *>(v[u]) = *reinterpret_cast<const uint2 *>(p);
			else
				*reinterpret_cast<uint4 *>(v[u]) = *reinterpret_cast<const uint4 *>(p);
		}
#pragma unroll
		for (int u = 0; u < UNROLL; u++)
#pragma unroll
			for (int w = 0; w < WORDS; w++)
				acc += v[u][w];
	}
	if (acc == 0x123456789abcull)
		out[0] = acc;
}

template <int LPP, int WORDS, int UNROLL>
static void run(const uint32_t *tab, uint64_t tab_words, unsigned long long *out, const char *where)
{
	const uint32_t grid = 256 * 2, iters = 64;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	k<LPP, WORDS, UNROLL><<<grid, 1024>>>(tab, tab_words, iters, out);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < 3; r++)
		k<LPP, WORDS, UNROLL><<<grid, 1024>>>(tab, tab_words, iters, out);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	ms /= 3;
	const double instr = (double)grid * 16 * iters * UNROLL, pieces = instr * (64 / LPP);
	printf("%-6s lanes/piece %2d  %2d B/lane  %2d in flight: %7.3f ms  %6.1f G wave-instr/s  %7.1f G pieces/s  %6.0f GB/s useful\n", where, LPP, WORDS * 4, UNROLL, ms,
	       instr / (ms * 1e-3) / 1e9, pieces / (ms * 1e-3) / 1e9, instr * 64 * WORDS * 4 / (ms * 1e-3) / 1e9);
}

template <int UNROLL>
static void sweep(const uint32_t *tab, uint64_t words, unsigned long long *out, const char *where)
{
	run<1, 1, UNROLL>(tab, words, out, where);
	run<1, 4, UNROLL>(tab, words, out, where);
	run<2, 2, UNROLL>(tab, words, out, where);
	run<4, 1, UNROLL>(tab, words, out, where);
	run<4, 2, UNROLL>(tab, words, out, where);
	run<8, 1, UNROLL>(tab, words, out, where);
	run<8, 2, UNROLL>(tab, words, out, where);
	run<16, 1, UNROLL>(tab, words, out, where);
	run<64, 1, UNROLL>(tab, words, out, where);
	run<64, 4, UNROLL>(tab, words, out, where);
}

int main()
{
	uint32_t *tab;
	unsigned long long *out;
	const uint64_t big = (uint64_t)2 << 30;
	hipMalloc(&tab, big);
	hipMalloc(&out, 8);
	hipMemset(tab, 1, big);
	sweep<8>(tab, ((uint64_t)16 << 20) / 4, out, "L2");
	sweep<8>(tab, big / 4, out, "HBM");
	sweep<2>(tab, big / 4, out, "HBM");
	return 0;
}
