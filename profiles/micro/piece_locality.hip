// Micro-benchmark (round 6): is the rate of small scattered pieces (the row-order join's leaf: 25 x 10^6 pieces of 16 - 48 bytes per stream) set by
// the NUMBER of pieces, or by where they lie?  Every wave-instruction reads 8 pieces of 8 lanes x BYTES bytes; the pieces' starts are pseudo-random
//   * anywhere in a table of 16 MiB ... 2 GiB                                   ("table N MiB")
//   * anywhere in a 2 GiB table, but all 8 pieces of an instruction inside ONE 2 MiB / 64 KiB window    ("window")
//   * one tile stride (384 KiB) apart, as the leaf's are: piece p of an instruction at base + p * stride + small random    ("strided")
// (address arithmetic is a mask and a multiply - no division: profiles/micro/piece_loads.hip spent its time in a 64-bit modulo)
//   hipcc --offload-arch=gfx950 -O3 -o piece_locality piece_locality.hip && ./piece_locality
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int WORDS, int UNROLL, int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ tab, uint64_t mask_words /* table words - 1 */, uint64_t win_words /* window words - 1 */, uint32_t iters,
					 unsigned long long *out)
{
	const uint32_t lane = threadIdx.x & 63, piece = lane / 8, in_piece = lane % 8;
	uint64_t xw = ((uint64_t)blockIdx.x * 1024 + (threadIdx.x & ~63u)) * 0x9E3779B97F4A7C15ull + 12345;	// per wave
	unsigned long long acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		uint32_t v[UNROLL][WORDS];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			xw = xw * 6364136223846793005ull + 1442695040888963407ull;
			const uint64_t xp = (xw ^ (piece * 0xC2B2AE3D27D4EB4Full)) * 0xD6E8FEB86659FD93ull;	// per piece
			uint64_t start;
			if (MODE == 0)		// anywhere in the table
				start = (xp >> 17) & mask_words;
			else if (MODE == 1)	// the instruction's pieces inside one window
				start = (((xw >> 17) & mask_words) & ~win_words) | ((xp >> 17) & win_words);
			else			// one tile stride apart
				start = (((xw >> 17) & mask_words) + (uint64_t)piece * (98304 + 80) + ((xp >> 40) & 63u) * 4) & mask_words;
			start &= ~3ull;		// 16-byte aligned
			const uint32_t *p = tab + start + (uint64_t)in_piece * WORDS;
			if (WORDS == 1)
				v[u][0] = __builtin_nontemporal_load(p);
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

template <int WORDS, int UNROLL, int MODE>
static void run(const uint32_t *tab, uint64_t tab_bytes, uint64_t win_bytes, unsigned long long *out, const char *what)
{
	const uint32_t grid = 256, iters = 128;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const uint64_t mw = tab_bytes / 4 - 1 - 65536 /* room behind the last start */, ww = win_bytes / 4 - 1;
	k<WORDS, UNROLL, MODE><<<grid, 1024>>>(tab, mw, ww, iters, out);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < 3; r++)
		k<WORDS, UNROLL, MODE><<<grid, 1024>>>(tab, mw, ww, iters, out);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	ms /= 3;
	const double instr = (double)grid * 16 * iters * UNROLL, pieces = instr * 8;
	printf("%-34s %2d B/lane %2d in flight, 16 waves per CU: %7.3f ms  %6.2f G wave-instr/s  %6.1f G pieces/s  %6.0f GB/s useful\n", what, WORDS * 4, UNROLL, ms, instr / (ms * 1e-3) / 1e9,
	       pieces / (ms * 1e-3) / 1e9, instr * 64 * WORDS * 4 / (ms * 1e-3) / 1e9);
}

int main()
{
	uint32_t *tab;
	unsigned long long *out;
	const uint64_t big = (uint64_t)2 << 30;
	hipMalloc(&tab, big + (1 << 20));
	hipMalloc(&out, 8);
	hipMemset(tab, 1, big);
	char name[64];
	for (uint64_t mib : { 16ull, 64ull, 256ull, 1024ull, 2048ull }) {
		snprintf(name, sizeof(name), "table %llu MiB", (unsigned long long)mib);
		run<1, 8, 0>(tab, mib << 20, 0, out, name);
		run<2, 8, 0>(tab, mib << 20, 0, out, name);
	}
	run<1, 8, 1>(tab, big, 2 << 20, out, "2 GiB, instruction in 2 MiB");
	run<2, 8, 1>(tab, big, 2 << 20, out, "2 GiB, instruction in 2 MiB");
	run<1, 8, 1>(tab, big, 64 << 10, out, "2 GiB, instruction in 64 KiB");
	run<2, 8, 1>(tab, big, 64 << 10, out, "2 GiB, instruction in 64 KiB");
	run<1, 8, 1>(tab, big, 4 << 10, out, "2 GiB, instruction in 4 KiB");
	run<2, 8, 1>(tab, big, 4 << 10, out, "2 GiB, instruction in 4 KiB");
	run<1, 8, 2>(tab, big, 0, out, "2 GiB, pieces a tile stride apart");
	run<2, 8, 2>(tab, big, 0, out, "2 GiB, pieces a tile stride apart");
	run<1, 16, 0>(tab, big, 0, out, "table 2048 MiB");
	run<1, 2, 0>(tab, big, 0, out, "table 2048 MiB");
	run<1, 16, 1>(tab, big, 2 << 20, out, "2 GiB, instruction in 2 MiB");
	return 0;
}
