// Micro-benchmark: read-only streaming bandwidth (sum of a 0.8 GB int64 array) for several launch geometries.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int THREADS, int ROUNDS>
__global__ __launch_bounds__(THREADS) void k(const ulonglong2 *__restrict__ p, uint64_t n2, unsigned long long *out)
{
	const uint64_t base = (uint64_t)blockIdx.x * THREADS * ROUNDS;
	unsigned long long acc = 0;
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint64_t i = base + (uint64_t)r * THREADS + threadIdx.x;
		if (i < n2) {
			const ulonglong2 q = p[i];
			acc += (q.x > 5) + (q.y > 5);
		}
	}
	const unsigned long long m = __ballot(acc & 1);
	if (threadIdx.x == 0 && m == 0x123456789ull)
		out[0] = acc;
}
template <int THREADS, int ROUNDS>
static void run(const ulonglong2 *d, uint64_t n2, unsigned long long *out)
{
	const uint32_t grid = (uint32_t)((n2 + (uint64_t)THREADS * ROUNDS - 1) / ((uint64_t)THREADS * ROUNDS));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	k<THREADS, ROUNDS><<<grid, THREADS>>>(d, n2, out);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int it = 0; it < 5; it++)
		k<THREADS, ROUNDS><<<grid, THREADS>>>(d, n2, out);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	ms /= 5;
	printf("threads %4d rounds %2d (%5d B/thread, %6u blocks): %.3f ms  %.0f GB/s\n", THREADS, ROUNDS, ROUNDS * 16, grid, ms,
	       n2 * 16 / (ms * 1e-3) / 1e9);
}
int main()
{
	const uint64_t n2 = 50000000;	// 0.8 GB
	ulonglong2 *d;
	unsigned long long *out;
	hipMalloc(&d, n2 * 16);
	hipMalloc(&out, 8);
	hipMemset(d, 1, n2 * 16);
	run<256, 1>(d, n2, out);
	run<256, 4>(d, n2, out);
	run<256, 8>(d, n2, out);
	run<256, 16>(d, n2, out);
	run<512, 4>(d, n2, out);
	run<512, 8>(d, n2, out);
	run<1024, 4>(d, n2, out);
	run<1024, 8>(d, n2, out);
	return 0;
}
