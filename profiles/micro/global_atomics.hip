// Micro-benchmark: what global atomics and random accesses cost on this chip, in the shapes the round-3 experiments ran into
// (the pruned pass as a streaming kernel, the row-order probe form).
//   hipcc --offload-arch=gfx950 -O3 -o global_atomics global_atomics.hip && ./global_atomics
// N operations by full waves (every lane one address), addresses from a multiplicative hash of the operation number:
//   returning / non-returning atomicAdd on K distinct words (packed, or one word per 128-byte line),
//   one atomicAdd per WORKGROUP on a single word (what a per-block "joined rows" counter does),
//   random 2-, 4- and 8-byte loads and 4-byte stores over a table of T bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ static inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool RETURNING>
__global__ void k_atomics(uint32_t *words, uint32_t k_mask, uint32_t stride, uint32_t n, uint32_t *sink)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	uint32_t *p = words + (size_t)(mix(i) & k_mask) * stride;
	if (RETURNING) {
		const uint32_t old = atomicAdd(p, 1u);
		if (old == 0xFFFFFFFFu)
			sink[0] = i;
	} else {
		__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

__global__ void k_one_word_per_block(uint32_t *word, uint32_t *sink)
{
	__shared__ uint32_t s;
	if (threadIdx.x == 0)
		s = 0;
	__syncthreads();
	atomicAdd(&s, 1u);
	__syncthreads();
	if (threadIdx.x == 0)
		atomicAdd(word, s);
}

template <typename T>
__global__ void k_random_loads(const T *tab, uint32_t mask, uint32_t n, unsigned long long *sink)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const T v = tab[mix(i) & mask];
	if (v == (T)0x5A5A5A5A)
		sink[0] = i;
}

__global__ void k_random_stores(uint32_t *tab, uint32_t mask, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		tab[mix(i) & mask] = i;
}

template <typename F>
static float timed(F f, int reps = 5)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	f();
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < reps; r++)
		f();
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	return ms / reps;
}

int main()
{
	const uint32_t N = 6250000;	/* (the survivors of variant D's pruned left table) */
	uint32_t *words, *sink;
	unsigned long long *sink64;
	hipMalloc(&words, (size_t)1 << 30);
	hipMalloc(&sink, 64);
	hipMalloc(&sink64, 64);
	hipMemset(words, 0, (size_t)1 << 30);
	const uint32_t grid = (N + 255) / 256;
	printf("%u operations by full waves\n", N);
	for (uint32_t k : { 1u << 12, 1u << 16, 1u << 20, 1u << 23 })
		for (uint32_t stride : { 1u, 32u }) {
			if ((size_t)k * stride * 4 > ((size_t)1 << 30))
				continue;
			const float a = timed([&] { k_atomics<true><<<grid, 256>>>(words, k - 1, stride, N, sink); });
			const float b = timed([&] { k_atomics<false><<<grid, 256>>>(words, k - 1, stride, N, sink); });
			printf("  atomicAdd on %8u words, %3u-byte apart: returning %.3f ms (%.1f ns per same-word add in a row), not returning %.3f ms\n", k, stride * 4, a,
			       a * 1e6 / ((double)N / k), b);
		}
	for (uint32_t blocks : { 4096u, 24414u, 100000u }) {
		const float a = timed([&] { k_one_word_per_block<<<blocks, 256>>>(words, sink); });
		printf("  one atomicAdd per workgroup on ONE word, %6u workgroups: %.3f ms\n", blocks, a);
	}
	for (uint32_t tbits : { 22u, 24u, 25u, 27u, 30u }) {
		const uint32_t bytes_mask = (1u << tbits) - 1u;
		const float l2 = timed([&] { k_random_loads<uint16_t><<<grid, 256>>>((const uint16_t *)words, bytes_mask >> 1, N, sink64); });
		const float l4 = timed([&] { k_random_loads<uint32_t><<<grid, 256>>>((const uint32_t *)words, bytes_mask >> 2, N, sink64); });
		const float l8 = timed([&] { k_random_loads<unsigned long long><<<grid, 256>>>((const unsigned long long *)words, bytes_mask >> 3, N, sink64); });
		const float s4 = timed([&] { k_random_stores<<<grid, 256>>>(words, bytes_mask >> 2, N); });
		printf("  table of %4u MiB: random 2-byte loads %.3f ms, 4-byte %.3f, 8-byte %.3f (%.1f G/s); random 4-byte stores %.3f ms\n", 1u << (tbits - 20), l2, l4, l8,
		       N / (l8 * 1e6), s4);
	}
	return 0;
}
