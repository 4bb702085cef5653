// Micro-benchmark: throughput of LDS atomic wave-instructions at random slot addresses (the access
// pattern of the leaf hash table), 2 workgroups x 1024 threads per CU like k_leaf_group_count.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define SLOTS 3833u
#define ITERS 256
template <int OP>
__global__ __launch_bounds__(1024) void k(uint64_t *out, int dep)
{
	__shared__ unsigned long long s64[SLOTS];
	__shared__ uint32_t s32[SLOTS];
	for (uint32_t i = threadIdx.x; i < SLOTS; i += 1024) {
		s64[i] = 0;
		s32[i] = 0;
	}
	__syncthreads();
	uint64_t x = (uint64_t)(blockIdx.x * 1024 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	uint64_t acc = 0;
	for (int it = 0; it < ITERS; it++) {
		x ^= x >> 33;
		x *= 0xff51afd7ed558ccdull;
		x ^= x >> 29;
		uint32_t s = (uint32_t)(x >> 20) % SLOTS;
		if (dep)
			s = (s + (uint32_t)acc) % SLOTS;	/* next address depends on the previous result: latency chain */
		if (OP == 0)
			acc += atomicAdd(&s32[s], 1u);				/* ds_add_rtn_u32 */
		else if (OP == 1)
			acc += atomicCAS(&s32[s], 0u, (uint32_t)x | 1u);		/* ds_cmpst_rtn_b32 */
		else if (OP == 2)
			acc += atomicCAS(&s64[s], 0ull, x | 1ull);		/* ds_cmpst_rtn_b64 */
		else if (OP == 3)
			atomicAdd(&s64[s], 1ull);				/* ds_add_u64 (no return) */
		else if (OP == 4)
			atomicMin(&s32[s], (uint32_t)x);			/* ds_min_u32 (no return) */
		else if (OP == 5)
			acc += s64[s];						/* ds_read_b64 */
		else if (OP == 6)
			acc += atomicAdd(&s64[s], 1ull);			/* ds_add_rtn_u64 */
		else if (OP == 7)
			acc += x & 3;						/* no LDS: the loop itself */
	}
	if (acc == 0x1234567)
		out[0] = acc;
}
template <int OP>
static void run(const char *name, int dep)
{
	uint64_t *d;
	hipMalloc(&d, 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int grid = 512 * 8;
	k<OP><<<grid, 1024>>>(d, dep);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<OP><<<grid, 1024>>>(d, dep);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double wave_ops = (double)grid * 16 * ITERS;
	const double per_cu_cycles = ms * 1e-3 * 2.4e9 / (wave_ops / 256.0);
	printf("%-28s dep=%d  %.3f ms  %.1f cycles per wave-instruction per CU  (%.2f lane-ops/clk/CU)\n", name, dep, ms, per_cu_cycles,
	       64.0 / per_cu_cycles);
	hipFree(d);
}
int main()
{
	for (int dep = 0; dep < 2; dep++) {
		run<7>("loop only", dep);
		run<5>("ds_read_b64", dep);
		run<0>("ds_add_rtn_u32", dep);
		run<1>("ds_cmpst_rtn_b32", dep);
		run<2>("ds_cmpst_rtn_b64", dep);
		run<6>("ds_add_rtn_u64", dep);
		run<3>("ds_add_u64 (no rtn)", dep);
		run<4>("ds_min_u32 (no rtn)", dep);
	}
	return 0;
}
