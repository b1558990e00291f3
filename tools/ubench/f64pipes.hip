// micro-benchmark: do v_mfma_f64_16x16x4_f64 and v_fma_f64 share an execution pipe on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 run role A, waves 4-7 run role B (waves w and w+4 sit on
// the same SIMD).  Roles: 0 idle, 1 MFMA f64 chain x4 accumulators, 2 FMA f64 x8 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int roleA, int roleB, int iters, double* out)
{
	const int wave = threadIdx.x >> 6;
	const int role = wave < 4 ? roleA : roleB;
	double r = 0.0;
	if (role == 1)
	{
		d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
		double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
		for (int i = 0; i < iters; i++)
		{
			a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
			a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
			a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
			a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
		}
		r = a0[0] + a1[1] + a2[2] + a3[3];
	}
	else if (role == 2)
	{
		double c[8];
		for (int j = 0; j < 8; j++) c[j] = threadIdx.x * 1e-3 + j;
		const double m = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
		for (int i = 0; i < iters; i++)
		{
#pragma unroll
			for (int u = 0; u < 8; u++)
#pragma unroll
				for (int j = 0; j < 8; j++) c[j] = __builtin_fma(c[j], m, b);
		}
		for (int j = 0; j < 8; j++) r += c[j];
	}
	else if (role == 3)
	{
		// dependent MFMA chain on ONE accumulator
		d4 a0 = {0, 0, 0, 0};
		double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
		for (int i = 0; i < iters; i++)
		{
			a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
			a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
			a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
			a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
		}
		r = a0[0];
	}
	if (r == 1.2345e300) out[threadIdx.x] = r;
}
int main()
{
	double* out;
	hipMalloc(&out, 4096);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 4000;
	const int cases[][2] = {{1, 0}, {2, 0}, {1, 2}, {1, 1}, {2, 2}, {3, 0}, {3, 2}};
	for (auto& c : cases)
	{
		k<<<256, 512>>>(c[0], c[1], iters, out);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		k<<<256, 512>>>(c[0], c[1], iters, out);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		// role 1: 4 MFMA per iter; role 2: 64 FMA per iter
		printf("roles %d,%d: %.3f ms  (per iter %.1f ns = %.0f cycles at 2.4 GHz)\n", c[0], c[1], ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
	}
	return 0;
}
