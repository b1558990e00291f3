// micro-benchmark: issue rates of fp64 / int32 VALU instructions on gfx950 at 1, 2, 4 waves per SIMD,
// with 1..8 independent dependency chains per wave
#include <hip/hip_runtime.h>
#include <cstdio>
template<int CH, int OP>
__global__ void k(int iters, double* out)
{
	double c[CH];
	unsigned u[CH];
	for (int j = 0; j < CH; j++) { c[j] = threadIdx.x * 1e-3 + j; u[j] = threadIdx.x + j; }
	const double m = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 64 / CH; r++)
#pragma unroll
			for (int j = 0; j < CH; j++)
			{
				if (OP == 0) c[j] = __builtin_fma(c[j], m, b);
				else if (OP == 1) c[j] = c[j] + m;
				else if (OP == 2) c[j] = c[j] * m;
				else if (OP == 3) u[j] = (u[j] ^ (u[j] >> 4)) ;   // 2 int ops
			}
	}
	double r = 0.0;
	for (int j = 0; j < CH; j++) r += c[j] + u[j];
	if (r == 1.2345e300) out[threadIdx.x] = r;
}
template<int CH, int OP>
void run(const char* name, int threads, double* out)
{
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0);
	(void) hipEventCreate(&e1);
	const int iters = 2000;
	k<CH, OP><<<256, threads>>>(iters, out);
	(void) hipDeviceSynchronize();
	(void) hipEventRecord(e0);
	k<CH, OP><<<256, threads>>>(iters, out);
	(void) hipEventRecord(e1);
	(void) hipEventSynchronize(e1);
	float ms;
	(void) hipEventElapsedTime(&ms, e0, e1);
	const double ops = (OP == 3 ? 128.0 : 64.0) * iters; // instructions per wave
	const double waves_per_simd = threads / 256.0;
	printf("%-8s chains %d waves/SIMD %.0f: %.3f ms, %.2f cycles/instr/wave, SIMD cycles per instr %.2f (at 2.4 GHz)\n", name, CH,
		waves_per_simd, ms, ms * 1e-3 * 2.4e9 / ops, ms * 1e-3 * 2.4e9 / (ops * waves_per_simd));
}
int main()
{
	double* out;
	(void) hipMalloc(&out, 8192);
	for (int threads : {256, 512, 1024})
	{
		run<8, 0>("fma64", threads, out);
		run<4, 0>("fma64", threads, out);
		run<2, 0>("fma64", threads, out);
		run<1, 0>("fma64", threads, out);
		run<8, 1>("add64", threads, out);
		run<8, 2>("mul64", threads, out);
		run<8, 3>("int32", threads, out);
		run<1, 3>("int32", threads, out);
	}
	return 0;
}
