// micro-benchmark: a pure fp64 FMA loop on every SIMD for ~2.5 s per configuration while tools/attic/fma_power.sh
// samples rocm-smi: the shader clock and board power the chip sustains under fp64 vector load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template<int CH>
__global__ void k(int iters, double* out)
{
	double c[CH];
	for (int j = 0; j < CH; j++) c[j] = threadIdx.x * 1e-3 + j;
	const double m = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 64 / CH; r++)
#pragma unroll
			for (int j = 0; j < CH; j++) c[j] = __builtin_fma(c[j], m, b);
	}
	double r = 0.0;
	for (int j = 0; j < CH; j++) r += c[j];
	if (r == 1.2345e300) out[threadIdx.x] = r;
}
int main(int argc, char** argv)
{
	double* out;
	(void) hipMalloc(&out, 8192);
	for (int threads : {256, 512, 1024})
	{
		const int iters = 20000;
		auto t0 = std::chrono::steady_clock::now();
		int n = 0;
		double ms_sum = 0.0;
		while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5)
		{
			hipEvent_t e0, e1;
			(void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
			(void) hipEventRecord(e0);
			for (int r = 0; r < 10; r++) k<8><<<256, threads>>>(iters, out);
			(void) hipEventRecord(e1);
			(void) hipEventSynchronize(e1);
			float ms; (void) hipEventElapsedTime(&ms, e0, e1);
			ms_sum += ms; n += 10;
			(void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
		}
		const double instr_per_simd = 64.0 * iters * (threads / 256.0); // wave instructions per SIMD per launch
		const double ms = ms_sum / n;
		printf("fma64 waves/SIMD %d: %.3f ms per launch, %.2f ns per instruction per SIMD -> %.1f TFLOP/s\n", threads / 256, ms,
			ms * 1e6 / instr_per_simd, 256.0 * 4 * instr_per_simd * 64 * 2 / (ms * 1e-3) / 1e12);
		fflush(stdout);
	}
	return 0;
}
