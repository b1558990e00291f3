// tools/ubench/vgpr_banks.hip -- does the VGPR bank of the operands matter for v_fma_f64 on gfx950?
// A 64-bit operand v[2k:2k+1] lives in banks (0,1) when k is even and (2,3) when k is odd (bank = register index mod 4).
// Each kernel runs 64 x 2000 fp64 FMAs per wave with explicitly allocated registers:
//   spread : accumulator, multiplier and multiplicand in different bank pairs where possible
//   same   : all three source operands in the same bank pair
// 1, 2 and 4 waves per SIMD; reports ns (and cycles at 2.4 GHz) per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template<int CASE>
__global__ __launch_bounds__(1024) void k_bank(int iters, double* out)
{
	// accumulators v[0:1] .. v[14:15] (8 chains), operands chosen per case
	asm volatile(
		"v_mov_b32 v0, 0\n v_mov_b32 v1, 0x3ff00000\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n"
		"v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0x3ff00000\n"
		"v_mov_b32 v8, 0\n v_mov_b32 v9, 0x3ff00000\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n"
		"v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n"
		"v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n"
		"v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3e000000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3e000000\n"
		::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
		"v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
		{
			if (CASE == 0)
				// same: acc v[4k:4k+1] (banks 0,1), a = v[16:17] (0,1), b = v[20:21] (0,1)
				asm volatile(
					"v_fma_f64 v[0:1], v[16:17], v[20:21], v[0:1]\n v_fma_f64 v[4:5], v[16:17], v[20:21], v[4:5]\n"
					"v_fma_f64 v[8:9], v[16:17], v[20:21], v[8:9]\n v_fma_f64 v[12:13], v[16:17], v[20:21], v[12:13]\n"
					"v_fma_f64 v[0:1], v[16:17], v[20:21], v[0:1]\n v_fma_f64 v[4:5], v[16:17], v[20:21], v[4:5]\n"
					"v_fma_f64 v[8:9], v[16:17], v[20:21], v[8:9]\n v_fma_f64 v[12:13], v[16:17], v[20:21], v[12:13]\n"
					::: "v0", "v1", "v4", "v5", "v8", "v9", "v12", "v13");
			else if (CASE == 1)
				// two in one pair: acc (0,1), a (0,1), b v[22:23] (2,3)
				asm volatile(
					"v_fma_f64 v[0:1], v[16:17], v[22:23], v[0:1]\n v_fma_f64 v[4:5], v[16:17], v[22:23], v[4:5]\n"
					"v_fma_f64 v[8:9], v[16:17], v[22:23], v[8:9]\n v_fma_f64 v[12:13], v[16:17], v[22:23], v[12:13]\n"
					"v_fma_f64 v[0:1], v[16:17], v[22:23], v[0:1]\n v_fma_f64 v[4:5], v[16:17], v[22:23], v[4:5]\n"
					"v_fma_f64 v[8:9], v[16:17], v[22:23], v[8:9]\n v_fma_f64 v[12:13], v[16:17], v[22:23], v[12:13]\n"
					::: "v0", "v1", "v4", "v5", "v8", "v9", "v12", "v13");
			else
				// spread as far as two bank pairs allow: acc (0,1), a v[18:19] (2,3), b v[22:23] (2,3)
				asm volatile(
					"v_fma_f64 v[0:1], v[18:19], v[22:23], v[0:1]\n v_fma_f64 v[4:5], v[18:19], v[22:23], v[4:5]\n"
					"v_fma_f64 v[8:9], v[18:19], v[22:23], v[8:9]\n v_fma_f64 v[12:13], v[18:19], v[22:23], v[12:13]\n"
					"v_fma_f64 v[0:1], v[18:19], v[22:23], v[0:1]\n v_fma_f64 v[4:5], v[18:19], v[22:23], v[4:5]\n"
					"v_fma_f64 v[8:9], v[18:19], v[22:23], v[8:9]\n v_fma_f64 v[12:13], v[18:19], v[22:23], v[12:13]\n"
					::: "v0", "v1", "v4", "v5", "v8", "v9", "v12", "v13");
		}
	}
	double r;
	asm volatile("v_add_f64 %0, v[0:1], v[4:5]" : "=v"(r));
	if (r == 1.2345e300) out[threadIdx.x] = r;
}

template<int CASE>
void run(const char* name, int threads, double* out)
{
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0);
	(void) hipEventCreate(&e1);
	const int iters = 2000;
	k_bank<CASE><<<256, threads>>>(iters, out);
	(void) hipDeviceSynchronize();
	(void) hipEventRecord(e0);
	k_bank<CASE><<<256, threads>>>(iters, out);
	(void) hipEventRecord(e1);
	(void) hipEventSynchronize(e1);
	float ms;
	(void) hipEventElapsedTime(&ms, e0, e1);
	const double ops = 64.0 * iters, wps = threads / 256.0;
	printf("%-28s waves/SIMD %.0f: %.3f ms, %.2f ns = %.2f cycles (2.4 GHz) per instruction and SIMD\n", name, wps, ms,
		ms * 1e6 / (ops * wps), ms * 1e-3 * 2.4e9 / (ops * wps));
}

int main()
{
	double* out;
	(void) hipMalloc(&out, 8192);
	for (int threads : {256, 512, 1024})
	{
		run<0>("all three in one bank pair", threads, out);
		run<1>("two of three in one pair", threads, out);
		run<2>("acc apart from a and b", threads, out);
	}
	return 0;
}
