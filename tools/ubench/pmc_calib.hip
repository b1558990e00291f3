// tools/ubench/pmc_calib.hip -- calibration kernels for the SQ counters used in profiles/*_pmc_summary.txt.
// Each wave executes a known number of one kind of VALU instruction: 64 x 2000 of v_fma_f64 (full rate on
// gfx950: one wave instruction per 4 cycles), of v_rcp_f64 (quarter rate) or of v_xor_b32.  Collected with the same
// counter groups as the product kernels (tools/pmc4.sh), they show what the counters count:
//   SQ_INSTS_VALU per wave           = 128 000 for all three;
//   SQ_ACTIVE_INST_VALU per wave     = the time those instructions occupy the vector ALU, in units of FOUR cycles
//                                      (so it EQUALS SQ_INSTS_VALU for full-rate instructions -- fp64 FMA included --
//                                      and is ~4x larger for v_rcp_f64);
//   SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_BUSY_CU_CYCLES are in the same unit.
#include <hip/hip_runtime.h>
#include <cstdio>

template<int OP>
__global__ __launch_bounds__(256) void k_calib(int iters, double* out)
{
	double c[8];
	unsigned u[8];
	for (int j = 0; j < 8; j++) { c[j] = 1.0 + threadIdx.x * 1e-3 + j; u[j] = threadIdx.x + j; }
	const double m = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int j = 0; j < 8; j++)
			{
				if (OP == 0) c[j] = __builtin_fma(c[j], m, b);
				else if (OP == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(c[j]));
				else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(i));
			}
	}
	double r = 0.0;
	for (int j = 0; j < 8; j++) r += c[j] + u[j];
	if (r == 1.2345e300) out[threadIdx.x] = r;
}

template<int OP>
void run(const char* name, double* out)
{
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0);
	(void) hipEventCreate(&e1);
	k_calib<OP><<<512, 256>>>(2000, out); // two workgroups per CU: two waves per SIMD
	(void) hipDeviceSynchronize();
	(void) hipEventRecord(e0);
	k_calib<OP><<<512, 256>>>(2000, out);
	(void) hipEventRecord(e1);
	(void) hipEventSynchronize(e1);
	float ms;
	(void) hipEventElapsedTime(&ms, e0, e1);
	printf("%-8s 128000 instructions per wave, 2 waves per SIMD: %.3f ms = %.2f ns per instruction and SIMD\n", name, ms,
		ms * 1e6 / 256000.0);
}

int main()
{
	double* out;
	(void) hipMalloc(&out, 8192);
	run<0>("fma_f64", out);
	run<1>("rcp_f64", out);
	run<2>("xor_b32", out);
	return 0;
}
