// tools/ubench/store_coalesce.hip -- how the vector-memory path of gfx950 treats 16-byte stores (and 8-byte loads)
// whose lanes cover a contiguous kilobyte but not in lane order.  The pair kernel's interpolator assigns phase
// pairs to lanes by LDS bank class, so consecutive lanes store 16-byte pieces far apart inside a 1280-byte group.
//   order 0: lane l -> slot l (natural)              order 1: quads of lanes contiguous, quads bit-reversed
//   order 2: slot = bit-reversed lane (every 16-byte piece on its own)   order 3: pairs of lanes contiguous
// Reports time per launch and effective store bandwidth; run under rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum for the
// number of write requests the L1 sends to L2.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int rev6(int x)
{
	return ((x & 1) << 5) | ((x & 2) << 3) | ((x & 4) << 1) | ((x & 8) >> 1) | ((x & 16) >> 3) | ((x & 32) >> 5);
}

template<int ORDER>
__global__ __launch_bounds__(256) void k_store(double2* out, int iters, double v)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int slot;
	if (ORDER == 0) slot = lane;
	else if (ORDER == 1) slot = (rev6(lane >> 2) >> 2) * 4 + (lane & 3);
	else if (ORDER == 2) slot = rev6(lane);
	else slot = (rev6(lane >> 1) >> 1) * 2 + (lane & 1);
	double2* p = out + ((size_t) blockIdx.x * 4 + wave) * (size_t) iters * 64 + slot;
	for (int i = 0; i < iters; i++)
	{
		double2 x;
		x.x = v + i;
		x.y = v - i;
		p[(size_t) i * 64] = x;
	}
}

template<int ORDER>
void run(const char* name, double2* out, int wgs, int iters)
{
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0);
	(void) hipEventCreate(&e1);
	k_store<ORDER><<<wgs, 256>>>(out, iters, 1.0);
	(void) hipDeviceSynchronize();
	(void) hipEventRecord(e0);
	for (int r = 0; r < 10; r++) k_store<ORDER><<<wgs, 256>>>(out, iters, 1.0 + r);
	(void) hipEventRecord(e1);
	(void) hipEventSynchronize(e1);
	float ms;
	(void) hipEventElapsedTime(&ms, e0, e1);
	ms /= 10;
	const double bytes = (double) wgs * 256 * iters * 16;
	printf("%-28s %.4f ms per launch, %.0f MB, %.2f TB/s\n", name, ms, bytes / 1e6, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
	const int wgs = 6318, iters = 11; // ~ the cfg2 launch: 6318 workgroups x 4 waves x ~11 KB each = 285 MB
	double2* out;
	(void) hipMalloc(&out, (size_t) wgs * 256 * iters * 16);
	run<0>("natural order", out, wgs, iters);
	run<1>("quads contiguous", out, wgs, iters);
	run<3>("lane pairs contiguous", out, wgs, iters);
	run<2>("every lane on its own", out, wgs, iters);
	run<0>("natural order", out, wgs, iters);
	return 0;
}
