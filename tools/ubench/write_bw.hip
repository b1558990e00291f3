// tools/ubench/write_bw.hip -- what HBM takes as pure WRITES: a store-only kernel (every wave instruction 1 KB of consecutive
// bytes, 16 bytes per lane, the pattern of k_hbcascade's last stage) over buffers of 128 MB ... 2 GB, ten launches each over
// the same buffer.  Below the 256 MB of the Infinity Cache a rewritten line never has to reach HBM; above it every launch's
// bytes do.  Beside it the same bytes as a copy (read + write).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_fill(double2* out, size_t n, double v)
{
	const size_t stride = (size_t) gridDim.x * 256;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
	{
		double2 x;
		x.x = v;
		x.y = v + 1.0;
		out[i] = x;
	}
}
__global__ __launch_bounds__(256) void k_copy(double2* out, const double2* in, size_t n)
{
	const size_t stride = (size_t) gridDim.x * 256;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

int main()
{
	double2 *a, *b;
	const size_t maxb = (size_t) 2048 << 20;
	(void) hipMalloc(&a, maxb);
	(void) hipMalloc(&b, maxb);
	(void) hipMemset(b, 0, maxb);
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0);
	(void) hipEventCreate(&e1);
	for (size_t mb : { 128, 256, 512, 1024, 2048 })
	{
		const size_t n = (mb << 20) / 16;
		for (int wgs : { 2048, 8192 })
		{
			float ms;
			k_fill<<<wgs, 256>>>(a, n, 1.0);
			(void) hipEventRecord(e0);
			for (int r = 0; r < 10; r++) k_fill<<<wgs, 256>>>(a, n, 1.0 + r);
			(void) hipEventRecord(e1);
			(void) hipEventSynchronize(e1);
			(void) hipEventElapsedTime(&ms, e0, e1);
			const double fill = (double) (mb << 20) / (ms / 10 * 1e-3) / 1e12;
			k_copy<<<wgs, 256>>>(a, b, n / 2);
			(void) hipEventRecord(e0);
			for (int r = 0; r < 10; r++) k_copy<<<wgs, 256>>>(a, b, n / 2);
			(void) hipEventRecord(e1);
			(void) hipEventSynchronize(e1);
			(void) hipEventElapsedTime(&ms, e0, e1);
			const double copy = (double) (mb << 20) / (ms / 10 * 1e-3) / 1e12;
			printf("%5zu MB, %5d workgroups: fill %.2f TB/s   copy (half read, half written) %.2f TB/s\n", mb, wgs, fill, copy);
		}
	}
	return 0;
}
