#!/usr/bin/env python3
"""tools/ubench/gen_stream_bench.py : builds tools/ubench/_build/stream_bench, a stand-alone timing harness
for the streaming kernels of r8b_kernels.hip (half-band cascade, 2x decimator, polynomial interpolator,
unfused whole-step interpolator).  Note: it launches the same buffers over and over, so read-heavy kernels see
part of their input in the 256 MB last-level cache (k_hbdown: 5.1 TB/s here, 4.0 in the product).

The kernel bodies are cut out of r8b_kernels.hip as they are (so the harness times the product's code) and
compiled with r8b_kernel_phases.h, without the convolver families -- seconds instead of the ten minutes
of the full library, which is what makes per-kernel experiments (`-D` flags, cycle stamps) affordable.
Usage: gen_stream_bench.py [extra hipcc flags...]; run the binary on the GPU box (no arguments).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "r8brain-free-src_amd", "csrc")
src = open(os.path.join(CSRC, "r8b_kernels.hip")).read()


def cut(start, end):
    a = src.index(start)
    return src[a:src.index(end, a)]


kernels = cut("// ------------------------------------------------------------------ whole-step polyphase FIR", "// ------------------------------------------------------------------ polynomial-interpolated bank") + cut("// positions of the (up to 64) outputs of tile t, lane = output, once per wave", "// ------------------------------------------------------------------ decimating half-band cascade")
main = r'''
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace r8bhip;

template<class F> static float time_ms(F f, int reps)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 5; i++) f();
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) f();
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	return ms / reps;
}

int main()
{
	const int nch = 1024;
	// ---- half-band cascade, cfg5 x 1024 channels: 5 stages, 2048 -> 65536 samples per channel and call
	{
		const long long in_per = 2048, out_per = 65536, calls = 20;
		double *x, *y, *ring;
		CK(hipMalloc(&x, sizeof(double) * nch * in_per * calls));
		CK(hipMalloc(&y, sizeof(double) * nch * out_per));
		CK(hipMalloc(&ring, sizeof(double) * nch * 4096));
		CK(hipMemset(x, 0, sizeof(double) * nch * in_per * calls));
		CK(hipMemset(ring, 0, sizeof(double) * nch * 4096));
		HBCascadeLaunch L = {};
		L.nst = 5;
		const int nt[5] = { 14, 8, 8, 4, 4 };
		for (int s = 0; s < 5; s++)
		{
			L.ntaps[s] = nt[s];
			for (int k = 0; k < 14; k++) L.taps[s][k] = k < nt[s] - 1 ? 0.3 / (k + 1) : 0.0;
		}
		hbc_fill_ranges(L);
		// the second call of a stream: inputs [2048, 4096), outputs [65536, 131072)
		L.a = out_per; L.b = 2 * out_per;
		L.nch = nch;
		L.in_end = 2 * in_per;
		L.pair_ok = 1;
		L.src.ring = ring; L.src.ring_stride = 4096; L.src.ring_mask = 4095;
		L.src.cur = x; L.src.cur_stride = in_per * calls; L.src.cur_base = 0; L.src.cur_fmt = 0;
		L.dst.p = y; L.dst.stride = out_per; L.dst.mask = -1; L.dst.off = -out_per; L.dst.fmt = 0;
		for (int tile : { 4096, 8192 })
		{
			L.tile = tile; L.buf = tile / 2 + 96; L.buf2 = tile / 4 + 96;
			const unsigned tiles = (unsigned) ((L.b - L.a + tile - 1) / tile);
			const size_t lds = (size_t) (L.buf + L.buf2 + 3 * kHbcSlack) * sizeof(double);
			CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hbcascade), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_hbcascade, dim3(tiles, nch), dim3(256), lds, 0, L); }, 50);
			CK(hipGetLastError());
			const double bytes = 8.0 * nch * (in_per + out_per);
			printf("k_hbcascade tile %d: %.4f ms  %.2f TB/s\n", tile, ms, bytes / ms * 1e-9);
		}
		// shallower runs (the heavy up-sampling chains of the rate table: 2 or 3 stages, taps 14 / 8 / 4 first)
		for (int nst : { 2, 3 })
		{
			HBCascadeLaunch M = L;
			M.nst = nst;
			hbc_fill_ranges(M);
			M.a = out_per; M.b = 2 * out_per; M.in_end = (2 * out_per) >> nst; M.nch = nch;
			M.src.cur_stride = in_per * calls; // (x holds 6144 samples per channel: only the first tiles' spans are real; timing only)
			M.tile = 8192; M.buf = M.tile / 2 + 96; M.buf2 = M.tile / 4 + 96;
			const unsigned tiles = (unsigned) ((M.b - M.a + M.tile - 1) / M.tile);
			const size_t lds = (size_t) (M.buf + M.buf2 + 3 * kHbcSlack) * sizeof(double);
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_hbcascade, dim3(tiles, nch), dim3(256), lds, 0, M); }, 50);
			CK(hipGetLastError());
			printf("k_hbcascade %d stages, tile 8192: %.4f ms  %.2f TB/s\n", nst, ms, 8.0 * nch * (out_per + (out_per >> nst)) / ms * 1e-9);
		}
		// the same run at BASELINE config 5's own size: 64 channels x 1024 -> 32768 outputs, 4096-sample tiles
		L.nch = 64; L.a = 32768; L.b = 65536; L.in_end = 2048; L.dst.off = -32768;
		for (int tile : { 1024, 2048, 4096, 8192 })
		{
			L.tile = tile; L.buf = L.tile / 2 + 96; L.buf2 = L.tile / 4 + 96;
			const size_t lds = (size_t) (L.buf + L.buf2 + 3 * kHbcSlack) * sizeof(double);
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_hbcascade, dim3(32768 / tile, 64), dim3(256), lds, 0, L); }, 200);
			CK(hipGetLastError());
			printf("k_hbcascade 64 ch x 32768, tile %d: %.4f ms\n", tile, ms);
		}
		hipFree(x); hipFree(y); hipFree(ring);
	}
	// ---- 2x decimator, 1024 channels x 16384 -> 8192
	{
		const long long in_per = 16384, out_per = 8192;
		double *x, *y, *ring;
		CK(hipMalloc(&x, sizeof(double) * nch * in_per * 2));
		CK(hipMalloc(&y, sizeof(double) * nch * out_per));
		CK(hipMalloc(&ring, sizeof(double) * nch * 4096));
		CK(hipMemset(x, 0, sizeof(double) * nch * in_per * 2));
		HBLaunch L = {};
		L.ntaps = 11;
		for (int k = 0; k < 11; k++) L.taps[k] = 0.3 / (k + 1);
		L.a = 4000; L.b = 4000 + out_per; L.nch = nch;
		L.src.ring = ring; L.src.ring_stride = 4096; L.src.ring_mask = 4095;
		L.src.cur = x; L.src.cur_stride = in_per * 2; L.src.cur_base = 0; L.src.cur_fmt = 0;
		L.dst.p = y; L.dst.stride = out_per; L.dst.mask = -1; L.dst.off = -4000; L.dst.fmt = 0;
		for (int tile : { 512, 1024, 2048, 4096 })
		{
			L.tile = tile;
			const unsigned tiles = (unsigned) ((out_per + L.tile - 1) / L.tile);
			const size_t lds = (size_t) hbdown_lds_doubles(L.tile, L.ntaps) * sizeof(double);
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_hbdown, dim3(tiles, nch), dim3(256), lds, 0, L); }, 50);
			CK(hipGetLastError());
			printf("k_hbdown 11 taps, tile %d: %.4f ms  %.2f TB/s\n", tile, ms, 8.0 * nch * (in_per + out_per) / ms * 1e-9);
		}
		hipFree(x); hipFree(y); hipFree(ring);
	}
	// ---- 2x up-sampler alone, 1024 channels x 32768 -> 65536 (8000 -> 32000's second stage)
	{
		const long long in_per = 32768 + 64, out_per = 65536;
		double *x, *y, *ring;
		CK(hipMalloc(&x, sizeof(double) * nch * in_per));
		CK(hipMalloc(&y, sizeof(double) * nch * (out_per + 2)));
		CK(hipMalloc(&ring, sizeof(double) * nch * 4096));
		CK(hipMemset(x, 0, sizeof(double) * nch * in_per));
		HBLaunch L = {};
		L.ntaps = 11;
		for (int k = 0; k < 11; k++) L.taps[k] = 0.3 / (k + 1);
		L.tile = 1024; L.nch = nch;
		L.src.ring = ring; L.src.ring_stride = 4096; L.src.ring_mask = 4095;
		L.src.cur = x; L.src.cur_stride = in_per; L.src.cur_base = 0; L.src.cur_fmt = 0;
		L.dst.p = y; L.dst.stride = out_per + 2; L.dst.mask = -1; L.dst.fmt = 0;
		for (int odd : { 0, 1 })
		{
			// outputs [40, 40 + out_per - 100) land at element 0 (aligned pairs) or 1 (odd offset: single stores)
			L.a = 40; L.b = 40 + out_per - 100; L.dst.off = -40 + odd;
			const long long nin = (L.b + 1) / 2 - L.a / 2;
			const unsigned tiles = (unsigned) ((nin + L.tile - 1) / L.tile);
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_hbup, dim3(tiles, nch), dim3(256), (size_t) (L.tile + 2 * L.ntaps) * sizeof(double), 0, L); }, 50);
			CK(hipGetLastError());
			printf("k_hbup 11 taps, offset %d: %.4f ms  %.2f TB/s\n", odd, ms, 8.0 * nch * 1.5 * (out_per - 100) / ms * 1e-9);
		}
		hipFree(x); hipFree(y); hipFree(ring);
	}
	// ---- polynomial-interpolated bank, 88200 -> 44101, 1024 channels x 32768 -> 16385
	{
		const long long in_per = 32768 + 256, out_per = 16385;
		const int flen = 24, fracs = 864;
		double *x, *y, *ring, *tab;
		CK(hipMalloc(&x, sizeof(double) * nch * in_per));
		CK(hipMalloc(&y, sizeof(double) * nch * out_per));
		CK(hipMalloc(&ring, sizeof(double) * nch * 4096));
		CK(hipMalloc(&tab, sizeof(double) * (fracs + 1) * flen * 3));
		CK(hipMemset(x, 0, sizeof(double) * nch * in_per));
		CK(hipMemset(tab, 0, sizeof(double) * (fracs + 1) * flen * 3));
		PolyLaunch L = {};
		L.flen = flen; L.fl2 = flen / 2; L.fll = flen / 2 - 1; L.fracs = fracs; L.table = tab;
		L.ssr = 88200.0; L.dsr = 44101.0;
		L.rpos0 = 100; L.fpos0 = 0.0; L.counter0 = 0; L.pos_int0 = 0; L.shift = 0.0;
		L.a = 0; L.b = out_per; L.nch = nch;
		L.span_max = 128 + flen + 4 + 8 + 1;
		L.pitch = L.span_max; while ((L.pitch & 31) != 1) L.pitch++;
		L.src.ring = ring; L.src.ring_stride = 4096; L.src.ring_mask = 4095;
		L.src.cur = x; L.src.cur_stride = in_per; L.src.cur_base = 0; L.src.cur_fmt = 0;
		L.dst.p = y; L.dst.stride = out_per; L.dst.mask = -1; L.dst.off = 0; L.dst.fmt = 0;
		const size_t lds = (size_t) poly_lds_doubles(L.pitch, flen) * sizeof(double);
		for (int k : { 0, 1 })
		{
			L.front = k;
			const dim3 grid((unsigned) ((out_per + kPolyTO - 1) / kPolyTO), (unsigned) ((nch + kPolyTC - 1) / kPolyTC));
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_poly_tiled, grid, dim3(256), lds, 0, L); }, 50);
			CK(hipGetLastError());
			printf("k_poly_tiled front %d: %.4f ms  %.2f TB/s (lds %zu)\n", k, ms, 8.0 * nch * (32768 + out_per) / ms * 1e-9, lds);
		}
		// a step that is not close to an integer (96000 -> 44111): time against the row pitch's residue mod 32
		L.ssr = 96000.0; L.dsr = 44111.0; L.front = 1;
		L.span_max = 140 + flen + 4 + 8;
		for (int res : { 1, 16 })
		{
			L.pitch = L.span_max; while ((L.pitch & 31) != res) L.pitch++;
			const size_t lds2 = (size_t) poly_lds_doubles(L.pitch, flen) * sizeof(double);
			const long long outs = 15000;
			L.b = outs;
			const dim3 grid((unsigned) ((outs + kPolyTO - 1) / kPolyTO), (unsigned) ((nch + kPolyTC - 1) / kPolyTC));
			const float ms = time_ms([&] { hipLaunchKernelGGL(k_poly_tiled, grid, dim3(256), lds2, 0, L); }, 20);
			CK(hipGetLastError());
			printf("k_poly_tiled step 2.176 pitch %d (res %d): %.4f ms\n", L.pitch, res, ms);
		}
		hipFree(x); hipFree(y); hipFree(ring); hipFree(tab);
	}
	// ---- whole-step polyphase FIR (unfused interpolator), 88200 -> 96000 (147/160, 24 taps), 1024 ch x 32768 -> 35666
	{
		const long long in_per = 32768 + 256, out_per = 35666;
		double *x, *y, *ring, *tab;
		CK(hipMalloc(&x, sizeof(double) * nch * in_per));
		CK(hipMalloc(&y, sizeof(double) * nch * out_per));
		CK(hipMalloc(&ring, sizeof(double) * nch * 4096));
		CK(hipMalloc(&tab, sizeof(double) * 160 * 24));
		CK(hipMemset(x, 0, sizeof(double) * nch * in_per));
		CK(hipMemset(tab, 0, sizeof(double) * 160 * 24));
		WholeLaunch L = {};
		L.in_step = 147; L.out_step = 160; L.flen = 24; L.fl2 = 12; L.fll = 11; L.pos0 = 0; L.table = tab;
		L.wtab = tab; L.inv_in = 123; // 147 * 123 = 18081 = 113 * 160 + 1
		L.a = 200; L.b = 200 + out_per - 400; L.tile = 1024; L.nch = nch;
		L.span_max = (int) ((long long) L.tile * 147 / 160) + 24 + 4 + 32;
		L.src.ring = ring; L.src.ring_stride = 4096; L.src.ring_mask = 4095;
		L.src.cur = x; L.src.cur_stride = in_per; L.src.cur_base = 0; L.src.cur_fmt = 0;
		L.dst.p = y; L.dst.stride = out_per; L.dst.mask = -1; L.dst.off = 0; L.dst.fmt = 0;
		for (int wt : { 0, 1 })
			for (int tile : { 1024, 2048, 4096 })
			{
				L.wtab = wt ? tab : nullptr; const int thr = 256;
				L.tile = tile;
				L.span_max = (int) ((long long) L.tile * 147 / 160) + 24 + 4 + 32;
				const unsigned tiles = (unsigned) ((L.b - L.a + L.tile - 1) / L.tile);
				const float ms = time_ms([&] { hipLaunchKernelGGL(k_whole, dim3(tiles, nch), dim3(thr), (size_t) L.span_max * sizeof(double), 0, L); }, 50);
				CK(hipGetLastError());
				printf("k_whole wtab %d, tile %d: %.4f ms  %.2f TB/s\n", L.wtab != nullptr, tile, ms, 8.0 * nch * (32768 + out_per) / ms * 1e-9);
			}
		hipFree(x); hipFree(y); hipFree(ring); hipFree(tab);
	}
	return 0;
}
'''
out_dir = os.path.join(ROOT, "tools", "ubench", "_build")
os.makedirs(out_dir, exist_ok=True)
path = os.path.join(out_dir, "stream_bench.hip")
open(path, "w").write('#include <hip/hip_runtime.h>\n#define R8B_NO_PCM_FUSE\n#define R8B_HD __device__ __forceinline__\n'
                      '#include "r8b_kernel_phases.h"\nnamespace r8bhip {\n' + kernels + "}\n" + main)
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-I" + CSRC, path,
       "-o", os.path.join(out_dir, "stream_bench")] + sys.argv[1:]
subprocess.run(cmd, check=True)
print("built", os.path.join(out_dir, "stream_bench"))
