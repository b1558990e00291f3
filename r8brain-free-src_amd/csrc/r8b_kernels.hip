// r8b_kernels.hip -- gfx950 kernels and launchers of the batched resampler (generic, unfused
// stage kernels: one launch per stage per process() call).  The arithmetic of each phase lives in
// r8b_kernel_phases.h; this file only arranges phases, barriers, LDS and grids.
//
// Grid convention: blockIdx.y = channel, blockIdx.x = tile (FFT block / output tile) of that
// channel's stream.  All global accesses of a workgroup are contiguous runs of one channel's
// stream (coalesced 8-byte lanes); tables (kernel spectrum, twiddles, polyphase bank) are small
// and stay L2 resident.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>

// This file is compiled twice (Makefile).  As is: fp64 caller buffers only, the stage kernels carry
// no format test at all.  With R8B_PCM_VARIANT: src_load / dst_store also decode / encode planar PCM
// caller buffers in place (r8b_pcm_codec.h; the test costs the fp64 path 1-8 % when compiled in,
// hence the twin).  The fp64 object exports the launchers the engine calls and forwards to the
// *_pcm twins when a view carries a PCM format; helpers and the transposing PCM kernels live in
// the fp64 object only.
#ifdef R8B_PCM_VARIANT
#define R8B_LAUNCH(name) name##_pcm
#else
#define R8B_LAUNCH(name) name##_f64
#define R8B_NO_PCM_FUSE
#endif
// ... and, for the build's sake, in parts (Makefile: six compilations side by side instead of one of six minutes):
//   -DR8B_TU_PAIR=k  the pair kernels (r8b_convp.h) of part k only -- the geometry lists R8B_CONVP_GEOMS* come from the
//                    command line (with -DR8B_DEV_GEOMS) -- and their dispatcher launch_convp_part<k>_f64
//   -DR8B_TU_NOPAIR=n everything else; launch_convp_f64 tries the n (= 8) part dispatchers
//   neither          one object with everything (development builds: tools/variant.sh, tools/isa_dev.sh)
// The PCM twin carries the streaming kernels and the generic convolver only: a fast-path convolver at a PCM edge is
// fed through the staging rows (r8b_capi.cpp batch_process_pcm, Engine::pcm_fused_in / _out), so the hundreds of
// fast-path instantiations exist once.
#if defined(R8B_TU_PAIR)
#define R8B_HAS_REST 0
#define R8B_HAS_PAIR 1
#elif defined(R8B_TU_NOPAIR)
#define R8B_HAS_REST 1
#define R8B_HAS_PAIR 0
#else
#define R8B_HAS_REST 1
#define R8B_HAS_PAIR 1
#endif
#ifdef R8B_PCM_VARIANT
#define R8B_HAS_FAST 0
#else
#define R8B_HAS_FAST 1
#endif

#define R8B_HD __device__ __forceinline__
// r8b_convp.h: pass addresses as absolute LDS byte addresses (its arrays start on multiples of 256 bytes)
#define R8B_LDS_ABS 1
// 8 consecutive doubles from LDS as 8 separate ds_read_b64 (the compiler would pair them into
// ds_read2_b64, half the LDS rate); completion is awaited by the wait statement naming the
// registers (cdna_hip_programming.md 5.7, form ii)
#define R8B_LDS_READ8(v, o, a) \
	asm volatile("ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\t" \
		"ds_read_b64 %2, %8 offset:%11\n\tds_read_b64 %3, %8 offset:%12\n\t" \
		"ds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\t" \
		"ds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16" \
		: "=v"(v[o]), "=v"(v[o + 1]), "=v"(v[o + 2]), "=v"(v[o + 3]), "=v"(v[o + 4]), \
		  "=v"(v[o + 5]), "=v"(v[o + 6]), "=v"(v[o + 7]) \
		: "v"(a), "i"(8 * (o)), "i"(8 * (o) + 8), "i"(8 * (o) + 16), "i"(8 * (o) + 24), \
		  "i"(8 * (o) + 32), "i"(8 * (o) + 40), "i"(8 * (o) + 48), "i"(8 * (o) + 56) : "memory")
#ifndef R8B_STAGED_WAIT
#define R8B_STAGED_WAIT 1
#endif
#if R8B_STAGED_WAIT
// The reads are only issued here; R8B_LDS_ARRIVED(N, v, o) placed before the first use of
// v[o..o+7] waits until those eight have returned (LDS returns in order), so the multiply-adds on
// the first taps overlap the return of the later ones.
#define R8B_LDS_WINDOW(N, v, p) \
	{ \
		const unsigned a_ = (unsigned) (unsigned long long) (p); \
		R8B_LDS_READ8(v, 0, a_); R8B_LDS_READ8(v, 8, a_); R8B_LDS_READ8(v, 16, a_); \
		if constexpr ((N) > 24) R8B_LDS_READ8(v, 24, a_); \
	}
#define R8B_LDS_ARRIVED(N, v, o) \
	asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(v[o]), "+v"(v[(o) + 1]), "+v"(v[(o) + 2]), \
		"+v"(v[(o) + 3]), "+v"(v[(o) + 4]), "+v"(v[(o) + 5]), "+v"(v[(o) + 6]), "+v"(v[(o) + 7]) \
		: "n"((N) - 8 - (o) > 15 ? 15 : (N) - 8 - (o))) /* the counter has 4 bits */
#else
#define R8B_LDS_ARRIVED(N, v, o)
#define R8B_LDS_WINDOW(N, v, p) \
	{ \
		const unsigned a_ = (unsigned) (unsigned long long) (p); \
		R8B_LDS_READ8(v, 0, a_); R8B_LDS_READ8(v, 8, a_); R8B_LDS_READ8(v, 16, a_); \
		if constexpr ((N) > 24) R8B_LDS_READ8(v, 24, a_); \
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), \
			"+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), \
			"+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])); \
		asm volatile("" : "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]), "+v"(v[20]), \
			"+v"(v[21]), "+v"(v[22]), "+v"(v[23])); \
		if constexpr ((N) > 24) asm volatile("" : "+v"(v[(N) - 8]), "+v"(v[(N) - 7]), \
			"+v"(v[(N) - 6]), "+v"(v[(N) - 5]), "+v"(v[(N) - 4]), "+v"(v[(N) - 3]), \
			"+v"(v[(N) - 2]), "+v"(v[(N) - 1])); \
	}
#endif
#define R8B_FORCE4(a, b, c, d) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d))
#define R8B_OPAQUE2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// the interpolator's 16-byte output stores (r8b_convp.h R8B_OUT_STORE16): R8B_NT_STORE = 1 marks them non-temporal
// (streaming: the outputs are not read again by this kernel and need not displace tables and history in L2)
#ifndef R8B_NT_STORE
#define R8B_NT_STORE 0
#endif
#if R8B_NT_STORE
typedef double r8b_d2_t __attribute__((ext_vector_type(2)));
#define R8B_OUT_STORE16(ptr, v) { r8b_d2_t t_; t_.x = (v).re; t_.y = (v).im; \
	__builtin_nontemporal_store(t_, reinterpret_cast<r8b_d2_t*>(ptr)); }
#endif
#ifndef R8B_NO_STORE16U
struct __attribute__((aligned(8))) r8b_cd8_t { double re, im; };
#define R8B_OUT_STORE16U(ptr, v) { r8b_cd8_t t_; t_.re = (v).re; t_.im = (v).im; *reinterpret_cast<r8b_cd8_t*>(ptr) = t_; }
#define R8B_IN_LOAD16U(ptr, a, b) { const r8b_cd8_t t_ = *reinterpret_cast<const r8b_cd8_t*>(ptr); (a) = t_.re; (b) = t_.im; }
#endif
// nothing is scheduled across this point (no instruction is emitted)
#ifndef R8B_NO_SCHED_FENCE
#define R8B_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#include "r8b_kernel_phases.h"
#include "r8b_convx.h"
#include "r8b_convp.h"
#include "r8b_convq.h"
#include "r8b_pcm.h"

namespace r8bhip {

namespace {

void check(hipError_t e, const char* what)
{
	if (e != hipSuccess)
		throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// "k_convp_walk<11, 1, 4, 24>": a kernel template's instance as rocprofv3 names it (launch_symbol_note)
[[maybe_unused]] std::string symbol4(const char* base, int a, int b, int c, int d)
{
	return std::string(base) + "<" + std::to_string(a) + ", " + std::to_string(b) + ", " + std::to_string(c) + ", " +
		std::to_string(d) + ">";
}

// Opt-in to more than 64 KB of dynamic LDS.  Function attributes are per DEVICE, so the memo is keyed
// by (function, device ordinal): a process that drives several GPUs opts in on each of them.
void lds_opt_in(const void* fn, const char* what)
{
	static std::mutex mu;
	static std::set<std::pair<const void*, int>> done;
	int dev = 0;
	check(hipGetDevice(&dev), "hipGetDevice");
	std::lock_guard<std::mutex> lock(mu);
	if (done.count(std::make_pair(fn, dev))) return;
	check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), what);
	done.insert(std::make_pair(fn, dev));
}

#if R8B_HAS_REST
// ------------------------------------------------------------------ overlap-save block convolver
// one workgroup = one FFT block of one channel, everything between the global load of the input
// samples and the global store of the valid outputs stays in LDS
__global__ __launch_bounds__(256) void k_conv(const ConvLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double* const ra = reinterpret_cast<double*>(smem);
	cd* const za = reinterpret_cast<cd*>(smem);
	double* const rb = ra + (L.inplace ? 0 : L.n_in);
	cd* const zb = reinterpret_cast<cd*>(rb);
	const int tid = threadIdx.x, nthr = blockDim.x;
	const long long k = L.k0 + blockIdx.x;
	const int ch = blockIdx.y;

	conv_load(L, ra, k, ch, tid, nthr);
	__syncthreads();
	const int N = L.n_in / 2;
	int n = N;
	for (int p = 0; p < L.n_fwd; p++)
	{
		fft_pass(za, N, n, L.fwd_radix[p], false, L.tw, L.tw_len, tid, nthr);
		n /= L.fwd_radix[p];
		__syncthreads();
	}
	conv_spectral(L, za, zb, tid, nthr);
	__syncthreads();
	const int N2 = L.n_out / 2;
	n = 1;
	for (int p = 0; p < L.n_inv; p++)
	{
		n *= L.inv_radix[p];
		fft_pass(zb, N2, n, L.inv_radix[p], true, L.tw, L.tw_len, tid, nthr);
		__syncthreads();
	}
	conv_store(L, rb, k, ch, tid, nthr);
}

// ... blocks whose forward array does not fit LDS (the reference's 32768-point blocks in front of a decimation in the
// spectrum, which has to be taken on the reference's own block length -- r8b_plan.cpp).  The first radix-2 DIF stage of
// the forward transform is taken in the load, which leaves two independent sub-blocks of N / 2 = 8192 complex (128 KB):
// each is loaded, transformed and multiplied into its parity class of backward bins IN LDS, one after the other
// (r8b_kernel_phases.h conv_load_r2 / conv_spectral_half); only the packed backward spectrum (n_out doubles, one array
// per workgroup, which walks the launch's (block, channel) items) passes through global memory -- written in
// bit-reversed slots, read back as a run -- before the backward transform runs in the same LDS.  Global memory is
// coherent inside a workgroup across __syncthreads().
__global__ __launch_bounds__(512) void k_conv_big(const ConvLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	cd* const zl = reinterpret_cast<cd*>(smem);
	double* const rb = reinterpret_cast<double*>(smem);
	cd* const zb = reinterpret_cast<cd*>(smem);
	cd* const zg = reinterpret_cast<cd*>(L.work + (size_t) blockIdx.x * (size_t) L.n_out);
	const int tid = threadIdx.x, nthr = blockDim.x;
	const long long items = (long long) L.nblk * L.nch;
	const int Nh = L.n_in / 4, N2 = L.n_out / 2;
	for (long long it = blockIdx.x; it < items; it += gridDim.x)
	{
		// (a channel's blocks side by side: their windows overlap)
		const int ch = (int) (it / L.nblk);
		const long long k = L.k0 + (it - (long long) ch * L.nblk);
		for (int h = 0; h < 2; h++)
		{
			conv_load_r2(L, zl, k, ch, h, tid, nthr);
			__syncthreads();
			int n = Nh;
			for (int p = 1; p < L.n_fwd; p++)
			{
				fft_pass_sw<SwXor>(zl, Nh, n, L.fwd_radix[p], false, L.tw, L.tw_len, tid, nthr);
				n /= L.fwd_radix[p];
				__syncthreads();
			}
			conv_spectral_half(L, zl, zg, h, tid, nthr);
			__syncthreads();
		}
		// (both LDS arrays swizzled -- SwXor: the passes with short sub-lengths would otherwise put up to 32 lanes on
		// one bank group)
		for (int i = tid; i < N2; i += nthr) zb[SwXor::at(i)] = zg[i];
		__syncthreads();
		int n = 1;
		for (int p = 0; p < L.n_inv; p++)
		{
			n *= L.inv_radix[p];
			fft_pass_sw<SwXor>(zb, N2, n, L.inv_radix[p], true, L.tw, L.tw_len, tid, nthr);
			__syncthreads();
		}
		conv_store_sw<SwXor>(L, rb, k, ch, tid, nthr);
		__syncthreads();
	}
}

// ------------------------------------------------------------------ whole-step polyphase FIR
__global__ __launch_bounds__(256) void k_whole(const WholeLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double* const xs = reinterpret_cast<double*>(smem);
	const int tid = threadIdx.x, nthr = blockDim.x;
	const int ch = blockIdx.y;
	const long long j0 = L.a + (long long) blockIdx.x * L.tile;
	long long j1 = j0 + L.tile;
	if (j1 > L.b) j1 = L.b;
	long long lo;
	int len;
	whole_tile_span(L, j0, j1, &lo, &len);
	whole_load(L, xs, lo, len, ch, tid, nthr);
	__syncthreads();
	whole_compute(L, xs, lo, j0, j1, ch, tid, nthr);
}

// ------------------------------------------------------------------ polynomial-interpolated bank
__global__ __launch_bounds__(256) void k_poly(const PolyLaunch L)
{
	const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
	const int ch = blockIdx.y;
	if (L.a + i < L.b) dst_store(L.dst, ch, L.a + i, poly_one(L, ch, i));
}

// (four workgroups per CU: capping the registers at 128 costs a few spilled values and is 17 % faster than
// 137 registers at three waves per SIMD)
// positions of the (up to 64) outputs of tile t, lane = output, once per wave: the span's ends are lanes 0 and
// nout - 1 (read across the wave) -- one fp64 division sequence per wave and tile
struct PolyTilePos
{
	long long rpos, lo;
	double fpos;
	int len, nout;
};

__device__ __forceinline__ PolyTilePos poly_tile_positions(const PolyLaunch& L, long long t, long long n, int tid)
{
	PolyTilePos P;
	const long long i0 = t * kPolyTO;
	P.nout = (int) (n - i0 < kPolyTO ? n - i0 : kPolyTO);
	const int o = (tid & 63) < P.nout ? (tid & 63) : P.nout - 1;
	poly_position(L, i0 + o, &P.rpos, &P.fpos);
	const long long r0 = ((long long) __builtin_amdgcn_readfirstlane((int) (P.rpos >> 32)) << 32) |
		(unsigned) __builtin_amdgcn_readfirstlane((int) P.rpos);
	const long long r1 = ((long long) __builtin_amdgcn_readlane((int) (P.rpos >> 32), P.nout - 1) << 32) |
		(unsigned) __builtin_amdgcn_readlane((int) P.rpos, P.nout - 1);
	P.lo = r0 - L.fll; // (poly_tile_span)
	P.len = (int) (r1 + L.fl2 - P.lo + 1);
	return P;
}

__global__ __launch_bounds__(256, 4) void k_poly_tiled(const PolyLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double* const xs = reinterpret_cast<double*>(smem);
	const int tid = threadIdx.x, nthr = blockDim.x;
	const int ch0 = (int) blockIdx.y * kPolyTC;
	const long long n = L.b - L.a;
	const int pitch = L.pitch;
	double* const cf = xs + pitch * kPolyTC;
	double* const xoff = cf + kPolyTO * poly_cf_pitch(L.flen);
#ifdef R8B_POLY_STAMPS
	long long ts[8]; int nts = 0;
	ts[nts++] = clock64();
#define R8B_PSTAMP() ts[nts++] = clock64()
#else
#define R8B_PSTAMP()
#endif
	const long long t = blockIdx.x;
	const PolyTilePos P = poly_tile_positions(L, t, n, tid);
	R8B_PSTAMP();
	const long long i0 = t * kPolyTO, i1 = i0 + P.nout;
	if (L.front)
	{
		// samples and bank entries fetched together, one barrier (spans up to 16 lanes x kPolyNV registers)
		// (every lane holds the position of output lane of the tile: another output's is a shuffle away)
		const double fp = P.fpos;
		poly_tile_front(L, xs, pitch, cf, xoff, P.lo, P.len, P.nout, P.rpos, P.fpos,
			[fp](int o) { return __shfl(fp, o); }, ch0, tid, nthr);
		R8B_PSTAMP();
		R8B_PSTAMP();
		R8B_PSTAMP();
	}
	else
	{
		poly_tile_load(L, xs, pitch, P.lo, P.len, ch0, tid, nthr);
		if (tid < P.nout) poly_tile_pos_write(L, xoff, P.lo, tid, P.rpos, P.fpos);
		R8B_PSTAMP();
		__syncthreads();
		R8B_PSTAMP();
		poly_tile_coefs(L, cf, xoff, i0, i1, tid, nthr);
		R8B_PSTAMP();
	}
	__syncthreads();
	R8B_PSTAMP();
	poly_tile_compute(L, xs, pitch, cf, xoff, i0, i1, ch0, tid, nthr);
	R8B_PSTAMP();
#ifdef R8B_POLY_STAMPS
	if ((tid & 63) == 0 && blockIdx.y == 17 && blockIdx.x == 100)
		printf("poly stamps wave %d: pos %lld load %lld bar %lld coefs %lld bar %lld compute %lld\n", tid >> 6,
			ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5]);
#endif
}

// ------------------------------------------------------------------ half-band stages
__global__ __launch_bounds__(256) void k_hbup(const HBLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double* const xs = reinterpret_cast<double*>(smem);
	const int tid = threadIdx.x, nthr = blockDim.x;
	const int ch = blockIdx.y;
	const int T = L.ntaps;
	if (blockIdx.z != 0)
	{
		// (a carried history copy: HBLaunch::tail)
		tail_copy(L.tail, (int) blockIdx.x, (int) gridDim.x, ch, tid, nthr);
		return;
	}
	const long long nb = L.a / 2, ne = (L.b + 1) / 2;
	const long long n0 = nb + (long long) blockIdx.x * L.tile;
	long long n1 = n0 + L.tile;
	if (n1 > ne) n1 = ne;
	const long long lo = n0 - (T - 1);
	const int len = (int) (n1 - n0) + 2 * T - 1;
	{
		const SrcBlock sb = src_block(L.src, ch, lo);
		src_block_stage<4>(sb, xs, len, len, tid, nthr);
	}
	__syncthreads();
	hbup_compute(L, xs, n0, n1, ch, tid, nthr);
}

__global__ __launch_bounds__(256) void k_hbdown(const HBLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	double* const xs = reinterpret_cast<double*>(smem);
	const int tid = threadIdx.x, nthr = blockDim.x;
	const int ch = blockIdx.y;
	if (blockIdx.z != 0)
	{
		// (a carried history copy: HBLaunch::tail)
		tail_copy(L.tail, (int) blockIdx.x, (int) gridDim.x, ch, tid, nthr);
		return;
	}
	const long long n0 = L.a + (long long) blockIdx.x * L.tile;
	long long n1 = n0 + L.tile;
	if (n1 > L.b) n1 = L.b;
	hbdown_load(L, xs, n0, n1, ch, tid, nthr);
	__syncthreads();
	hbdown_compute(L, xs, n0, n1, ch, tid, nthr);
}

// ------------------------------------------------------------------ half-band cascade (cfg5: 5 stages, 32x)
__global__ __launch_bounds__(256) void k_hbcascade(const HBCascadeLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	// two buffers, alternating; the last stage's input (tile/2 samples) lands in the big one, so
	// the other never holds more than tile/4
	// (kHbcSlack free elements in front of each and behind the second: hbc_stage_t)
	double* const big = reinterpret_cast<double*>(smem) + kHbcSlack;
	double* const small = big + L.buf + kHbcSlack;
	const int tid = threadIdx.x, nthr = blockDim.x;
	const int ch = blockIdx.y;
	if (blockIdx.z != 0)
	{
		// (a carried history copy: HBLaunch::tail)
		tail_copy(L.tail, (int) blockIdx.x, (int) gridDim.x, ch, tid, nthr);
		return;
	}
	const long long q0 = L.a + (long long) blockIdx.x * L.tile;
	long long q1 = q0 + L.tile;
	if (q1 > L.b) q1 = L.b;
#ifdef R8B_HBC_STAMPS
	long long ts[12]; int nts = 0;
	ts[nts++] = clock64();
#endif
	double* xin = ((L.nst - 1) & 1) ? small : big;
	double* yout = ((L.nst - 1) & 1) ? big : small;
	{
		long long lo, hi;
		hbc_stage_range(L, q0, q1, L.nst, &lo, &hi);
		hbc_load_span(L, lo, hi, xin, ch, tid, nthr);
	}
	__syncthreads();
#ifdef R8B_HBC_STAMPS
	ts[nts++] = clock64();
#endif
	// Rolled: the stage number is a run-time value, what it indexes (tap counts, taps) are scalar loads of
	// kernel arguments, and a stage's range comes in closed form from the tile (hbc_stage_range) instead of a
	// table (a run-time index would push the table into scratch memory).  One copy of each stage routine instead of
	// one per stage: the unrolled form was five times the code (78 KB, beyond the instruction cache) and
	// slower.  Measured and not kept (tools/ubench/gen_stream_bench.py, cfg5 x 1024 channels, 0.138 ms as
	// built): tap counts / taps held in scalar registers across the loop or fetched a stage ahead (0.150 -
	// 0.161 ms: scalar register pressure, and scalar loads in flight make every LDS wait a full drain); a
	// workgroup walking 2 - 8 consecutive tiles with the next tile's input prefetched into registers (0.142 -
	// 0.187 ms: with three workgroups per CU the load of one already overlaps the stages of the others).
#pragma unroll 1
	for (int s = 0; s < L.nst; s++)
	{
		long long lo, hi;
		hbc_stage_range(L, q0, q1, s, &lo, &hi);
		const long long in_lo = floor_half(lo) - (L.ntaps[s] - 1);
		hbc_stage(L, s, xin, in_lo, lo, hi, yout, s + 1 == L.nst, ch, tid, nthr);
		__syncthreads();
#ifdef R8B_HBC_STAMPS
		if (nts < 12) ts[nts++] = clock64();
#endif
		double* t = xin;
		xin = yout;
		yout = t;
	}
#ifdef R8B_HBC_STAMPS
	if (tid == 0 && ch == 517 && blockIdx.x == 3) // (tools/ubench/gen_stream_bench.py -DR8B_HBC_STAMPS)
		printf("hbc stamps tile %d nst %d: load %lld st %lld %lld %lld %lld %lld\n", L.tile, L.nst,
			ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5]);
#endif
}

// ------------------------------------------------------------------ decimating half-band cascade
__global__ __launch_bounds__(256) void k_hbdcascade(const HBCascadeLaunch L)
{
	extern __shared__ __align__(16) unsigned char smem[];
	// inputs of the even stages live in the first buffer, of the odd stages in the second
	double* const even = reinterpret_cast<double*>(smem);
	double* const odd = even + L.buf;
	const int tid = threadIdx.x, nthr = blockDim.x;
	const int ch = blockIdx.y;
	if (blockIdx.z != 0)
	{
		// (a carried history copy: HBLaunch::tail)
		tail_copy(L.tail, (int) blockIdx.x, (int) gridDim.x, ch, tid, nthr);
		return;
	}
	const long long q0 = L.a + (long long) blockIdx.x * L.tile;
	long long q1 = q0 + L.tile;
	if (q1 > L.b) q1 = L.b;
	HBCRanges R;
	hbd_ranges(L, q0, q1, R);
	const int len = (int) (R.in_hi - R.in_lo);
	{
		const SrcBlock sb = src_block(L.src, ch, R.in_lo);
		const int end = clamp_rel(L.in_end - R.in_lo);
		src_block_stage<4>(sb, even, len, end, tid, nthr);
	}
	__syncthreads();
	long long in_lo = R.in_lo;
#pragma unroll
	for (int s = 0; s < kMaxCascade; s++)
	{
		if (s >= L.nst) break;
		hbd_stage(L, s, (s & 1) ? odd : even, in_lo, R.lo[s], R.hi[s], (s & 1) ? even : odd,
			s + 1 == L.nst, ch, tid, nthr);
		__syncthreads();
		in_lo = R.lo[s] - (s + 1 < L.nst ? L.skip[s] : 0);
	}
}

// ------------------------------------------------------------------ history tail of the caller's buffer
__global__ __launch_bounds__(256) void k_tail(const TailLaunch L)
{
	tail_copy(L, (int) blockIdx.x, (int) gridDim.x, (int) blockIdx.y, (int) threadIdx.x, (int) blockDim.x);
}

// ------------------------------------------------------------------ PCM ingest / egress (r8b_pcm.h)
#ifndef R8B_PCM_VARIANT
__global__ __launch_bounds__(256) void k_pcm_in(const PcmLaunch L)
{
	__shared__ double tile[kPcmTile * kPcmPitch];
	const long long f0 = (long long) blockIdx.x * kPcmTile;
	const int c0 = (int) blockIdx.y * kPcmTile;
	if (!L.interleaved)
	{
		pcm_in_direct(L, f0, c0, threadIdx.x, 256);
		return;
	}
	pcm_in_gather(L, tile, f0, c0, threadIdx.x, 256);
	__syncthreads();
	pcm_in_scatter(L, tile, f0, c0, threadIdx.x, 256);
}

__global__ __launch_bounds__(256) void k_pcm_out(const PcmLaunch L)
{
	__shared__ double tile[kPcmTile * kPcmPitch];
	const long long f0 = (long long) blockIdx.x * kPcmTile;
	const int c0 = (int) blockIdx.y * kPcmTile;
	if (!L.interleaved)
	{
		pcm_out_direct(L, f0, c0, threadIdx.x, 256);
		return;
	}
	pcm_out_gather(L, tile, f0, c0, threadIdx.x, 256);
	__syncthreads();
	pcm_out_scatter(L, tile, f0, c0, threadIdx.x, 256);
}

// planar buffers: a workgroup per (chunk of a row, channel)
__global__ __launch_bounds__(256) void k_pcm_rows_in(const PcmLaunch L)
{
	pcm_row_in(L, (long long) blockIdx.x * kPcmRowChunk, (int) blockIdx.y, threadIdx.x, 256);
}

__global__ __launch_bounds__(256) void k_pcm_rows_out(const PcmLaunch L)
{
	pcm_row_out(L, (long long) blockIdx.x * kPcmRowChunk, (int) blockIdx.y, threadIdx.x, 256);
}
#endif

#endif // R8B_HAS_REST

// ------------------------------------------------------------------ fast path (r8b_convx.h)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory,
// i.e. drains vmcnt before every barrier; the phases exchange data through LDS exclusively, and
// the table fetches a phase issues for the NEXT phase must stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#if R8B_HAS_REST && R8B_HAS_FAST
template<int LOGN, int UPLOG>
struct GpuExec
{
	ConvxState<LOGN, UPLOG> st;
	unsigned rot = 0; // per-workgroup rotation of the logical wave roles
	// two wave-local steps in one phase: the second reads LDS words written by other lanes of the
	// SAME wave in the first; LDS serves a wave's accesses in issue order, so only the compiler
	// must be kept from reordering them
	template<class FA, class FB>
	__device__ __forceinline__ void wave_phase2(FA fa, FB fb)
	{
		const int ltid = (int) ((threadIdx.x + 64u * rot) & (kConvxThreads - 1));
		fa(ltid, st);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
		fb(ltid, st);
		lds_barrier();
	}
	template<class F>
	__device__ __forceinline__ void each(F f) // no barrier
	{
		f((int) threadIdx.x, st);
	}
	template<class F>
	__device__ __forceinline__ void phase(F f)
	{
		// Logical thread id rotated by whole waves per workgroup: the phases that keep only the
		// first one or two logical waves busy (radix-16 passes, interpolation) then land on
		// different SIMDs for the workgroups sharing a CU (measured +3.5 %).
		f((int) ((threadIdx.x + 64u * rot) & (kConvxThreads - 1)), st);
		lds_barrier();
	}
};

template<int LOGN, int UPLOG, int MODE, int FLENP>
#ifndef R8B_CONVX_MINWAVES
#define R8B_CONVX_MINWAVES 3
#endif
// (the 8192 / 4096-point arrays leave room for one / two workgroups per CU anyway: no point in
// capping the registers below that)
__global__ __launch_bounds__(kConvxThreads, (LOGN + (UPLOG > 0 ? UPLOG : 0) >= 13 ? 1 :
	(LOGN + (UPLOG > 0 ? UPLOG : 0) == 12 ? 2 : R8B_CONVX_MINWAVES))) void k_convx(const ConvxLaunch X)
{
	extern __shared__ __align__(16) unsigned char smem[];
	// XCD-aware work mapping: the dispatcher hands consecutive workgroup ids to the 8 XCDs round
	// robin, each XCD with its own L2.  Consecutive blocks of a channel share PrevInputLen input
	// samples (overlap-save), so channel c is served by XCD c % 8 and an XCD walks the blocks of
	// one channel back to back: the overlap is re-read from that XCD's L2 instead of HBM.
	const unsigned w = blockIdx.x, nblk = (unsigned) X.c.nblk, nch = (unsigned) X.c.nch;
	unsigned blk, ch;
	if ((nch & 7u) == 0)
	{
		const unsigned i = w >> 3;
		blk = i % nblk;
		ch = ((i / nblk) << 3) + (w & 7u);
	}
	else
	{
		blk = w % nblk;
		ch = w / nblk;
	}
	// wave-uniform by construction; tell the compiler (keeps block/channel arithmetic on the SALU)
	blk = (unsigned) __builtin_amdgcn_readfirstlane((int) blk);
	ch = (unsigned) __builtin_amdgcn_readfirstlane((int) ch);
	GpuExec<LOGN, UPLOG> ex;
	ex.rot = (blk + ch) & 3u;
	convx_body<LOGN, UPLOG, MODE, FLENP>(ex, X, reinterpret_cast<double*>(smem), X.c.k0 + blk, (int) ch);
}

#endif // R8B_HAS_REST && R8B_HAS_FAST

#if R8B_HAS_PAIR && R8B_HAS_FAST
// ------------------------------------------------------------------ fast path, pair form (r8b_convp.h)
#if defined(R8B_CP_STAMPS) || defined(R8B_TIMELINE)
// (development builds only -- tools/stamps_probe.py, tools/timeline_probe.py: per-phase cycle stamps / where and when every
// workgroup ran; the shipped library compiles none of it)
#include "r8b_dev_probes.h"
#endif
template<int LN, int UL>
struct GpuExecP
{
	ConvpState<LN, UL> st;
	int tid_ = (int) threadIdx.x;
	unsigned* flags_; // one word per wave behind the array (r8b_convp.h kConvpFlagBytes)
	unsigned* lv_;    // ... and the blocks' level words behind them (kConvpLevelWords)
	// (abytes: the array's size -- the half-array form's is another, r8b_convp.h convp_ha_array_bytes)
	__device__ __forceinline__ explicit GpuExecP(unsigned char* smem, int abytes = convp_array_bytes<LN, UL>())
		: flags_(reinterpret_cast<unsigned*>(smem + abytes)), lv_(flags_ + 16) {}
	// workgroup-wide OR of a small bit set: every thread posts before a barrier, anybody collects after it
	__device__ __forceinline__ void post_bits(int, unsigned v)
	{
		const unsigned w = (__builtin_amdgcn_ballot_w64((v & 1u) != 0) != 0 ? 1u : 0u) |
			(__builtin_amdgcn_ballot_w64((v & 2u) != 0) != 0 ? 2u : 0u);
		if ((threadIdx.x & 63u) == 0) flags_[threadIdx.x >> 6] = w;
	}
	__device__ __forceinline__ unsigned collect_bits() const
	{
		unsigned r = 0;
#pragma unroll
		for (int w = 0; w < ConvpGeom<LN, UL>::WT / 64; w++) r |= flags_[w];
		return (unsigned) __builtin_amdgcn_readfirstlane((int) r);
	}
	// The levels of a block's two channels (r8b_convp.h cp_level_words), combined over the block's threads with a
	// maximum per channel: every thread posts before a barrier, anybody collects after it.  A block of whole waves: inside
	// a wave by DPP -- lane pairs, quads (quad_perm), the two quads of eight lanes (row_half_mirror), the two halves of a
	// row (row_mirror) --, across its four rows through scalar registers, one packed word per wave; several blocks in a
	// wave (short transforms): lane exchanges over the block's lanes, one word per block.
	template<int CTRL>
	static __device__ __forceinline__ unsigned lv_dpp_max(unsigned v)
	{
		const unsigned o = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) v, CTRL, 0xf, 0xf, false);
		return o > v ? o : v;
	}
	static __device__ __forceinline__ unsigned lv_wave_max(unsigned v)
	{
		v = lv_dpp_max<0xB1>(v);   // quad_perm [1, 0, 3, 2]
		v = lv_dpp_max<0x4E>(v);   // quad_perm [2, 3, 0, 1]
		v = lv_dpp_max<0x141>(v);  // row_half_mirror
		v = lv_dpp_max<0x140>(v);  // row_mirror
		const unsigned r0 = (unsigned) __builtin_amdgcn_readlane((int) v, 0), r1 = (unsigned) __builtin_amdgcn_readlane((int) v, 16);
		const unsigned r2 = (unsigned) __builtin_amdgcn_readlane((int) v, 32), r3 = (unsigned) __builtin_amdgcn_readlane((int) v, 48);
		const unsigned m0 = r0 > r1 ? r0 : r1, m1 = r2 > r3 ? r2 : r3;
		return m0 > m1 ? m0 : m1;
	}
	__device__ __forceinline__ void post_levels(int tid, int sub, CpLevels v)
	{
		typedef ConvpGeom<LN, UL> G;
		if constexpr (G::NT >= 64)
		{
			CpLevels w;
			w.a = lv_wave_max(v.a);
			w.b = lv_wave_max(v.b);
			if ((threadIdx.x & 63u) == 0) lv_[threadIdx.x >> 6] = cp_level_pack(w);
		}
		else
		{
			unsigned u = cp_level_pack(v);
#pragma unroll
			for (int m = 1; m < G::NT; m <<= 1) u = cp_level_max(u, (unsigned) __shfl_xor((int) u, m, 64));
			if ((tid & (G::NT - 1)) == 0) lv_[sub] = u;
		}
	}
	__device__ __forceinline__ unsigned collect_levels(int sub) const
	{
		typedef ConvpGeom<LN, UL> G;
		if constexpr (G::NT >= 64)
		{
			constexpr int NWB = G::NT / 64; // waves per block
			unsigned r = 0;
#pragma unroll
			for (int w = 0; w < NWB; w++) r = cp_level_max(r, lv_[sub * NWB + w]);
			// (one block per workgroup: the same in every lane)
			if constexpr (G::SUB == 1) r = (unsigned) __builtin_amdgcn_readfirstlane((int) r);
			return r;
		}
		else return lv_[sub];
	}
	// the block's level shift (cp_level_shift), left in a word of its own by the first pass for the body's last phases
	// (workgroup barriers lie in between)
	__device__ __forceinline__ void post_shift(int, int sub, int lt, int d)
	{
		if (lt == 0) lv_[kConvpLevelWords + sub] = (unsigned) d;
	}
	__device__ __forceinline__ int collect_shift(int sub) const
	{
		const int d = (int) lv_[kConvpLevelWords + sub];
		if constexpr (ConvpGeom<LN, UL>::SUB == 1) return __builtin_amdgcn_readfirstlane(d);
		else return d;
	}
	// a value that is the same in every lane, kept in a vector register across phases: back to the scalar unit
	__device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
	// steps that exchange data between the lanes of ONE wave only (r8b_convp.h: forward passes 1..,
	// middle pass, first backward pass): LDS serves a wave's accesses in issue order, so between the
	// steps only the compiler must be kept from reordering them; a workgroup barrier ends the sequence
	__device__ __forceinline__ void wave_sync()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
	}
#ifdef R8B_CP_STAMPS
	// (development: cycle stamps of eight workgroups' phases -- before / after each barrier; written straight to
	// global memory so that the stamped workgroups keep their registers and the others pay one scalar branch)
	int nts_ = 0;
	__device__ __forceinline__ void stamp()
	{
		if (blockIdx.x >= R8B_CP_STAMPS && blockIdx.x < R8B_CP_STAMPS + 8 && (threadIdx.x & 63) == 0)
		{
			long long* o = g_cp_stamps + ((blockIdx.x - R8B_CP_STAMPS) * 8 + (threadIdx.x >> 6)) * 32;
			nts_++;
			o[nts_] = clock64();
			o[0] = nts_;
		}
	}
	__device__ void dump() const {}
	__device__ __forceinline__ void stamp2() { stamp(); }
#else
	__device__ __forceinline__ void stamp2() {}
	__device__ __forceinline__ void stamp() {}
	__device__ __forceinline__ void dump() const {}
#endif
	template<class F0, class... F>
	__device__ __forceinline__ void wave_steps(F0 f0, F... f)
	{
		f0(tid_, st);
		((wave_sync(), f(tid_, st)), ...);
		stamp();
		lds_barrier();
		stamp();
	}
	// (walk form: nothing to reset -- the flag words are rewritten per block.  The thread index is made opaque once per
	// block: everything a phase derives from it -- LDS addresses, table offsets, dozens of values -- is loop invariant,
	// and hoisted out of the block loop it would stay live across ALL phases: 150 registers more than the one-block
	// kernel needs, i.e. spills)
	__device__ __forceinline__ void next_block() { asm volatile("" : "+v"(tid_)); }
	template<class F>
	__device__ __forceinline__ void each(F f) // no barrier
	{
		f(tid_, st);
		stamp();
	}
	template<class F>
	__device__ __forceinline__ void phase(F f)
	{
		f(tid_, st);
		stamp();
		lds_barrier();
		stamp();
	}
};

// the launch descriptor's hot scalars pinned in scalar registers (k_convp, below, says why)
template<int MODE>
__device__ __forceinline__ void convp_pin(ConvxLaunch& H, const ConvxLaunch& X)
{
	H.c = X.c;
	H.in_step = X.in_step; H.out_step = X.out_step; H.flen = X.flen; H.fl2w = X.fl2w; H.fllw = X.fllw;
	H.table = X.table; H.wtab = X.wtab; H.wa = X.wa; H.wb = X.wb; H.wdst = X.wdst;
	H.run_off = X.run_off; H.ptab = X.ptab; H.ctab = X.ctab; H.nsets = X.nsets;
	H.nblk_magic = X.nblk_magic;
	H.park_n = X.park_n; H.park_out = X.park_out; H.park_slices = X.park_slices;
	H.walk = X.walk;
	// (integers are made opaque -- "+s" --, pointers are only USED here -- "s" --: a pointer that went through an asm
	// output loses its kernel-argument provenance and would be dereferenced with flat instructions)
	asm volatile("" : "+s"(H.c.k0), "+s"(H.c.blk_stride), "+s"(H.c.blk_offset), "+s"(H.c.in_len), "+s"(H.c.fl2),
		"+s"(H.c.up_pow2), "+s"(H.c.src.cur_stride), "+s"(H.c.src.cur_base), "+s"(H.c.src.cur_fmt), "+s"(H.c.rot),
		"+s"(H.c.fl2r), "+s"(H.c.tail_flags), "+s"(H.c.tail_bf)
		: "s"(H.c.src.cur), "s"(H.c.hp), "s"(H.c.ptw));
	if constexpr (MODE == 4 || MODE == 5 || MODE == 16 || MODE == 17 || MODE == 23 || MODE == 25 || MODE == 29 || MODE == 30 ||
		MODE == 33)
		asm volatile("" : "+s"(H.run_off), "+s"(H.in_step), "+s"(H.out_step), "+s"(H.nsets), "+s"(H.wdst.stride),
			"+s"(H.wdst.mask), "+s"(H.wdst.off), "+s"(H.wdst.fmt), "+s"(H.park_n), "+s"(H.park_out), "+s"(H.park_slices),
			"+s"(H.c.t_zero)
			: "s"(H.ptab), "s"(H.ctab), "s"(H.wdst.p));
	else if constexpr (MODE == 18)
		// (one-channel form + one phase per thread: what its last two phases read)
		asm volatile("" : "+s"(H.in_step), "+s"(H.out_step), "+s"(H.flen), "+s"(H.wdst.stride), "+s"(H.wdst.mask),
			"+s"(H.wdst.off), "+s"(H.wdst.fmt), "+s"(H.park_n), "+s"(H.park_out), "+s"(H.park_slices), "+s"(H.c.t_zero)
			: "s"(H.wtab), "s"(H.wdst.p));
	else if constexpr (MODE == 1) {}
	else
		asm volatile("" : "+s"(H.c.a), "+s"(H.c.b), "+s"(H.c.dst.stride), "+s"(H.c.dst.mask), "+s"(H.c.dst.off),
			"+s"(H.c.dst.fmt), "+s"(H.c.down), "+s"(H.c.down_pow2), "+s"(H.c.up) : "s"(H.c.dst.p));
}

// 64 KB of LDS per workgroup: two workgroups per CU, i.e. two waves per SIMD and 256 registers each
#ifndef R8B_DEV_MINBLK
#define R8B_DEV_MINBLK 2 // (development builds: the register budget of the 256-thread kernels cut for more workgroups per CU)
#endif
#ifndef R8B_SPLIT_MINBLOCKS
#define R8B_SPLIT_MINBLOCKS 3 // (R8B_SPLIT_UP2 development builds: workgroups per CU the register budget is cut for)
#endif
template<int LN, int UL, int MODE, int FLENP>
__global__ __launch_bounds__((ConvpGeom<LN, UL>::WT), (convp_mode_ha(MODE) ? convp_ha_minblocks(MODE, (ConvpGeom<LN, UL>::WT)) : kSplit<LN, UL> ? R8B_SPLIT_MINBLOCKS : (ConvpGeom<LN, UL>::WT) > 256 ? 1 : R8B_DEV_MINBLK)) void k_convp(const ConvxLaunch X)
{
	extern __shared__ __align__(256) unsigned char smem_[];
	// (development builds, the control of the occupancy experiment with a truncated array -- R8B_FAKE_LDS --: the array
	// starts R8B_DEV_LDS_SHIFT bytes into a FULL-size allocation, so the same share of its accesses falls outside the
	// allocation while the workgroups per CU stay what they are; timing only, results are wrong)
#ifdef R8B_DEV_LDS_SHIFT
	unsigned char* const smem = smem_ + R8B_DEV_LDS_SHIFT;
#else
	unsigned char* const smem = smem_;
#endif
#ifdef R8B_TIMELINE
	const long long tl_t0 = (long long) __builtin_readcyclecounter();
#endif
	// XCD-aware mapping as in k_convx, over channel PAIRS and groups of SUB consecutive blocks: item w of a launch
	// is block group bg of channel pair pr
	constexpr int SUB = ConvpGeom<LN, UL>::SUB;
	// (the one-channel form -- convp_mode_solo --: an item is a channel, not a pair)
	constexpr bool SOLO = convp_mode_solo(MODE);
	const unsigned npair = SOLO ? (unsigned) X.c.nch : ((unsigned) X.c.nch + 1u) >> 1;
	const unsigned nbg = ((unsigned) X.c.nblk + SUB - 1u) / SUB;
	auto decode = [&](unsigned wi, unsigned& bg, unsigned& pr)
	{
		if (nbg == 1)
		{
			bg = 0;
			pr = wi;
		}
		else if ((npair & 7u) == 0)
		{
			const unsigned i = wi >> 3, qd = convp_div(i, X.nblk_magic);
			bg = i - qd * nbg;
			pr = (qd << 3) + (wi & 7u);
		}
		else
		{
			pr = convp_div(wi, X.nblk_magic);
			bg = wi - pr * nbg;
		}
		bg = (unsigned) __builtin_amdgcn_readfirstlane((int) bg);
		pr = (unsigned) __builtin_amdgcn_readfirstlane((int) pr);
	};
	GpuExecP<LN, UL> ex(smem, convp_mode_array_bytes<LN, UL, MODE>());
	// Kernel arguments live in memory: left to itself the compiler fetches each one where it is first needed --
	// chains of dependent scalar loads at the start of the workgroup (measured: 3 300 cycles before the first sample
	// load is issued) and one more load in front of most phases, whose wait is a wait on the LDS counter too, i.e. a
	// full drain of the phase's LDS traffic.  The scalars the phases use are therefore copied into a local descriptor
	// and pinned in scalar registers HERE, all loads in flight together and one wait; the phases read the copy.
	// (Pointers to rarely used data -- the history ring of a call's first blocks -- stay in memory.)
	ConvxLaunch H;
	convp_pin<MODE>(H, X);
	ex.stamp();
	// (One workgroup per item.  Persistent workgroups that take their items from per-XCD work queues -- no start-up
	// between items -- were built and measured 25-30 % SLOWER, as was a 512-thread workgroup carrying two block pairs
	// in step: DESIGN.md section 5; the code is in the history of this file, the round-3 commits "Pair kernel: persistent workgroups on per-XCD work queues" ... "twin-block experiment".)
	unsigned bg, pr;
	decode(blockIdx.x, bg, pr);
	const int chA = SOLO ? (int) pr : (int) (2u * pr);
	const bool bvalid = !SOLO && chA + 1 < X.c.nch;
	ConvpItem cur;
	const int b0 = (int) bg * SUB;
	cur.k = X.c.k0 + b0;
	cur.nvalid = X.c.nblk - b0 < SUB ? X.c.nblk - b0 : SUB;
	cur.chA = chA;
	cur.chB = bvalid ? chA + 1 : chA;
	cur.bvalid = bvalid;
	// (MODE 1 -- one phase per thread -- already fills the scalar file with its span bookkeeping: it reads the
	// arguments where it needs them, as before)
	if constexpr (MODE == 1) convp_body<LN, UL, MODE, FLENP>(ex, X, X, reinterpret_cast<cd*>(smem), cur);
	else convp_body<LN, UL, MODE, FLENP>(ex, H, X, reinterpret_cast<cd*>(smem), cur);
#ifdef R8B_TIMELINE
	if (threadIdx.x == 0 && blockIdx.x < 16384)
	{
		unsigned hwid, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
		long long* o = g_timeline + 4 * (size_t) blockIdx.x;
		o[0] = tl_t0;
		o[1] = (long long) __builtin_readcyclecounter();
		o[2] = (long long) hwid | ((long long) xcc << 32);
		o[3] = X.c.k0; // (first block of the launch: tells the launches apart)
	}
#endif
#ifdef R8B_CP_STAMPS
	// (how long the workgroup's last stores take to be acknowledged: the slot stays occupied until then)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	ex.stamp();
#endif
	ex.dump();
}

// Walk form of the fused two-phase modes (r8b_convp.h convp_walk, convp_walk_range): ONE launch, two kinds of workgroups.
// The first nslice x npair take the call's INTERIOR blocks [walk_i0, walk_i1) -- workgroup wi blocks [slice walk_len,
// (slice + 1) walk_len) of them for channel pair wi mod npair -- on the lean walk body; the others one EDGE block each (a
// call's first block, whose window reaches back into the history ring, and its last ones, which own the history tail
// and the parked outputs) on the general body, exactly as k_convp runs it.  Consecutive workgroups are consecutive
// pairs in both parts, so a pair's blocks (whose windows overlap) meet in one XCD's L2.  The walk workgroups come
// first in the grid: they are resident before any edge workgroup is dispatched and all end together.
template<int LN, int UL, int MODE, int FLENP>
__global__ __launch_bounds__((ConvpGeom<LN, UL>::WT), 2) void k_convp_walk(const ConvxLaunch X)
{
	extern __shared__ __align__(256) unsigned char smem[];
	if constexpr (convp_walk_ok<LN, UL, MODE>())
	{
		GpuExecP<LN, UL> ex(smem);
		ConvxLaunch H;
		convp_pin<MODE>(H, X);
		const unsigned npair = ((unsigned) X.c.nch + 1u) >> 1;
		const unsigned nwi = (unsigned) (X.walk_i1 - X.walk_i0);
		const unsigned nslice = (nwi + (unsigned) X.walk_len - 1u) / (unsigned) X.walk_len;
		const unsigned nwalk = nslice * npair;
		ConvpItem cur;
		cur.nvalid = 1;
		if (blockIdx.x < nwalk)
		{
			unsigned slice = blockIdx.x / npair, pr = blockIdx.x - slice * npair;
			slice = (unsigned) __builtin_amdgcn_readfirstlane((int) slice);
			pr = (unsigned) __builtin_amdgcn_readfirstlane((int) pr);
			const int chA = (int) (2u * pr);
			const bool bvalid = chA + 1 < X.c.nch;
			const int b0 = (int) slice * X.walk_len;
			cur.k = X.c.k0 + X.walk_i0 + b0;
			cur.chA = chA;
			cur.chB = bvalid ? chA + 1 : chA;
			cur.bvalid = bvalid;
			const int left = (int) nwi - b0;
			convp_walk<LN, UL, MODE, FLENP>(ex, H, X, reinterpret_cast<cd*>(smem), cur, left < X.walk_len ? left : X.walk_len);
		}
		else
		{
			const unsigned e = blockIdx.x - nwalk;
			unsigned j = e / npair, pr = e - j * npair;
			j = (unsigned) __builtin_amdgcn_readfirstlane((int) j);
			pr = (unsigned) __builtin_amdgcn_readfirstlane((int) pr);
			const int blk = (int) j < X.walk_i0 ? (int) j : X.walk_i1 + ((int) j - X.walk_i0);
			const int chA = (int) (2u * pr);
			const bool bvalid = chA + 1 < X.c.nch;
			cur.k = X.c.k0 + blk;
			cur.chA = chA;
			cur.chB = bvalid ? chA + 1 : chA;
			cur.bvalid = bvalid;
			convp_body<LN, UL, MODE, FLENP>(ex, H, X, reinterpret_cast<cd*>(smem), cur);
		}
	}
}

// ---- eight elements per thread (r8b_convq.h): the 2048 -> 4096-point block pair on 512 threads
struct GpuExecQ
{
	ConvqState st;
	int tid_ = (int) threadIdx.x;
	unsigned* flags_;
	unsigned* lv_;
	__device__ __forceinline__ explicit GpuExecQ(unsigned char* smem)
		: flags_(reinterpret_cast<unsigned*>(smem + kConvqN2 * 16)), lv_(flags_ + 16) {}
	__device__ __forceinline__ void post_bits(int, unsigned v)
	{
		const unsigned w = (__builtin_amdgcn_ballot_w64((v & 1u) != 0) != 0 ? 1u : 0u) |
			(__builtin_amdgcn_ballot_w64((v & 2u) != 0) != 0 ? 2u : 0u);
		if ((threadIdx.x & 63u) == 0) flags_[threadIdx.x >> 6] = w;
	}
	__device__ __forceinline__ unsigned collect_bits() const
	{
		unsigned r = 0;
#pragma unroll
		for (int w = 0; w < kConvqThreads / 64; w++) r |= flags_[w];
		return (unsigned) __builtin_amdgcn_readfirstlane((int) r);
	}
	__device__ __forceinline__ void post_levels(int, int, CpLevels v)
	{
		CpLevels w;
		w.a = GpuExecP<11, 1>::lv_wave_max(v.a);
		w.b = GpuExecP<11, 1>::lv_wave_max(v.b);
		if ((threadIdx.x & 63u) == 0) lv_[threadIdx.x >> 6] = cp_level_pack(w);
	}
	__device__ __forceinline__ unsigned collect_levels(int) const
	{
		unsigned r = 0;
#pragma unroll
		for (int w = 0; w < kConvqThreads / 64; w++) r = cp_level_max(r, lv_[w]);
		return (unsigned) __builtin_amdgcn_readfirstlane((int) r);
	}
	__device__ __forceinline__ void post_shift(int, int, int lt, int d)
	{
		if (lt == 0) lv_[kConvpLevelWords] = (unsigned) d;
	}
	__device__ __forceinline__ int collect_shift(int) const
	{
		return __builtin_amdgcn_readfirstlane((int) lv_[kConvpLevelWords]);
	}
	__device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
	__device__ __forceinline__ void wave_sync()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
	}
	template<class F0, class... F>
	__device__ __forceinline__ void wave_steps(F0 f0, F... f)
	{
		f0(tid_, st);
		((wave_sync(), f(tid_, st)), ...);
		lds_barrier();
	}
	template<class F>
	__device__ __forceinline__ void each(F f) { f(tid_, st); }
	template<class F>
	__device__ __forceinline__ void phase(F f)
	{
		f(tid_, st);
		lds_barrier();
	}
};

// one workgroup per block pair, the workgroup map of k_convp (pair-major over the launch's blocks, XCD-interleaved)
__global__ __launch_bounds__(kConvqThreads, 2) void k_convq(const ConvxLaunch X)
{
	extern __shared__ __align__(256) unsigned char smem[];
	const unsigned npair = ((unsigned) X.c.nch + 1u) >> 1, nbg = (unsigned) X.c.nblk;
	unsigned bg, pr;
	const unsigned wi = blockIdx.x;
	if (nbg == 1) { bg = 0; pr = wi; }
	else if ((npair & 7u) == 0)
	{
		const unsigned i = wi >> 3, qd = convp_div(i, X.nblk_magic);
		bg = i - qd * nbg;
		pr = (qd << 3) + (wi & 7u);
	}
	else
	{
		pr = convp_div(wi, X.nblk_magic);
		bg = wi - pr * nbg;
	}
	bg = (unsigned) __builtin_amdgcn_readfirstlane((int) bg);
	pr = (unsigned) __builtin_amdgcn_readfirstlane((int) pr);
	GpuExecQ ex(smem);
	ConvxLaunch H;
	convp_pin<0>(H, X);
	ConvpItem cur;
	cur.k = X.c.k0 + (int) bg;
	cur.nvalid = 1;
	cur.chA = (int) (2u * pr);
	cur.bvalid = cur.chA + 1 < X.c.nch;
	cur.chB = cur.bvalid ? cur.chA + 1 : cur.chA;
	convq_body(ex, H, X, reinterpret_cast<cd*>(smem), cur);
}

template<int LN, int UL, int MODE, int FLENP>
void launch_convp_t(const ConvxLaunch& X0, hipStream_t stream)
{
	{
	ConvxLaunch X = X0;
	// (one block group: floor(2^32 / 1) + 1 does not fit; 0 makes convp_div return 0, handled by the kernel)
	constexpr unsigned SUB = ConvpGeom<LN, UL>::SUB;
	const unsigned nbg = ((unsigned) X.c.nblk + SUB - 1u) / SUB;
	X.nblk_magic = nbg > 1 ? (unsigned) (0x100000000ull / nbg) + 1u : 0u;
	{
		// convp_div(i, magic) is floor(i / nbg) only while i * nbg < 2^32; i runs up to the grid size (an eighth of
		// it in the XCD-interleaved mapping).  Far out of reach of audio batches -- tens of millions of input
		// samples per call and channel pair --, refused rather than mapped wrongly
		const unsigned long long np = convp_mode_solo(MODE) ? (unsigned long long) X.c.nch :
			((unsigned long long) X.c.nch + 1ull) >> 1;
		const unsigned long long imax = (np & 7ull) == 0 ? (np >> 3) * nbg : np * nbg;
		if (nbg > 1 && imax * nbg >= 0x100000000ull)
			throw std::runtime_error("launch_convp: too many blocks per call for the workgroup map (split the call)");
	}
	convp_prepare<LN, UL>(X, MODE != 1 && MODE != 18, convp_mode_sp(MODE), convp_mode_solo(MODE), convp_mode_p3(MODE));
	auto kern = k_convp<LN, UL, MODE, FLENP>;
	size_t lds = (size_t) convp_lds_bytes<LN, UL>();
	// (mode 21: the half-array form -- four workgroups per CU)
	if constexpr (convp_mode_ha(MODE)) lds = (size_t) convp_ha_lds_bytes<LN, UL, MODE>();
	// (mode 20: the half-band front stages its raw samples over the array and what lies behind it)
	if constexpr (MODE == 20) lds = lds > (size_t) kHbfLdsBytes ? lds : (size_t) kHbfLdsBytes;
#ifdef R8B_DEV_ONLY_MODE
	// (development builds: occupancy experiments with a truncated array -- timing only, results are wrong)
	if (const char* e = getenv("R8B_FAKE_LDS")) lds = (size_t) atoi(e);
#endif
	lds_opt_in(reinterpret_cast<const void*>(kern), "hipFuncSetAttribute(k_convp)");
	const unsigned npair = convp_mode_solo(MODE) ? (unsigned) X.c.nch : ((unsigned) X.c.nch + 1u) >> 1;
	unsigned grid = nbg * npair;
	if constexpr (LN == 12 && UL == 0 && MODE == 5)
	{
		// ... of the 4096 -> 4096-point 1:1 geometry (kernel mode 33: both transforms' exchanges by parts; BASELINE's cfg3)
		if (X.half_fused != 0 && convp_ha_fused_fits(X.run_off, X.c.in_len, X.in_step))
		{
			launch_convp_t<LN, UL, 33, FLENP>(X0, stream);
			return;
		}
	}
	if constexpr (LN == 11 && UL == 1 && (MODE == 16 || MODE == 17))
	{
		// ... with a complex kernel spectrum (kernel modes 29 / 30: minimum-phase chains)
		if (X.half_fused != 0 && convp_ha_fused_fits(X.run_off, X.c.in_len, X.in_step))
		{
			launch_convp_t<LN, UL, MODE == 16 ? 29 : 30, FLENP>(X0, stream);
			return;
		}
	}
	if constexpr ((LN == 11 || LN == 12) && UL == 1 && (MODE == 6 || MODE == 7))
	{
		// ... convolver-only (kernel modes 31 / 32)
		if (X.half != 0)
		{
			launch_convp_t<LN, UL, MODE == 6 ? 31 : 32, FLENP>(X0, stream);
			return;
		}
	}
	if constexpr (LN == 11 && UL == 1 && (MODE == 4 || MODE == 5))
	{
		// half-array form with the whole-step interpolator fused in (kernel modes 23 / 25): 49 KB, three workgroups per CU;
		// in place of mode 4 / 5 and of the walk form
		if (X.half_fused != 0 && convp_ha_fused_fits(X.run_off, X.c.in_len, X.in_step))
		{
			launch_convp_t<LN, UL, MODE == 4 ? 23 : 25, FLENP>(X0, stream);
			return;
		}
	}
	if constexpr (convp_walk_ok<LN, UL, MODE>())
	{
		// (X.walk: the engine allows the walk form, at most that many blocks per workgroup; the launch's interior blocks
		// decide whether it is taken)
		int i0 = 0, i1 = 0;
		if (X.walk > 0 && convp_walk_range<LN, UL>(X, &i0, &i1) && i1 - i0 >= 2)
		{
			X.walk_i0 = i0;
			X.walk_i1 = i1;
			X.walk_len = X.walk < i1 - i0 ? X.walk : i1 - i0;
			auto wkern = k_convp_walk<LN, UL, MODE, FLENP>;
			lds_opt_in(reinterpret_cast<const void*>(wkern), "hipFuncSetAttribute(k_convp_walk)");
			const unsigned nwi = (unsigned) (i1 - i0), nslice = (nwi + (unsigned) X.walk_len - 1u) / (unsigned) X.walk_len;
			hipLaunchKernelGGL(wkern, dim3((nslice + (unsigned) X.c.nblk - nwi) * npair), dim3(ConvpGeom<LN, UL>::WT), lds,
				stream, X);
			check(hipGetLastError(), "launch k_convp_walk");
			static const std::string wsym = symbol4("k_convp_walk", LN, UL, MODE, FLENP);
			launch_symbol_note(wsym.c_str());
			launch_walk_blocks_add((long long) nwi);
			return;
		}
	}
	if constexpr (LN == 12 && UL == -1 && (MODE == 0 || MODE == 3))
	{
		// ... of the 4096 -> 2048-point decimating geometry (kernel modes 27 / 28: the FORWARD transform's exchanges by parts)
		if (X.half != 0)
		{
			launch_convp_t<LN, UL, MODE == 0 ? 27 : 28, FLENP>(X0, stream);
			return;
		}
	}
	if constexpr ((LN == 11 || LN == 12) && UL == 1 && (MODE == 0 || MODE == 3))
	{
		// half-array form (r8b_convp.h cp_ha_*, kernel modes 21 / 22): the same block pair in 32 KB of LDS, four workgroups
		// per CU (4096 -> 8192 points: 64 KB, two workgroups of 512 threads)
		if (X.half != 0 && X.quad == 0)
		{
			launch_convp_t<LN, UL, MODE == 0 ? 21 : 22, FLENP>(X0, stream);
			return;
		}
	}
	if constexpr (LN == 11 && UL == 1 && MODE == 0)
	{
		// eight elements per thread (r8b_convq.h): the same work on 512 threads per block pair, four waves per SIMD
		if (X.quad != 0)
		{
			lds_opt_in(reinterpret_cast<const void*>(k_convq), "hipFuncSetAttribute(k_convq)");
			hipLaunchKernelGGL(k_convq, dim3(grid), dim3(kConvqThreads), (size_t) convq_lds_bytes(), stream, X);
			check(hipGetLastError(), "launch k_convq");
			launch_symbol_note("k_convq");
			return;
		}
	}
	hipLaunchKernelGGL(kern, dim3(grid), dim3(ConvpGeom<LN, UL>::WT), lds, stream, X);
	check(hipGetLastError(), "launch k_convp");
	static const std::string sym = symbol4("k_convp", LN, UL, MODE, FLENP);
	launch_symbol_note(sym.c_str());
	}
}

// (the half-band front, mode 20: the 4096 -> 2048-point decimating geometry)
template<int LN, int DL>
void launch_convp_hbf(const ConvxLaunch& X, hipStream_t stream)
{
	if constexpr (LN == 12 && DL == 1) launch_convp_t<LN, -DL, 20, 24>(X, stream);
	else throw std::runtime_error("launch_convp: half-band front on a geometry it is not built for");
}

// (the polyphase 3x form: 1:1 geometries of 1024 ... 4096 points)
template<int LN, int UL>
void launch_convp_p3(const ConvxLaunch& X, hipStream_t stream)
{
	if constexpr (UL == 0 && LN >= 10 && LN <= 12) launch_convp_t<LN, UL, 19, 24>(X, stream);
	else throw std::runtime_error("launch_convp: polyphase 3x form on a geometry it is not built for");
}

// (the one-channel form decimating by 2: geometry <13, -1>)
template<int LN, int DL>
void launch_convp_solo_down(const ConvxLaunch& X, int mode, hipStream_t stream)
{
	if constexpr (LN == 13 && DL == 1)
	{
		if (mode == 10) launch_convp_t<LN, -DL, 10, 24>(X, stream);
		else if (mode == 11) launch_convp_t<LN, -DL, 11, 24>(X, stream);
		else if (mode == 14) launch_convp_t<LN, -DL, 14, 24>(X, stream);
		else launch_convp_t<LN, -DL, 15, 24>(X, stream);
	}
	// (decimating by 4: real spectra only)
	if constexpr (LN == 13 && DL == 2)
	{
		if (mode == 10) launch_convp_t<LN, -DL, 10, 24>(X, stream);
		else launch_convp_t<LN, -DL, 11, 24>(X, stream);
	}
}

template<int LN, int UL>
void launch_convp_sp(const ConvxLaunch& X, int mode, hipStream_t stream)
{
	if constexpr (LN == 13 && UL == 0)
	{
		if (mode == 8) launch_convp_t<LN, UL, 8, 24>(X, stream);
		else if (mode == 9) launch_convp_t<LN, UL, 9, 24>(X, stream);
		else if (mode == 10) launch_convp_t<LN, UL, 10, 24>(X, stream);
		else if (mode == 11) launch_convp_t<LN, UL, 11, 24>(X, stream);
		else if (mode == 18 && X.flen > 24) launch_convp_t<LN, UL, 18, 32>(X, stream);
		else if (mode == 18) launch_convp_t<LN, UL, 18, 24>(X, stream);
		else if (mode == 12) launch_convp_t<LN, UL, 12, 24>(X, stream);
		else if (mode == 13) launch_convp_t<LN, UL, 13, 24>(X, stream);
		else if (mode == 14) launch_convp_t<LN, UL, 14, 24>(X, stream);
		else launch_convp_t<LN, UL, 15, 24>(X, stream);
	}
}

#endif // R8B_HAS_PAIR && R8B_HAS_FAST

#if R8B_HAS_REST && R8B_HAS_FAST
template<int LOGN, int UPLOG, int MODE, int FLENP>
void launch_convx_t(const ConvxLaunch& X, hipStream_t stream)
{
	auto kern = k_convx<LOGN, UPLOG, MODE, FLENP>;
	// work array; the linear output run (in_len + kConvxRunPad doubles) aliases its start
	const size_t lds = (size_t) convx_lds_need(UPLOG > 0 ? LOGN + UPLOG : LOGN, X.c.in_len, MODE) * sizeof(double);
	lds_opt_in(reinterpret_cast<const void*>(kern), "hipFuncSetAttribute(k_convx)");
	hipLaunchKernelGGL(kern, dim3((unsigned) X.c.nblk * (unsigned) X.c.nch), dim3(kConvxThreads),
		lds, stream, X);
	check(hipGetLastError(), "launch k_convx");
	static const std::string sym = symbol4("k_convx", LOGN, UPLOG, MODE, FLENP);
	launch_symbol_note(sym.c_str());
}

#endif // R8B_HAS_REST && R8B_HAS_FAST

#if R8B_HAS_REST
void set_lds_attrs()
{
	lds_opt_in(reinterpret_cast<const void*>(k_conv), "hipFuncSetAttribute(k_conv)");
	lds_opt_in(reinterpret_cast<const void*>(k_conv_big), "hipFuncSetAttribute(k_conv_big)");
	lds_opt_in(reinterpret_cast<const void*>(k_whole), "hipFuncSetAttribute(k_whole)");
	lds_opt_in(reinterpret_cast<const void*>(k_hbdcascade), "hipFuncSetAttribute(k_hbdcascade)");
}
#endif // R8B_HAS_REST

} // namespace

#if R8B_HAS_REST
void R8B_LAUNCH(launch_conv)(const ConvLaunch& L, void* stream)
{
	set_lds_attrs();
	if (L.work != nullptr)
	{
		// (LDS: a forward sub-block of n_in / 4 complex, then the backward array of n_out doubles)
		const size_t lds_big = (size_t) std::max(L.n_in / 2, L.n_out) * sizeof(double);
		hipLaunchKernelGGL(k_conv_big, dim3((unsigned) L.work_slots), dim3((unsigned) L.threads),
			lds_big, (hipStream_t) stream, L);
		check(hipGetLastError(), "launch k_conv_big");
		launch_symbol_note("k_conv_big");
		return;
	}
	const size_t lds = (size_t) (L.inplace ? L.n_in : L.n_in + L.n_out) * sizeof(double);
	hipLaunchKernelGGL(k_conv, dim3((unsigned) L.nblk, (unsigned) L.nch), dim3((unsigned) L.threads),
		lds, (hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_conv");
	launch_symbol_note("k_conv");
}

void R8B_LAUNCH(launch_whole)(const WholeLaunch& L, void* stream)
{
	set_lds_attrs();
	const long long n = L.b - L.a;
	const unsigned tiles = (unsigned) ((n + L.tile - 1) / L.tile);
	// A thread keeps one phase (whole_compute_t): no more waves than the phases fill (147 phases: three waves, not four
	// with one idle but for its share of the staging -- 96000 -> 44100: 0.0838 -> 0.0777 ms; MORE waves, whole sets of
	// phases on 448 / 512 threads, took 0.1256: profiles/r04_experiments.txt)
	// (fewer phases than that: as many whole sets of phases as 256 threads hold, rounded up to whole waves -- a small
	// Out keeps its four waves, which also share the tile's staging)
	unsigned nthr = 256;
	if (L.out_step <= 256) nthr = ((256u / (unsigned) L.out_step) * (unsigned) L.out_step + 63u) & ~63u;
	hipLaunchKernelGGL(k_whole, dim3(tiles, (unsigned) L.nch), dim3(nthr),
		(size_t) L.span_max * sizeof(double), (hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_whole");
	launch_symbol_note("k_whole");
}

void R8B_LAUNCH(launch_poly)(const PolyLaunch& L, void* stream)
{
	const long long n = L.b - L.a;
	if (L.span_max > 0)
	{
		// x rows + interpolated taps + row offsets: poly_lds_doubles()
		const size_t lds = (size_t) poly_lds_doubles(L.pitch, L.flen) * sizeof(double);
		hipLaunchKernelGGL(k_poly_tiled, dim3((unsigned) ((n + kPolyTO - 1) / kPolyTO),
			(unsigned) ((L.nch + kPolyTC - 1) / kPolyTC)), dim3(256), lds, (hipStream_t) stream, L);
		check(hipGetLastError(), "launch k_poly_tiled");
		launch_symbol_note("k_poly_tiled");
		return;
	}
	hipLaunchKernelGGL(k_poly, dim3((unsigned) ((n + 255) / 256), (unsigned) L.nch), dim3(256), 0,
		(hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_poly");
	launch_symbol_note("k_poly");
}

void R8B_LAUNCH(launch_hbup)(const HBLaunch& L, void* stream)
{
	const long long n = (L.b + 1) / 2 - L.a / 2;
	const unsigned tiles = (unsigned) ((n + L.tile - 1) / L.tile);
	hipLaunchKernelGGL(k_hbup, dim3(tiles, (unsigned) L.nch, L.carry_tail ? 2u : 1u), dim3(256),
		(size_t) (L.tile + 2 * L.ntaps) * sizeof(double), (hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_hbup");
	launch_symbol_note("k_hbup");
}

void R8B_LAUNCH(launch_hbdown)(const HBLaunch& L, void* stream)
{
	const long long n = L.b - L.a;
	const unsigned tiles = (unsigned) ((n + L.tile - 1) / L.tile);
	hipLaunchKernelGGL(k_hbdown, dim3(tiles, (unsigned) L.nch, L.carry_tail ? 2u : 1u), dim3(256),
		(size_t) hbdown_lds_doubles(L.tile, L.ntaps) * sizeof(double), (hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_hbdown");
	launch_symbol_note("k_hbdown");
}

#if !R8B_HAS_FAST
// (the PCM twin has no fast-path convolvers: see the top of the file)
void R8B_LAUNCH(launch_convx)(const ConvxLaunch&, int, void*)
{
	throw std::logic_error("launch_convx: a PCM view reached a fast-path convolver (it is fed through the staging rows)");
}
void R8B_LAUNCH(launch_convp)(const ConvxLaunch&, int, void*)
{
	throw std::logic_error("launch_convp: a PCM view reached a fast-path convolver (it is fed through the staging rows)");
}
#else
void R8B_LAUNCH(launch_convx)(const ConvxLaunch& X, int mode, void* stream)
{
	int logn = 0;
	while ((2 << logn) < X.c.n_in) logn++;
	// mode 3: a 3x zero-stuffed input / 3x strided output is 1:1 as far as the transforms go
	const int up = X.c.up_pow2 ? X.c.up : 1;
	const bool wide = X.flen > 24;
#define R8B_CONVX_DISPATCH_DOWN(LN, DL) \
	if (logn == LN && X.c.down == (1 << DL)) \
	{ \
		if (mode == 3) launch_convx_t<LN, -DL, 3, 24>(X, (hipStream_t) stream); \
		else launch_convx_t<LN, -DL, 0, 24>(X, (hipStream_t) stream); \
		return; \
	}
	if (X.c.down_pow2 && X.c.down > 1)
	{
		R8B_CONVX_GEOMS_DOWN(R8B_CONVX_DISPATCH_DOWN)
	}
#undef R8B_CONVX_DISPATCH_DOWN
#define R8B_CONVX_DISPATCH(LN, UL) \
	if (logn == LN && up == (1 << UL)) \
	{ \
		if (mode == 0) launch_convx_t<LN, UL, 0, 24>(X, (hipStream_t) stream); \
		else if (mode == 3) launch_convx_t<LN, UL, 3, 24>(X, (hipStream_t) stream); \
		else if (wide) launch_convx_t<LN, UL, 1, 32>(X, (hipStream_t) stream); \
		else launch_convx_t<LN, UL, 1, 24>(X, (hipStream_t) stream); \
		return; \
	}
	R8B_CONVX_GEOMS(R8B_CONVX_DISPATCH)
#undef R8B_CONVX_DISPATCH
	throw std::runtime_error("launch_convx: geometry not instantiated");
}
#endif // R8B_HAS_FAST
#endif // R8B_HAS_REST

#if R8B_HAS_FAST
#if R8B_HAS_PAIR
// the dispatcher over the pair kernels of this object: the whole set (one-object builds: throws when the geometry is
// not among them), or part R8B_TU_PAIR of it (returns false then)
#ifdef R8B_TU_PAIR
#define R8B_PAIR_CAT2(a, b) a##b
#define R8B_PAIR_CAT(a, b) R8B_PAIR_CAT2(a, b)
#define R8B_PAIR_RET bool
#define R8B_PAIR_DONE return true
bool R8B_PAIR_CAT(launch_convp_part, R8B_TU_PAIR)(const ConvxLaunch& X, int mode, void* stream)
#else
#define R8B_PAIR_RET void
#define R8B_PAIR_DONE return
void R8B_LAUNCH(launch_convp)(const ConvxLaunch& X, int mode, void* stream)
#endif
{
	int ln = 0;
	while ((1 << ln) < X.c.n_in) ln++;
	const bool wide = X.flen > 24;
	const int up = X.c.up_pow2 ? X.c.up : 1; // (mode 3: a 3x zero-stuffed input is 1:1 for the transforms)
#ifdef R8B_DEV_ONLY_MODE
	// development build (tools/variant.sh): one mode of the listed geometries only
#define R8B_CONVP_DISPATCH(LN, UL) \
	if (ln == LN && up == (1 << UL) && mode == R8B_DEV_ONLY_MODE) \
	{ \
		launch_convp_t<LN, UL, R8B_DEV_ONLY_MODE, 24>(X, (hipStream_t) stream); \
		R8B_PAIR_DONE; \
	}
#else
#define R8B_CONVP_DISPATCH(LN, UL) \
	if (ln == LN && up == (1 << UL)) \
	{ \
		if (mode == 0) launch_convp_t<LN, UL, 0, 24>(X, (hipStream_t) stream); \
		else if (mode == 19) launch_convp_p3<LN, UL>(X, (hipStream_t) stream); \
		else if (mode == 3) launch_convp_t<LN, UL, 3, 24>(X, (hipStream_t) stream); \
		else if (mode == 6) launch_convp_t<LN, UL, 6, 24>(X, (hipStream_t) stream); \
		else if (mode == 7) launch_convp_t<LN, UL, 7, 24>(X, (hipStream_t) stream); \
		else if (mode == 4) launch_convp_t<LN, UL, 4, 24>(X, (hipStream_t) stream); \
		else if (mode == 5) launch_convp_t<LN, UL, 5, 24>(X, (hipStream_t) stream); \
		else if (mode == 16) launch_convp_t<LN, UL, 16, 24>(X, (hipStream_t) stream); \
		else if (mode == 17) launch_convp_t<LN, UL, 17, 24>(X, (hipStream_t) stream); \
		else if (wide) launch_convp_t<LN, UL, 1, 32>(X, (hipStream_t) stream); \
		else launch_convp_t<LN, UL, 1, 24>(X, (hipStream_t) stream); \
		R8B_PAIR_DONE; \
	}
#endif
	if (X.c.down_pow2 && X.c.down > 1)
	{
#define R8B_CONVP_DISPATCH_DOWN(LN, DL) \
		if (LN == 13 && ln == 14 && X.c.down == (1 << DL) && ((DL == 1 && convp_mode_solo(mode)) || \
			(DL == 2 && (mode == 10 || mode == 11)))) \
		{ \
			launch_convp_solo_down<LN, DL>(X, mode, (hipStream_t) stream); \
			R8B_PAIR_DONE; \
		} \
		if (ln == LN && X.c.down == (1 << DL) && mode == 20) \
		{ \
			launch_convp_hbf<LN, DL>(X, (hipStream_t) stream); \
			R8B_PAIR_DONE; \
		} \
		if (ln == LN && X.c.down == (1 << DL) && mode < 8) \
		{ \
			if (mode == 3) launch_convp_t<LN, -DL, 3, 24>(X, (hipStream_t) stream); \
			else if (mode == 6) launch_convp_t<LN, -DL, 6, 24>(X, (hipStream_t) stream); \
			else if (mode == 7) launch_convp_t<LN, -DL, 7, 24>(X, (hipStream_t) stream); \
			else launch_convp_t<LN, -DL, 0, 24>(X, (hipStream_t) stream); \
			R8B_PAIR_DONE; \
		}
		R8B_CONVP_GEOMS_DOWN(R8B_CONVP_DISPATCH_DOWN)
#undef R8B_CONVP_DISPATCH_DOWN
#ifdef R8B_TU_PAIR
		return false;
#else
		throw std::runtime_error("launch_convp: decimating geometry not instantiated");
#endif
	}
	// (modes 8 / 9 / 12 / 13: the split 2x up-sampling form -- r8b_convp.h cp_sp_* --, modes 10 / 11 / 14 / 15: the one-channel form -- cp_solo_*,
	// 16384-point blocks -- live on the 8192-point 1:1 geometry)
#define R8B_CONVP_DISPATCH_BIG(LN, UL) \
	if (LN == 13 && UL == 0 && ((ln == 13 && convp_mode_sp(mode)) || (ln == 14 && convp_mode_solo(mode)))) \
	{ \
		launch_convp_sp<LN, UL>(X, mode, (hipStream_t) stream); \
		R8B_PAIR_DONE; \
	} \
	if (ln == LN && up == (1 << UL) && (mode < 8 || mode == 16 || mode == 17)) \
	{ \
		if (mode == 3) launch_convp_t<LN, UL, 3, 24>(X, (hipStream_t) stream); \
		else if (mode == 6) launch_convp_t<LN, UL, 6, 24>(X, (hipStream_t) stream); \
		else if (mode == 7) launch_convp_t<LN, UL, 7, 24>(X, (hipStream_t) stream); \
		else if (mode == 0) launch_convp_t<LN, UL, 0, 24>(X, (hipStream_t) stream); \
		else if (mode == 4) launch_convp_t<LN, UL, 4, 24>(X, (hipStream_t) stream); \
		else if (mode == 5) launch_convp_t<LN, UL, 5, 24>(X, (hipStream_t) stream); \
		else if (mode == 16) launch_convp_t<LN, UL, 16, 24>(X, (hipStream_t) stream); \
		else if (mode == 17) launch_convp_t<LN, UL, 17, 24>(X, (hipStream_t) stream); \
		else if (wide) launch_convp_t<LN, UL, 1, 32>(X, (hipStream_t) stream); \
		else launch_convp_t<LN, UL, 1, 24>(X, (hipStream_t) stream); \
		R8B_PAIR_DONE; \
	}
	R8B_CONVP_GEOMS_BIG(R8B_CONVP_DISPATCH_BIG)
#undef R8B_CONVP_DISPATCH_BIG
	R8B_CONVP_GEOMS(R8B_CONVP_DISPATCH)
#undef R8B_CONVP_DISPATCH
#ifdef R8B_TU_PAIR
	return false;
#else
	throw std::runtime_error("launch_convp: geometry not instantiated");
#endif
}
#else // !R8B_HAS_PAIR: the pair kernels live in R8B_TU_NOPAIR part objects
#define R8B_PAIR_DECL(k) bool launch_convp_part##k(const ConvxLaunch& X, int mode, void* stream);
R8B_PAIR_DECL(1) R8B_PAIR_DECL(2) R8B_PAIR_DECL(3) R8B_PAIR_DECL(4) R8B_PAIR_DECL(5) R8B_PAIR_DECL(6) R8B_PAIR_DECL(7)
R8B_PAIR_DECL(8)
#undef R8B_PAIR_DECL
void R8B_LAUNCH(launch_convp)(const ConvxLaunch& X, int mode, void* stream)
{
	static_assert(R8B_TU_NOPAIR == 8, "the Makefile builds eight pair-kernel parts");
	if (launch_convp_part1(X, mode, stream) || launch_convp_part2(X, mode, stream) ||
		launch_convp_part3(X, mode, stream) || launch_convp_part4(X, mode, stream) ||
		launch_convp_part5(X, mode, stream) || launch_convp_part6(X, mode, stream) ||
		launch_convp_part7(X, mode, stream) || launch_convp_part8(X, mode, stream)) return;
	throw std::runtime_error("launch_convp: geometry not instantiated");
}
#endif // R8B_HAS_PAIR
#endif // R8B_HAS_FAST

#if R8B_HAS_REST

void R8B_LAUNCH(launch_hbcascade)(const HBCascadeLaunch& L, void* stream)
{
	const long long n = L.b - L.a;
	if (n <= 0)
	{
		if (L.carry_tail) throw std::logic_error("launch_hbcascade: a history copy on a launch without tiles");
		return;
	}
	const unsigned tiles = (unsigned) ((n + L.tile - 1) / L.tile);
	hipLaunchKernelGGL(k_hbcascade, dim3(tiles, (unsigned) L.nch, L.carry_tail ? 2u : 1u), dim3(256),
		(size_t) (L.buf + L.buf2 + 3 * kHbcSlack) * sizeof(double), (hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_hbcascade");
	launch_symbol_note("k_hbcascade");
}

void R8B_LAUNCH(launch_hbdcascade)(const HBCascadeLaunch& L, void* stream)
{
	set_lds_attrs();
	const long long n = L.b - L.a;
	if (n <= 0)
	{
		if (L.carry_tail) throw std::logic_error("launch_hbdcascade: a history copy on a launch without tiles");
		return;
	}
	const unsigned tiles = (unsigned) ((n + L.tile - 1) / L.tile);
	hipLaunchKernelGGL(k_hbdcascade, dim3(tiles, (unsigned) L.nch, L.carry_tail ? 2u : 1u), dim3(256),
		(size_t) (L.buf + L.buf2) * sizeof(double), (hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_hbdcascade");
	launch_symbol_note("k_hbdcascade");
}

void R8B_LAUNCH(launch_tail)(const TailLaunch& L, void* stream)
{
	const long long n = L.p1 - L.p0;
	if (n <= 0) return;
	hipLaunchKernelGGL(k_tail, dim3((unsigned) ((n + 255) / 256), (unsigned) L.nch), dim3(256), 0,
		(hipStream_t) stream, L);
	check(hipGetLastError(), "launch k_tail");
	launch_symbol_note("k_tail");
}

#ifndef R8B_PCM_VARIANT
// ------------------------------------------------------------------ format dispatch
void launch_conv_pcm(const ConvLaunch& L, void* stream);
void launch_whole_pcm(const WholeLaunch& L, void* stream);
void launch_poly_pcm(const PolyLaunch& L, void* stream);
void launch_hbup_pcm(const HBLaunch& L, void* stream);
void launch_hbdown_pcm(const HBLaunch& L, void* stream);
void launch_convx_pcm(const ConvxLaunch& X, int mode, void* stream);
void launch_convp_pcm(const ConvxLaunch& X, int mode, void* stream);
void launch_hbcascade_pcm(const HBCascadeLaunch& L, void* stream);
void launch_hbdcascade_pcm(const HBCascadeLaunch& L, void* stream);
void launch_tail_pcm(const TailLaunch& L, void* stream);

#define R8B_DISPATCH(name, T) \
	void name(const T& L, void* stream) \
	{ \
		if ((L.src.cur_fmt | L.dst.fmt) != kPcmF64) name##_pcm(L, stream); \
		else name##_f64(L, stream); \
	}
R8B_DISPATCH(launch_conv, ConvLaunch)
R8B_DISPATCH(launch_whole, WholeLaunch)
R8B_DISPATCH(launch_poly, PolyLaunch)
R8B_DISPATCH(launch_hbup, HBLaunch)
R8B_DISPATCH(launch_hbdown, HBLaunch)
R8B_DISPATCH(launch_hbcascade, HBCascadeLaunch)
R8B_DISPATCH(launch_hbdcascade, HBCascadeLaunch)
#undef R8B_DISPATCH

void launch_convx(const ConvxLaunch& X, int mode, void* stream)
{
	if ((X.c.src.cur_fmt | X.c.dst.fmt | X.wdst.fmt) != kPcmF64) launch_convx_pcm(X, mode, stream);
	else launch_convx_f64(X, mode, stream);
}

// (per THREAD: an engine counts what its own launches walked as the difference around them -- Engine::launch_pair)
namespace { thread_local long long t_walk_blocks = 0; }
long long launch_walk_blocks() { return t_walk_blocks; }
void launch_walk_blocks_add(long long n) { t_walk_blocks += n; }

namespace { thread_local const char* t_last_symbol = nullptr; }
void launch_symbol_note(const char* symbol) { t_last_symbol = symbol; }
const char* launch_symbol_last() { return t_last_symbol; }

void launch_convp(const ConvxLaunch& X, int mode, void* stream)
{
	if ((X.c.src.cur_fmt | X.c.dst.fmt | X.wdst.fmt) != kPcmF64) launch_convp_pcm(X, mode, stream);
	else launch_convp_f64(X, mode, stream);
}

void launch_tail(const TailLaunch& L, void* stream)
{
	if (L.src.cur_fmt != kPcmF64) launch_tail_pcm(L, stream);
	else launch_tail_f64(L, stream);
}

static void launch_pcm(const PcmLaunch& L, bool in, void* stream)
{
	if (L.n <= 0 || L.nch <= 0) return;
	if (!L.interleaved)
	{
		const dim3 rows((unsigned) ((L.n + kPcmRowChunk - 1) / kPcmRowChunk), (unsigned) L.nch);
		if (in) hipLaunchKernelGGL(k_pcm_rows_in, rows, dim3(256), 0, (hipStream_t) stream, L);
		else hipLaunchKernelGGL(k_pcm_rows_out, rows, dim3(256), 0, (hipStream_t) stream, L);
		check(hipGetLastError(), in ? "launch k_pcm_rows_in" : "launch k_pcm_rows_out");
		return;
	}
	const dim3 grid((unsigned) ((L.n + kPcmTile - 1) / kPcmTile),
		(unsigned) ((L.nch + kPcmTile - 1) / kPcmTile));
	if (in) hipLaunchKernelGGL(k_pcm_in, grid, dim3(256), 0, (hipStream_t) stream, L);
	else hipLaunchKernelGGL(k_pcm_out, grid, dim3(256), 0, (hipStream_t) stream, L);
	check(hipGetLastError(), in ? "launch k_pcm_in" : "launch k_pcm_out");
}

void launch_pcm_in(const PcmLaunch& L, void* stream) { launch_pcm(L, true, stream); }
void launch_pcm_out(const PcmLaunch& L, void* stream) { launch_pcm(L, false, stream); }

// ------------------------------------------------------------------ memory helpers

int dev_resolve(int device)
{
	int n = 0;
	check(hipGetDeviceCount(&n), "hipGetDeviceCount (is a HIP device visible? there is no CPU fallback)");
	if (device < 0) check(hipGetDevice(&device), "hipGetDevice");
	if (device >= n) throw std::runtime_error("device ordinal out of range");
	return device;
}

int dev_swap(int device)
{
	int cur = -1;
	check(hipGetDevice(&cur), "hipGetDevice");
	if (cur != device) check(hipSetDevice(device), "hipSetDevice");
	return cur;
}

void dev_restore(int device) noexcept
{
	// (a failure here -- the runtime already unloaded at interpreter exit, a device lost to an earlier sticky error --
	// must not turn a destructor into std::terminate or mask the exception being unwound)
	int cur = -1;
	if (hipGetDevice(&cur) == hipSuccess && cur != device) (void) hipSetDevice(device);
}

void* dev_alloc(size_t bytes)
{
	void* p = nullptr;
	if (bytes == 0) bytes = 8;
	check(hipMalloc(&p, bytes), "hipMalloc");
	check(hipMemset(p, 0, bytes), "hipMemset");
	return p;
}

void dev_free(void* p)
{
	if (p) (void) hipFree(p);
}

void dev_zero(void* p, size_t bytes, void* stream)
{
	check(hipMemsetAsync(p, 0, bytes, (hipStream_t) stream), "hipMemsetAsync");
}

void dev_upload(void* dst, const void* src, size_t bytes)
{
	check(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
}

void dev_upload_async(void* dst, const void* src, size_t bytes, void* stream)
{
	check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t) stream),
		"hipMemcpyAsync H2D");
}

void dev_download(void* dst, const void* src, size_t bytes, void* stream)
{
	check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream),
		"hipMemcpyAsync D2H");
	check(hipStreamSynchronize((hipStream_t) stream), "hipStreamSynchronize");
}

void dev_sync(void* stream)
{
	check(hipStreamSynchronize((hipStream_t) stream), "hipStreamSynchronize");
}

void dev_check_last(const char* what)
{
	check(hipGetLastError(), what);
}

void* dev_event_create()
{
	hipEvent_t e;
	check(hipEventCreate(&e), "hipEventCreate");
	return e;
}

void dev_event_destroy(void* ev)
{
	if (ev) (void) hipEventDestroy((hipEvent_t) ev);
}

void dev_event_record(void* ev, void* stream)
{
	check(hipEventRecord((hipEvent_t) ev, (hipStream_t) stream), "hipEventRecord");
}

float dev_event_elapsed_ms(void* start, void* stop)
{
	float ms = 0.f;
	check(hipEventSynchronize((hipEvent_t) stop), "hipEventSynchronize");
	check(hipEventElapsedTime(&ms, (hipEvent_t) start, (hipEvent_t) stop), "hipEventElapsedTime");
	return ms;
}
#endif // !R8B_PCM_VARIANT
#endif // R8B_HAS_REST

} // namespace r8bhip
