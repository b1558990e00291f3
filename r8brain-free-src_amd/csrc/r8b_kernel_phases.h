// r8b_kernel_phases.h -- the arithmetic of every kernel, written as per-thread "phase" functions.
//
// A kernel in r8b_kernels.hip is a sequence of phases separated by __syncthreads(); each phase is
// a function of (thread id, thread count, LDS pointers, launch parameters) defined here.  The
// includer defines R8B_HD (`__device__ __forceinline__` in r8b_kernels.hip).  tests/emul/ includes
// the same header with R8B_HD empty and runs the phases thread by thread on the host, so the
// index arithmetic can be unit-tested in a container without a GPU; the product library never
// contains that build.
//
// Stream semantics each kernel reproduces (reference file:line):
//   overlap-save block convolver   CDSPBlockConvolver.h:252-354, 512-593, 606-629,
//                                  CDSPRealFFT.h:98-170, 289-385 (K1..K7 of SURVEY.md 2.1)
//   whole-step polyphase FIR       CDSPFracInterpolator.h:991-1060
//   polynomial-interpolated bank   CDSPFracInterpolator.h:1069-1179
//   half-band 2x up / down         CDSPHBUpsampler.h:773-786, CDSPHBDownsampler.h:282-295
#ifndef R8B_KERNEL_PHASES_H
#define R8B_KERNEL_PHASES_H

#ifndef R8B_HD
	#error "define R8B_HD before including r8b_kernel_phases.h"
#endif

#include "r8b_launch.h"
#include "r8b_pcm_codec.h"

namespace r8bhip {

struct alignas(16) cd
{
	double re, im;
};

// A tap window in LDS.  On the GPU the pointer is volatile in the LDS address space: every tap becomes ONE
// ds_read_b64 (256 B/clk per CU) with the compiler's own per-read wait counts; left alone, the compiler
// pairs neighbouring taps into ds_read2_b64, which moves half as many bytes per LDS cycle
// (MI355X_MICROARCH.md, LDS table) -- the half-band and interpolator loops are LDS-read bound.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const volatile __attribute__((address_space(3))) double* LdsWin;
#else
typedef const double* LdsWin;
#endif
R8B_HD LdsWin lds_win(const double* p) { return (LdsWin) p; }

// ------------------------------------------------------------------------------------ views

R8B_HD double src_load(const SrcView& s, int ch, long long pos)
{
#ifndef R8B_NO_PCM_FUSE // defined by the fp64-only build of r8b_kernels.hip
	if (s.cur_fmt != kPcmF64)
	{
		// planar PCM caller buffer, decoded in place: element sizes differ from the ring's, so
		// the select is a branch here
		double v;
		if (pos >= s.cur_base)
			v = pcm_decode(reinterpret_cast<const unsigned char*>(s.cur) +
				((long long) ch * s.cur_stride + (pos - s.cur_base)) * pcm_bytes(s.cur_fmt),
				s.cur_fmt);
		else
			v = s.ring[(long long) ch * s.ring_stride + (pos & s.ring_mask)];
		return pos < 0 ? 0.0 : v;
	}
#endif
	// one load through a selected address (no branches: a thread's loads stay in flight together)
	const double* pr = s.ring + ((long long) ch * s.ring_stride + (pos & s.ring_mask));
	const double* pc = s.cur + ((long long) ch * s.cur_stride + (pos - s.cur_base));
	const double* p = pos >= s.cur_base ? pc : pr;
	const double v = *p;
	return pos < 0 ? 0.0 : v;
}

R8B_HD void dst_store(const DstView& d, int ch, long long q, double v)
{
#ifndef R8B_NO_PCM_FUSE
	if (d.fmt != kPcmF64)
	{
		pcm_encode(reinterpret_cast<unsigned char*>(d.p) +
			((long long) ch * d.stride + ((q + d.off) & d.mask)) * pcm_bytes(d.fmt), d.fmt, v);
		return;
	}
#endif
	d.p[(long long) ch * d.stride + ((q + d.off) & d.mask)] = v;
}

// All loads of a block are positions base + rel with a workgroup-uniform 64-bit base and a small
// per-lane offset: the uniform part (row pointers, the fresh/history boundary, the stream start)
// is folded once on the scalar unit, each load is left with 32-bit arithmetic.  (Through
// src_load every load repeats ~70 instructions of 64-bit address and select arithmetic.)
struct SrcBlock
{
	const double* pc; // where position `base` would sit in the caller's buffer
	const double* pr; // the channel's ring row
	unsigned b_lo, mask; // low half of base; ring mask (ring sizes are far below 2^32)
	int c_rel;        // rel >= c_rel  <=>  position >= cur_base (fresh sample)
	int z_rel;        // rel <  z_rel  <=>  position < 0 (reads as zero)
#ifndef R8B_NO_PCM_FUSE
	// PCM-aware build: single-sample loads of a PCM caller buffer go through src_load()
	const SrcView* sv;
	int ch;
	long long base;
#endif
};

R8B_HD int clamp_rel(long long d)
{
	return d > 0x3fffffffLL ? 0x3fffffff : (d < -0x3fffffffLL ? -0x3fffffff : (int) d);
}

R8B_HD SrcBlock src_block(const SrcView& s, int ch, long long base)
{
	SrcBlock b;
	b.pr = s.ring + (long long) ch * s.ring_stride;
	b.b_lo = (unsigned) base;
	b.mask = (unsigned) s.ring_mask;
	// cur_base is LLONG_MAX when the stage has no caller buffer, and base may be negative (first
	// tile of a stream): nothing here may overflow or form a wild pointer
	const bool far = s.cur_base > base + 0x3fffffffLL;
	b.c_rel = far ? 0x3fffffff : clamp_rel(s.cur_base - base);
	b.pc = far ? s.ring : s.cur + ((long long) ch * s.cur_stride + (base - s.cur_base));
	b.z_rel = clamp_rel(-base);
#ifndef R8B_NO_PCM_FUSE
	b.sv = &s;
	b.ch = ch;
	b.base = base;
#endif
	return b;
}

R8B_HD cd src_block_load2(const SrcBlock& b, int rel)
{
	const double* pr = b.pr + ((b.b_lo + (unsigned) rel) & b.mask);
	const double* pc = b.pc + rel;
	const cd v = *reinterpret_cast<const cd*>(rel >= b.c_rel ? pc : pr);
	cd r;
	r.re = rel < b.z_rel ? 0.0 : v.re;
	r.im = rel < b.z_rel ? 0.0 : v.im;
	return r;
}

R8B_HD double src_block_load1(const SrcBlock& b, int rel)
{
#ifndef R8B_NO_PCM_FUSE
	if (b.sv->cur_fmt != kPcmF64) return src_load(*b.sv, b.ch, b.base + rel);
#endif
	const double* pr = b.pr + ((b.b_lo + (unsigned) rel) & b.mask);
	const double* pc = b.pc + rel;
	const double v = *(rel >= b.c_rel ? pc : pr);
	return rel < b.z_rel ? 0.0 : v;
}

// Staging loop of the tile kernels: positions base + [0, len) of one channel into LDS (zeros from `end`
// on).  A thread issues U loads before its first LDS store, so U of its HBM round trips overlap; the
// plain loop (load, wait, store per pass) keeps ONE load per lane in flight, which holds a streaming kernel
// at the latency-bandwidth product of its resident waves instead of the HBM rate.  Out-of-range lanes load
// a clamped (valid) element and store nothing.
struct SlotLinear
{
	R8B_HD int operator()(int i) const { return i; }
};

// even elements first, odd elements from `odd` on: a stride-2 reader (2x decimator) finds its taps at
// consecutive slots
struct SlotParity
{
	int odd;
	R8B_HD int operator()(int i) const { return (i >> 1) + ((i & 1) ? odd : 0); }
};

template<int U, class Slot>
R8B_HD void src_block_stage_to(const SrcBlock& b, double* xs, const Slot& slot, int len, int end, int tid,
	int nthr)
{
	const int lim = end < len ? end : len;
	if (lim <= 0)
	{
		for (int i = tid; i < len; i += nthr) xs[slot(i)] = 0.0;
		return;
	}
	for (int i0 = tid; i0 < len; i0 += U * nthr)
	{
		double v[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			v[u] = src_block_load1(b, i < lim ? i : lim - 1);
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			if (i < len) xs[slot(i)] = i < lim ? v[u] : 0.0;
		}
	}
}

template<int U>
R8B_HD void src_block_stage(const SrcBlock& b, double* xs, int len, int end, int tid, int nthr)
{
	src_block_stage_to<U>(b, xs, SlotLinear(), len, end, tid, nthr);
}

// ------------------------------------------------------------------------------------ small DFTs

// (r,i) *= w16^E with w16 = exp(-2 pi i / 16); CONJ selects exp(+...)
template<int E, bool CONJ>
R8B_HD void mul_w16(double& r, double& i)
{
	constexpr double C1 = 0.92387953251128674; // cos(pi/8)
	constexpr double S1 = 0.38268343236508977; // sin(pi/8)
	constexpr double C2 = 0.70710678118654752; // cos(pi/4)
	if constexpr (E == 0)
	{
	}
	else if constexpr (E == 4)
	{
		const double t = r;
		if constexpr (CONJ) { r = -i; i = t; }
		else { r = i; i = -t; }
	}
	else
	{
		constexpr double c = E == 1 ? C1 : (E == 2 ? C2 : (E == 3 ? S1 : (E == 5 ? -S1 :
			(E == 6 ? -C2 : -C1))));
		constexpr double s0 = E == 1 ? S1 : (E == 2 ? C2 : (E == 3 ? C1 : (E == 5 ? C1 :
			(E == 6 ? C2 : S1))));
		constexpr double s = CONJ ? -s0 : s0; // w = c - i*s
		const double tr = r * c + i * s;
		const double ti = i * c - r * s;
		r = tr;
		i = ti;
	}
}

template<int R, int H, int IDX>
struct DifBf
{
	static R8B_HD void run(double* vr, double* vi)
	{
		constexpr int g = (IDX / H) * 2 * H, u = IDX % H, E = u * (8 / H);
		constexpr int i0 = g + u, i1 = g + u + H;
		const double ar = vr[i0], ai = vi[i0], br = vr[i1], bi = vi[i1];
		vr[i0] = ar + br;
		vi[i0] = ai + bi;
		double dr = ar - br, di = ai - bi;
		mul_w16<E, false>(dr, di);
		vr[i1] = dr;
		vi[i1] = di;
		if constexpr (IDX + 1 < R / 2) DifBf<R, H, IDX + 1>::run(vr, vi);
	}
};

template<int R, int H>
struct DifSt
{
	static R8B_HD void run(double* vr, double* vi)
	{
		DifBf<R, H, 0>::run(vr, vi);
		if constexpr (H > 1) DifSt<R, H / 2>::run(vr, vi);
	}
};

template<int R, int H, int IDX>
struct DitBf
{
	static R8B_HD void run(double* vr, double* vi)
	{
		constexpr int g = (IDX / H) * 2 * H, u = IDX % H, E = u * (8 / H);
		constexpr int i0 = g + u, i1 = g + u + H;
		const double ar = vr[i0], ai = vi[i0];
		double br = vr[i1], bi = vi[i1];
		mul_w16<E, true>(br, bi);
		vr[i0] = ar + br;
		vi[i0] = ai + bi;
		vr[i1] = ar - br;
		vi[i1] = ai - bi;
		if constexpr (IDX + 1 < R / 2) DitBf<R, H, IDX + 1>::run(vr, vi);
	}
};

template<int R, int H>
struct DitSt
{
	static R8B_HD void run(double* vr, double* vi)
	{
		DitBf<R, H, 0>::run(vr, vi);
		if constexpr (2 * H < R) DitSt<R, 2 * H>::run(vr, vi);
	}
};

// R-point forward DFT in registers, natural order in, bit-reversed order out
template<int R> R8B_HD void dif_regs(double* vr, double* vi) { DifSt<R, R / 2>::run(vr, vi); }
// R-point backward (conjugate, unnormalised) DFT, bit-reversed order in, natural order out
template<int R> R8B_HD void dit_regs(double* vr, double* vi) { DitSt<R, 1>::run(vr, vi); }

template<int R> constexpr int bitrev_c(int p)
{
	int r = 0;
	for (int b = 1; b < R; b <<= 1)
	{
		r = (r << 1) | (p & 1);
		p >>= 1;
	}
	return r;
}

R8B_HD int ilog2(int v)
{
	int b = 0;
	while ((1 << b) < v) b++;
	return b;
}

R8B_HD unsigned bitrev32(unsigned v)
{
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	return (v >> 16) | (v << 16);
}

R8B_HD int bitrev_n(int v, int bits) { return bits == 0 ? 0 : (int) (bitrev32((unsigned) v) >> (32 - bits)); }

// ------------------------------------------------------------------------------------ FFT passes
//
// In-place decimation-in-frequency forward transform: a pass with sub-transform length n and
// radix R performs log2(R) radix-2 DIF stages of every length-n sub-block, so after all passes the
// spectrum sits in plain bit-reversed order (bin k at position bitrev(k)).  The backward transform
// is the transposed flow graph (decimation in time, conjugate twiddles): bit-reversed in, natural
// out, unnormalised.  No reordering pass exists anywhere: the spectral stage in between addresses
// bins through bitrev().  `tw` holds exp(-2 pi i e / tw_len), e < tw_len.

// Where element i of a transform array sits: as it is, or with index bits 0-3 XOR bits 4-7 (SwXor, r8b_convp.h pswz: the
// passes whose lanes are 8 or 16 elements apart -- sub-lengths 64 and below, up to 32 lanes on one bank group unswizzled
// -- then spread over the banks).  The long-block kernel (k_conv_big) uses SwXor for both of its LDS arrays.
struct SwNone
{
	static R8B_HD int at(int i) { return i; }
};
struct SwXor
{
	static R8B_HD int at(int i) { return i ^ ((i >> 4) & 15); }
};

template<int R, class SW = SwNone>
R8B_HD void dif_pass_one(cd* buf, int n, int bidx, const cd* tw, int tw_len)
{
	const int q = n / R;
	const int blk = bidx / q, j = bidx - blk * q;
	const int i0 = blk * n + j;
	double vr[R], vi[R];
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[SW::at(i0 + p * q)];
		vr[p] = v.re;
		vi[p] = v.im;
	}
	dif_regs<R>(vr, vi);
	if (q > 1)
	{
		const int ts = tw_len / n * j;
#pragma unroll
		for (int p = 1; p < R; p++)
		{
			const cd w = tw[ts * bitrev_c<R>(p)];
			const double tr = vr[p] * w.re - vi[p] * w.im;
			const double ti = vr[p] * w.im + vi[p] * w.re;
			vr[p] = tr;
			vi[p] = ti;
		}
	}
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		buf[SW::at(i0 + p * q)] = v;
	}
}

template<int R, class SW = SwNone>
R8B_HD void dit_pass_one(cd* buf, int n, int bidx, const cd* tw, int tw_len)
{
	const int q = n / R;
	const int blk = bidx / q, j = bidx - blk * q;
	const int i0 = blk * n + j;
	double vr[R], vi[R];
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[SW::at(i0 + p * q)];
		vr[p] = v.re;
		vi[p] = v.im;
	}
	if (q > 1)
	{
		const int ts = tw_len / n * j;
#pragma unroll
		for (int p = 1; p < R; p++)
		{
			const cd w = tw[ts * bitrev_c<R>(p)]; // multiply by conj(w)
			const double tr = vr[p] * w.re + vi[p] * w.im;
			const double ti = vi[p] * w.re - vr[p] * w.im;
			vr[p] = tr;
			vi[p] = ti;
		}
	}
	dit_regs<R>(vr, vi);
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		buf[SW::at(i0 + p * q)] = v;
	}
}

// one whole pass over a length-N complex buffer (all sub-blocks), strided over the threads
template<class SW>
R8B_HD void fft_pass_sw(cd* buf, int N, int n, int radix, bool inverse, const cd* tw, int tw_len,
	int tid, int nthr)
{
	const int nb = N / radix;
	for (int b = tid; b < nb; b += nthr)
	{
		if (!inverse)
		{
			if (radix == 16) dif_pass_one<16, SW>(buf, n, b, tw, tw_len);
			else if (radix == 8) dif_pass_one<8, SW>(buf, n, b, tw, tw_len);
			else if (radix == 4) dif_pass_one<4, SW>(buf, n, b, tw, tw_len);
			else dif_pass_one<2, SW>(buf, n, b, tw, tw_len);
		}
		else
		{
			if (radix == 16) dit_pass_one<16, SW>(buf, n, b, tw, tw_len);
			else if (radix == 8) dit_pass_one<8, SW>(buf, n, b, tw, tw_len);
			else if (radix == 4) dit_pass_one<4, SW>(buf, n, b, tw, tw_len);
			else dit_pass_one<2, SW>(buf, n, b, tw, tw_len);
		}
	}
}
R8B_HD void fft_pass(cd* buf, int N, int n, int radix, bool inverse, const cd* tw, int tw_len,
	int tid, int nthr)
{
	fft_pass_sw<SwNone>(buf, N, n, radix, inverse, tw, tw_len, tid, nthr);
}

// history copy (k_tail; carried by a half-band launch's second grid layer): workgroup w of nw of channel ch copies its
// share of the stream positions into the ring the next call reads
R8B_HD void tail_copy(const TailLaunch& T, int w, int nw, int ch, int tid, int nthr)
{
	for (long long i = T.p0 + (long long) w * nthr + tid; i < T.p1; i += (long long) nw * nthr)
		T.ring[(long long) ch * T.ring_stride + (i & T.ring_mask)] = src_load(T.src, ch, i);
}

// ------------------------------------------------------------------------------------ convolver

// K1: assemble the circular input block k of channel ch as n_in reals in LDS.
// (reference CDSPBlockConvolver.h:283-305 and, for non-2^k up-sampling, copyUpsample :414-496)
R8B_HD void conv_load(const ConvLaunch& L, double* a, long long k, int ch, int tid, int nthr)
{
	if (L.up_pow2)
	{
		const int iln = L.in_len / L.up; // new input samples per block
		const long long base = k * iln;
		for (int i = tid; i < L.n_in; i += nthr)
		{
			const long long pos = i < iln ? base + i : base + i - L.n_in;
			a[i] = src_load(L.src, ch, pos);
		}
	}
	else
	{
		const long long base = k * (long long) L.in_len;
		for (int i = tid; i < L.n_in; i += nthr)
		{
			const long long t = i < L.in_len ? base + i : base + i - L.bl2;
			double v = 0.0;
			if (t >= 0 && t % L.up == 0) v = src_load(L.src, ch, t / L.up);
			a[i] = v;
		}
	}
}

// where the forward result's bins are: the whole transform bit-reversed in one array (bin k at bitrev(k)) ...
struct ZFull
{
	const cd* za;
	int logN;
	R8B_HD cd operator()(int k) const { return za[bitrev_n(k, logN)]; }
};
// ... or ONE of the two sub-blocks the first radix-2 DIF stage leaves -- bins k = h (mod 2) --, transformed as an array
// of its own (k_conv_big): bin k at bitrev(k >> 1) of N / 2 positions.  Bins k and N - k, kp and N2 - kp have the same
// parity (N, N2 even), so the spectral stage of a parity class needs its own sub-block only.
struct ZHalf
{
	const cd* zl;
	int logN;
	R8B_HD cd operator()(int k) const { return zl[SwXor::at(bitrev_n(k >> 1, logN - 1))]; }
};

// spectrum bin k (0 <= k <= N) of the 2N-point real sequence whose packed N-point complex DFT
// sits bit-reversed behind `z`
template<class Z>
R8B_HD cd real_bin(const Z& z, int k, int N, const cd* tw, int tw_len)
{
	const int k1 = k & (N - 1), k2 = (N - k) & (N - 1);
	const cd z1 = z(k1), z2 = z(k2);
	const double er = 0.5 * (z1.re + z2.re), ei = 0.5 * (z1.im - z2.im);
	// O = (Z1 - conj Z2) / (2i)
	const double dr = z1.re - z2.re, di = z1.im + z2.im;
	const double orr = 0.5 * di, oi = -0.5 * dr;
	const cd w = tw[(long) k * (tw_len / (2 * N))];
	cd r;
	r.re = er + (w.re * orr - w.im * oi);
	r.im = ei + (w.re * oi + w.im * orr);
	return r;
}

// bin m (0 <= m <= bl2/2) of the spectrum of the zero-stuffed block: the 2N-point spectrum
// repeated with period 2N (K3, reference CDSPBlockConvolver.h:606-629)
template<class Z>
R8B_HD cd stuffed_bin(const Z& z, int m, int N, const cd* tw, int tw_len)
{
	const int mm = m & (2 * N - 1);
	if (mm <= N) return real_bin(z, mm, N, tw, tw_len);
	cd r = real_bin(z, 2 * N - mm, N, tw, tw_len);
	r.im = -r.im;
	return r;
}

// bin m (0 <= m <= N2) of the product spectrum handed to the inverse transform (K4, K5):
// multiply by the real zero-phase kernel (reference CDSPRealFFT.h:289-385); when decimating by
// 2^k the Nyquist bin of the shortened transform is the reference's fix-up
// (reference CDSPBlockConvolver.h:329-342).
template<class Z>
R8B_HD cd product_bin(const ConvLaunch& L, const Z& z, int m, int N, int N2)
{
	cd r = stuffed_bin(z, m, N, L.tw, L.tw_len);
	if (L.Hc != nullptr)
	{
		// complex kernel spectrum (minimum phase; reference CDSPRealFFT.h:186-274 multiplyBlocks)
		const cd h = L.Hc[m];
		cd o;
		o.re = h.re * r.re - h.im * r.im;
		o.im = (m == N2 && L.down_pow2 && L.down > 1) ? 0.0 : h.re * r.im + h.im * r.re;
		return o;
	}
	const double h = L.H[m];
	if (m == N2 && L.down_pow2 && L.down > 1)
	{
		r.re = h * (r.re + r.im);
		r.im = 0.0;
		return r;
	}
	r.re *= h;
	r.im *= h;
	return r;
}

// K3+K4+K5 and the packing for the half-length complex backward transform: reads the forward
// result (bit-reversed behind `z`, N complex), writes N2 complex values bit-reversed into zb -- the slots
// kp = first, first + step, ... <= N2 / 2 (a slot is the pair of backward bins kp, N2 - kp).
template<class Z>
R8B_HD void conv_spectral_z(const ConvLaunch& L, const Z& z, cd* zb, int first, int step, int tid, int nthr)
{
	const int N = L.n_in / 2, N2 = L.n_out / 2;
	const int logN2 = ilog2(N2);
	const int tsh = L.tw_len / (2 * N2);
	for (int kp = first + step * tid; kp <= N2 / 2; kp += step * nthr)
	{
		const cd sa = product_bin(L, z, kp, N, N2);
		const cd sb = product_bin(L, z, N2 - kp, N, N2);
		// Z'[k] = (Sa + conj Sb) + i * conj(w^k) * (Sa - conj Sb),  w = exp(-2 pi i / (2 N2))
		{
			const cd w = L.tw[(long) kp * tsh];
			const double er = sa.re + sb.re, ei = sa.im - sb.im;
			const double dr = sa.re - sb.re, di = sa.im + sb.im;
			// conj(w) * D
			const double pr = w.re * dr + w.im * di, pi = w.re * di - w.im * dr;
			cd zz;
			zz.re = er - pi;
			zz.im = ei + pr;
			zb[bitrev_n(kp, logN2)] = zz;
		}
		const int k2 = N2 - kp;
		if (kp != 0 && k2 != kp)
		{
			const cd w = L.tw[(long) k2 * tsh];
			const double er = sb.re + sa.re, ei = sb.im - sa.im;
			const double dr = sb.re - sa.re, di = sb.im + sa.im;
			const double pr = w.re * dr + w.im * di, pi = w.re * di - w.im * dr;
			cd zz;
			zz.re = er - pi;
			zz.im = ei + pr;
			zb[bitrev_n(k2, logN2)] = zz;
		}
	}
}

R8B_HD void conv_spectral(const ConvLaunch& L, const cd* za, cd* zb, int tid, int nthr)
{
	ZFull z;
	z.za = za;
	z.logN = ilog2(L.n_in / 2);
	conv_spectral_z(L, z, zb, 0, 1, tid, nthr);
}

// ---- blocks whose forward array does not fit LDS (k_conv_big: the reference's 32768-point blocks in front of a
// decimation in the spectrum).  The forward transform's first radix-2 DIF stage is taken in the load -- every sample of
// the block is a function of the source alone --, which leaves two independent sub-blocks of N / 2 complex: each is
// loaded, transformed and multiplied into its parity class of backward bins (ZHalf) in LDS, one after the other; only
// the packed backward spectrum (n_out doubles) goes through global memory.

// sample i of block k's circular input (conv_load's element).  Zero stuffing: virtual time t = base + d, d = i or
// i - bl2, in 32-bit arithmetic relative to the block's base (q0 = floor(base / up), r0 = base mod up, once per block):
// t is a source sample's time iff (r0 + d) mod up == 0, the sample q0 + floor((r0 + d) / up)
struct ConvBlockBase
{
	long long q0;
	int r0;
};
R8B_HD ConvBlockBase conv_block_base(const ConvLaunch& L, long long k)
{
	ConvBlockBase B;
	if (L.up_pow2)
	{
		B.q0 = k * (L.in_len / L.up);
		B.r0 = 0;
	}
	else
	{
		const long long base = k * (long long) L.in_len;
		B.q0 = base / L.up;
		B.r0 = (int) (base - B.q0 * L.up);
	}
	return B;
}
R8B_HD double conv_sample(const ConvLaunch& L, const ConvBlockBase& B, int ch, int i)
{
	if (L.up_pow2)
	{
		const int iln = L.in_len / L.up;
		return src_load(L.src, ch, i < iln ? B.q0 + i : B.q0 + i - L.n_in);
	}
	const int d = i < L.in_len ? i : i - L.bl2; // (>= -bl2)
	const int bias = (L.bl2 / L.up + 1) * L.up;   // a multiple of up that makes the dividend positive
	const unsigned e = (unsigned) (B.r0 + d + bias);
	const unsigned q = L.up == 3 ? e / 3u : e / (unsigned) L.up;
	if (e - q * (unsigned) L.up != 0u) return 0.0;
	const long long pos = B.q0 + (long long) q - bias / L.up;
	return pos >= 0 ? src_load(L.src, ch, pos) : 0.0;
}

// sub-block h (0 / 1) of block k behind the first radix-2 DIF stage (dif_pass_one<2> at n = N, in the same order of
// operations), N / 2 complex into zl
R8B_HD void conv_load_r2(const ConvLaunch& L, cd* zl, long long k, int ch, int h, int tid, int nthr)
{
	const int N = L.n_in / 2, Nh = N / 2;
	const int ts = L.tw_len / N;
	const ConvBlockBase B = conv_block_base(L, k);
	for (int j = tid; j < Nh; j += nthr)
	{
		const double ar = conv_sample(L, B, ch, 2 * j), ai = conv_sample(L, B, ch, 2 * j + 1);
		const double br = conv_sample(L, B, ch, 2 * (j + Nh)), bi = conv_sample(L, B, ch, 2 * (j + Nh) + 1);
		cd o;
		if (h == 0)
		{
			o.re = ar + br;
			o.im = ai + bi;
		}
		else
		{
			const double dr = ar - br, di = ai - bi;
			const cd w = L.tw[(long) ts * j];
			o.re = dr * w.re - di * w.im;
			o.im = dr * w.im + di * w.re;
		}
		zl[SwXor::at(j)] = o;
	}
}

// the spectral stage of sub-block h: backward bins kp = h (mod 2) from the sub-block's transform in zl
R8B_HD void conv_spectral_half(const ConvLaunch& L, const cd* zl, cd* zb, int h, int tid, int nthr)
{
	ZHalf z;
	z.zl = zl;
	z.logN = ilog2(L.n_in / 2);
	conv_spectral_z(L, z, zb, h, 2, tid, nthr);
}

// K7: emit the valid part of block k that falls into the call's output range [L.a, L.b)
// (reference CDSPBlockConvolver.h:512-593)
template<class SW>
R8B_HD void conv_store_sw(const ConvLaunch& L, const double* y, long long k, int ch, int tid, int nthr)
{
	const long long t0 = k * (long long) L.in_len - L.fl2; // first time (virtual rate) of the block
	const long long t1 = t0 + L.in_len;
	long long q0 = t0 <= 0 ? 0 : (t0 + L.down - 1) / L.down;
	long long q1 = t1 <= 0 ? 0 : (t1 + L.down - 1) / L.down;
	if (q0 < L.a) q0 = L.a;
	if (q1 > L.b) q1 = L.b;
	const long long kb = k * (long long) L.in_len;
	for (long long q = q0 + tid; q < q1; q += nthr)
	{
		long long c = q * L.down - kb;
		if (c < 0) c += L.bl2;
		const int idx = L.down_pow2 ? (int) (c / L.down) : (int) c;
		// (real sample idx = part idx & 1 of complex element idx >> 1)
		dst_store(L.dst, ch, q, y[2 * SW::at(idx >> 1) + (idx & 1)]);
	}
}
R8B_HD void conv_store(const ConvLaunch& L, const double* y, long long k, int ch, int tid, int nthr)
{
	conv_store_sw<SwNone>(L, y, k, ch, tid, nthr);
}

// ------------------------------------------------------------------------------------ interpolators

// whole-step polyphase FIR: tile of outputs [j0, j1) of channel ch; x tile staged in `xs`
R8B_HD void whole_tile_span(const WholeLaunch& L, long long j0, long long j1, long long* lo,
	int* len)
{
	const long long r0 = (j0 * L.in_step + L.pos0) / L.out_step - L.fll;
	const long long r1 = ((j1 - 1) * L.in_step + L.pos0) / L.out_step + L.fl2;
	*lo = r0;
	*len = (int) (r1 - r0 + 1);
}

static const int kWholePad = 32; // zeros behind a tile's input span (read by the zero-padded rows)

R8B_HD void whole_load(const WholeLaunch& L, double* xs, long long lo, int len, int ch, int tid,
	int nthr)
{
	const SrcBlock sb = src_block(L.src, ch, lo);
	src_block_stage<4>(sb, xs, len, len, tid, nthr);
	for (int i = tid; i < kWholePad; i += nthr) xs[len + i] = 0.0;
}

// A thread keeps ONE phase (outputs j0 + t, j0 + t + Out, ...: the same polyphase row, input positions In
// apart): the position arithmetic (a 64-bit division) and the row fetch happen once per thread and tile,
// not once per output, and the row lives in registers.  Fewer phases than threads: lanes share a phase in
// group sets; more: a thread walks phases t, t + nthr, ...
#ifndef R8B_SCHED_FENCE
#define R8B_SCHED_FENCE() // (a scheduling barrier in the GPU build, r8b_kernels.hip)
#endif

template<int FLENP>
R8B_HD void whole_compute_t(const WholeLaunch& L, const double* xs, long long lo, long long j0,
	long long j1, int ch, int tid, int nthr)
{
	const int Out = L.out_step, In = L.in_step;
	const int nsets = nthr >= Out ? nthr / Out : 1;
	const int set = nthr >= Out ? tid / Out : 0;
	if (set >= nsets) return;
	const int ustep = nsets * In;
	const long long jstep = (long long) nsets * Out;
	for (int t = nthr >= Out ? tid - set * Out : tid; t < Out; t += nthr)
	{
		long long j = j0 + t + (long long) set * Out;
		if (j >= j1) continue;
		const long long p = j * In + L.pos0;
		const long long r = p / Out;
		const int ph = (int) (p - r * Out);
		double row[FLENP];
		if (L.wtab != nullptr)
		{
			// transposed table in output order: the lanes of a wave (consecutive outputs) read consecutive
			// doubles per tap (row-major, every lane fetched its own 192-byte row: 64 cache lines per load)
			const double* tc = L.wtab + (unsigned) (ph * L.inv_in) % (unsigned) Out;
#pragma unroll
			for (int i = 0; i < FLENP; i++) row[i] = i < L.flen ? tc[(long) i * Out] : 0.0;
		}
		else
		{
			const double* tr = L.table + (long) ph * L.flen;
#pragma unroll
			for (int i = 0; i < FLENP; i++) row[i] = i < L.flen ? tr[i] : 0.0;
		}
		int u = (int) (r - L.fll - lo);
		for (; j < j1; j += jstep, u += ustep)
		{
			const LdsWin x = lds_win(xs + u);
			// the window first, then the sums: left to itself the compiler gave all taps ONE register and
			// drained the LDS queue before every multiply-add (an exposed round trip per tap); in two halves from
			// 24 taps on (the whole window live would cost a resident wave per SIMD)
			constexpr int CH = FLENP >= 24 ? FLENP / 2 : FLENP;
			double s0 = 0.0, s1 = 0.0;
#pragma unroll
			for (int c = 0; c < FLENP; c += CH)
			{
				double xv[CH];
#pragma unroll
				for (int i = 0; i < CH; i++) xv[i] = x[c + i];
				R8B_SCHED_FENCE();
#pragma unroll
				for (int i = 0; i < CH; i += 2)
				{
					s0 += row[c + i] * xv[i];
					s1 += row[c + i + 1] * xv[i + 1];
				}
				R8B_SCHED_FENCE();
			}
			dst_store(L.dst, ch, j, s0 + s1);
		}
	}
}

R8B_HD void whole_compute(const WholeLaunch& L, const double* xs, long long lo, long long j0,
	long long j1, int ch, int tid, int nthr)
{
	if (L.flen <= 8) whole_compute_t<8>(L, xs, lo, j0, j1, ch, tid, nthr);
	else if (L.flen <= 16) whole_compute_t<16>(L, xs, lo, j0, j1, ch, tid, nthr);
	else if (L.flen <= 24) whole_compute_t<24>(L, xs, lo, j0, j1, ch, tid, nthr);
	else whole_compute_t<32>(L, xs, lo, j0, j1, ch, tid, nthr);
}

// polynomial-interpolated bank: output number i of this call (absolute index L.a + i)
R8B_HD void poly_position(const PolyLaunch& L, long long i, long long* rpos, double* fpos)
{
	// bit-identical to the host plan's counter (r8b_plan.cpp): no fused multiply-add here
#pragma clang fp contract(off)
	if (i == 0)
	{
		*rpos = L.rpos0;
		*fpos = L.fpos0;
		return;
	}
	const double nxt = ((double) (L.counter0 + i) + L.shift) * L.ssr / L.dsr;
	const long long ni = (long long) nxt;
	*rpos = L.rpos0 + (ni - L.pos_int0);
	*fpos = nxt - (double) ni;
}

R8B_HD double poly_one(const PolyLaunch& L, int ch, long long i)
{
	long long rpos;
	double fpos;
	poly_position(L, i, &rpos, &fpos);
	double x, x2;
	int fti;
	{
#pragma clang fp contract(off)
		x = fpos * L.fracs;
		fti = (int) x;
		x -= fti;
		x2 = x * x;
	}
	const double* c = L.table + (long) fti * L.flen * 3;
	double s = 0.0;
	for (int t = 0; t < L.flen; t++)
	{
		const double coef = c[0] + c[1] * x + c[2] * x2;
		s += coef * src_load(L.src, ch, rpos - L.fll + t);
		c += 3;
	}
	return s;
}

// Tiled form: a workgroup takes kPolyTC channels x kPolyTO consecutive outputs.  All channels
// follow the same position schedule, so the interpolated taps c0 + c1 x + c2 x^2 (reference
// CDSPFracInterpolator.h:1088-1150) depend on the output index only: they are evaluated once per
// output into LDS (cf[o * poly_cf_pitch + t], plus the x-row offset of output o) and shared by the 16
// channels; each channel's input span is staged in LDS (row pitch odd against bank conflicts).
// In the compute phase thread (o, g) pulls output o's taps into registers once and walks channels
// g, g+4, ...: one LDS read per multiply-add, and a wave stores 64 consecutive outputs of one
// channel (lanes = channels, as before, wrote 64 scattered 8-byte words per store).
#ifndef R8B_POLY_TC
#define R8B_POLY_TC 16
#endif
static const int kPolyTC = R8B_POLY_TC; // channels per workgroup
static const int kPolyTO = 64; // outputs per workgroup
static const int kPolyPad = 8; // zeroed doubles behind a row's span (engine: span_max includes them)

// doubles per output in the tap table cf[o * pitch + t]: odd, so that the 16 consecutive outputs a lane group
// reads for one tap and the consecutive taps of an output a lane group writes both spread over the banks
constexpr int poly_cf_pitch(int flen) { return flen | 1; }

constexpr int poly_lds_doubles(int pitch, int flen)
{
	return pitch * kPolyTC + kPolyTO * poly_cf_pitch(flen) + 3 * kPolyTO;
}

// input span [lo, lo + len) needed by outputs i0 .. i1-1 of this call
R8B_HD void poly_tile_span(const PolyLaunch& L, long long i0, long long i1, long long* lo, int* len)
{
	long long r0, r1;
	double f;
	poly_position(L, i0, &r0, &f);
	poly_position(L, i1 - 1, &r1, &f);
	*lo = r0 - L.fll;
	*len = (int) (r1 + L.fl2 - *lo + 1);
}

R8B_HD void poly_tile_load(const PolyLaunch& L, double* xs, int pitch, long long lo, int len, int ch0,
	int tid, int nthr)
{
	// nthr / kPolyTC lanes per channel row, all rows at once: a thread's loads (5 per round) are in flight
	// together (one row after the other was 16 dependent HBM round trips per workgroup)
	const int lpc = nthr / kPolyTC;
	const int c = tid / lpc;
	if (c >= kPolyTC || ch0 + c >= L.nch) return;
	const SrcBlock sb = src_block(L.src, ch0 + c, lo);
	src_block_stage<5>(sb, xs + c * pitch, len, len, tid - c * lpc, lpc);
	// the compute loop runs over the tap count rounded up to 8 (zero taps): what it reads behind the span
	// must be finite
	const int l = tid - c * lpc;
	if (l < kPolyPad) xs[c * pitch + len + l] = 0.0;
}

// The same in two halves for the pipelined kernel: the loads of the NEXT tile are issued into registers
// before the current tile's arithmetic and stored to LDS after it (spans up to kPolyNV elements per lane).
static const int kPolyNV = 12;

// (16 lanes per channel row, all 16 rows at once.  Measured and not kept: a wave reading whole rows, 64
// consecutive samples per load -- 5 cache lines instead of up to 32 per instruction, and twice as slow: a
// load then goes to one or two memory channels instead of sixteen.)
R8B_HD void poly_tile_fetch(const PolyLaunch& L, long long lo, int len, int ch0, int tid, int nthr,
	double (&v)[kPolyNV])
{
	const int lpc = nthr / kPolyTC;
	const int c = tid / lpc, l = tid - c * lpc;
	if (c >= kPolyTC || ch0 + c >= L.nch) return;
	const SrcBlock sb = src_block(L.src, ch0 + c, lo);
#pragma unroll
	for (int u = 0; u < kPolyNV; u++)
	{
		const int i = l + u * lpc;
		v[u] = src_block_load1(sb, i < len ? i : len - 1);
	}
}

R8B_HD void poly_tile_commit(const PolyLaunch& L, double* xs, int pitch, int len, int ch0, int tid, int nthr,
	const double (&v)[kPolyNV])
{
	const int lpc = nthr / kPolyTC;
	const int c = tid / lpc, l = tid - c * lpc;
	if (c >= kPolyTC || ch0 + c >= L.nch) return;
#pragma unroll
	for (int u = 0; u < kPolyNV; u++)
	{
		const int i = l + u * lpc;
		if (i < len) xs[c * pitch + i] = v[u];
	}
	if (l < kPolyPad) xs[c * pitch + len + l] = 0.0;
}

// per output: x-row offset, bank entry and its fractional argument (xoff[o], xoff[64 + o],
// xoff[128 + o]) -- the position arithmetic (an fp64 division among it) once per output, not once
// per tap
R8B_HD void poly_tile_pos_write(const PolyLaunch& L, double* xoff, long long lo, int o, long long rpos,
	double fpos)
{
	double x;
	int fti;
	{
#pragma clang fp contract(off)
		x = fpos * L.fracs;
		fti = (int) x;
		x -= fti;
	}
	xoff[o] = (double) (rpos - L.fll - lo);
	xoff[kPolyTO + o] = (double) fti;
	xoff[2 * kPolyTO + o] = x;
}

R8B_HD void poly_tile_pos(const PolyLaunch& L, double* xoff, long long lo, long long i0, long long i1,
	int tid, int nthr)
{
	const int nout = (int) (i1 - i0);
	for (int o = tid; o < nout; o += nthr)
	{
		long long rpos;
		double fpos;
		poly_position(L, i0 + o, &rpos, &fpos);
		poly_tile_pos_write(L, xoff, lo, o, rpos, fpos);
	}
}

R8B_HD void poly_tile_coefs(const PolyLaunch& L, double* cf, const double* xoff, long long i0,
	long long i1, int tid, int nthr)
{
	// record tid, tid + nthr, ... of the tile's nout x flen bank records: consecutive lanes read consecutive
	// 24-byte records of a row (see poly_tile_front)
	const int nout = (int) (i1 - i0);
	const int nrec = nout * L.flen;
	const int cfp = poly_cf_pitch(L.flen);
	const float rfl = 1.0f / (float) L.flen;
	constexpr int U = 4;
	for (int T0 = tid; T0 < nrec; T0 += U * nthr)
	{
		double c0[U], c1[U], c2[U], xr[U];
		int slot[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int T = T0 + u * nthr;
			const int Tc = T < nrec ? T : T0;
			const int o = (int) (((float) Tc + 0.5f) * rfl);
			const int t = Tc - o * L.flen;
			const int fti = (int) xoff[kPolyTO + o];
			const double* c = L.table + ((long) fti * L.flen + t) * 3;
			c0[u] = c[0];
			c1[u] = c[1];
			c2[u] = c[2];
			xr[u] = xoff[2 * kPolyTO + o];
			slot[u] = T < nrec ? o * cfp + t : -1;
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			if (slot[u] < 0) continue;
			double x2;
			{
#pragma clang fp contract(off)
				x2 = xr[u] * xr[u];
			}
			cf[slot[u]] = c0[u] + c1[u] * xr[u] + c2[u] * x2;
		}
	}
}

// Front half of a tile as ONE phase: the samples and the bank entries are fetched together (both only need
// the output positions, which every wave has computed lane = output), then stored to LDS -- load, barrier,
// table fetch, barrier in turn exposed two memory round trips (8 000 + 5 500 cycles of a 18 000-cycle
// workgroup).  (rpos, fpos): position of output (tid % 64, clamped to the tile) as poly_position gives it;
// fpos_of(o): the fractional position of output o of the tile (on the GPU a lane shuffle within the wave).
// The bank entries of the tile -- nout rows of flen (c0, c1, c2) records, 24 bytes each -- are fetched record
// tid, tid + nthr, ...: consecutive lanes read consecutive records of a row.  (One thread per (output, tap
// class) read 64 different rows per load instruction whenever the step is not close to an integer -- the
// bank row then changes with every output -- and the fetch, not the samples, set the kernel's time:
// 96000 -> 44111 0.174 ms against 0.104 for 88200 -> 44101.)
template<class FposOf>
R8B_HD void poly_tile_front(const PolyLaunch& L, double* xs, int pitch, double* cf, double* xoff, long long lo,
	int len, int nout, long long rpos, double fpos, const FposOf& fpos_of, int ch0, int tid, int nthr)
{
	double v[kPolyNV];
	poly_tile_fetch(L, lo, len, ch0, tid, nthr, v);
	constexpr int UR = 8; // records per thread: 64 outputs x up to 32 taps over 256 threads
	const int nrec = nout * L.flen;
	const int cfp = poly_cf_pitch(L.flen);
	const float rfl = 1.0f / (float) L.flen;
	double c0[UR], c1[UR], c2[UR], xr[UR];
	int slot[UR];
#pragma unroll
	for (int u = 0; u < UR; u++)
	{
		slot[u] = -1;
		if (u * nthr >= nrec) continue; // (uniform)
		const int T = tid + u * nthr;
		const int Tc = T < nrec ? T : 0;
		const int o = (int) (((float) Tc + 0.5f) * rfl); // Tc / flen (exact: Tc < 2048, the quotient's fraction >= 1/64)
		const int t = Tc - o * L.flen;
		double x;
		int fti;
		{
#pragma clang fp contract(off)
			x = fpos_of(o) * L.fracs;
			fti = (int) x;
			x -= fti;
		}
		const double* c = L.table + ((long) fti * L.flen + t) * 3;
		c0[u] = c[0];
		c1[u] = c[1];
		c2[u] = c[2];
		xr[u] = x;
		if (T < nrec) slot[u] = o * cfp + t;
	}
	poly_tile_commit(L, xs, pitch, len, ch0, tid, nthr, v);
#pragma unroll
	for (int u = 0; u < UR; u++)
	{
		if (slot[u] < 0) continue;
		double x2;
		{
#pragma clang fp contract(off)
			x2 = xr[u] * xr[u];
		}
		cf[slot[u]] = c0[u] + c1[u] * xr[u] + c2[u] * x2;
	}
	if (tid < nout) poly_tile_pos_write(L, xoff, lo, tid, rpos, fpos);
}

template<int FLENP>
R8B_HD void poly_tile_compute_t(const PolyLaunch& L, const double* xs, int pitch, const double* cf,
	const double* xoff, long long i0, long long i1, int ch0, int tid, int nthr)
{
	// a wave = 16 consecutive outputs x 4 channel classes: the 32 lanes LDS serves together read two
	// ADJACENT channel rows, whose pitch (r8b_engine.cpp poly_row_pitch) interleaves their 8-byte slots for
	// this step -- 64 outputs of one row at a step of ~2 samples were a 2- to 3-way bank conflict on
	// every read; 16 lanes still store 128 consecutive bytes of a channel
	const int ow = kPolyTO / (nthr / 64); // outputs per wave
	const int lane = tid & 63;
	const int o = (tid >> 6) * ow + lane % ow, g = lane / ow, ng = 64 / ow;
	if (i0 + o >= i1) return;
	double row[FLENP];
#pragma unroll
	for (int t = 0; t < FLENP; t++) row[t] = t < L.flen ? lds_win(cf)[o * poly_cf_pitch(L.flen) + t] : 0.0;
	const int xo = (int) xoff[o];
	for (int c = g; c < kPolyTC && ch0 + c < L.nch; c += ng)
	{
		const LdsWin xv = lds_win(xs + c * pitch + xo);
		double s0 = 0.0, s1 = 0.0;
#pragma unroll
		for (int t = 0; t < FLENP; t += 2)
		{
			// (taps beyond flen are zero; their x slots hold samples or the zeros behind the span)
			s0 += row[t] * xv[t];
			s1 += row[t + 1] * xv[t + 1];
		}
		dst_store(L.dst, ch0 + c, L.a + i0 + o, s0 + s1);
	}
}

R8B_HD void poly_tile_compute(const PolyLaunch& L, const double* xs, int pitch, const double* cf,
	const double* xoff, long long i0, long long i1, int ch0, int tid, int nthr)
{
	if (L.flen <= 8) poly_tile_compute_t<8>(L, xs, pitch, cf, xoff, i0, i1, ch0, tid, nthr);
	else if (L.flen <= 16) poly_tile_compute_t<16>(L, xs, pitch, cf, xoff, i0, i1, ch0, tid, nthr);
	else if (L.flen <= 24) poly_tile_compute_t<24>(L, xs, pitch, cf, xoff, i0, i1, ch0, tid, nthr);
	else poly_tile_compute_t<32>(L, xs, pitch, cf, xoff, i0, i1, ch0, tid, nthr);
}

// ------------------------------------------------------------------------------------ half-band

// Inner loops of the half-band kernels: a thread takes kHbIlp of its outputs per round and issues the LDS
// reads of all of them tap by tap before the first sum is needed (one output per round left every read
// exposed: a round was one LDS round trip per tap).  The sums keep their order per output.
static const int kHbIlp = 4;

// 2x up: tile of input indices [n0, n1); xs holds x[n0 - T + 1 .. n1 + T - 1 + 1)
R8B_HD void hbup_compute(const HBLaunch& L, const double* xs, long long n0, long long n1, int ch,
	int tid, int nthr)
{
	constexpr int U = kHbIlp;
	const int T = L.ntaps;
	const int cnt = (int) (n1 - n0);
	// plain fp64 destination row: pq[2 i] is output 2 (n0 + i); 16-byte aligned when that element index is even
	const bool linear = L.dst.mask == -1 && L.dst.fmt == kPcmF64;
	double* const pq = linear ? L.dst.p + ((long long) ch * L.dst.stride + (2 * n0 + L.dst.off)) : nullptr;
	const bool pair16 = linear && ((size_t) pq & 15) == 0;
	// odd element index: the aligned pairs are (odd output of an input, even output of the next input = x[1])
	const bool pair16b = linear && !pair16 && ((size_t) (pq + 1) & 15) == 0;
	for (int i0 = tid; i0 < cnt; i0 += U * nthr)
	{
		LdsWin x[U]; // x[u][0] == stream x[n]
		double s[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			x[u] = lds_win(xs + (i < cnt ? i : i0) + (T - 1));
			s[u] = 0.0;
		}
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			if (k >= T) break;
			const double f = L.taps[k];
#pragma unroll
			for (int u = 0; u < U; u++) s[u] += f * (x[u][1 + k] + x[u][-k]);
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			if (i >= cnt) continue;
			const long long q = 2 * (n0 + i);
			const bool in0 = q >= L.a && q < L.b, in1 = q + 1 >= L.a && q + 1 < L.b;
			if (pair16 && in0 && in1)
			{
				// the even/odd output pair as one 16-byte store: a wave writes 1 KB of consecutive bytes (two
				// 8-byte stores at a 16-byte lane stride each fill half of every cache line they touch)
				cd v;
				v.re = x[u][0];
				v.im = s[u];
				*reinterpret_cast<cd*>(pq + 2 * i) = v;
				continue;
			}
			if (pair16b)
			{
				// an even output is written by the pair of the input before it, unless that pair starts
				// before the range
				if (in0 && q - 1 < L.a) pq[2 * i] = x[u][0];
				if (in1)
				{
					if (q + 2 < L.b)
					{
						cd v;
						v.re = s[u];
						v.im = x[u][1];
						*reinterpret_cast<cd*>(pq + 2 * i + 1) = v;
					}
					else pq[2 * i + 1] = s[u];
				}
				continue;
			}
			if (in0) dst_store(L.dst, ch, q, x[u][0]);
			if (in1) dst_store(L.dst, ch, q + 1, s[u]);
		}
	}
}

// 2x down: tile of output indices [n0, n1) from x[2 n0 - 2T + 1 .. 2 (n1-1) + 2T - 1].  The span is staged
// DE-INTERLEAVED: element i of the span (position 2 n0 - 2T + 1 + i) with i even -- the taps' samples
// x[2n +- (2k+1)] -- at xs[i / 2], with i odd -- the centre samples x[2n] -- behind them, so that the 32
// lanes LDS serves together read 32 consecutive doubles per tap (interleaved, the stride of two doubles was a
// 2-way bank conflict on every read: 47 % of the LDS cycles of a kernel whose LDS unit was 57 % busy).
// (constexpr: callable from the kernels and from the launchers)
constexpr int hbdown_odd_base(int tile, int T)
{
	// even-slot count of the longest span, moved to 8 (mod 16) doubles: a 16-lane group of the staging
	// store writes 8 even and 8 odd slots, which then fall into different halves of the 32 store banks
	return ((tile + 2 * T + 7) & ~15) + 8;
}

constexpr int hbdown_lds_doubles(int tile, int T) { return hbdown_odd_base(tile, T) + tile + 2 * T; }

R8B_HD void hbdown_load(const HBLaunch& L, double* xs, long long n0, long long n1, int ch, int tid, int nthr)
{
	const int T = L.ntaps;
	const long long lo = 2 * n0 - (2 * T - 1);
	const int len = (int) (2 * (n1 - n0 - 1) + 1) + 2 * (2 * T - 1);
	const SrcBlock sb = src_block(L.src, ch, lo);
	SlotParity slot;
	slot.odd = hbdown_odd_base(L.tile, T);
	src_block_stage_to<4>(sb, xs, slot, len, len, tid, nthr);
}

R8B_HD void hbdown_compute(const HBLaunch& L, const double* xs, long long n0, long long n1, int ch,
	int tid, int nthr)
{
	constexpr int U = kHbIlp;
	const int T = L.ntaps;
	const int cnt = (int) (n1 - n0);
	const double* const ctr = xs + hbdown_odd_base(L.tile, T) + (T - 1); // ctr[m] == stream x[2 (n0 + m)]
	for (int i0 = tid; i0 < cnt; i0 += U * nthr)
	{
		LdsWin e[U]; // e[u][k] == x[2n + 1 + 2k], e[u][-1 - k] == x[2n - 1 - 2k]
		double s[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			const int m = i < cnt ? i : i0;
			e[u] = lds_win(xs + m + T);
			s[u] = lds_win(ctr)[m];
		}
#pragma unroll
		for (int k = 0; k < 16; k++)
		{
			if (k >= T) break;
			const double f = L.taps[k];
#pragma unroll
			for (int u = 0; u < U; u++) s[u] += f * (e[u][k] + e[u][-1 - k]);
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			if (i < cnt) dst_store(L.dst, ch, n0 + i, s[u]);
		}
	}
}

// ------------------------------------------------------------------------------------ half-band cascade
//
// Tile of last-stage outputs [q0, q1).  Working backwards, stage s (0-based) must produce its
// outputs [lo[s], hi[s]) and reads its input over [in_lo[s], in_hi[s]):
//   y[2n] = x[n], y[2n+1] = sum_k f[k] (x[n+1+k] + x[n-k])  (reference CDSPHBUpsampler.h:773-786)
// so in_lo = floor(lo/2) - (T-1), in_hi = floor((hi-1)/2) + T + 1.  Stage s+1's input range is
// stage s's output range.

struct HBCRanges
{
	long long lo[kMaxCascade], hi[kMaxCascade]; // outputs of stage s
	long long in_lo, in_hi;                      // input of stage 0
};

R8B_HD long long floor_half(long long v) { return v >> 1; } // arithmetic shift == floor for negatives

R8B_HD void hbc_ranges(const HBCascadeLaunch& L, long long q0, long long q1, HBCRanges& R)
{
	// (fully unrolled with compile-time indices: a run-time index into R puts the whole structure into
	// scratch MEMORY, and every stage then waits for it -- measured ~3 000 cycles per stage)
	long long lo = q0, hi = q1;
#pragma unroll
	for (int s = kMaxCascade - 1; s >= 0; s--)
	{
		if (s >= L.nst) continue;
		R.lo[s] = lo;
		R.hi[s] = hi;
		const int T = L.ntaps[s];
		const long long ilo = floor_half(lo) - (T - 1), ihi = floor_half(hi - 1) + T + 1;
		// (input n of stage s is output n + skip[s - 1] of the stage in front of it)
		const long long sk = s > 0 ? L.skip[s - 1] : 0;
		lo = ilo + sk;
		hi = ihi + sk;
	}
	R.in_lo = lo;
	R.in_hi = hi;
}

// the output range [lo, hi) of stage s alone, walked back from the tile (run-time s: the kernel keeps its
// stage loop rolled -- one copy of each stage routine instead of one per stage -- and no table of ranges,
// which a run-time index would push into scratch memory)
R8B_HD void hbc_stage_range(const HBCascadeLaunch& L, long long q0, long long q1, int s, long long* lo,
	long long* hi)
{
	// closed form (r8b_launch.h hbc_fill_ranges; s == nst: the input span of stage 0)
	const int k = s < L.nst ? L.nst - 1 - s : L.nst;
	*lo = (q0 - L.rlo[s]) >> k;
	*hi = ((q1 - 1 + L.rhi[s]) >> k) + 1;
}

R8B_HD void hbc_load_span(const HBCascadeLaunch& L, long long in_lo, long long in_hi, double* buf, int ch,
	int tid, int nthr)
{
	const int len = (int) (in_hi - in_lo);
	const SrcBlock sb = src_block(L.src, ch, in_lo);
	const int end = clamp_rel(L.in_end - in_lo);
	src_block_stage<4>(sb, buf, len, end, tid, nthr);
}

R8B_HD void hbc_load(const HBCascadeLaunch& L, const HBCRanges& R, double* buf, int ch, int tid,
	int nthr)
{
	hbc_load_span(L, R.in_lo, R.in_hi, buf, ch, tid, nthr);
}

// one stage: input x[] (LDS, xin[0] = stream position in_lo) -> outputs [lo, hi) either into LDS
// (yout[0] = position lo) or, for the last stage, to the destination.  TP = tap count rounded up
// (4, 8 or 14): L.ntaps[] already holds the rounded count (so the input ranges cover the wider
// window) and the extra taps are zero.
// one input sample of a stage: x[0] == stream x[n0 + i]; even output x[0], odd output the filter's sum
template<int TP>
R8B_HD void hbc_point(const double (&f)[TP], LdsWin x, double& ev, double& od, double& nx)
{
	double a0 = 0.0, a1 = 0.0;
#pragma unroll
	for (int k = 0; k < TP; k += 2)
	{
		a0 += f[k] * (x[1 + k] + x[-k]);
		a1 += f[k + 1] * (x[2 + k] + x[-k - 1]);
	}
	ev = x[0];
	nx = x[1];
	od = a0 + a1;
}

// (kz: the stage's outputs below kz do not exist for the next stage -- 0, or the stage's out_skip in a chain with a
// fractional latency --: they are stored as zeros)
#ifndef R8B_HBC_SLACK
#define R8B_HBC_SLACK 1 // (0: the edge inputs of every stage through the general loop, as before -- development A/B)
#endif
// elements the cascade's LDS buffers keep free in front of their first stream element (and as many behind the last)
static const int kHbcSlack = 2;
template<int TP, bool LAST>
R8B_HD void hbc_stage_t(const HBCascadeLaunch& L, const double (&taps)[14], const double* xin, long long in_lo,
	long long lo, long long hi, double* yout, int ch, int tid, int nthr, int kz)
{
	constexpr int U = TP <= 4 ? 4 : (TP <= 8 ? 2 : 1); // inputs of a thread per round (see kHbIlp): ~32 reads in flight
	double f[TP];
#pragma unroll
	for (int k = 0; k < TP; k++) f[k] = taps[k];
	const long long n0 = floor_half(lo);
	const int cnt = (int) (floor_half(hi - 1) + 1 - n0);
	const int xoff = (int) (n0 - in_lo);  // index of x[n0] in xin
	const int qoff = (int) (2 * n0 - lo); // index of output 2*n0 relative to lo (0 or -1)
	const int nout = (int) (hi - lo);
	// A stage's stream starts at position kz (0): the outputs of input index i lie (partly) before it iff i < ineg (first
	// tile only); zlim - 2 i = how many of that input's outputs 2 (n0 + i), + 1, + 2 do
	const long long zlim = (long long) kz - 2 * n0;
	const long long izero = (zlim + 1) >> 1; // inputs i < izero have an output below kz
	const int ineg = izero <= 0 ? 0 : (izero > cnt ? cnt : (int) izero);
	// last stage into a plain fp64 row: pq[2 i] is output 2 (n0 + i)
	const bool linear = LAST && L.dst.mask == -1 && L.dst.fmt == kPcmF64;
	double* const pq = linear ? L.dst.p + ((long long) ch * L.dst.stride + (2 * n0 + L.dst.off)) : nullptr;
	const double* const xb = xin + xoff;

	// Interior of the tile, [ia, ib): both outputs of an input exist and lie inside the tile, the store is
	// one unconditional 16-byte (pair) or two 8-byte stores at a 32-bit offset -- no per-sample range tests,
	// masks or 64-bit positions (they were 7 integer instructions per useful multiply-add).  The (at most
	// two) inputs outside it and tiles at a stream's start take the general loop below.
	int ia = 0, ib = 0;
	const int mode = !LAST ? 0 : (!linear ? -1 : L.pair_ok);
	if (ineg == 0 && mode >= 0)
	{
		if (!LAST && R8B_HBC_SLACK)
		{
			// (a stage that feeds the next one through LDS: EVERY input is interior -- the two outputs that fall just outside
			// [lo, hi) land in the slack element in front of / behind the stream's buffer (kHbcSlack, k_hbcascade) and nobody
			// reads them.  The general loop below was a second, dependent LDS round trip per stage for two inputs.)
			ia = 0;
			ib = cnt;
		}
		else if (mode == 2)
		{
			// the pair (odd output of i, even output of i + 1) ends inside the tile; input 0 also owns the
			// tile's first even output: general loop
			ia = 1;
			ib = (nout - qoff - 1) / 2;
		}
		else
		{
			ia = qoff < 0 ? 1 : 0;
			ib = (nout - qoff) / 2;
		}
		if (ib > cnt) ib = cnt;
		if (ib < ia) ib = ia;
		int i = ia + tid;
		for (; i + (U - 1) * nthr < ib; i += U * nthr)
		{
			double ev[U], od[U], nx[U];
#pragma unroll
			for (int u = 0; u < U; u++) hbc_point<TP>(f, lds_win(xb + i + u * nthr), ev[u], od[u], nx[u]);
#pragma unroll
			for (int u = 0; u < U; u++)
			{
				const int o = qoff + 2 * (i + u * nthr);
				cd v;
				if (mode == 2)
				{
					v.re = od[u];
					v.im = nx[u];
					*reinterpret_cast<cd*>(pq + 2 * (i + u * nthr) + 1) = v;
				}
				else if (mode == 1)
				{
					v.re = ev[u];
					v.im = od[u];
					*reinterpret_cast<cd*>(pq + 2 * (i + u * nthr)) = v;
				}
				else if (LAST)
				{
					pq[2 * (i + u * nthr)] = ev[u];
					pq[2 * (i + u * nthr) + 1] = od[u];
				}
				else
				{
					yout[o] = ev[u];
					yout[o + 1] = od[u];
				}
			}
		}
		for (; i < ib; i += nthr)
		{
			double ev, od, nx;
			hbc_point<TP>(f, lds_win(xb + i), ev, od, nx);
			const int o = qoff + 2 * i;
			cd v;
			if (mode == 2)
			{
				v.re = od;
				v.im = nx;
				*reinterpret_cast<cd*>(pq + 2 * i + 1) = v;
			}
			else if (mode == 1)
			{
				v.re = ev;
				v.im = od;
				*reinterpret_cast<cd*>(pq + 2 * i) = v;
			}
			else if (LAST)
			{
				pq[2 * i] = ev;
				pq[2 * i + 1] = od;
			}
			else
			{
				yout[o] = ev;
				yout[o + 1] = od;
			}
		}
	}
	// everything else: inputs j = 0 .. cnt - (ib - ia) - 1 of the list with [ia, ib) cut out
	const int rest = cnt - (ib - ia);
	for (int j = tid; j < rest; j += nthr)
	{
		const int i = j < ia ? j : j + (ib - ia);
		double ev, od, nx;
		hbc_point<TP>(f, lds_win(xb + i), ev, od, nx);
		// output 2 (n0 + i) + h exists iff it is not below kz
		const double e = 2LL * i < zlim ? 0.0 : ev;
		const double d = 2LL * i + 1 < zlim ? 0.0 : od;
		const int o = qoff + 2 * i;
		if (LAST)
		{
			if (mode == 2)
			{
				// odd destination offset: the aligned pairs are (odd output of this input, even output of
				// the next = its input sample x[1]); the tile's first even output goes alone
				double* const p = pq + 2 * i;
				if (i == 0 && o >= 0 && o < nout) p[0] = e;
				if (o + 2 < nout)
				{
					cd v;
					v.re = d;
					v.im = 2LL * i + 2 < zlim ? 0.0 : nx;
					*reinterpret_cast<cd*>(p + 1) = v;
				}
				else if (o + 1 < nout) p[1] = d;
				continue;
			}
			if (mode == 1 && o >= 0 && o + 1 < nout)
			{
				// the even/odd output pair as one 16-byte store (full 128-byte lines per wave
				// instruction instead of two half-used ones)
				cd v;
				v.re = e;
				v.im = d;
				*reinterpret_cast<cd*>(pq + 2 * i) = v;
				continue;
			}
			const long long q = 2 * (n0 + i);
			if (o >= 0 && o < nout) dst_store(L.dst, ch, q, e);
			if (o + 1 >= 0 && o + 1 < nout) dst_store(L.dst, ch, q + 1, d);
		}
		else
		{
			if (o >= 0 && o < nout) yout[o] = e;
			if (o + 1 >= 0 && o + 1 < nout) yout[o + 1] = d;
		}
	}
}

// stage with T taps (already rounded up to 4 / 8 / 14) whose values the caller holds
R8B_HD void hbc_stage_f(const HBCascadeLaunch& L, const double (&taps)[14], int T, const double* xin,
	long long in_lo, long long lo, long long hi, double* yout, bool last, int ch, int tid, int nthr, int kz = 0)
{
	if (T <= 4)
	{
		if (last) hbc_stage_t<4, true>(L, taps, xin, in_lo, lo, hi, yout, ch, tid, nthr, kz);
		else hbc_stage_t<4, false>(L, taps, xin, in_lo, lo, hi, yout, ch, tid, nthr, kz);
	}
	else if (T <= 8)
	{
		if (last) hbc_stage_t<8, true>(L, taps, xin, in_lo, lo, hi, yout, ch, tid, nthr, kz);
		else hbc_stage_t<8, false>(L, taps, xin, in_lo, lo, hi, yout, ch, tid, nthr, kz);
	}
	else
	{
		if (last) hbc_stage_t<14, true>(L, taps, xin, in_lo, lo, hi, yout, ch, tid, nthr, kz);
		else hbc_stage_t<14, false>(L, taps, xin, in_lo, lo, hi, yout, ch, tid, nthr, kz);
	}
}

R8B_HD void hbc_stage(const HBCascadeLaunch& L, int s, const double* xin, long long in_lo,
	long long lo, long long hi, double* yout, bool last, int ch, int tid, int nthr)
{
	double taps[14];
	for (int k = 0; k < 14; k++) taps[k] = L.taps[s][k];
	// (what the next stage of the run does not see of this one's stream: HBCascadeLaunch::skip)
	// (has_skip: one more scalar load per stage only where there is something to load -- in a linear-phase chain it cost
	// cfg5 1.4 %: scalar loads in flight make the stage's first LDS wait a full drain)
	hbc_stage_f(L, taps, L.ntaps[s], xin, in_lo, lo, hi, yout, last, ch, tid, nthr,
		L.has_skip != 0 && !last ? L.skip[s] : 0);
}

// ------------------------------------------------------------------------------------ decimating cascade
//
// A run of 2x half-band decimators (deep decimation chains, e.g. 2822400 -> 176400 of the
// reference's bench/sacd.cpp) executed by one kernel.  Tile of last-stage outputs [q0, q1); stage s
// produces [lo[s], hi[s]) from its input over [2 lo - (2T-1), 2 (hi-1) + (2T-1)]:
//   y[n] = x[2n] + sum_k f[k] (x[2n+1+2k] + x[2n-1-2k])   (reference CDSPHBDownsampler.h:282-295)
// Uses HBCascadeLaunch (taps zero-padded to 4 / 8 / 14 like the up-sampling cascade).

R8B_HD void hbd_ranges(const HBCascadeLaunch& L, long long q0, long long q1, HBCRanges& R)
{
	long long lo = q0, hi = q1;
#pragma unroll
	for (int s = kMaxCascade - 1; s >= 0; s--)
	{
		if (s >= L.nst) continue;
		R.lo[s] = lo;
		R.hi[s] = hi;
		const int T = L.ntaps[s];
		const long long ilo = 2 * lo - (2 * T - 1), ihi = 2 * (hi - 1) + (2 * T - 1) + 1;
		// (chains with a fractional latency: input n of stage s is output n + skip[s - 1] of the stage in front of it)
		const long long sk = s > 0 ? L.skip[s - 1] : 0;
		lo = ilo + sk;
		hi = ihi + sk;
	}
	R.in_lo = lo;
	R.in_hi = hi;
}

// (in_lo: the INPUT index of xin[0] -- the previous stage's first output less its skip)
template<int TP>
R8B_HD void hbd_stage_t(const HBCascadeLaunch& L, int s, const double* xin, long long in_lo,
	long long lo, long long hi, double* yout, bool last, int ch, int tid, int nthr)
{
	// the stage's outputs below kz do not exist for the next stage (0; its out_skip in a chain with a fractional latency)
	const long long kz = last || L.has_skip == 0 ? 0 : L.skip[s];
	double f[TP];
#pragma unroll
	for (int k = 0; k < TP; k++) f[k] = L.taps[s][k];
	const int cnt = (int) (hi - lo);
	const int xoff = (int) (2 * lo - in_lo); // index of x[2 lo] in xin
	constexpr int U = TP <= 4 ? 4 : (TP <= 8 ? 2 : 1); // outputs of a thread per round (see kHbIlp)
	for (int i0 = tid; i0 < cnt; i0 += U * nthr)
	{
		double v[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			const LdsWin x = lds_win(xin + xoff + 2 * (i < cnt ? i : i0)); // x[0] == stream x[2 (lo + i)]
			double a0 = x[0], a1 = 0.0;
#pragma unroll
			for (int k = 0; k < TP; k += 2)
			{
				a0 += f[k] * (x[1 + 2 * k] + x[-1 - 2 * k]);
				a1 += f[k + 1] * (x[3 + 2 * k] + x[-3 - 2 * k]);
			}
			v[u] = a0 + a1;
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int i = i0 + u * nthr;
			if (i >= cnt) continue;
			// a stage's stream starts at position 0: earlier outputs do not exist for the next stage
			const double w = lo + i < kz ? 0.0 : v[u];
			if (last) dst_store(L.dst, ch, lo + i, w);
			else yout[i] = w;
		}
	}
}

R8B_HD void hbd_stage(const HBCascadeLaunch& L, int s, const double* xin, long long in_lo,
	long long lo, long long hi, double* yout, bool last, int ch, int tid, int nthr)
{
	const int T = L.ntaps[s];
	if (T <= 4) hbd_stage_t<4>(L, s, xin, in_lo, lo, hi, yout, last, ch, tid, nthr);
	else if (T <= 8) hbd_stage_t<8>(L, s, xin, in_lo, lo, hi, yout, last, ch, tid, nthr);
	else hbd_stage_t<14>(L, s, xin, in_lo, lo, hi, yout, last, ch, tid, nthr);
}

} // namespace r8bhip

#endif
