// r8b_convq.h -- the pair form with EIGHT elements per thread: the 2048 -> 4096-point block pair of r8b_convp.h on 512
// threads instead of 256 (round 6; VERDICT r5 "E = 8").
//
// Why.  A thread of r8b_convp.h owns 16 elements of the backward transform, i.e. 256 bytes of the workgroup's LDS
// array: a CU's 160 KB hold 640 such threads -- two 256-thread workgroups, two waves per SIMD, whatever the registers
// allow.  The convolver-only kernel k_convp<11, 1, 0, 24> (115 registers) measured on MI355X with its array truncated so
// that four workgroups fit a CU (wrong results, same instruction stream) runs 20 % faster than with two, and the same
// truncation at two workgroups per CU changes nothing (profiles/r06_experiments.txt): the pair kernels are bound by how
// few waves a SIMD has to choose from.  With eight elements per thread the same 64 KB array belongs to 512 threads, two
// workgroups per CU are sixteen waves -- four per SIMD -- and the register budget is 128.
//
// What it costs.  A pass moves log2(elements per thread) bits, so the transforms take more passes through LDS:
//   forward, 2048 points, 4 elements per thread: radix 4 x 5 (sub-lengths 2048, 512, 128, 32, 8) + the last radix-2 stage
//     in the middle pass (r8b_convp.h: 8 x 8 x 8 + radix 4 in the middle);
//   backward, 4096 points, 8 per thread: the middle pass (the folded up-sampling stage + radix 4 over the thread's eight
//     consecutive positions), then radix 8 x 3 (sub-lengths 64, 512, 4096)  (r8b_convp.h: middle + 16 x 16).
// Wave w of the eight owns forward positions [256 w, 256 w + 256) = backward positions [512 w, 512 w + 512), in one part of
// the array (forward position p at slot (p / 256) 512 + p mod 256): the passes with sub-lengths <= 256 forward and <= 512
// backward stay inside a wave and are separated by wave-level ordering points only.  Workgroup barriers per block pair:
// behind the loads (levels), behind the first forward pass, behind the second one (sub-length 512 spans two waves),
// behind the wave-local passes: 4 (r8b_convp.h with the level exchange: 3), then the last backward pass and the stores.
//
// LDS addressing.  The XOR swizzle of r8b_convp.h is built for 16 consecutive elements per lane; here the lanes of a
// 16-lane service group differ in bits {0,1,2,3} (consecutive butterflies), {0,1,2,5} (sub-length 32), {0,3,4,5} (sub-length
// 8), {2,3,4,5} (four consecutive per lane), {3,4,5,6} (eight consecutive per lane) or {0,1,2,6} (backward sub-length 64)
// of the element index.  qswz() XORs 0010 / 1101 / 1001 into the low four bits for index bits 4 / 5 / 6: every one of those
// bit sets then maps one to one onto the sixteen 16-byte bank groups (found by search over all such maps; checked by
// tests/test_emul.py test_convq_swizzle_is_conflict_free).  It is linear over XOR like pswz(), so a pass still computes
// one address per thread and reaches its other elements with an XOR constant in address bits 4-7 plus the
// instruction's immediate offset.
//
// Arithmetic: the same transform as r8b_convp.h (DIF forward to bit-reversed order, one real multiplication per bin
// with the same constants Hs = H[k] + H[k+N], Hd = H[k] - H[k+N] -- ConvLaunch::hp, read at another index --, DIT
// backward), other radices: results differ from r8b_convp.h's by rounding (1e-16), and are bitwise independent of how the
// stream is cut into calls for the same reason they are there (blocks anchored at absolute positions).
//
// Reference semantics reproduced: CDSPBlockConvolver.h:252-354, 512-593, 606-629; CDSPRealFFT.h:289-385.
#ifndef R8B_CONVQ_H
#define R8B_CONVQ_H

#include "r8b_convp.h"

namespace r8bhip {

static const int kConvqThreads = 512;
static const int kConvqN = 2048, kConvqN2 = 4096;

R8B_HD constexpr int qswz(int e)
{
	return e ^ (((e >> 4) & 1) * 2) ^ (((e >> 5) & 1) * 13) ^ (((e >> 6) & 1) * 9);
}
// forward position -> element of the (backward) array: a wave's forward data in the first half of its own part
R8B_HD constexpr int qfmap(int p) { return ((p >> 8) << 9) | (p & 255); }
R8B_HD constexpr int qsw_xc(int m) { return qswz(m) & 15; }
R8B_HD constexpr int qsw_hi(int m) { return m & ~15; }

// (the per-pass address register and the element accesses: r8b_convp.h SwBase / sw_ld / sw_st with this file's swizzle)
#if defined(R8B_LDS_ABS) && defined(__HIP_DEVICE_COMPILE__)
R8B_HD cd qsw_ld(SwBase b, int m)
{
	const lds_d2_t t = *(const lds_d2a_t*) (size_t) ((b.a ^ (unsigned) (qsw_xc(m) << 4)) + (unsigned) (qsw_hi(m) << 4));
	cd v;
	v.re = t.x;
	v.im = t.y;
	return v;
}
R8B_HD void qsw_st(SwBase b, int m, cd v)
{
	lds_d2_t t;
	t.x = v.re;
	t.y = v.im;
	*(lds_d2a_t*) (size_t) ((b.a ^ (unsigned) (qsw_xc(m) << 4)) + (unsigned) (qsw_hi(m) << 4)) = t;
}
#else
R8B_HD cd qsw_ld(SwBase b, int m)
{
	return *reinterpret_cast<const cd*>(b.p + ((b.bb ^ (qsw_xc(m) << 4)) + (qsw_hi(m) << 4)));
}
R8B_HD void qsw_st(SwBase b, int m, cd v)
{
	*reinterpret_cast<cd*>(b.p + ((b.bb ^ (qsw_xc(m) << 4)) + (qsw_hi(m) << 4))) = v;
}
#endif

struct ConvqState
{
	double vr[8], vi[8];
	double pr[4], pi[4]; // the block's input samples (channel A, channel B) of the first pass
	cd tw[4];            // base powers of the pass at hand (w, w^2, w^3, w^4)
	cd hp[4];            // (Hs, Hd) of the thread's four forward positions
	double pk[2];        // the thread's element of the previous call's parked outputs and where it goes (cp_park_slice_*)
	double* pka;
	int pf;
	double tk[2];        // the thread's element of the history tail behind the last block's window (cp_tail_slice_*)
	double* tka;
};

// LDS: the array, the flag / level words of r8b_convp.h
constexpr int convq_lds_bytes() { return kConvqN2 * 16 + kConvpFlagBytes; }

// ---- passes ---------------------------------------------------------------------------------------------------------
// forward DIF pass of radix 4 over sub-length NSUB, butterfly b of the block pair (positions are forward positions: qfmap)
template<int NSUB, bool TW>
R8B_HD void qdif4_n(cd* buf, int b, const cd* twr)
{
	constexpr int q = NSUB / 4;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * NSUB + j;
	const SwBase bb = sw_base(buf, qswz(qfmap(e0)));
	double vr[4], vi[4];
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		const cd v = qsw_ld(bb, qfmap(p * q));
		vr[p] = v.re;
		vi[p] = v.im;
	}
	dif_regs<4>(vr, vi);
	if constexpr (TW)
	{
#pragma unroll
		for (int p = 1; p < 4; p++)
		{
			const cd w = tw_get(twr, bitrev_c<4>(p));
			const double tr = vr[p] * w.re - vi[p] * w.im;
			const double ti = vr[p] * w.im + vi[p] * w.re;
			vr[p] = tr;
			vi[p] = ti;
		}
	}
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		qsw_st(bb, qfmap(p * q), v);
	}
}

// backward DIT pass of radix 8 over sub-length NSUB, butterfly b; results to vr / vi (the caller stores them or keeps them)
template<int NSUB>
R8B_HD void qdit8_regs(const cd* buf, int b, const cd* twr, double* vr, double* vi)
{
	constexpr int q = NSUB / 8;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * NSUB + j;
	const SwBase bb = sw_base(buf, qswz(e0));
#pragma unroll
	for (int p = 0; p < 8; p++)
	{
		const cd v = qsw_ld(bb, p * q);
		vr[p] = v.re;
		vi[p] = v.im;
	}
#pragma unroll
	for (int p = 1; p < 8; p++)
	{
		const cd w = tw_get(twr, bitrev_c<8>(p));
		const double tr = vr[p] * w.re + vi[p] * w.im;
		const double ti = vi[p] * w.re - vr[p] * w.im;
		vr[p] = tr;
		vi[p] = ti;
	}
	dit_regs<8>(vr, vi);
}
template<int NSUB>
R8B_HD void qdit8(cd* buf, int b, const cd* twr)
{
	constexpr int q = NSUB / 8;
	double vr[8], vi[8];
	qdit8_regs<NSUB>(buf, b, twr, vr, vi);
	const int blk = b / q, j = b - blk * q;
	const SwBase bb = sw_base(buf, qswz(blk * NSUB + j));
#pragma unroll
	for (int p = 0; p < 8; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		qsw_st(bb, p * q, v);
	}
}

// base powers w_n^(j m), m = 1 .. 4, from the shared exp() table (ConvLaunch::tw: exp(-2 pi i e / tw_len))
template<int NB>
R8B_HD void qtw_fetch(cd* twr, const ConvLaunch& L, int n, int j)
{
	const int ts = L.tw_len / n * j;
#pragma unroll
	for (int c = 0; c < NB; c++) twr[c] = R8B_TAB_LD_T(L.tw, 0, ts * (c + 1));
}

// ---- phases ---------------------------------------------------------------------------------------------------------
// K1: thread lt owns elements lt + 512 p of the first pass (cf. cp_load: the block is one window of N consecutive samples,
// rotated so that the valid outputs start at circular position fl2 mod 2)
R8B_HD void cq_load(const ConvLaunch& L, ConvqState& st, long long k, int chA, int chB, int lt)
{
	constexpr int N = kConvqN, q = N / 4;
	const int iln = L.in_len >> 1;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) >> 1;
	const int wr = (L.rot + N - iln) & (N - 1);
	if (L.src.cur_fmt == kPcmF64 && base - (N - iln) >= L.src.cur_base && base - (N - iln) >= 0)
	{
		const long long w0 = base - (N - iln) - L.src.cur_base;
		const double* const pa = L.src.cur + ((long long) chA * L.src.cur_stride + w0);
		const double* const pb = L.src.cur + ((long long) chB * L.src.cur_stride + w0);
		const unsigned l0 = (unsigned) (lt + wr);
#pragma unroll
		for (int p = 0; p < 4; p++)
		{
			const unsigned w = (l0 + (unsigned) (p * q)) & (unsigned) (N - 1);
			st.pr[p] = pa[w];
			st.pi[p] = pb[w];
		}
		return;
	}
	const SrcBlock sa = src_block(L.src, chA, base), sb = src_block(L.src, chB, base);
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		const int i = (lt + p * q + L.rot) & (N - 1);
		const int rel = i < iln ? i : i - N;
		st.pr[p] = src_block_load1(sa, rel);
		st.pi[p] = src_block_load1(sb, rel);
	}
}

R8B_HD unsigned cq_nonzero_bits(const ConvqState& st)
{
	unsigned a = 0, b = 0;
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		unsigned long long ua, ub;
		__builtin_memcpy(&ua, &st.pr[p], 8);
		__builtin_memcpy(&ub, &st.pi[p], 8);
		a |= (unsigned) ua | ((unsigned) (ua >> 32) << 1);
		b |= (unsigned) ub | ((unsigned) (ub >> 32) << 1);
	}
	return (a != 0 ? 1u : 0u) | (b != 0 ? 2u : 0u);
}
R8B_HD CpLevels cq_level_words(const ConvqState& st)
{
	CpLevels v;
	v.a = v.b = 0;
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		unsigned long long ua, ub;
		__builtin_memcpy(&ua, &st.pr[p], 8);
		__builtin_memcpy(&ub, &st.pi[p], 8);
		const unsigned ha = (unsigned) (ua >> 32) & 0x7fffffffu, hb = (unsigned) (ub >> 32) & 0x7fffffffu;
		v.a = ha > v.a ? ha : v.a;
		v.b = hb > v.b ? hb : v.b;
	}
	return v;
}

// first forward pass (sub-length 2048, radix 4) from the registers cq_load() filled; st.tw: w, w^2, w^3 of butterfly lt
R8B_HD void cq_first(cd* buf, const ConvqState& st, int lt, int lsh)
{
	double vr[4], vi[4];
	// (the quieter channel at its partner's level: r8b_convp.h cp_scale_in)
	const double sa = lsh < 0 ? cp_pow2(-lsh) : 1.0, sb = lsh > 0 ? cp_pow2(lsh) : 1.0;
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		vr[p] = lsh < 0 ? st.pr[p] * sa : st.pr[p];
		vi[p] = lsh > 0 ? st.pi[p] * sb : st.pi[p];
	}
	dif_regs<4>(vr, vi);
#pragma unroll
	for (int p = 1; p < 4; p++)
	{
		const cd w = tw_get(st.tw, bitrev_c<4>(p));
		const double tr = vr[p] * w.re - vi[p] * w.im;
		const double ti = vr[p] * w.im + vi[p] * w.re;
		vr[p] = tr;
		vi[p] = ti;
	}
	const SwBase bb = sw_base(buf, qswz(qfmap(lt)));
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		qsw_st(bb, qfmap(p * 512), v);
	}
}

// middle pass: the last forward stage (radix 2 over the thread's positions 4 lt + (0, 1), (2, 3)), the multiplication by
// the kernel with the first backward stage folded in -- forward position p feeds backward positions 2p, 2p + 1 as
// Z (H[k] + H[k+N]), Z (H[k] - H[k+N]) --, and the backward stages over the thread's eight consecutive positions
R8B_HD void cq_middle(cd* buf, ConvqState& st, int lt)
{
	double zr[4], zi[4];
	const SwBase bbf = sw_base(buf, qswz(qfmap(4 * lt)));
#pragma unroll
	for (int c = 0; c < 4; c++)
	{
		const cd v = qsw_ld(bbf, c);
		zr[c] = v.re;
		zi[c] = v.im;
	}
	dif_regs<2>(zr, zi);
	dif_regs<2>(zr + 2, zi + 2);
#pragma unroll
	for (int c = 0; c < 4; c++)
	{
		st.vr[2 * c] = zr[c] * st.hp[c].re;
		st.vi[2 * c] = zi[c] * st.hp[c].re;
		st.vr[2 * c + 1] = zr[c] * st.hp[c].im;
		st.vi[2 * c + 1] = zi[c] * st.hp[c].im;
	}
	DitSt<8, 2>::run(st.vr, st.vi);
}
R8B_HD void cq_middle_write(cd* buf, const ConvqState& st, int lt)
{
	const SwBase bb = sw_base(buf, qswz(8 * lt));
#pragma unroll
	for (int p = 0; p < 8; p++)
	{
		cd v;
		v.re = st.vr[p];
		v.im = st.vi[p];
		qsw_st(bb, p, v);
	}
}
// (Hs, Hd) of forward position 4 lt + c from the table of r8b_convp.h: entry (c', t) = position 8 t + c'
R8B_HD void cq_hp_prefetch(const ConvLaunch& L, ConvqState& st, int lt)
{
#pragma unroll
	for (int c = 0; c < 4; c++) st.hp[c] = R8B_TAB_LD_H(L.hp, (4 * (lt & 1) + c) * 256, (lt >> 1));
}

// K7 from the registers: the thread's element p is circular time lt + 512 p (cf. cp_store_conv, mode 0)
R8B_HD void cq_store_conv(const ConvLaunch& L, const ConvqState& st, long long k, int chA, int chB, bool bvalid, int lt,
	const DstView& pd, long long pend)
{
	const long long t0 = cx_block_t0(L, k);
	auto run = [&](const CpStoreView& v)
	{
#pragma unroll
		for (int p = 0; p < 8; p++)
			cp_store1(v, (unsigned) ((lt + 512 * p + L.fl2r) & (kConvqN2 - 1)), st.vr[p], st.vi[p], bvalid);
	};
	run(cp_store_view(L.dst, chA, chB, t0, L.a, L.b, L.in_len));
	if (pend > L.b) run(cp_store_view(pd, chA, chB, t0, L.b, pend, L.in_len));
}

// History for the next call out of the registers cq_load() filled (cf. cp_tail_owned)
R8B_HD void cq_tail_owned(const ConvLaunch& L, const ConvqState& st, long long k, int chA, int chB, bool bvalid, int lt)
{
	constexpr int N = kConvqN, q = N / 4;
	if (k < L.k0 + L.tail_bf) return;
	const int iln = L.in_len >> 1;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) >> 1;
	const long long nbase = ((k + 1) * (long long) L.blk_stride + L.blk_offset) >> 1;
	const long long lo = k == L.k0 + L.tail_bf || base < L.tail_c0 ? L.tail_c0 : base;
	const long long hi = k == L.k0 + L.nblk - 1 || nbase > L.tail_c1 ? L.tail_c1 : nbase;
	if (lo >= hi) return;
	const long long w0 = base - (N - iln);
	const unsigned lo_r = (unsigned) (lo - w0), n_r = (unsigned) (hi - lo);
	const unsigned m = (unsigned) L.src.ring_mask, w0m = (unsigned) (w0 & L.src.ring_mask);
	double* const ra = L.tail_ring + (long long) chA * L.src.ring_stride;
	double* const rb = L.tail_ring + (long long) chB * L.src.ring_stride;
	const unsigned l0 = (unsigned) (lt + ((L.rot + N - iln) & (N - 1)));
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		const unsigned w = (l0 + (unsigned) (p * q)) & (unsigned) (N - 1);
		if (w - lo_r < n_r)
		{
			const unsigned e = (w0m + w) & m;
			ra[e] = st.pr[p];
			if (bvalid) rb[e] = st.pi[p];
		}
	}
}

// ---- the body: one block pair by a 512-thread workgroup (convolver-only mode: the outputs go from the registers to the
// destination view).  Exec: GpuExecQ (r8b_kernels.hip) / EmulExecQ (tests/emul) -- phase(), wave_steps(), each(),
// post_bits / collect_bits, post_levels / collect_levels / post_shift / collect_shift as in r8b_convp.h
template<class Exec>
R8B_HD void convq_body(Exec& ex, const ConvxLaunch& X, const ConvxLaunch& XM, cd* buf, const ConvpItem& cur)
{
	typedef ConvqState St;
	const ConvLaunch& L = X.c;
	const int chA = cur.chA, chB = cur.chB;
	const bool bvalid = cur.bvalid;
	constexpr int WT = kConvqThreads;
	ex.phase([&](int tid, St& st)
	{
		// (the first pass's twiddles -- L2 -- requested ahead of the samples -- HBM)
		qtw_fetch<3>(st.tw, L, kConvqN, tid);
		cq_load(L, st, cur.k, chA, chB, tid);
		if ((L.tail_flags & 2) != 0 && cur.k + 1 > L.k0 + L.tail_bf)
		{
			if ((L.tail_flags & 8) == 0 && cur.k + 1 == L.k0 + L.nblk)
				cp_tail_rest<WT>(L, L.tail_c1, L.tail_p1, chA, chB, bvalid, tid);
			cq_tail_owned(L, st, cur.k, chA, chB, bvalid, tid);
		}
		st.tka = nullptr;
		if ((L.tail_flags & 8) != 0) cp_tail_slice_load<WT>(L, st, (int) (cur.k - L.k0), chA, chB, tid);
		st.pf = X.park_out != 0 && cur.k + 1 == L.k0 + L.nblk ? 1 : 0;
		st.pka = nullptr;
		if (X.park_n > 0)
		{
			if (X.park_slices != 0) cp_park_slice_load<WT>(XM, X.wdst, st, (int) (cur.k - L.k0), chA, chB, tid);
			else if (cur.k == L.k0) cp_park_back<WT>(XM, X.wdst, chA, chB, bvalid, tid);
		}
		ex.post_bits(tid, cq_nonzero_bits(st));
		ex.post_levels(tid, 0, cq_level_words(st));
	});
	ex.phase([&](int tid, St& st)
	{
		const int lsh = cp_level_shift(ex.collect_levels(0));
		ex.post_shift(tid, 0, tid, lsh);
		cq_first(buf, st, tid, lsh);
		qtw_fetch<3>(st.tw, L, 512, tid & 127);
	});
	// second forward pass: sub-length 512 spans two waves' parts of the array
	ex.phase([&](int tid, St& st)
	{
		qdif4_n<512, true>(buf, tid, st.tw);
		qtw_fetch<3>(st.tw, L, 128, tid & 31);
	});
	auto s_f3 = [&](int tid, St& st)
	{
		qdif4_n<128, true>(buf, tid, st.tw);
		qtw_fetch<3>(st.tw, L, 32, tid & 7);
	};
	auto s_f4 = [&](int tid, St& st)
	{
		qdif4_n<32, true>(buf, tid, st.tw);
		qtw_fetch<3>(st.tw, L, 8, tid & 1);
		cq_hp_prefetch(L, st, tid);
	};
	auto s_f5 = [&](int tid, St& st) { qdif4_n<8, true>(buf, tid, st.tw); };
	auto s_midc = [&](int tid, St& st)
	{
		cq_middle(buf, st, tid);
		qtw_fetch<4>(st.tw, L, 64, tid & 7);
	};
	auto s_midw = [&](int tid, St& st) { cq_middle_write(buf, st, tid); };
	auto s_b1 = [&](int tid, St& st)
	{
		qdit8<64>(buf, tid, st.tw);
		qtw_fetch<4>(st.tw, L, 512, tid & 63);
	};
	auto s_b2 = [&](int tid, St& st)
	{
		qdit8<512>(buf, tid, st.tw);
		qtw_fetch<4>(st.tw, L, kConvqN2, tid);
	};
	ex.wave_steps(s_f3, s_f4, s_f5, s_midc, s_midw, s_b1, s_b2);
	// History for the next call, the cases the registers do not cover (cf. convp_body)
	if ((L.tail_flags & 5) != 0)
	{
		const unsigned long long tn = (unsigned long long) (L.tail_p1 - L.tail_p0), nb = (unsigned long long) L.nblk;
		const unsigned long long bi = (unsigned long long) (cur.k - L.k0), be = bi + 1u;
		long long s0 = L.tail_p0 + (long long) (tn * bi / nb), s1 = L.tail_p0 + (long long) (tn * be / nb);
		if ((L.tail_flags & 2) != 0)
		{
			s0 = L.tail_p0;
			s1 = bi == 0 ? L.tail_c0 : L.tail_p0;
		}
		ex.each([&](int tid, St&)
		{
			constexpr int TB = 8;
			const SrcBlock sba = src_block(L.src, chA, s0), sbb = src_block(L.src, chB, s0);
			for (long long i0 = s0 + tid; i0 < s1; i0 += (long long) TB * WT)
			{
				double va[TB], vb[TB];
#pragma unroll
				for (int j = 0; j < TB; j++)
				{
					const long long i = i0 + (long long) j * WT;
					const int rel = (int) ((i < s1 ? i : s1 - 1) - s0);
					va[j] = src_block_load1(sba, rel);
					vb[j] = src_block_load1(sbb, rel);
				}
#pragma unroll
				for (int j = 0; j < TB; j++)
				{
					const long long i = i0 + (long long) j * WT;
					if (i < s1)
					{
						L.tail_ring[(long long) chA * L.src.ring_stride + (i & L.src.ring_mask)] = va[j];
						if (bvalid) L.tail_ring[(long long) chB * L.src.ring_stride + (i & L.src.ring_mask)] = vb[j];
					}
				}
			}
		});
	}
	ex.each([&](int tid, St& st)
	{
		cp_park_slice_store(X.wdst, st, chA, chB, bvalid);
		cp_tail_slice_store(L, st, chA, chB, bvalid);
		qdit8_regs<kConvqN2>(buf, tid, st.tw, st.vr, st.vi);
		cp_scale_out<8>(st.vr, st.vi, ex.collect_shift(0));
		const unsigned nzb = ex.collect_bits();
		if (nzb != 3u)
		{
#pragma unroll
			for (int p = 0; p < 8; p++)
			{
				if (!(nzb & 1u)) st.vr[p] = 0.0;
				if (!(nzb & 2u)) st.vi[p] = 0.0;
			}
		}
		DstView pd = L.dst;
		long long pend = L.b;
		// (the call's last block parks what lies behind the call's range: r8b_convp.h cp_park_view)
		if (ex.uniform(st.pf) != 0)
		{
			pd.p = XM.park_dst;
			pd.stride = XM.park_stride;
			pd.mask = -1;
			pd.off = -XM.c.b;
			pd.fmt = kPcmF64;
			pend = XM.park_blk.jhi;
		}
		cq_store_conv(L, st, cur.k, chA, chB, bvalid, tid, pd, pend);
	});
}

} // namespace r8bhip

#endif
