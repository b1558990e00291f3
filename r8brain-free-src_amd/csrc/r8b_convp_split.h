// r8b_convp_split.h -- DEVELOPMENT BUILDS ONLY (-DR8B_SPLIT_UP2, tools/variant.sh): the occupancy experiment of round 4.
// Included by r8b_convp.h behind that macro; the shipped library does not compile it.
//
// The 2x up-sampling pair kernel with 2048 -> 4096-point transforms (convolver-only modes) with its backward transform
// split into the two 2048-point transforms of the even and the odd outputs, one after the other in a 32 KB array: three
// or four workgroups fit a CU instead of two.  Correct (RMS 2.8e-16 / peak 1.6e-15 against the oracle on the emulator
// and on the GPU).  Measured (profiles/r04_experiments.txt, profiles/r04_up2_{base,split2,split3}_pmc_summary.txt):
// 25 % slower than the shipped form at two workgroups per CU (its twiddles come from the shared exp() table: 2.6x the
// L1 <- L2 requests; three more LDS round trips and two more barriers per block), and a THIRD resident workgroup buys
// 2.4 % (bench) / nothing (counter runs): waves wait longer -- lifetime 29 900 -> 40 700 cycles, "waiting to issue" 28 %
// -> 38 % of it -- while the L1's pending-request stall stays where it was.  The kernel is not occupancy bound.
#ifndef R8B_CONVP_SPLIT_H
#define R8B_CONVP_SPLIT_H

namespace r8bhip {

// ---- split backward transform (R8B_SPLIT_UP2, kSplit) -----------------------------------------------------
// Thread lt owns forward positions 8 lt .. 8 lt + 7 (bit-reversed order: bin k = bitrev3(c) 256 + bitrev8(lt) for
// position 8 lt + c).  Even outputs y[2m] = IDFT_N(Z Hs)[m], odd outputs y[2m+1] = IDFT_N(Z Hd th^k)[m], th =
// e^{+2 pi i / 2N}: th^k = conj(tw[bitrev8(lt)]) conj(tw[256 bitrev3(c)]) -- one fetched entry of the exp(-2 pi i e / 4096)
// table per thread and the 16th roots of unity as constants.  An N-point half transform (N = 2048, 8 elements per
// thread): radix 8 in registers over the thread's consecutive positions, radix 8 with sub-length 64, radix 8 with
// sub-length 512 (both inside the wave's 512 positions), barrier, radix 4 with sub-length 2048 (two butterflies per
// thread: b = lt, lt + 256 -> outputs m = b + 512 p = lt + 256 i, i = 2 p + (b >= 256)).
template<int LN, int UL>
R8B_HD void cp_split_middle(const ConvLaunch& L, const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	static_assert(G::E1 == 8 && G::RM == 4 && G::NBF == 2 && G::NT == 256, "split form: 2048 -> 4096 points");
	double zr[8], zi[8];
	const SwBase bbf = sw_base(buf, pswz(8 * lt));
#pragma unroll
	for (int c = 0; c < 8; c++)
	{
		const cd v = sw_ld(bbf, c);
		zr[c] = v.re;
		zi[c] = v.im;
	}
	dif_regs<4>(zr, zi);
	dif_regs<4>(zr + 4, zi + 4);
	// th^(bitrev8(lt)): conj of the table's entry
	unsigned r = (unsigned) lt;
	r = ((r & 0xf0u) >> 4) | ((r & 0x0fu) << 4);
	r = ((r & 0xccu) >> 2) | ((r & 0x33u) << 2);
	r = ((r & 0xaau) >> 1) | ((r & 0x55u) << 1);
	const cd wt = L.tw[(L.tw_len >> 12) * (int) r];
	// e^{+2 pi i j / 16}, j = bitrev3(c)
	constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H2 = 0.70710678118654752440;
	const double wr[8] = { 1.0, 0.0, H2, -H2, C1, -S1, S1, -C1 };
	const double wi[8] = { 0.0, 1.0, H2, H2, S1, C1, C1, S1 };
#pragma unroll
	for (int c = 0; c < 8; c++)
	{
		const double hs = st.hp[c].re, hd = st.hp[c].im;
		st.vr[c] = zr[c] * hs;
		st.vi[c] = zi[c] * hs;
		// conj(wt) * (wr + i wi)
		const double tr = wt.re * wr[c] + wt.im * wi[c], ti = wt.re * wi[c] - wt.im * wr[c];
		const double dr = zr[c] * hd, di = zi[c] * hd;
		st.vr[8 + c] = dr * tr - di * ti;
		st.vi[8 + c] = dr * ti + di * tr;
	}
}

// one half (vr / vi: 8 values at the thread's consecutive positions): radix 8 in registers, to LDS
template<int LN, int UL>
R8B_HD void cp_split_first(cd* buf, double* vr, double* vi, int lt)
{
	dit_regs<8>(vr, vi);
	const SwBase bb = sw_base(buf, pswz(8 * lt));
#pragma unroll
	for (int c = 0; c < 8; c++)
	{
		cd v;
		v.re = vr[c];
		v.im = vi[c];
		sw_st(bb, c, v);
	}
}
// radix-8 pass with sub-length n (64 or 512), in place
template<int N_>
R8B_HD void cp_split_pass(cd* buf, const cd* twr, int lt)
{
	double vr[8], vi[8];
	pdit_regs<8, true>(buf, N_, lt, twr, vr, vi);
	constexpr int q = N_ / 8;
	const int blk = lt / q, j = lt - blk * q;
	const SwBase bb = sw_base(buf, pswz(blk * N_ + j));
#pragma unroll
	for (int p = 0; p < 8; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		sw_st(bb, p * q, v);
	}
}
// last pass: radix 4, sub-length 2048, butterflies lt and lt + 256; out[i], i = 2 p + bf: the half's output lt + 256 i
template<int LN, int UL>
R8B_HD void cp_split_last(const cd* buf, const cd* twr, double* outr, double* outi, int lt)
{
#pragma unroll
	for (int bf = 0; bf < 2; bf++)
	{
		double ar[4], ai[4];
		pdit_regs<4, true>(buf, 2048, lt + 256 * bf, twr + 3 * bf, ar, ai);
#pragma unroll
		for (int p = 0; p < 4; p++)
		{
			outr[2 * p + bf] = ar[p];
			outi[2 * p + bf] = ai[p];
		}
	}
}
// outputs of both halves from the registers: E[i] = y at circular position 2 (lt + 256 i), O[i] at that + 1
template<int LN, int UL>
R8B_HD void cp_split_store(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int chA, int chB, bool bvalid,
	int lt, const DstView& pd, long long pend)
{
	constexpr int mask = ConvpGeom<LN, UL>::N2 - 1;
	const long long t0 = cx_block_t0(L, k);
	auto run = [&](const CpStoreView& v)
	{
#pragma unroll
		for (int i = 0; i < 8; i++)
		{
			const int c0 = 2 * (lt + 256 * i);
			cp_store1(v, (unsigned) ((c0 + L.fl2r) & mask), st.vr[i], st.vi[i], bvalid);
			cp_store1(v, (unsigned) ((c0 + 1 + L.fl2r) & mask), st.vr[8 + i], st.vi[8 + i], bvalid);
		}
	};
	run(cp_store_view(L.dst, chA, chB, t0, L.a, L.b, L.in_len));
	if (pend > L.b) run(cp_store_view(pd, chA, chB, t0, L.b, pend, L.in_len));
}

} // namespace r8bhip

#endif
