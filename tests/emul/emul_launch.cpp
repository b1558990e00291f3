// tests/emul/emul_launch.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host stand-in for r8b_kernels.hip so that the engine's schedule and the kernels' index
// arithmetic can be unit-tested in a container without a GPU: it includes the very same
// r8b_kernel_phases.h and runs every phase for tid = 0..nthr-1 in a loop, one loop per
// barrier-separated phase ("device memory" is host memory).  It is built only by
// tests/emul/Makefile into tests/emul/_build/libr8bsrc_emul.so and loaded only by tests/ --
// never by the package, bench.py or __graft_entry__.  It is NOT a CPU fallback of the product:
// libr8bsrc_hip.so does not contain it and fails loudly without a HIP device.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <cstring>
#include <stdexcept>
#include <vector>

#define R8B_HD inline
// the GPU reads the tap window with inline-asm LDS loads; here it is a plain copy
#define R8B_LDS_WINDOW(N, v, p) { for (int i_ = 0; i_ < (N); i_++) (v)[i_] = (p)[i_]; }
#define R8B_LDS_ARRIVED(N, v, o)
#include "r8b_kernel_phases.h"
#include "r8b_convx.h"
#include "r8b_convp.h"
#include "r8b_convq.h"
#include "r8b_pcm.h"

namespace r8bhip {

// the long-block form (k_conv_big): the same phases in the kernel's order
static void launch_conv_big(const ConvLaunch& L)
{
	const int Nh = L.n_in / 4, N2 = L.n_out / 2, nthr = L.threads;
	std::vector<cd> zl((size_t) std::max(Nh, N2)), zg((size_t) N2);
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < L.nblk; bx++)
		{
			const long long k = L.k0 + bx;
			for (int h = 0; h < 2; h++)
			{
				for (int t = 0; t < nthr; t++) conv_load_r2(L, zl.data(), k, ch, h, t, nthr);
				int n = Nh;
				for (int p = 1; p < L.n_fwd; p++)
				{
					for (int t = 0; t < nthr; t++) fft_pass_sw<SwXor>(zl.data(), Nh, n, L.fwd_radix[p], false, L.tw, L.tw_len, t, nthr);
					n /= L.fwd_radix[p];
				}
				for (int t = 0; t < nthr; t++) conv_spectral_half(L, zl.data(), zg.data(), h, t, nthr);
			}
			for (int i = 0; i < N2; i++) zl[(size_t) SwXor::at(i)] = zg[(size_t) i];
			int n = 1;
			for (int p = 0; p < L.n_inv; p++)
			{
				n *= L.inv_radix[p];
				for (int t = 0; t < nthr; t++) fft_pass_sw<SwXor>(zl.data(), N2, n, L.inv_radix[p], true, L.tw, L.tw_len, t, nthr);
			}
			for (int t = 0; t < nthr; t++)
				conv_store_sw<SwXor>(L, reinterpret_cast<const double*>(zl.data()), k, ch, t, nthr);
		}
}

void launch_conv(const ConvLaunch& L, void*)
{
	if (L.work != nullptr)
	{
		launch_conv_big(L);
		return;
	}
	std::vector<double> lds((size_t) (L.n_in + L.n_out) + 2);
	// 16-byte alignment for the cd views
	double* base = lds.data();
	if (((size_t) base & 15) != 0) base++;
	const int nthr = L.threads;
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < L.nblk; bx++)
		{
			double* ra = base;
			cd* za = reinterpret_cast<cd*>(ra);
			double* rb = ra + (L.inplace ? 0 : L.n_in);
			cd* zb = reinterpret_cast<cd*>(rb);
			const long long k = L.k0 + bx;
			for (int t = 0; t < nthr; t++) conv_load(L, ra, k, ch, t, nthr);
			const int N = L.n_in / 2;
			int n = N;
			for (int p = 0; p < L.n_fwd; p++)
			{
				for (int t = 0; t < nthr; t++)
					fft_pass(za, N, n, L.fwd_radix[p], false, L.tw, L.tw_len, t, nthr);
				n /= L.fwd_radix[p];
			}
			for (int t = 0; t < nthr; t++) conv_spectral(L, za, zb, t, nthr);
			const int N2 = L.n_out / 2;
			n = 1;
			for (int p = 0; p < L.n_inv; p++)
			{
				n *= L.inv_radix[p];
				for (int t = 0; t < nthr; t++)
					fft_pass(zb, N2, n, L.inv_radix[p], true, L.tw, L.tw_len, t, nthr);
			}
			for (int t = 0; t < nthr; t++) conv_store(L, rb, k, ch, t, nthr);
		}
}

void launch_whole(const WholeLaunch& L, void*)
{
	std::vector<double> xs((size_t) L.span_max);
	const int nthr = 256;
	const long long n = L.b - L.a;
	const int tiles = (int) ((n + L.tile - 1) / L.tile);
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < tiles; bx++)
		{
			const long long j0 = L.a + (long long) bx * L.tile;
			long long j1 = j0 + L.tile;
			if (j1 > L.b) j1 = L.b;
			long long lo;
			int len;
			whole_tile_span(L, j0, j1, &lo, &len);
			if (len + kWholePad > L.span_max) throw std::runtime_error("emul: whole-step tile overflows LDS");
			for (int t = 0; t < nthr; t++) whole_load(L, xs.data(), lo, len, ch, t, nthr);
			for (int t = 0; t < nthr; t++) whole_compute(L, xs.data(), lo, j0, j1, ch, t, nthr);
		}
}

void launch_poly(const PolyLaunch& L, void*)
{
	if (L.span_max > 0)
	{
		const int nthr = 256, pitch = L.pitch;
		std::vector<double> xs((size_t) poly_lds_doubles(L.pitch, L.flen));
		double* const cf = xs.data() + pitch * kPolyTC;
		double* const xoff = cf + kPolyTO * poly_cf_pitch(L.flen);
		const long long n = L.b - L.a;
		for (int by = 0; by < (L.nch + kPolyTC - 1) / kPolyTC; by++)
			for (long long i0 = 0; i0 < n; i0 += kPolyTO)
			{
				for (double& v : xs) v = std::numeric_limits<double>::quiet_NaN();
				const long long i1 = std::min(n, i0 + kPolyTO);
				long long lo;
				int len;
				poly_tile_span(L, i0, i1, &lo, &len);
				if (len + kPolyPad > L.span_max) throw std::runtime_error("emul: poly tile span overflows LDS");
				if (L.front)
				{
					// (the kernel's single front phase; a lane's own output position as the wave computes it)
					if (len > 16 * kPolyNV) throw std::runtime_error("emul: poly tile span beyond the front phase");
					const int nout = (int) (i1 - i0);
					for (int t = 0; t < nthr; t++)
					{
						const int o = (t & 63) < nout ? (t & 63) : nout - 1;
						long long rpos;
						double fpos;
						poly_position(L, i0 + o, &rpos, &fpos);
						auto fpos_of = [&](int oo)
						{
							long long r;
							double f;
							poly_position(L, i0 + oo, &r, &f);
							return f;
						};
						poly_tile_front(L, xs.data(), pitch, cf, xoff, lo, len, nout, rpos, fpos, fpos_of,
							by * kPolyTC, t, nthr);
					}
				}
				else
				{
					for (int t = 0; t < nthr; t++)
						poly_tile_load(L, xs.data(), pitch, lo, len, by * kPolyTC, t, nthr);
					for (int t = 0; t < nthr; t++) poly_tile_pos(L, xoff, lo, i0, i1, t, nthr);
					for (int t = 0; t < nthr; t++) poly_tile_coefs(L, cf, xoff, i0, i1, t, nthr);
				}
				for (int t = 0; t < nthr; t++)
					poly_tile_compute(L, xs.data(), pitch, cf, xoff, i0, i1, by * kPolyTC, t, nthr);
			}
		return;
	}
	for (int ch = 0; ch < L.nch; ch++)
		for (long long i = 0; L.a + i < L.b; i++) dst_store(L.dst, ch, L.a + i, poly_one(L, ch, i));
}

// the second grid layer of a launch that carries a history copy (HBLaunch::tail), or k_tail's own grid
static void emul_carried_tail(const TailLaunch& T, int blocks, int nch)
{
	for (int ch = 0; ch < nch; ch++)
		for (int w = 0; w < blocks; w++)
			for (int t = 0; t < 256; t++) tail_copy(T, w, blocks, ch, t, 256);
}

void launch_hbup(const HBLaunch& L, void*)
{
	const int nthr = 256, T = L.ntaps;
	std::vector<double> xs((size_t) (L.tile + 2 * T));
	const long long nb = L.a / 2, ne = (L.b + 1) / 2;
	const int tiles = (int) ((ne - nb + L.tile - 1) / L.tile);
	if (L.carry_tail) emul_carried_tail(L.tail, tiles, L.nch);
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < tiles; bx++)
		{
			const long long n0 = nb + (long long) bx * L.tile;
			long long n1 = n0 + L.tile;
			if (n1 > ne) n1 = ne;
			const long long lo = n0 - (T - 1);
			const int len = (int) (n1 - n0) + 2 * T - 1;
			for (int i = 0; i < len; i++) xs[(size_t) i] = src_load(L.src, ch, lo + i);
			for (int t = 0; t < nthr; t++) hbup_compute(L, xs.data(), n0, n1, ch, t, nthr);
		}
}

void launch_hbdown(const HBLaunch& L, void*)
{
	const int nthr = 256, T = L.ntaps;
	std::vector<double> xs((size_t) hbdown_lds_doubles(L.tile, T));
	const long long n = L.b - L.a;
	const int tiles = (int) ((n + L.tile - 1) / L.tile);
	if (L.carry_tail) emul_carried_tail(L.tail, tiles, L.nch);
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < tiles; bx++)
		{
			const long long n0 = L.a + (long long) bx * L.tile;
			long long n1 = n0 + L.tile;
			if (n1 > L.b) n1 = L.b;
			for (int t = 0; t < nthr; t++) hbdown_load(L, xs.data(), n0, n1, ch, t, nthr);
			for (int t = 0; t < nthr; t++) hbdown_compute(L, xs.data(), n0, n1, ch, t, nthr);
		}
}

template<int LOGN, int UPLOG>
struct EmulExec
{
	std::vector<ConvxState<LOGN, UPLOG>> st;
	EmulExec() : st((size_t) kConvxThreads) {}
	template<class F>
	void phase(F f)
	{
		for (int t = 0; t < kConvxThreads; t++) f(t, st[(size_t) t]);
	}
	// wave-local pair: all lanes of a wave finish the first step before any starts the second;
	// waves run one after the other (they must not depend on each other inside the pair)
	template<class FA, class FB>
	void wave_phase2(FA fa, FB fb)
	{
		for (int w = 0; w < kConvxThreads / 64; w++)
		{
			for (int t = 64 * w; t < 64 * w + 64; t++) fa(t, st[(size_t) t]);
			for (int t = 64 * w; t < 64 * w + 64; t++) fb(t, st[(size_t) t]);
		}
	}
	template<class F>
	void each(F f)
	{
		for (int t = 0; t < kConvxThreads; t++) f(t, st[(size_t) t]);
	}

};

template<int LOGN, int UPLOG, int MODE, int FLENP>
void emul_convx_t(const ConvxLaunch& X)
{
	std::vector<double> lds((size_t) convx_lds_need(UPLOG > 0 ? LOGN + UPLOG : LOGN, X.c.in_len, MODE) + 2);
	double* base = lds.data();
	if (((size_t) base & 15) != 0) base++;
	for (int ch = 0; ch < X.c.nch; ch++)
		for (int bx = 0; bx < X.c.nblk; bx++)
		{
			// poison the LDS (pad slots are never written by the kernel and must never matter)
			for (double& v : lds) v = std::numeric_limits<double>::quiet_NaN();
			EmulExec<LOGN, UPLOG> ex;
			convx_body<LOGN, UPLOG, MODE, FLENP>(ex, X, base, X.c.k0 + bx, ch);
		}
}

// pair form: a phase runs for all 256 threads before the next one starts
template<int LN, int UL>
struct EmulExecP
{
	static constexpr int WT = ConvpGeom<LN, UL>::WT;
	std::vector<ConvpState<LN, UL>> st;
	unsigned bits = 0;
	// the blocks' level words (r8b_convp.h cp_level_bits): field-wise maximum over a block's threads
	unsigned lv[ConvpGeom<LN, UL>::SUB];
	EmulExecP() : st((size_t) WT) { next_block(); }
	void stamp2() {}
	void post_bits(int, unsigned v) { bits |= v; }
	unsigned collect_bits() const { return bits; }
	int shift[ConvpGeom<LN, UL>::SUB];
	void post_levels(int, int sub, CpLevels v) { lv[sub] = cp_level_max(lv[sub], cp_level_pack(v)); }
	unsigned collect_levels(int sub) const { return lv[sub]; }
	void post_shift(int, int sub, int, int d) { shift[sub] = d; }
	int collect_shift(int sub) const { return shift[sub]; }
	int uniform(int v) const { return v; }
	void next_block()
	{
		bits = 0;
		for (unsigned& v : lv) v = 0;
		for (int& v : shift) v = 0;
	}
	template<class F>
	void phase(F f)
	{
		for (int t = 0; t < WT; t++) f(t, st[(size_t) t]);
	}
	template<class F>
	void each(F f)
	{
		for (int t = 0; t < WT; t++) f(t, st[(size_t) t]);
	}
	// wave-local steps: a wave runs ALL its steps before the next wave starts (so any dependence on
	// another wave's data inside the sequence shows as a wrong result); within a step the lanes run
	// in DESCENDING order in every second wave, so that read-before-write inside a step is exercised
	// (the hardware issues a step's reads for all lanes before its writes -- modelled by two passes
	// where a step both reads and overwrites other lanes' slots, see run_step)
	template<class F>
	void run_step(F& f, int w)
	{
		for (int t = 64 * w; t < 64 * w + 64; t++) f(t, st[(size_t) t]);
	}
	template<class... F>
	void wave_steps(F... f)
	{
		for (int w = 0; w < WT / 64; w++) (run_step(f, w), ...);
	}
};

// eight elements per thread (r8b_convq.h): 512 threads per block pair; waves of 64 run their wave-local steps one wave
// after the other, like EmulExecP
struct EmulExecQ
{
	static constexpr int WT = kConvqThreads;
	std::vector<ConvqState> st;
	unsigned bits = 0, lv = 0;
	int shift = 0;
	EmulExecQ() : st((size_t) WT) {}
	void post_bits(int, unsigned v) { bits |= v; }
	unsigned collect_bits() const { return bits; }
	void post_levels(int, int, CpLevels v) { lv = cp_level_max(lv, cp_level_pack(v)); }
	unsigned collect_levels(int) const { return lv; }
	void post_shift(int, int, int, int d) { shift = d; }
	int collect_shift(int) const { return shift; }
	int uniform(int v) const { return v; }
	template<class F>
	void phase(F f)
	{
		for (int t = 0; t < WT; t++) f(t, st[(size_t) t]);
	}
	template<class F>
	void each(F f)
	{
		for (int t = 0; t < WT; t++) f(t, st[(size_t) t]);
	}
	template<class F>
	void run_step(F& f, int w)
	{
		// (lanes in DESCENDING order in every second wave: read-before-write inside a step is exercised)
		if (w & 1)
			for (int t = 64 * w + 63; t >= 64 * w; t--) f(t, st[(size_t) t]);
		else
			for (int t = 64 * w; t < 64 * w + 64; t++) f(t, st[(size_t) t]);
	}
	template<class... F>
	void wave_steps(F... f)
	{
		for (int w = 0; w < WT / 64; w++) (run_step(f, w), ...);
	}
};

void emul_convq(const ConvxLaunch& X0)
{
	ConvxLaunch X = X0;
	convp_prepare<11, 1>(X, true, false, false, false);
	std::vector<double> lds((size_t) convq_lds_bytes() / sizeof(double) + 2);
	double* base = lds.data();
	if (((size_t) base & 15) != 0) base++;
	const long long items = (long long) X.c.nblk * ((X.c.nch + 1) / 2);
	for (long long i = 0; i < items; i++)
	{
		EmulExecQ ex;
		for (double& v : lds) v = std::numeric_limits<double>::quiet_NaN();
		for (auto& s : ex.st)
			for (int j = 0; j < 8; j++) s.vr[j] = s.vi[j] = std::numeric_limits<double>::quiet_NaN();
		convq_body(ex, X, X, reinterpret_cast<cd*>(base), convp_item<1>(X.c, i, false));
	}
}

template<int LN, int UL, int MODE, int FLENP>
void emul_convp_t(const ConvxLaunch& X0)
{
	if constexpr (LN == 12 && UL == 0 && MODE == 5)
	{
		// (kernel mode 33: the 1:1 geometry, both transforms by parts)
		if (X0.half_fused != 0 && convp_ha_fused_fits(X0.run_off, X0.c.in_len, X0.in_step))
		{
			emul_convp_t<LN, UL, 33, FLENP>(X0);
			return;
		}
	}
	if constexpr (LN == 11 && UL == 1 && (MODE == 16 || MODE == 17))
	{
		if (X0.half_fused != 0 && convp_ha_fused_fits(X0.run_off, X0.c.in_len, X0.in_step))
		{
			emul_convp_t<LN, UL, MODE == 16 ? 29 : 30, FLENP>(X0);
			return;
		}
	}
	if constexpr ((LN == 11 || LN == 12) && UL == 1 && (MODE == 6 || MODE == 7))
	{
		if (X0.half != 0)
		{
			emul_convp_t<LN, UL, MODE == 6 ? 31 : 32, FLENP>(X0);
			return;
		}
	}
	if constexpr (LN == 11 && UL == 1 && (MODE == 4 || MODE == 5))
	{
		// half-array form with the interpolator fused in (kernel modes 23 / 25)
		if (X0.half_fused != 0 && convp_ha_fused_fits(X0.run_off, X0.c.in_len, X0.in_step))
		{
			emul_convp_t<LN, UL, MODE == 4 ? 23 : 25, FLENP>(X0);
			return;
		}
	}
	if constexpr (LN == 11 && UL == 1 && MODE == 0)
	{
		if (X0.quad != 0)
		{
			emul_convq(X0);
			return;
		}
	}
	if constexpr (LN == 12 && UL == -1 && (MODE == 0 || MODE == 3))
	{
		// ... of the decimating geometry (kernel modes 27 / 28)
		if (X0.half != 0)
		{
			emul_convp_t<LN, UL, MODE == 0 ? 27 : 28, FLENP>(X0);
			return;
		}
	}
	if constexpr ((LN == 11 || LN == 12) && UL == 1 && (MODE == 0 || MODE == 3))
	{
		// half-array form (r8b_convp.h cp_ha_*, kernel modes 21 / 22)
		if (X0.half != 0 && X0.quad == 0)
		{
			emul_convp_t<LN, UL, MODE == 0 ? 21 : 22, FLENP>(X0);
			return;
		}
	}
	{
		// (the kernel instance a GPU launch would have started, as rocprofv3 names it: Engine::stage symbol)
		static const std::string sym = "k_convp<" + std::to_string(LN) + ", " + std::to_string(UL) + ", " + std::to_string(MODE) +
			", " + std::to_string(FLENP) + ">";
		launch_symbol_note(sym.c_str());
	}
	ConvxLaunch X = X0;
	constexpr bool SOLO = convp_mode_solo(MODE);
	convp_prepare<LN, UL>(X, MODE != 1 && MODE != 18, convp_mode_sp(MODE), SOLO, convp_mode_p3(MODE));
	// (the half-array form gets ITS allocation: an access beyond it is an error the poisoned vector does not hide)
	int lds_bytes = std::max(convp_lds_bytes<LN, UL>(), MODE == 20 ? kHbfLdsBytes : 0);
	if constexpr (convp_mode_ha(MODE)) lds_bytes = convp_ha_lds_bytes<LN, UL, MODE>();
	std::vector<double> lds((size_t) lds_bytes / sizeof(double) + 2);
	double* base = lds.data();
	if (((size_t) base & 15) != 0) base++;
	constexpr int SUB = ConvpGeom<LN, UL>::SUB;
	if constexpr (convp_walk_ok<LN, UL, MODE>())
	{
		int i0 = 0, i1 = 0;
		if (X.walk > 0 && convp_walk_range<LN, UL>(X, &i0, &i1) && i1 - i0 >= 2)
		{
			// walk form (r8b_convp.h convp_walk; the GPU's k_convp_walk): walk workgroups over the interior blocks, then a
			// workgroup per edge block on the general body
			X.walk_i0 = i0;
			X.walk_i1 = i1;
			X.walk_len = std::min(X.walk, i1 - i0);
			const int npair = (X.c.nch + 1) / 2, nwi = i1 - i0, nslice = (nwi + X.walk_len - 1) / X.walk_len;
			const int nwalk = nslice * npair, ntot = (nslice + X.c.nblk - nwi) * npair;
			launch_walk_blocks_add(nwi);
			for (int wi = 0; wi < ntot; wi++)
			{
				EmulExecP<LN, UL> ex;
				for (double& v : lds) v = std::numeric_limits<double>::quiet_NaN();
				for (auto& s : ex.st)
					for (int j = 0; j < 16; j++) s.vr[j] = s.vi[j] = std::numeric_limits<double>::quiet_NaN();
				ConvpItem cur;
				cur.nvalid = 1;
				if (wi < nwalk)
				{
					const int slice = wi / npair, pr = wi % npair, b0 = slice * X.walk_len;
					cur.k = X.c.k0 + i0 + b0;
					cur.chA = 2 * pr;
					cur.bvalid = cur.chA + 1 < X.c.nch;
					cur.chB = cur.bvalid ? cur.chA + 1 : cur.chA;
					convp_walk<LN, UL, MODE, FLENP>(ex, X, X, reinterpret_cast<cd*>(base), cur, std::min(X.walk_len, nwi - b0));
				}
				else
				{
					const int e = wi - nwalk, j = e / npair, pr = e % npair;
					cur.k = X.c.k0 + (j < i0 ? j : i1 + (j - i0));
					cur.chA = 2 * pr;
					cur.bvalid = cur.chA + 1 < X.c.nch;
					cur.chB = cur.bvalid ? cur.chA + 1 : cur.chA;
					convp_body<LN, UL, MODE, FLENP>(ex, X, X, reinterpret_cast<cd*>(base), cur);
				}
			}
			return;
		}
	}
	const long long items = (long long) ((X.c.nblk + SUB - 1) / SUB) * (SOLO ? X.c.nch : (X.c.nch + 1) / 2);
	for (long long i = 0; i < items; i++)
	{
		EmulExecP<LN, UL> ex;
		for (double& v : lds) v = std::numeric_limits<double>::quiet_NaN();
		for (auto& s : ex.st)
			for (int j = 0; j < 16; j++) s.vr[j] = s.vi[j] = std::numeric_limits<double>::quiet_NaN();
		convp_body<LN, UL, MODE, FLENP>(ex, X, reinterpret_cast<cd*>(base), convp_item<SUB>(X.c, i, SOLO));
	}
}

template<int LN, int DL>
void emul_convp_hbf(const ConvxLaunch& X)
{
	if constexpr (LN == 12 && DL == 1) emul_convp_t<LN, -DL, 20, 24>(X);
	else throw std::runtime_error("emul launch_convp: half-band front on a geometry it is not built for");
}

template<int LN, int DL>
void emul_convp_solo_down(const ConvxLaunch& X, int mode)
{
	if constexpr (LN == 13 && DL == 1)
	{
		if (mode == 10) emul_convp_t<LN, -DL, 10, 24>(X);
		else if (mode == 11) emul_convp_t<LN, -DL, 11, 24>(X);
		else if (mode == 14) emul_convp_t<LN, -DL, 14, 24>(X);
		else emul_convp_t<LN, -DL, 15, 24>(X);
	}
	if constexpr (LN == 13 && DL == 2)
	{
		if (mode == 10) emul_convp_t<LN, -DL, 10, 24>(X);
		else emul_convp_t<LN, -DL, 11, 24>(X);
	}
}

template<int LN, int UL>
void emul_convp_sp(const ConvxLaunch& X, int mode)
{
	if constexpr (LN == 13 && UL == 0)
	{
		if (mode == 8) emul_convp_t<LN, UL, 8, 24>(X);
		else if (mode == 9) emul_convp_t<LN, UL, 9, 24>(X);
		else if (mode == 10) emul_convp_t<LN, UL, 10, 24>(X);
		else if (mode == 11) emul_convp_t<LN, UL, 11, 24>(X);
		else if (mode == 18 && X.flen > 24) emul_convp_t<LN, UL, 18, 32>(X);
		else if (mode == 18) emul_convp_t<LN, UL, 18, 24>(X);
		else if (mode == 12) emul_convp_t<LN, UL, 12, 24>(X);
		else if (mode == 13) emul_convp_t<LN, UL, 13, 24>(X);
		else if (mode == 14) emul_convp_t<LN, UL, 14, 24>(X);
		else emul_convp_t<LN, UL, 15, 24>(X);
	}
}

template<int LN, int UL>
void emul_convp_p3(const ConvxLaunch& X)
{
	if constexpr (UL == 0 && LN >= 10 && LN <= 12) emul_convp_t<LN, UL, 19, 24>(X);
	else throw std::runtime_error("launch_convp: polyphase 3x form on a geometry it is not built for");
}

static thread_local long long t_walk_blocks = 0;
long long launch_walk_blocks() { return t_walk_blocks; }
void launch_walk_blocks_add(long long n) { t_walk_blocks += n; }
static thread_local const char* t_last_symbol = nullptr;
void launch_symbol_note(const char* symbol) { t_last_symbol = symbol; }
const char* launch_symbol_last() { return t_last_symbol; }

void launch_convp(const ConvxLaunch& X, int mode, void*)
{
	int ln = 0;
	while ((1 << ln) < X.c.n_in) ln++;
	const bool wide = X.flen > 24;
	const int up = X.c.up_pow2 ? X.c.up : 1;
#define R8B_CONVP_DISPATCH(LN, UL) \
	if (ln == LN && up == (1 << UL)) \
	{ \
		if (mode == 0) emul_convp_t<LN, UL, 0, 24>(X); \
		else if (mode == 19) emul_convp_p3<LN, UL>(X); \
		else if (mode == 3) emul_convp_t<LN, UL, 3, 24>(X); \
		else if (mode == 6) emul_convp_t<LN, UL, 6, 24>(X); \
		else if (mode == 7) emul_convp_t<LN, UL, 7, 24>(X); \
		else if (mode == 4) emul_convp_t<LN, UL, 4, 24>(X); \
		else if (mode == 5) emul_convp_t<LN, UL, 5, 24>(X); \
		else if (mode == 16) emul_convp_t<LN, UL, 16, 24>(X); \
		else if (mode == 17) emul_convp_t<LN, UL, 17, 24>(X); \
		else if (wide) emul_convp_t<LN, UL, 1, 32>(X); \
		else emul_convp_t<LN, UL, 1, 24>(X); \
		return; \
	}
	if (X.c.down_pow2 && X.c.down > 1)
	{
#define R8B_CONVP_DISPATCH_DOWN(LN, DL) \
		if (LN == 13 && ln == 14 && X.c.down == (1 << DL) && ((DL == 1 && convp_mode_solo(mode)) || \
			(DL == 2 && (mode == 10 || mode == 11)))) \
		{ \
			emul_convp_solo_down<LN, DL>(X, mode); \
			return; \
		} \
		if (ln == LN && X.c.down == (1 << DL) && mode == 20) \
		{ \
			emul_convp_hbf<LN, DL>(X); \
			return; \
		} \
		if (ln == LN && X.c.down == (1 << DL) && mode < 8) \
		{ \
			if (mode == 3) emul_convp_t<LN, -DL, 3, 24>(X); \
			else if (mode == 6) emul_convp_t<LN, -DL, 6, 24>(X); \
			else if (mode == 7) emul_convp_t<LN, -DL, 7, 24>(X); \
			else emul_convp_t<LN, -DL, 0, 24>(X); \
			return; \
		}
		R8B_CONVP_GEOMS_DOWN(R8B_CONVP_DISPATCH_DOWN)
#undef R8B_CONVP_DISPATCH_DOWN
		throw std::runtime_error("launch_convp: decimating geometry not instantiated");
	}
#define R8B_CONVP_DISPATCH_BIG(LN, UL) \
	if (LN == 13 && UL == 0 && ((ln == 13 && convp_mode_sp(mode)) || (ln == 14 && convp_mode_solo(mode)))) \
	{ \
		emul_convp_sp<LN, UL>(X, mode); \
		return; \
	} \
	if (ln == LN && up == (1 << UL) && (mode < 8 || mode == 16 || mode == 17)) \
	{ \
		if (mode == 3) emul_convp_t<LN, UL, 3, 24>(X); \
		else if (mode == 6) emul_convp_t<LN, UL, 6, 24>(X); \
		else if (mode == 7) emul_convp_t<LN, UL, 7, 24>(X); \
		else if (mode == 0) emul_convp_t<LN, UL, 0, 24>(X); \
		else if (mode == 4) emul_convp_t<LN, UL, 4, 24>(X); \
		else if (mode == 5) emul_convp_t<LN, UL, 5, 24>(X); \
		else if (mode == 16) emul_convp_t<LN, UL, 16, 24>(X); \
		else if (mode == 17) emul_convp_t<LN, UL, 17, 24>(X); \
		else if (wide) emul_convp_t<LN, UL, 1, 32>(X); \
		else emul_convp_t<LN, UL, 1, 24>(X); \
		return; \
	}
	R8B_CONVP_GEOMS_BIG(R8B_CONVP_DISPATCH_BIG)
#undef R8B_CONVP_DISPATCH_BIG
	R8B_CONVP_GEOMS(R8B_CONVP_DISPATCH)
#undef R8B_CONVP_DISPATCH
	throw std::runtime_error("emul launch_convp: geometry not instantiated");
}

void launch_convx(const ConvxLaunch& X, int mode, void*)
{
	int logn = 0;
	while ((2 << logn) < X.c.n_in) logn++;
	const int up = X.c.up_pow2 ? X.c.up : 1;
	const bool wide = X.flen > 24;
#define R8B_CONVX_DISPATCH_DOWN(LN, DL) \
	if (logn == LN && X.c.down == (1 << DL)) \
	{ \
		if (mode == 3) emul_convx_t<LN, -DL, 3, 24>(X); \
		else emul_convx_t<LN, -DL, 0, 24>(X); \
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
		if (mode == 0) emul_convx_t<LN, UL, 0, 24>(X); \
		else if (mode == 3) emul_convx_t<LN, UL, 3, 24>(X); \
		else if (wide) emul_convx_t<LN, UL, 1, 32>(X); \
		else emul_convx_t<LN, UL, 1, 24>(X); \
		return; \
	}
	R8B_CONVX_GEOMS(R8B_CONVX_DISPATCH)
#undef R8B_CONVX_DISPATCH
	throw std::runtime_error("emul launch_convx: geometry not instantiated");
}

void launch_hbcascade(const HBCascadeLaunch& L, void*)
{
	const int nthr = 256;
	std::vector<double> lds((size_t) (L.buf + L.buf2 + 3 * kHbcSlack));
	const long long n = L.b - L.a;
	if (n <= 0)
	{
		if (L.carry_tail) throw std::logic_error("emul: a history copy on a cascade launch without tiles");
		return;
	}
	const int tiles = (int) ((n + L.tile - 1) / L.tile);
	if (L.carry_tail) emul_carried_tail(L.tail, tiles, L.nch);
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < tiles; bx++)
		{
			for (double& v : lds) v = std::numeric_limits<double>::quiet_NaN();
			const long long q0 = L.a + (long long) bx * L.tile;
			long long q1 = q0 + L.tile;
			if (q1 > L.b) q1 = L.b;
			HBCRanges R;
			hbc_ranges(L, q0, q1, R);
			// big / small buffer as in k_hbcascade; every stream must fit the one it lands in
			double* const big = lds.data() + kHbcSlack;
			double* const small = big + L.buf + kHbcSlack;
			double* xin = ((L.nst - 1) & 1) ? small : big;
			double* yout = ((L.nst - 1) & 1) ? big : small;
			if (R.in_hi - R.in_lo > (xin == big ? L.buf : L.buf2))
				throw std::runtime_error("emul: cascade LDS");
			for (int t = 0; t < nthr; t++) hbc_load(L, R, xin, ch, t, nthr);
			long long in_lo = R.in_lo;
			for (int s = 0; s < L.nst; s++)
			{
				if (s + 1 < L.nst && R.hi[s] - R.lo[s] > (yout == big ? L.buf : L.buf2))
					throw std::runtime_error("emul: cascade LDS");
				for (int t = 0; t < nthr; t++)
					hbc_stage(L, s, xin, in_lo, R.lo[s], R.hi[s], yout, s + 1 == L.nst, ch, t, nthr);
				// (the next stage's input n is this stage's output n + skip: the buffer's first element is its input R.lo - skip)
				in_lo = R.lo[s] - (s + 1 < L.nst ? L.skip[s] : 0);
				std::swap(xin, yout);
			}
		}
}

void launch_hbdcascade(const HBCascadeLaunch& L, void*)
{
	const int nthr = 256;
	std::vector<double> lds((size_t) (L.buf + L.buf2));
	const long long n = L.b - L.a;
	if (n <= 0)
	{
		if (L.carry_tail) throw std::logic_error("emul: a history copy on a cascade launch without tiles");
		return;
	}
	const int tiles = (int) ((n + L.tile - 1) / L.tile);
	if (L.carry_tail) emul_carried_tail(L.tail, tiles, L.nch);
	for (int ch = 0; ch < L.nch; ch++)
		for (int bx = 0; bx < tiles; bx++)
		{
			for (double& v : lds) v = std::numeric_limits<double>::quiet_NaN();
			const long long q0 = L.a + (long long) bx * L.tile;
			long long q1 = q0 + L.tile;
			if (q1 > L.b) q1 = L.b;
			HBCRanges R;
			hbd_ranges(L, q0, q1, R);
			double* const even = lds.data();
			double* const odd = even + L.buf;
			if (R.in_hi - R.in_lo > L.buf) throw std::runtime_error("emul: decimating cascade LDS");
			for (long long i = 0; i < R.in_hi - R.in_lo; i++)
				even[i] = R.in_lo + i < L.in_end ? src_load(L.src, ch, R.in_lo + i) : 0.0;
			long long in_lo = R.in_lo;
			for (int s = 0; s < L.nst; s++)
			{
				if (s + 1 < L.nst && R.hi[s] - R.lo[s] > ((s & 1) ? L.buf : L.buf2))
					throw std::runtime_error("emul: decimating cascade LDS");
				for (int t = 0; t < nthr; t++)
					hbd_stage(L, s, (s & 1) ? odd : even, in_lo, R.lo[s], R.hi[s],
						(s & 1) ? even : odd, s + 1 == L.nst, ch, t, nthr);
				in_lo = R.lo[s] - (s + 1 < L.nst ? L.skip[s] : 0);
			}
		}
}

void launch_tail(const TailLaunch& L, void*)
{
	const int blocks = (int) ((L.p1 - L.p0 + 255) / 256);
	emul_carried_tail(L, blocks, L.nch);
}

static void emul_pcm(const PcmLaunch& L, bool in)
{
	if (!L.interleaved)
	{
		// planar buffers: the row kernels (r8b_pcm.h pcm_row_in / _out)
		for (int c = 0; c < L.nch; c++)
			for (long long f0 = 0; f0 < L.n; f0 += kPcmRowChunk)
				for (int t = 0; t < 256; t++)
				{
					if (in) pcm_row_in(L, f0, c, t, 256);
					else pcm_row_out(L, f0, c, t, 256);
				}
		return;
	}
	std::vector<double> tile((size_t) kPcmTile * kPcmPitch);
	const int nthr = 256;
	for (int c0 = 0; c0 < L.nch; c0 += kPcmTile)
		for (long long f0 = 0; f0 < L.n; f0 += kPcmTile)
		{
			for (double& v : tile) v = std::numeric_limits<double>::quiet_NaN();
			if (!L.interleaved)
			{
				for (int t = 0; t < nthr; t++)
				{
					if (in) pcm_in_direct(L, f0, c0, t, nthr);
					else pcm_out_direct(L, f0, c0, t, nthr);
				}
				continue;
			}
			for (int t = 0; t < nthr; t++)
			{
				if (in) pcm_in_gather(L, tile.data(), f0, c0, t, nthr);
				else pcm_out_gather(L, tile.data(), f0, c0, t, nthr);
			}
			for (int t = 0; t < nthr; t++)
			{
				if (in) pcm_in_scatter(L, tile.data(), f0, c0, t, nthr);
				else pcm_out_scatter(L, tile.data(), f0, c0, t, nthr);
			}
		}
}

void launch_pcm_in(const PcmLaunch& L, void*) { emul_pcm(L, true); }
void launch_pcm_out(const PcmLaunch& L, void*) { emul_pcm(L, false); }

int dev_resolve(int device) { return device < 0 ? 0 : device; }
int dev_swap(int) { return -1; }
void dev_restore(int) noexcept {}

void* dev_alloc(size_t bytes)
{
	void* p = nullptr;
	if (bytes == 0) bytes = 8;
	if (posix_memalign(&p, 64, bytes) != 0) throw std::runtime_error("emul: out of memory");
	memset(p, 0, bytes);
	return p;
}

void dev_free(void* p) { free(p); }
void dev_zero(void* p, size_t bytes, void*) { memset(p, 0, bytes); }
void dev_upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
void dev_upload_async(void* dst, const void* src, size_t bytes, void*) { memcpy(dst, src, bytes); }
void dev_download(void* dst, const void* src, size_t bytes, void*) { memcpy(dst, src, bytes); }
void dev_sync(void*) {}
void dev_check_last(const char*) {}
void* dev_event_create() { return nullptr; }
void dev_event_destroy(void*) {}
void dev_event_record(void*, void*) {}
float dev_event_elapsed_ms(void*, void*) { return 0.f; }

} // namespace r8bhip
