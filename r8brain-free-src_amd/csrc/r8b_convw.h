// r8b_convw.h -- the fast path, wave-per-block form: ONE WAVEFRONT = one overlap-save block of one
// channel.  Same arithmetic as r8b_convx.h (fused convolver + whole-step interpolator, overlapped
// block stride, host-folded spectral constants), different mapping to the machine:
//
//  * r8b_convx.h spreads a block over 4 waves and walks ~12 barrier-separated phases; with 4
//    workgroups per CU the waves of a block sit on 4 SIMDs next to 12 unrelated waves, every phase
//    ends at the slowest of them, and measured throughput saturates at 2x a single resident
//    workgroup with neither the LDS nor the vector ALU above 55 %.
//  * here a block lives in the REGISTERS of one wave: the 1024 / 2048-point transforms are three
//    in-register radix passes (32 or 16, 16, 4 points per lane) with a lane transposition through
//    LDS between them.  Nothing waits for another wave: no s_barrier anywhere, LDS traffic of a
//    wave is ordered by the LDS queue itself.  LDS per block: 21.7 KB (cfg2) -> 7 blocks per CU.
//
// Index algebra (forward, N = 64*M points, M in {16, 32}; lane l = 4a + j1):
//   pass 1  lane l holds x[l + 64p], p < M        radix M over p   -> r1 = bitrev(slot), * w_N^(l r1)
//   X1      lane (a, j1) holds sub-transform r1 in {a, a+16}, elements j = j1 + 4q
//   pass 2  radix 16 over q                        -> r2 = bitrev(slot), * w_64^(j1 r2)
//   X2      lane (a, j1') holds the 4-point groups (r1, r2 = 4 j1' + i), i < 4
//   pass 3  radix 4                                -> r3 = bitrev(slot); bin k = r1 + M (r2 + 16 r3)
// The backward transform (N2 = 64*M2 points) is the transposed graph with conjugate twiddles and
// ends with time index e = l + 64p in lane l.  In between, output bin k of the backward transform
// is ca(k) Z[k mod N] + cb(k) conj(Z[-k mod N]) -- every case of cx_spec_write() reduces to this
// form -- with (ca, cb) from a host table laid out [slot][lane] for this kernel's bins.
//
// Reference semantics reproduced: as r8b_convx.h.
#ifndef R8B_CONVW_H
#define R8B_CONVW_H

#include "r8b_convx.h"

namespace r8bhip {

static const int kWaveLanes = 64;

// compiler scheduling fence (no instruction): the includer may define it; default none
#ifndef R8B_SCHED_FENCE
#define R8B_SCHED_FENCE()
#endif

template<int LOGN, int UPLOG>
struct ConvwGeom
{
	static constexpr int N = 1 << LOGN, N2 = N << UPLOG;
	static constexpr int M = N / 64, M2 = N2 / 64, MX = M > M2 ? M : M2;
	static constexpr int LM2 = LOGN + UPLOG - 6;
	static_assert(UPLOG == 0 || UPLOG == 1, "up-sampling 1 or 2");
	static_assert((M == 16 || M == 32) && (M2 == 16 || M2 == 32), "1024 or 2048 points");
};

// ---- LDS planes (doubles unless noted) -----------------------------------------------------------
// X1: element (r1, j), j < 64.  Row pitch 68: lanes walking j are consecutive, lanes (a, j1)
// reading j = j1 + 4q of rows a..a+3 hit banks 4a + j1.
R8B_HD int cw_ax1(int r1, int j) { return r1 * 68 + j; }
// X2: element (r1, r2, j1).  Strides chosen so that both the writers (lanes (a, j1), fixed r2) and
// the readers (lanes (a, j1'), r2 = 4 j1' + i, fixed j1) touch 16 distinct banks per 16 lanes:
// r1 stride 1, r2 stride MM+1 (= 1 mod 16), j1 stride 16 (MM+1) + 4 (= 4 mod 16).
template<int MM> R8B_HD int cw_ax2(int r1, int r2, int j1)
{
	return j1 * (16 * (MM + 1) + 4) + r2 * (MM + 1) + r1;
}
// Z: forward bin k as one complex; one spare complex every 2^LM2 so that the spectral stage's reads
// (lane stride 4*M2 bins) spread over the banks
template<int LM2> R8B_HD int cw_az(int k) { return k + (k >> LM2); }

template<int LOGN, int UPLOG>
constexpr int convw_plane_doubles()
{
	typedef ConvwGeom<LOGN, UPLOG> G;
	const int x1 = G::MX * 68, x2 = 4 * (16 * (G::MX + 1) + 4);
	const int z = 2 * (G::N + (G::N >> G::LM2) + 1);
	return x1 > x2 ? (x1 > z ? x1 : z) : (x2 > z ? x2 : z);
}

inline int convw_lds_need(int planes, int in_len)
{
	const int run = in_len + kConvxRunPad;
	return planes > run ? planes : run;
}

template<int LOGN, int UPLOG>
struct ConvwState
{
	double vr[ConvwGeom<LOGN, UPLOG>::MX], vi[ConvwGeom<LOGN, UPLOG>::MX];
};

// ---- register butterflies ---------------------------------------------------------------------
// cos / sin of 2 pi j / 32, j < 16
R8B_HD double cw_c32(int j)
{
	constexpr double c[16] = { 1.0, 0.98078528040323044913, 0.92387953251128675613,
		0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
		0.38268343236508977173, 0.19509032201612826785, 0.0, -0.19509032201612826785,
		-0.38268343236508977173, -0.55557023301960222474, -0.70710678118654752440,
		-0.83146961230254523708, -0.92387953251128675613, -0.98078528040323044913 };
	return c[j];
}
R8B_HD double cw_s32(int j) { return j < 8 ? cw_c32(8 - j) : cw_c32(j - 8); }

// R-point DFT in registers, natural in, bit-reversed out (R = 4, 16, 32)
template<int R>
R8B_HD void cw_dif(double* vr, double* vi)
{
	if constexpr (R == 32)
	{
#pragma unroll
		for (int j = 0; j < 16; j++)
		{
			const double ar = vr[j], ai = vi[j], br = vr[j + 16], bi = vi[j + 16];
			vr[j] = ar + br;
			vi[j] = ai + bi;
			const double dr = ar - br, di = ai - bi;
			// * exp(-2 pi i j / 32) = c - i s
			const double c = cw_c32(j), s = cw_s32(j);
			vr[j + 16] = dr * c + di * s;
			vi[j + 16] = di * c - dr * s;
		}
		dif_regs<16>(vr, vi);
		dif_regs<16>(vr + 16, vi + 16);
	}
	else dif_regs<R>(vr, vi);
}

// backward (conjugate, unnormalised), bit-reversed in, natural out
template<int R>
R8B_HD void cw_dit(double* vr, double* vi)
{
	if constexpr (R == 32)
	{
		dit_regs<16>(vr, vi);
		dit_regs<16>(vr + 16, vi + 16);
#pragma unroll
		for (int j = 0; j < 16; j++)
		{
			const double c = cw_c32(j), s = cw_s32(j);
			// b * exp(+2 pi i j / 32) = b (c + i s)
			const double br = vr[j + 16] * c - vi[j + 16] * s;
			const double bi = vi[j + 16] * c + vr[j + 16] * s;
			const double ar = vr[j], ai = vi[j];
			vr[j] = ar + br;
			vi[j] = ai + bi;
			vr[j + 16] = ar - br;
			vi[j + 16] = ai - bi;
		}
	}
	else dit_regs<R>(vr, vi);
}

// ---- twiddles: w^(j*i), i < R, from the base powers i in {1,2,3}, {4,8,12}, {16} --------------------
template<int R>
R8B_HD void cw_tw_fetch(cd* twr, const cd* tw, int tw_len, int n, int j)
{
	tw_fetch<(R > 16 ? 16 : R)>(twr, tw, tw_len, n, j);
	if constexpr (R > 16) twr[6] = tw[tw_len / n * j * 16];
}

R8B_HD cd cw_tw_get(const cd* twr, int i)
{
	if (i < 16) return tw_get(twr, i);
	if (i == 16) return twr[6];
	const cd a = tw_get(twr, i - 16), b = twr[6];
	cd r;
	r.re = a.re * b.re - a.im * b.im;
	r.im = a.re * b.im + a.im * b.re;
	return r;
}

// slot p (bit-reversed order) *= w^(j * bitrev(p)), or its conjugate
template<int R, bool CONJ>
R8B_HD void cw_twiddle(double* vr, double* vi, const cd* twr)
{
#pragma unroll
	for (int p = 1; p < R; p++)
	{
		const cd w = cw_tw_get(twr, bitrev_c<R>(p));
		const double wi = CONJ ? -w.im : w.im;
		const double tr = vr[p] * w.re - vi[p] * wi;
		const double ti = vr[p] * wi + vi[p] * w.re;
		vr[p] = tr;
		vi[p] = ti;
		// (left alone, the scheduler forms all R-1 twiddles first: 2(R-1) more live registers)
		if constexpr (R > 16)
		{
			if ((p & 3) == 3) R8B_SCHED_FENCE();
		}
	}
}

// ---- the block, as wave-synchronous steps ---------------------------------------------------------
//
// `Exec::step(f)` runs f(lane, state) on all 64 lanes and orders its LDS traffic before the next
// step's: on the GPU plain straight-line code plus a compiler fence (the LDS queue of a wave is in
// order), in the host emulation a loop over lanes.

// input: complex e = l + 64p is the real pair (2e, 2e+1) of the circular block
template<int LOGN, int UPLOG>
R8B_HD void cw_load(const ConvLaunch& L, ConvwState<LOGN, UPLOG>& st, long long k, int ch, int l)
{
	typedef ConvwGeom<LOGN, UPLOG> G;
	constexpr int NIN = 2 * G::N;
	const int iln = L.in_len / L.up;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) / L.up;
	const SrcBlock sb = src_block(L.src, ch, base);
#pragma unroll
	for (int p = 0; p < G::M; p++)
	{
		const int i = 2 * (l + 64 * p);
		const long long pos = i < iln ? base + i : base + i - NIN;
		if (L.vec_ok)
		{
			const cd v = src_block_load2(sb, i < iln ? i : i - NIN);
			st.vr[p] = v.re;
			st.vi[p] = v.im;
		}
		else
		{
			const long long pos1 = i + 1 < iln ? base + i + 1 : base + i + 1 - NIN;
			st.vr[p] = src_load(L.src, ch, pos);
			st.vi[p] = src_load(L.src, ch, pos1);
		}
	}
}

// X1, forward direction: from "lane j, slot = bitrev(r1)" to "lane (a, j1), slot 16s + q"
template<int MM>
R8B_HD void cw_x1_put(double* lds, const double* v, int l)
{
#pragma unroll
	for (int p = 0; p < MM; p++) lds[cw_ax1(bitrev_c<MM>(p), l)] = v[p];
}
template<int MM>
R8B_HD void cw_x1_get(const double* lds, double* v, int l)
{
	const int a = l >> 2, j1 = l & 3;
#pragma unroll
	for (int s = 0; s < MM / 16; s++)
#pragma unroll
		for (int q = 0; q < 16; q++) v[16 * s + q] = lds[cw_ax1(a + 16 * s, j1 + 4 * q)];
}
// X1, backward direction: the reverse movement
template<int MM>
R8B_HD void cw_x1_put_inv(double* lds, const double* v, int l)
{
	const int a = l >> 2, j1 = l & 3;
#pragma unroll
	for (int s = 0; s < MM / 16; s++)
#pragma unroll
		for (int q = 0; q < 16; q++) lds[cw_ax1(a + 16 * s, j1 + 4 * q)] = v[16 * s + q];
}
template<int MM>
R8B_HD void cw_x1_get_inv(const double* lds, double* v, int l)
{
#pragma unroll
	for (int p = 0; p < MM; p++) v[p] = lds[cw_ax1(bitrev_c<MM>(p), l)];
}

// X2, forward: from "lane (a, j1), slot 16s + bitrev(r2)" to "lane (a, j1'), slot 16s + 4i + j1"
template<int MM>
R8B_HD void cw_x2_put(double* lds, const double* v, int l)
{
	const int a = l >> 2, j1 = l & 3;
#pragma unroll
	for (int s = 0; s < MM / 16; s++)
#pragma unroll
		for (int q = 0; q < 16; q++) lds[cw_ax2<MM>(a + 16 * s, bitrev_c<16>(q), j1)] = v[16 * s + q];
}
template<int MM>
R8B_HD void cw_x2_get(const double* lds, double* v, int l)
{
	const int a = l >> 2, jp = l & 3;
#pragma unroll
	for (int s = 0; s < MM / 16; s++)
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int t = 0; t < 4; t++)
				v[16 * s + 4 * i + t] = lds[cw_ax2<MM>(a + 16 * s, 4 * jp + i, t)];
}
template<int MM>
R8B_HD void cw_x2_put_inv(double* lds, const double* v, int l)
{
	const int a = l >> 2, jp = l & 3;
#pragma unroll
	for (int s = 0; s < MM / 16; s++)
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int t = 0; t < 4; t++)
				lds[cw_ax2<MM>(a + 16 * s, 4 * jp + i, t)] = v[16 * s + 4 * i + t];
}
template<int MM>
R8B_HD void cw_x2_get_inv(const double* lds, double* v, int l)
{
	const int a = l >> 2, j1 = l & 3;
#pragma unroll
	for (int s = 0; s < MM / 16; s++)
#pragma unroll
		for (int q = 0; q < 16; q++) v[16 * s + q] = lds[cw_ax2<MM>(a + 16 * s, bitrev_c<16>(q), j1)];
}

// forward bins into the Z plane: lane (a, j1') slot 16s + 4i + t holds bin
// k = (a + 16s) + M ((4 j1' + i) + 16 bitrev2(t))
template<int LOGN, int UPLOG>
R8B_HD void cw_z_put(cd* z, const ConvwState<LOGN, UPLOG>& st, int l)
{
	typedef ConvwGeom<LOGN, UPLOG> G;
	const int a = l >> 2, jp = l & 3;
#pragma unroll
	for (int s = 0; s < G::M / 16; s++)
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int t = 0; t < 4; t++)
			{
				const int k = (a + 16 * s) + G::M * ((4 * jp + i) + 16 * bitrev_c<4>(t));
				cd v;
				v.re = st.vr[16 * s + 4 * i + t];
				v.im = st.vi[16 * s + 4 * i + t];
				z[cw_az<G::LM2>(k)] = v;
			}
}

// backward bin owned by lane l at (s, i, s3): k = (a + 16s) + M2 ((4 j1' + i) + 16 s3)
template<int LOGN, int UPLOG>
R8B_HD int cw_out_bin(int l, int s, int i, int s3)
{
	typedef ConvwGeom<LOGN, UPLOG> G;
	return ((l >> 2) + 16 * s) + G::M2 * ((4 * (l & 3) + i) + 16 * s3);
}

// spectral stage: backward inputs straight into the registers of the first backward pass (slot
// 16s + 4i + bitrev2(s3)).  Constants: wspec[((16s + 4i + s3) * 2 + {0, 1}) * 64 + lane]: the
// eight of a group are fetched one group ahead (R8B_SCHED_FENCE keeps the compiler from hoisting
// all 64 fetches to the top, which costs 250 registers).
template<int LOGN, int UPLOG>
R8B_HD void cw_spec_fetch(const cd* wspec, cd* c, int g, int l)
{
#pragma unroll
	for (int e = 0; e < 8; e++) c[e] = wspec[(long) (8 * g + e) * kWaveLanes + l];
}

template<int LOGN, int UPLOG, int G>
R8B_HD void cw_spec_group(const cd* z, const cd* c, ConvwState<LOGN, UPLOG>& st, int l)
{
	typedef ConvwGeom<LOGN, UPLOG> GE;
	constexpr int s = G / 4, i = G % 4;
	cd u[4], v[4];
	// bins k and k + N share their forward pair when the backward transform is 2N long
	constexpr int NP = UPLOG == 1 ? 2 : 4;
#pragma unroll
	for (int s3 = 0; s3 < NP; s3++)
	{
		const int k = cw_out_bin<LOGN, UPLOG>(l, s, i, s3);
		u[s3] = z[cw_az<GE::LM2>(k & (GE::N - 1))];
		v[s3] = z[cw_az<GE::LM2>((GE::N - k) & (GE::N - 1))];
	}
#pragma unroll
	for (int s3 = 0; s3 < 4; s3++)
	{
		const int src = UPLOG == 1 ? (s3 & 1) : s3;
		const cd ca = c[2 * s3], cb = c[2 * s3 + 1];
		const cd uu = u[src], vv = v[src];
		// ca * u + cb * conj(v)
		const int slot = 16 * s + 4 * i + bitrev_c<4>(s3);
		st.vr[slot] = ca.re * uu.re - ca.im * uu.im + cb.re * vv.re + cb.im * vv.im;
		st.vi[slot] = ca.re * uu.im + ca.im * uu.re + cb.im * vv.re - cb.re * vv.im;
	}
}

template<int LOGN, int UPLOG, int G>
R8B_HD void cw_spec_chain(const cd* wspec, const cd* z, const cd* c, ConvwState<LOGN, UPLOG>& st,
	int l)
{
	constexpr int NG = ConvwGeom<LOGN, UPLOG>::M2 / 4;
	cd nxt[8];
	if constexpr (G + 1 < NG) cw_spec_fetch<LOGN, UPLOG>(wspec, nxt, G + 1, l);
	R8B_SCHED_FENCE();
	cw_spec_group<LOGN, UPLOG, G>(z, c, st, l);
	R8B_SCHED_FENCE();
	if constexpr (G + 1 < NG) cw_spec_chain<LOGN, UPLOG, G + 1>(wspec, z, nxt, st, l);
}

template<int LOGN, int UPLOG>
R8B_HD void cw_spectral(const cd* wspec, const cd* z, ConvwState<LOGN, UPLOG>& st, int l)
{
	cd c[8];
	cw_spec_fetch<LOGN, UPLOG>(wspec, c, 0, l);
	cw_spec_chain<LOGN, UPLOG, 0>(wspec, z, c, st, l);
}

// the block's valid outputs as the linear run y[u]: lane l slot p holds time index e = l + 64p
template<int LOGN, int UPLOG, bool ZERO_NEG>
R8B_HD void cw_run_store(const ConvLaunch& L, double* y, const ConvwState<LOGN, UPLOG>& st,
	long long k, int l)
{
	typedef ConvwGeom<LOGN, UPLOG> G;
	const int mask = 2 * G::N2 - 1;
	const long long t0 = cx_block_t0(L, k);
	const int nzero = !ZERO_NEG || t0 >= 0 ? 0 : (-t0 > L.in_len ? L.in_len : (int) -t0);
#pragma unroll
	for (int p = 0; p < G::M2; p++)
	{
		const int e = l + 64 * p;
		const int u0 = (2 * e + L.fl2) & mask, u1 = (2 * e + 1 + L.fl2) & mask;
		if (u0 < L.in_len) y[u0] = u0 < nzero ? 0.0 : st.vr[p];
		if (u1 < L.in_len) y[u1] = u1 < nzero ? 0.0 : st.vi[p];
		// (keeps the address and mask arithmetic of all 2*M2 stores from being formed up front)
		if ((p & 3) == 3) R8B_SCHED_FENCE();
	}
	if (l < kConvxRunPad) y[L.in_len + l] = 0.0;
}

// convolver output alone (MODE 0)
R8B_HD void cw_store_conv(const ConvLaunch& L, const double* y, long long k, int ch, int l)
{
	const long long t0 = cx_block_t0(L, k);
	for (int u = l; u < L.in_len; u += kWaveLanes)
	{
		const long long q = t0 + u;
		if (q >= L.a && q < L.b) dst_store(L.dst, ch, q, y[u]);
	}
}

// fused whole-step interpolator (MODE 1): "thread" t = l + 64r < OutStep of cx_whole_compute(),
// three rounds for OutStep <= 192 ... four for 256; the row of each round is fetched from the
// transposed bank (coalesced) when the round starts
template<int FLENP>
R8B_HD void cw_interp(const ConvxLaunch& X, const double* y, long long k, int ch, int l)
{
	for (int t = l; t < X.out_step; t += kWaveLanes)
	{
		double row[FLENP];
		cx_whole_row<FLENP>(X, row, t);
		cx_whole_compute<FLENP>(X, y, row, k, ch, t);
	}
}

template<int LOGN, int UPLOG, int MODE, int FLENP, class Exec>
R8B_HD void convw_body(Exec& ex, const ConvxLaunch& X, double* lds, long long k, int ch)
{
	typedef ConvwGeom<LOGN, UPLOG> G;
	typedef ConvwState<LOGN, UPLOG> St;
	const ConvLaunch& L = X.c;
	cd* const z = reinterpret_cast<cd*>(lds);

	// ---- forward
	ex.step([&](int l, St& st)
	{
		cw_load<LOGN, UPLOG>(L, st, k, ch, l);
		cd twr[7];
		cw_tw_fetch<G::M>(twr, L.tw, L.tw_len, G::N, l);
		cw_dif<G::M>(st.vr, st.vi);
		cw_twiddle<G::M, false>(st.vr, st.vi, twr);
		cw_x1_put<G::M>(lds, st.vr, l);
	});
	ex.step([&](int l, St& st) { cw_x1_get<G::M>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x1_put<G::M>(lds, st.vi, l); });
	ex.step([&](int l, St& st)
	{
		cw_x1_get<G::M>(lds, st.vi, l);
		cd twr[7];
		cw_tw_fetch<16>(twr, L.tw, L.tw_len, 64, l & 3);
#pragma unroll
		for (int s = 0; s < G::M / 16; s++)
		{
			cw_dif<16>(st.vr + 16 * s, st.vi + 16 * s);
			cw_twiddle<16, false>(st.vr + 16 * s, st.vi + 16 * s, twr);
		}
	});
	ex.step([&](int l, St& st) { cw_x2_put<G::M>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x2_get<G::M>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x2_put<G::M>(lds, st.vi, l); });
	ex.step([&](int l, St& st)
	{
		cw_x2_get<G::M>(lds, st.vi, l);
#pragma unroll
		for (int g = 0; g < G::M / 4; g++) cw_dif<4>(st.vr + 4 * g, st.vi + 4 * g);
	});
	// ---- spectral stage
	ex.step([&](int l, St& st) { cw_z_put<LOGN, UPLOG>(z, st, l); });
	ex.step([&](int l, St& st)
	{
		cw_spectral<LOGN, UPLOG>(L.wspec, z, st, l);
		// ---- backward
#pragma unroll
		for (int g = 0; g < G::M2 / 4; g++) cw_dit<4>(st.vr + 4 * g, st.vi + 4 * g);
	});
	ex.step([&](int l, St& st) { cw_x2_put_inv<G::M2>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x2_get_inv<G::M2>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x2_put_inv<G::M2>(lds, st.vi, l); });
	ex.step([&](int l, St& st)
	{
		cw_x2_get_inv<G::M2>(lds, st.vi, l);
		cd twr[7];
		cw_tw_fetch<16>(twr, L.tw, L.tw_len, 64, l & 3);
#pragma unroll
		for (int s = 0; s < G::M2 / 16; s++)
		{
			cw_twiddle<16, true>(st.vr + 16 * s, st.vi + 16 * s, twr);
			cw_dit<16>(st.vr + 16 * s, st.vi + 16 * s);
		}
	});
	ex.step([&](int l, St& st) { cw_x1_put_inv<G::M2>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x1_get_inv<G::M2>(lds, st.vr, l); });
	ex.step([&](int l, St& st) { cw_x1_put_inv<G::M2>(lds, st.vi, l); });
	ex.step([&](int l, St& st)
	{
		cw_x1_get_inv<G::M2>(lds, st.vi, l);
		cd twr[7];
		cw_tw_fetch<G::M2>(twr, L.tw, L.tw_len, G::N2, l);
		cw_twiddle<G::M2, true>(st.vr, st.vi, twr);
		cw_dit<G::M2>(st.vr, st.vi);
	});
	// ---- output
	ex.step([&](int l, St& st) { cw_run_store<LOGN, UPLOG, MODE != 0>(L, lds, st, k, l); });
	if (L.tail_ring != nullptr && k == L.k0)
	{
		ex.step([&](int l, St&)
		{
			for (long long i = L.tail_p0 + l; i < L.tail_p1; i += kWaveLanes)
				L.tail_ring[(long long) ch * L.src.ring_stride + (i & L.src.ring_mask)] =
					src_load(L.src, ch, i);
		});
	}
	ex.step([&](int l, St&)
	{
		if constexpr (MODE == 1) cw_interp<FLENP>(X, lds, k, ch, l);
		else cw_store_conv(L, lds, k, ch, l);
	});
}

} // namespace r8bhip

#endif
