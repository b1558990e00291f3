// r8b_convp.h -- the fast path in PAIR form: one workgroup = one overlap-save block of TWO channels.
//
// The convolver's kernel is real (zero phase), so the block of channel A can ride in the real part
// and the block of channel B in the imaginary part of ONE complex transform: y_A + i y_B =
// IDFT(DFT(x_A + i x_B) * H).  Compared with the one-channel form of r8b_convx.h (a real transform
// packed into a half-length complex one) this removes the whole spectral stage -- real-FFT
// unpacking, spectrum replication, kernel multiplication and packing for the half-length backward
// transform collapse into ONE real multiplication per bin -- and with it the only place where a
// thread needs bins another thread holds.  What is left:
//   * forward DIF passes (natural in, bit-reversed out) with 8 or 16 elements per thread;
//   * a MIDDLE pass in registers: the last forward butterflies (stride 1: 8 or 16 consecutive
//     positions per thread), the multiplication by the kernel, and the first backward (DIT)
//     butterflies over the 16 consecutive positions the same thread owns in the backward array.
//     With 2x up-sampling the zero-stuffed spectrum is the forward spectrum repeated (reference
//     CDSPBlockConvolver.h:606-629): backward positions 2p and 2p+1 (bins k and k + N) both descend
//     from forward position p, so the first radix-2 stage of the backward transform is folded into
//     the constants Hs = H[k] + H[k+N], Hd = H[k] - H[k+N] (the even and odd polyphase branches of
//     the filter) and costs nothing;
//   * two radix-16 backward passes; the last one keeps its results in registers;
//   * MODE 1: the block's valid outputs of both channels as one linear run of (A, B) pairs in LDS;
//     the whole-step interpolator reads a pair per tap (one 16-byte LDS read feeds two
//     multiply-adds with the same coefficient);  MODE 0: straight from the registers to the
//     destination, nothing goes back to LDS.
// Per block and channel: half as many barrier-separated phases as the one-channel form, no spectral
// stage, ~45 % less LDS traffic in the transforms.  LDS: one array of N2 complex (64 KB for the
// 4096-point backward transform), two workgroups per CU; the XOR swizzle pswz() makes every pass
// conflict free for 16-byte accesses without pad slots.
//
// Geometries: 2048-point forward / 4096-point backward (2x up: BASELINE configs 2, 4, 5) and
// 4096 / 4096 (1:1: config 3); everything else stays on r8b_convx.h.
//
// Reference semantics reproduced: CDSPBlockConvolver.h:252-354, 512-593, 606-629;
// CDSPRealFFT.h:289-385; CDSPFracInterpolator.h:991-1060.
#ifndef R8B_CONVP_H
#define R8B_CONVP_H

#include "r8b_convx.h"

namespace r8bhip {

static const int kConvpThreads = 256;

// physical slot of complex element e: index bits 0-3 XOR bits 4-7.  Every access pattern of the
// passes below (lanes = consecutive elements, lanes = elements 4 / 8 / 16 apart, 8 or 16
// consecutive elements per lane) then touches 16 different 16-byte bank groups per 16-lane
// service group.
R8B_HD int pswz(int e) { return e ^ ((e >> 4) & 15); }

template<int LN, int UL>
struct ConvpGeom
{
	static constexpr int N = 1 << LN, LN2 = LN + UL, N2 = 1 << LN2;
	static constexpr int E1 = N / kConvpThreads, E2 = N2 / kConvpThreads; // elements per thread
	static constexpr int EB1 = LN - 8;
	static_assert(E2 == 16 && (E1 == 8 || E1 == 16), "pair kernel: 4096-point backward transform only");
	static constexpr int NPRE = (LN - 1) / EB1;   // forward passes before the middle one (radix E1)
	static constexpr int MB = LN - NPRE * EB1;    // log2 radix of the forward butterflies in the middle pass
	static constexpr int RM = 1 << MB, NBF = E1 / RM;
	// Wave w (threads 64 w ...) owns forward positions [w N/4, (w+1) N/4) after the first pass and
	// backward positions [w N2/4, (w+1) N2/4) up to the last pass: the passes in between never leave
	// that range (their butterflies span at most 512 / 1024 consecutive elements), so they need no
	// workgroup barrier -- provided the wave's forward data lives where its backward data will: forward
	// position p sits at slot fslot(p) = (p / (N/4)) * (N2/4) + p mod (N/4)  (the identity when N2 = N).
	static constexpr int FW = N / 4, BW = N2 / 4;
};

template<int LN, int UL>
R8B_HD int fslot(int p)
{
	typedef ConvpGeom<LN, UL> G;
	if constexpr (UL == 0) return pswz(p);
	else return pswz((p / G::FW) * G::BW + (p & (G::FW - 1)));
}

template<int LN, int UL>
struct ConvpState
{
	double vr[16], vi[16];
	double pr[16], pi[16]; // the block's input samples (channel A, channel B) of the first pass
	cd tw[6];
	cd hp[8];
	double row[32];
	double rows2[2 * 25]; // mode 4: the two rows of the thread's phase pair
};

constexpr int convp_lds_bytes(int logn2) { return (1 << logn2) * 16; }

// ---- passes over the swizzled array ---------------------------------------------------------------

template<int LN, int UL, int R, bool TW>
R8B_HD void pdif(cd* buf, int n, int b, const cd* twr)
{
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
	double vr[R], vi[R];
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[fslot<LN, UL>(e0 + p * q)];
		vr[p] = v.re;
		vi[p] = v.im;
	}
	dif_regs<R>(vr, vi);
	if constexpr (TW)
	{
#pragma unroll
		for (int p = 1; p < R; p++)
		{
			const cd w = tw_get(twr, bitrev_c<R>(p));
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
		buf[fslot<LN, UL>(e0 + p * q)] = v;
	}
}

template<int R, bool TW>
R8B_HD void pdit_regs(const cd* buf, int n, int b, const cd* twr, double* vr, double* vi)
{
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[pswz(e0 + p * q)];
		vr[p] = v.re;
		vi[p] = v.im;
	}
	if constexpr (TW)
	{
#pragma unroll
		for (int p = 1; p < R; p++)
		{
			const cd w = tw_get(twr, bitrev_c<R>(p));
			const double tr = vr[p] * w.re + vi[p] * w.im;
			const double ti = vi[p] * w.re - vr[p] * w.im;
			vr[p] = tr;
			vi[p] = ti;
		}
	}
	dit_regs<R>(vr, vi);
}

// ---- phases -----------------------------------------------------------------------------------------

// K1: thread t owns the radix-E1 butterfly over elements t + 256 p of the first pass; element i of the
// circular block is sample i of channel A (real part) and of channel B (imaginary part).  A wave
// reads 64 consecutive samples of each channel per load.  The samples wait in registers: the
// persistent form of the kernel issues these loads one block ahead (r8b_kernels.hip k_convp_loop).
template<int LN, int UL>
R8B_HD void cp_load(const ConvLaunch& L, ConvpState<LN, UL>& st, long long k, int chA, int chB, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
	const int iln = L.in_len / L.up;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) / L.up;
	const SrcBlock sa = src_block(L.src, chA, base), sb = src_block(L.src, chB, base);
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const int i = tid + p * q;
		const int rel = i < iln ? i : i - G::N;
#if defined(R8B_P_SKIP) && (R8B_P_SKIP & 1)
		st.pr[p] = 1e-3 * (rel & 255) + (double) sa.b_lo * 1e-9;
		st.pi[p] = 1e-3 * (rel & 127) + (double) sb.b_lo * 1e-9;
#else
		st.pr[p] = src_block_load1(sa, rel);
		st.pi[p] = src_block_load1(sb, rel);
#endif
	}
}

// ---- the persistent form stages the NEXT block's samples in LDS by LDS-DMA (no registers) -----------
// Staging area: 2048 (E1 = 8) or 4096 (E1 = 16) doubles of channel A, then as many of channel B, behind
// the run (slots the interpolation phase does not touch; the launch allocates 16 KB beyond the
// transform array for it).  One DMA operation moves two consecutive samples (16 bytes) per lane: op
// o = pair (o mod N/2) of channel A (o < N/2) or B, landing at byte 16 o of the area -- a wave's 64 ops
// are 64 consecutive 16-byte words, as LDS-DMA requires.
template<int LN, int UL>
struct ConvpStage
{
	typedef ConvpGeom<LN, UL> G;
	static constexpr int OPS = G::N, PER_THREAD = OPS / kConvpThreads; // ops per block, per thread
};

// can block k be staged?  needs 16-byte loads of sample pairs everywhere (L.vec_ok) and no position
// before the start of the stream (those read as zero, which a DMA cannot produce)
template<int LN, int UL>
R8B_HD bool cp_can_stage(const ConvLaunch& L, long long k)
{
	typedef ConvpGeom<LN, UL> G;
	const int iln = L.in_len / L.up;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) / L.up;
	return G::E1 == 8 && L.vec_ok && L.src.cur_fmt == kPcmF64 && base - (G::N - iln) >= 0;
}

// global address of the sample pair DMA op (r, tid) of block k fetches
template<int LN, int UL>
R8B_HD const double* cp_stage_src(const ConvLaunch& L, long long k, int chA, int chB, int r, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	const int o = tid + kConvpThreads * r;
	const int ch = o < G::N / 2 ? chA : chB;
	const int i = 2 * (o & (G::N / 2 - 1));
	const int iln = L.in_len / L.up;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) / L.up;
	const int rel = i < iln ? i : i - G::N;
	const long long pos = base + rel;
	const double* pr = L.src.ring + ((long long) ch * L.src.ring_stride + (pos & L.src.ring_mask));
	const double* pc = L.src.cur + ((long long) ch * L.src.cur_stride + (pos - L.src.cur_base));
	return pos >= L.src.cur_base ? pc : pr;
}

// first forward pass from the staging area
template<int LN, int UL>
R8B_HD void cp_unstage(const double* stage, ConvpState<LN, UL>& st, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		st.pr[p] = stage[tid + p * q];
		st.pi[p] = stage[G::N + tid + p * q];
	}
}

// first forward pass, from the registers cp_load() filled
template<int LN, int UL>
R8B_HD void cp_first(const ConvLaunch& L, cd* buf, const ConvpState<LN, UL>& st, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
	cd loc[6];
	tw_fetch<R>(loc, L.tw, L.tw_len, G::N, tid);
	double vr[R], vi[R];
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		vr[p] = st.pr[p];
		vi[p] = st.pi[p];
	}
	dif_regs<R>(vr, vi);
#pragma unroll
	for (int p = 1; p < R; p++)
	{
		const cd w = tw_get(loc, bitrev_c<R>(p));
		const double tr = vr[p] * w.re - vi[p] * w.im;
		const double ti = vr[p] * w.im + vi[p] * w.re;
		vr[p] = tr;
		vi[p] = ti;
	}
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		buf[fslot<LN, UL>(tid + p * q)] = v;
	}
}

// forward pass I (1 <= I < NPRE): sub-length N >> (I * EB1), radix E1, one butterfly per thread
template<int LN, int UL, int I>
struct ConvpPre
{
	typedef ConvpGeom<LN, UL> G;
	static constexpr int n = G::N >> (I * G::EB1);
	static R8B_HD void prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int tid)
	{
		tw_fetch<G::E1>(st.tw, L.tw, L.tw_len, n, tid & (n / G::E1 - 1));
	}
	static R8B_HD void run(cd* buf, const ConvpState<LN, UL>& st, int tid)
	{
		pdif<LN, UL, G::E1, true>(buf, n, tid, st.tw);
	}
};

// kernel constants of the middle pass, hp[c * 256 + t] (a wave reads 64 consecutive 16-byte entries):
//   2x up: (Hs, Hd) of forward position 8 t + c;   1:1: H of backward positions 16 t + 2 c, + 1
template<int LN, int UL>
R8B_HD void cp_hp_prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int tid)
{
#pragma unroll
	for (int c = 0; c < 8; c++)
	{
#if defined(R8B_P_SKIP) && (R8B_P_SKIP & 8)
		st.hp[c].re = 1e-3 * (c + 1) + 1e-9 * tid;
		st.hp[c].im = 1e-3 * (c + 2) - 1e-9 * tid;
#else
		st.hp[c] = L.hp[c * kConvpThreads + tid];
#endif
	}
}

// middle pass, compute part: results (the backward array's positions 16 t + p after the first
// backward pass) stay in st.vr / st.vi
template<int LN, int UL>
R8B_HD void cp_middle_compute(const cd* buf, ConvpState<LN, UL>& st, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	double zr[G::E1], zi[G::E1];
#pragma unroll
	for (int c = 0; c < G::E1; c++)
	{
		const cd v = buf[fslot<LN, UL>(G::E1 * tid + c)];
		zr[c] = v.re;
		zi[c] = v.im;
	}
#pragma unroll
	for (int f = 0; f < G::NBF; f++) dif_regs<G::RM>(zr + G::RM * f, zi + G::RM * f);
	if constexpr (UL > 0)
	{
		// forward position p -> backward positions 2p, 2p+1 already combined by the first radix-2
		// stage: Z (H[k] + H[k+N]), Z (H[k] - H[k+N])
#pragma unroll
		for (int c = 0; c < 8; c++)
		{
			st.vr[2 * c] = zr[c] * st.hp[c].re;
			st.vi[2 * c] = zi[c] * st.hp[c].re;
			st.vr[2 * c + 1] = zr[c] * st.hp[c].im;
			st.vi[2 * c + 1] = zi[c] * st.hp[c].im;
		}
		DitSt<16, 2>::run(st.vr, st.vi);
	}
	else
	{
#pragma unroll
		for (int c = 0; c < 8; c++)
		{
			st.vr[2 * c] = zr[2 * c] * st.hp[c].re;
			st.vi[2 * c] = zi[2 * c] * st.hp[c].re;
			st.vr[2 * c + 1] = zr[2 * c + 1] * st.hp[c].im;
			st.vi[2 * c + 1] = zi[2 * c + 1] * st.hp[c].im;
		}
		dit_regs<16>(st.vr, st.vi);
	}
}

template<int LN, int UL>
R8B_HD void cp_middle_write(cd* buf, const ConvpState<LN, UL>& st, int tid)
{
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		cd v;
		v.re = st.vr[p];
		v.im = st.vi[p];
		buf[pswz(16 * tid + p)] = v;
	}
}

// backward pass with sub-length 256 (radix 16, in place)
template<int LN, int UL>
R8B_HD void cp_back1(cd* buf, const ConvpState<LN, UL>& st, int tid)
{
	double vr[16], vi[16];
	pdit_regs<16, true>(buf, 256, tid, st.tw, vr, vi);
	const int e0 = (tid >> 4) * 256 + (tid & 15);
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		buf[pswz(e0 + p * 16)] = v;
	}
}

// last backward pass (sub-length 4096): element t + 256 p = (y_A, y_B) at circular time t + 256 p
template<int LN, int UL>
R8B_HD void cp_back2(const cd* buf, ConvpState<LN, UL>& st, int tid)
{
	pdit_regs<16, true>(buf, 4096, tid, st.tw, st.vr, st.vi);
}

// MODE 1: the block's valid outputs as one linear run of (A, B) pairs, y[u] = outputs at time t0 + u
template<int LN, int UL>
R8B_HD void cp_final_store(const ConvLaunch& L, cd* y, const ConvpState<LN, UL>& st, long long k, int tid)
{
	// (y already points at the run: buf + run_off)
	constexpr int mask = ConvpGeom<LN, UL>::N2 - 1;
	const long long t0 = cx_block_t0(L, k);
	// a stage's stream starts at t = 0: earlier outputs do not exist for the interpolator
	// (reference CDSPFracInterpolator.h:834-859)
	const int nzero = t0 >= 0 ? 0 : (-t0 > L.in_len ? L.in_len : (int) -t0);
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		const int u = (tid + kConvpThreads * p + L.fl2) & mask;
		if (u < L.in_len)
		{
			cd v;
			v.re = u < nzero ? 0.0 : st.vr[p];
			v.im = u < nzero ? 0.0 : st.vi[p];
			y[u] = v;
		}
	}
	// zero extension read (times zero taps) by the padded polyphase rows
	if (tid < kConvxRunPad)
	{
		cd z;
		z.re = z.im = 0.0;
		y[L.in_len + tid] = z;
	}
}

// MODE 0: K7 straight from the registers
template<int LN, int UL>
R8B_HD void cp_store_conv(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int chA,
	int chB, bool bvalid, int tid)
{
	constexpr int mask = ConvpGeom<LN, UL>::N2 - 1;
	const long long t0 = cx_block_t0(L, k);
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		const int u = (tid + kConvpThreads * p + L.fl2) & mask;
		const long long q = t0 + u;
#if defined(R8B_P_SKIP) && (R8B_P_SKIP & 2)
		if (st.vr[p] == 1.2345e300)
#else
		if (u < L.in_len && q >= L.a && q < L.b)
#endif
		{
			dst_store(L.dst, chA, q, st.vr[p]);
			if (bvalid) dst_store(L.dst, chB, q, st.vi[p]);
		}
	}
}

// MODE 1: K8 on the pair run (cf. cx_whole_compute): one 16-byte LDS read per tap feeds both channels
template<int FLEN>
R8B_HD void cp_whole_compute(const ConvxLaunch& X, const cd* y, const double* row, long long k,
	int chA, int chB, bool bvalid, int tid)
{
	const ConvLaunch& L = X.c;
	if (tid >= X.out_step) return;
	const SpanInfo& B = X.blk[k - L.k0];
	int d = tid - B.jlo_mod;
	if (d < 0) d += X.out_step;
	long long j = B.jlo + d;
	const long long jhi = B.jhi;
	if (j >= jhi) return;
	int u = B.u_lo + (int) ((unsigned) (B.ph_lo + d * X.in_step) / (unsigned) X.out_step);
	for (; j < jhi; j += X.out_step, u += X.in_step)
	{
		double sa[2] = { 0.0, 0.0 }, sb[2] = { 0.0, 0.0 };
		// taps in chunks of eight, each chunk's reads issued one chunk ahead of its multiply-adds
		cd v[2][8];
#pragma unroll
		for (int i = 0; i < 8; i++) v[0][i] = y[u + i];
#pragma unroll
		for (int c = 0; c < FLEN / 8; c++)
		{
			R8B_SCHED_FENCE();
			if (c + 1 < FLEN / 8)
			{
#pragma unroll
				for (int i = 0; i < 8; i++) v[(c + 1) & 1][i] = y[u + 8 * (c + 1) + i];
			}
#pragma unroll
			for (int i = 0; i < 8; i++)
			{
				sa[i & 1] += row[8 * c + i] * v[c & 1][i].re;
				sb[i & 1] += row[8 * c + i] * v[c & 1][i].im;
			}
		}
		dst_store(X.wdst, chA, j, sa[0] + sa[1]);
		if (bvalid) dst_store(X.wdst, chB, j, sb[0] + sb[1]);
	}
}

// MODE 2: K8 on the matrix cores, both channels of the pair at once.  Output j = Out g + ph (group g,
// phase ph) reads y[In g + r_ph - fll + i], r_ph = floor(ph In / Out).  A tile is D[16 x 16] =
// A[16 x K] B[K x 16] (v_mfma_f64_16x16x4_f64, K steps of 4) with
//   rows    m = 2 gl + c : channel c of group g0 + 8 ct + gl (column tile ct = 8 groups x 2 channels),
//           A[m][kk] = y_c[In g + r_(16 pt) - fll + kk]: one 8-byte LDS read per lane and K step;
//   columns n : phase 16 pt + n, B[kk][n] = T[row(ph)][kk - (r_ph - r_(16 pt))], zero outside the taps
//           (the banded table X.mf_atab, block independent: a wave keeps the tiles of its <= 3 phase
//           tiles in registers);
// so a lane's four results are one channel's outputs at 16 consecutive phases per 16-lane group: a
// store instruction writes four 128-byte runs.  Blocks keep their own (ragged) output ranges: groups
// cut by the block's range [jlo, jhi) are computed whole and masked at the store; every LDS slot a
// masked or zero-weighted product reads holds a finite value (the whole array was written by the
// transform passes).  Versus the vector form: no polyphase row in registers, 1/7 of the LDS reads,
// and the multiply-adds run on the matrix pipe next to the other workgroup's transform passes.
// Operand layout (cdna_hip_programming.md section 3): lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; D register i of lane l is row (l>>4) + 4 i, column l&15.
static const int kConvpSets = 3; // phase tiles a wave can hold

// phase tiles of wave w: [w * tiles / 4, ...) -- at most kConvpSets (checked by the host)
R8B_HD int cp_mfma_first_tile(const ConvxLaunch& X, int wave) { return wave * X.mf_tiles / 4; }

// index (in doubles) of lane's A operand of K step 0: run slot of (group, K column lane >> 4), channel
// lane & 1
R8B_HD int cp_mfma_a_index(const ConvxLaunch& X, const SpanInfo& B, int pt, int ct, int lane)
{
	int gl = 8 * ct + ((lane & 15) >> 1);
	gl = gl > B.ph_lo ? B.ph_lo : gl; // rows beyond the block's last group repeat it (masked anyway)
	return 2 * (B.u_lo + X.in_step * gl + X.mf_boff[pt] + (lane >> 4)) + (lane & 1);
}

R8B_HD void cp_mfma_store(const ConvxLaunch& X, const SpanInfo& B, int pt, int ct, int lane, int chA,
	int chB, bool bvalid, const double* d)
{
	const int ph = 16 * pt + (lane & 15);
	if (ph >= X.out_step) return;
	const long long jg = B.jlo - B.jlo_mod + ph;
#pragma unroll
	for (int i = 0; i < 4; i++)
	{
		const int m = (lane >> 4) + 4 * i;
		const long long j = jg + (long long) X.out_step * (8 * ct + (m >> 1));
#if defined(R8B_P_SKIP) && (R8B_P_SKIP & 2)
		if (d[i] == 1.2345e300)
#else
		if (j >= B.jlo && j < B.jhi && ((m & 1) == 0 || bvalid))
#endif
			dst_store(X.wdst, (m & 1) ? chB : chA, j, d[i]);
	}
}

// MODE 4: K8 on the vector ALU, two ADJACENT phases per thread.  With In <= Out the tap windows of
// phases 2q and 2q+1 start 0 or 1 samples apart, so 25 (A, B) pairs read from LDS feed four outputs
// (two phases x two channels): half the LDS reads per output of the one-phase form, and with
// floor(256 / pairs) group sets per workgroup nearly every lane works (cfg2: 80 phase pairs x 3 sets =
// 240 lanes; the one-phase form keeps 160 of 256 busy).  The second row is stored shifted by its
// window offset (25 entries, zero padded), so the loop has no per-thread branch.  Lanes are assigned
// to phase pairs through the host table X.ptab such that the 16 lanes LDS serves together start their
// windows in 16 different bank groups (window starts of consecutive phase pairs are ~1.84 slots
// apart: in natural order they collide two to three ways).  Groups cut by the block's output range are
// computed whole and masked at the store (slots outside the run hold finite transform data).
static const int kConvpTaps2 = 25;

R8B_HD void cp_rows2_fetch(const ConvxLaunch& X, double* rows, int tid)
{
#pragma unroll
	for (int i = 0; i < 2 * kConvpTaps2; i++)
	{
#if defined(R8B_P_SKIP) && (R8B_P_SKIP & 8)
		rows[i] = 1e-2 * (i + 1) + 1e-9 * tid;
#else
		rows[i] = X.ctab[i * kConvpThreads + tid];
#endif
	}
}

R8B_HD void cp_whole2_compute(const ConvxLaunch& X, const cd* y, const double* rows, long long k,
	int chA, int chB, bool bvalid, int tid)
{
	const int pt = X.ptab[tid];
	if (pt < 0) return;
	const int q = pt & 0xffff, set = pt >> 16;
	// (block constants into registers once: the kernel arguments live in memory)
	const SpanInfo& Bm = X.blk[k - X.c.k0];
	const int hi_mod = Bm.pad, gmax = Bm.ph_lo, lo_mod = Bm.jlo_mod, u_lo = Bm.u_lo;
	if (hi_mod == 0) return;
	const int in_step = X.in_step, out_step = X.out_step, nsets = X.nsets;
	const long long jg = Bm.jlo - lo_mod + 2 * q;
	// group 0 starts at the block's first output (phase lo_mod), the last group ends before phase hi_mod
	const bool f0 = 2 * q >= lo_mod, f1 = 2 * q + 1 >= lo_mod && 2 * q + 1 < out_step;
	const bool l0 = 2 * q < hi_mod, l1 = 2 * q + 1 < hi_mod && 2 * q + 1 < out_step;
	const int rq = (int) ((unsigned) (2 * q * in_step) / (unsigned) out_step);
	for (int gl = set; gl <= gmax; gl += nsets)
	{
		const cd* w = y + (u_lo + in_step * gl + rq);
		double a0[2] = { 0.0, 0.0 }, b0[2] = { 0.0, 0.0 }, a1[2] = { 0.0, 0.0 }, b1[2] = { 0.0, 0.0 };
		// the window in chunks of five taps, each chunk's reads issued one chunk ahead of its
		// multiply-adds (two chunks of 16-byte values are live: the scheduler, left alone, reads all 25
		// first -- 100 registers the prefetched samples of the next block then have to leave for)
		constexpr int CH = 5, NCH = kConvpTaps2 / CH;
		cd v[2][CH];
#pragma unroll
		for (int i = 0; i < CH; i++) v[0][i] = w[i];
#pragma unroll
		for (int c = 0; c < NCH; c++)
		{
			R8B_SCHED_FENCE();
			if (c + 1 < NCH)
			{
#pragma unroll
				for (int i = 0; i < CH; i++) v[(c + 1) & 1][i] = w[CH * (c + 1) + i];
			}
#pragma unroll
			for (int i = 0; i < CH; i++)
			{
				const int t = CH * c + i;
				a0[t & 1] += rows[t] * v[c & 1][i].re;
				b0[t & 1] += rows[t] * v[c & 1][i].im;
				a1[t & 1] += rows[kConvpTaps2 + t] * v[c & 1][i].re;
				b1[t & 1] += rows[kConvpTaps2 + t] * v[c & 1][i].im;
			}
		}
		const long long j = jg + (long long) out_step * gl;
		const bool v0 = (gl > 0 || f0) && (gl < gmax || l0);
		const bool v1 = (gl > 0 ? 2 * q + 1 < out_step : f1) && (gl < gmax || l1);
		if (v0)
		{
			dst_store(X.wdst, chA, j, a0[0] + a0[1]);
			if (bvalid) dst_store(X.wdst, chB, j, b0[0] + b0[1]);
		}
		if (v1)
		{
			dst_store(X.wdst, chA, j + 1, a1[0] + a1[1]);
			if (bvalid) dst_store(X.wdst, chB, j + 1, b1[0] + b1[1]);
		}
	}
}

// ---- the kernel body ---------------------------------------------------------------------------------

// one unit of work: block k of the channel pair (chA, chB); bvalid: chB is a real channel (an odd
// channel count leaves the last one without a partner: its block rides alone, chB = chA)
struct ConvpItem
{
	long long k;
	int chA, chB;
	bool bvalid;
};

// LOOP: the persistent form.  `staged`: the samples of `cur` wait in the LDS staging area (else they
// are loaded here, like in the one-block form).  If has_next and the next item can be staged, its
// DMA is issued behind the last barrier of the transforms, so that it flies during the
// interpolation; returns whether that happened.  `stage` = the staging area (LOOP only).
template<int LN, int UL, int MODE, int FLENP, bool LOOP, class Exec>
R8B_HD bool convp_body(Exec& ex, const ConvxLaunch& X, cd* buf, double* stage, const ConvpItem& cur,
	bool staged, const ConvpItem& nxt, bool has_next)
{
	typedef ConvpGeom<LN, UL> G;
	typedef ConvpState<LN, UL> St;
	const ConvLaunch& L = X.c;
	const long long k = cur.k;
	const int chA = cur.chA, chB = cur.chB;
	const bool bvalid = cur.bvalid;
	if constexpr (MODE == 2) ex.template mfma_pair_prefetch<(FLENP > 24 ? 12 : 10)>(X);
	// (the first pass writes wave 3's part of the array, which the staging area overlaps: every thread
	// has its samples in registers before any thread writes)
	if (LOOP && staged) ex.phase([&](int tid, St& st) { cp_unstage<LN, UL>(stage, st, tid); });
	ex.phase([&](int tid, St& st)
	{
		if (!(LOOP && staged)) cp_load<LN, UL>(L, st, k, chA, chB, tid);
		cp_first<LN, UL>(L, buf, st, tid);
		if constexpr (G::NPRE > 1) ConvpPre<LN, UL, 1>::prefetch(L, st, tid);
		else cp_hp_prefetch<LN, UL>(L, st, tid);
	});
	// forward passes 1 .., the middle pass and the first backward pass stay inside each wave's own range
	// of the array (ConvpGeom): wave-level ordering points instead of workgroup barriers between them
	auto s_pre1 = [&](int tid, St& st)
	{
		ConvpPre<LN, UL, 1>::run(buf, st, tid);
		if constexpr (G::NPRE > 2) ConvpPre<LN, UL, 2>::prefetch(L, st, tid);
		else cp_hp_prefetch<LN, UL>(L, st, tid);
	};
	auto s_pre2 = [&](int tid, St& st)
	{
		ConvpPre<LN, UL, 2>::run(buf, st, tid);
		cp_hp_prefetch<LN, UL>(L, st, tid);
	};
	// (two steps: every lane has read its forward data before any lane's backward data overwrites it --
	// on the GPU program order alone guarantees that, LDS serves a wave's accesses in issue order)
	auto s_midc = [&](int tid, St& st)
	{
		cp_middle_compute<LN, UL>(buf, st, tid);
		tw_fetch<16>(st.tw, L.tw, L.tw_len, 256, tid & 15);
	};
	auto s_midw = [&](int tid, St& st) { cp_middle_write<LN, UL>(buf, st, tid); };
	auto s_b1 = [&](int tid, St& st)
	{
		cp_back1<LN, UL>(buf, st, tid);
		tw_fetch<16>(st.tw, L.tw, L.tw_len, 4096, tid);
	};
	static_assert(G::NPRE == 2 || G::NPRE == 3, "pair kernel: two or three forward passes before the middle");
	if constexpr (G::NPRE == 3) ex.wave_steps(s_pre1, s_pre2, s_midc, s_midw, s_b1);
	else ex.wave_steps(s_pre1, s_midc, s_midw, s_b1);
	// history for the next call (stage 0 only): the first block's workgroup copies the tail of
	// the caller's buffers into the other history ring; the stores need no wait
	if (L.tail_ring != nullptr && k == L.k0)
	{
		ex.each([&](int tid, St&)
		{
			for (long long i = L.tail_p0 + tid; i < L.tail_p1; i += kConvpThreads)
			{
				L.tail_ring[(long long) chA * L.src.ring_stride + (i & L.src.ring_mask)] =
					src_load(L.src, chA, i);
				if (bvalid)
					L.tail_ring[(long long) chB * L.src.ring_stride + (i & L.src.ring_mask)] =
						src_load(L.src, chB, i);
			}
		});
	}
	bool next_staged = false;
	auto prefetch_next = [&]()
	{
		if constexpr (LOOP)
		{
			if (has_next && stage != nullptr && cp_can_stage<LN, UL>(L, nxt.k))
			{
				ex.stage_issue(L, stage, nxt);
				next_staged = true;
			}
		}
	};
	if constexpr (MODE == 0)
	{
		if constexpr (LOOP)
		{
			// (the staging area overlaps the array the last pass reads)
			ex.phase([&](int tid, St& st) { cp_back2<LN, UL>(buf, st, tid); });
			prefetch_next();
		}
		else ex.each([&](int tid, St& st) { cp_back2<LN, UL>(buf, st, tid); });
		ex.each([&](int tid, St& st) { cp_store_conv<LN, UL>(L, st, k, chA, chB, bvalid, tid); });
	}
	else if constexpr (MODE == 2)
	{
		constexpr int KS = FLENP > 24 ? 12 : 10;
		ex.phase([&](int tid, St& st) { cp_back2<LN, UL>(buf, st, tid); });
		ex.phase([&](int tid, St& st) { cp_final_store<LN, UL>(L, buf + X.run_off, st, k, tid); });
		prefetch_next();
#if !defined(R8B_P_SKIP) || !(R8B_P_SKIP & 4)
		ex.template mfma_pair_interp<KS>(X, buf, k, chA, chB, bvalid);
#endif
	}
	else if constexpr (MODE == 4)
	{
		ex.phase([&](int tid, St& st)
		{
			cp_back2<LN, UL>(buf, st, tid);
			cp_rows2_fetch(X, st.rows2, tid);
		});
		ex.phase([&](int tid, St& st) { cp_final_store<LN, UL>(L, buf + X.run_off, st, k, tid); });
		prefetch_next();
		ex.each([&](int tid, St& st)
		{
			cp_whole2_compute(X, buf, st.rows2, k, chA, chB, bvalid, tid);
		});
	}
	else
	{
		ex.phase([&](int tid, St& st)
		{
			cp_back2<LN, UL>(buf, st, tid);
			cx_whole_row<FLENP>(X, st.row, tid);
		});
		ex.phase([&](int tid, St& st) { cp_final_store<LN, UL>(L, buf, st, k, tid); });
		prefetch_next();
		ex.each([&](int tid, St& st)
		{
			cp_whole_compute<FLENP>(X, buf, st.row, k, chA, chB, bvalid, tid);
		});
	}
	return next_staged;
}

// item i of a launch, pair major: blocks of one channel pair are consecutive (a persistent
// workgroup walks them back to back: the overlap-save history is re-read from its own caches)
R8B_HD ConvpItem convp_item(const ConvLaunch& L, long long i)
{
	ConvpItem it;
	const int pr = (int) (i / L.nblk);
	it.k = L.k0 + (i - (long long) pr * L.nblk);
	it.chA = 2 * pr;
	it.bvalid = it.chA + 1 < L.nch;
	it.chB = it.bvalid ? it.chA + 1 : it.chA;
	return it;
}

} // namespace r8bhip

#endif
