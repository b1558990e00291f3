// r8b_convx.h -- the fast path: one workgroup = one overlap-save block of one channel, with
// compile-time transform sizes, and (MODE 1) the whole-step polyphase interpolator of the NEXT
// stage fused behind the inverse transform, so that the 2x-rate convolver output never leaves LDS.
//
// Covers block convolvers with a 2^k-point geometry, up-sampling 1 or 2 and no decimation -- the
// first stage of every up-sampling and mild down-sampling chain of the reference
// (CDSPResampler.h:218-330) including BASELINE configs 1-5.  Everything else takes the generic
// kernels of r8b_kernel_phases.h.
//
// Differences from the generic convolver kernel, all about where cycles and LDS bytes go:
//  * LDS holds ONE complex work array (in-place transforms), padded by one complex every 16 so
//    that both the stride-q passes and the contiguous-per-thread passes are bank-conflict free;
//  * the spectral stage runs in place: each thread first pulls its forward bins into registers,
//    then (after a barrier) writes the backward-transform inputs they determine;
//  * the last backward pass keeps its results in registers across a barrier and then writes the
//    block's VALID outputs only, rotated so that they form one linear run y[0..in_len) in LDS
//    (aliasing the work array: 34.8 KB of LDS per workgroup for the 2048-point backward FFT);
//  * twiddles of pass p+1 are fetched from the L2-resident table BEFORE the barrier that ends
//    pass p, so their latency overlaps the barrier wait;
//  * MODE 1: thread t < OutStep keeps polyphase row (t*InStep mod OutStep) in registers and
//    produces outputs j = t (mod OutStep) of the block, reading y straight from LDS.  In this
//    mode consecutive blocks start blk_stride = in_len - flen (rounded to the up factor) virtual
//    samples apart instead of in_len, i.e. their valid output ranges overlap by one interpolator
//    filter length, so every tap window lies inside one block and nothing is exchanged between
//    workgroups (cost: < 1 % more blocks).
//
// Reference semantics reproduced: CDSPBlockConvolver.h:252-354, 512-593, 606-629;
// CDSPRealFFT.h:289-385; CDSPFracInterpolator.h:991-1060.
#ifndef R8B_CONVX_H
#define R8B_CONVX_H

#include "r8b_kernel_phases.h"

namespace r8bhip {

static const int kConvxThreads = 256;

// compiler scheduling fence (no instruction is emitted): r8b_kernels.hip defines it as
// __builtin_amdgcn_sched_barrier(0); elsewhere it is nothing
#ifndef R8B_SCHED_FENCE
#define R8B_SCHED_FENCE()
#endif
static const int kConvxRunPad = 32; // zeros behind the linear output run (see cx_final_store)

// padded complex index: one spare slot after every 16
#ifndef R8B_CX_PADSHIFT
#define R8B_CX_PADSHIFT 4
#endif
R8B_HD int cpad(int e) { return e + (e >> R8B_CX_PADSHIFT); }
// padded index of real sample i when the reals are viewed as packed complex pairs
R8B_HD int rpad(int i) { return i + ((i >> 5) << 1); }

constexpr int convx_lds_doubles(int logn2)
{
	return 2 * ((1 << logn2) + ((1 << logn2) >> R8B_CX_PADSHIFT));
}
// LDS doubles a workgroup needs: the padded work array, which the linear output run aliases once the
// last backward pass sits in registers
inline int convx_lds_need(int logn2, int in_len, int /*mode*/)
{
	const int work = convx_lds_doubles(logn2);
	const int run = in_len + kConvxRunPad;
	return work > run ? work : run;
}

// log2 radix of the pass that touches the full length (first DIF pass / last DIT pass): sized so
// that one butterfly per thread covers the array when possible
#ifndef R8B_CX_MAXBITS
#define R8B_CX_MAXBITS 3 // largest log2 radix of a pass: radix 8 keeps the kernel under 128 VGPRs (4
                         // workgroups per CU); radix 16 saves an LDS round trip but allows only 3
#endif
constexpr int big_pass_bits(int logn)
{
	return logn - 8 < 1 ? 1 : (logn - 8 > R8B_CX_MAXBITS ? R8B_CX_MAXBITS : logn - 8);
}

// number of remaining passes and their log2 radices (as even as possible, each <= R8B_CX_MAXBITS)
constexpr int rest_passes(int logn)
{
	return (logn - big_pass_bits(logn) + R8B_CX_MAXBITS - 1) / R8B_CX_MAXBITS;
}
constexpr int rest_bits(int logn, int i)
{
	const int bits = logn - big_pass_bits(logn), np = rest_passes(logn);
	return (bits + np - 1 - i) / np;
}

// per-thread state that lives in registers across barriers
template<int LOGN, int UPLOG>
struct ConvxState
{
	// UPLOG = 1: 2x up-sampling, 0: 1:1, -1: 2x decimation (backward transform of half the length)
	static constexpr bool DOWN = UPLOG < 0;
	static constexpr int N = 1 << LOGN, LOGN2 = LOGN + UPLOG, N2 = 1 << LOGN2;
	// spectral slots: one per forward bin pair (kf, N-kf), or when decimating per backward pair
	// (k, N2-k), which needs four forward bins
	static constexpr int SLOTS = (DOWN ? N2 : N) / 2 + 1;
	static constexpr int SP = (SLOTS + kConvxThreads - 1) / kConvxThreads;
	static constexpr int NSP = DOWN ? 4 : 2;
	static constexpr int RL = 1 << big_pass_bits(LOGN2);
	static constexpr int FIN = (N2 / RL + kConvxThreads - 1) / kConvxThreads;
	static constexpr int RF = 1 << big_pass_bits(LOGN); // radix of the first forward pass
	static constexpr int FF = (N / RF + kConvxThreads - 1) / kConvxThreads; // its butterflies/thread
	cd tw[6];
	cd tw0[RF > 8 ? 6 : (RF > 4 ? 4 : 3)]; // first-pass twiddles: same for every block
	cd pre[FF][RF];                        // inputs of the block's first-pass butterflies
	cd sp[SP][NSP];
	// spectral stage by output position (R8B_CX_SPEC_BY_POSITION): one backward input per slot
	static constexpr int SO = DOWN ? 1 : (N2 + kConvxThreads - 1) / kConvxThreads;
	cd so[SO];
	double fr[FIN][RL], fi[FIN][RL];
	double row[32];
};

// ---- transform passes over the padded array -------------------------------------------------

// Twiddles of a radix-R butterfly are w^(j*i), i = 1..R-1.  Only the "base" powers
// i in {1,2,3} and {4,8,12} are fetched (6 values for R = 16, 4 for R = 8, 3 for R = 4); the
// others are products base[i & 3] * base[i & 12], formed on the fly: 24 registers instead of 60.
template<int R>
R8B_HD void tw_fetch(cd* twr, const cd* tw, int tw_len, int n, int j)
{
	const int ts = tw_len / n * j;
	twr[0] = tw[ts];
	if constexpr (R >= 4)
	{
		twr[1] = tw[ts * 2];
		twr[2] = tw[ts * 3];
	}
	if constexpr (R >= 8) twr[3] = tw[ts * 4];
	if constexpr (R >= 16)
	{
		twr[4] = tw[ts * 8];
		twr[5] = tw[ts * 12];
	}
}

// w^(j*i) from the base set (i is a compile-time constant after unrolling)
R8B_HD cd tw_get(const cd* twr, int i)
{
	const int lo = i & 3, hi = i >> 2;
	if (hi == 0) return twr[lo - 1];
	if (lo == 0) return twr[2 + hi];
	const cd a = twr[lo - 1], b = twr[2 + hi];
	cd r;
	r.re = a.re * b.re - a.im * b.im;
	r.im = a.re * b.im + a.im * b.re;
	return r;
}

template<int R, bool TW>
R8B_HD void xdif(cd* buf, int n, int b, const cd* twr)
{
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
	double vr[R], vi[R];
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[cpad(e0 + p * q)];
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
		buf[cpad(e0 + p * q)] = v;
	}
}

// backward butterfly; results stay in vr/vi (the caller stores them)
template<int R, bool TW>
R8B_HD void xdit_regs(const cd* buf, int n, int b, const cd* twr, double* vr, double* vi)
{
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[cpad(e0 + p * q)];
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

template<int R, bool TW>
R8B_HD void xdit(cd* buf, int n, int b, const cd* twr)
{
	double vr[R], vi[R];
	xdit_regs<R, TW>(buf, n, b, twr, vr, vi);
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		buf[cpad(e0 + p * q)] = v;
	}
}

// one full pass with sub-length n = 2^LOGSUB and radix 2^RB over an array of 2^LOGTOT complex.
// Twiddles come from `twr` when each thread owns exactly one butterfly (prefetched by the
// caller), otherwise they are fetched in place.
template<int LOGTOT, int LOGSUB, int RB, bool INV>
R8B_HD void xpass(cd* buf, const cd* twr, const cd* tw, int tw_len, int tid)
{
	constexpr int R = 1 << RB, n = 1 << LOGSUB, nb = (1 << LOGTOT) / R;
	constexpr bool TW = LOGSUB > RB;
	constexpr bool PRE = nb <= kConvxThreads;
	for (int b = tid; b < nb; b += kConvxThreads)
	{
		if constexpr (TW && !PRE)
		{
			cd loc[6];
			tw_fetch<R>(loc, tw, tw_len, n, b & (n / R - 1));
			if constexpr (INV) xdit<R, TW>(buf, n, b, loc);
			else xdif<R, TW>(buf, n, b, loc);
		}
		else
		{
			if constexpr (INV) xdit<R, TW>(buf, n, b, twr);
			else xdif<R, TW>(buf, n, b, twr);
		}
	}
}

template<int LOGTOT, int LOGSUB, int RB>
R8B_HD void xprefetch(cd* twr, const cd* tw, int tw_len, int tid)
{
	constexpr int R = 1 << RB, n = 1 << LOGSUB, nb = (1 << LOGTOT) / R;
	if constexpr (LOGSUB > RB && nb <= kConvxThreads)
	{
		if (tid < nb) tw_fetch<R>(twr, tw, tw_len, n, tid & (n / R - 1));
	}
}

// ---- forward: DIF, first pass big_pass_bits, then the rest -----------------------------------
// The kernel body calls these in order with a barrier after each; `FwdPass<LOGN, I>` is pass I.

template<int LOGN, int I>
struct FwdPass
{
	static constexpr int B0 = big_pass_bits(LOGN);
	static constexpr int NP = 1 + rest_passes(LOGN);
	// sub-length and radix bits of pass I
	static constexpr int sub()
	{
		int s = LOGN;
		if (I >= 1) s -= B0;
		for (int i = 0; i + 1 < I; i++) s -= rest_bits(LOGN, i);
		return s;
	}
	static constexpr int rb() { return I == 0 ? B0 : rest_bits(LOGN, I - 1); }
	static R8B_HD void prefetch(cd* twr, const cd* tw, int tw_len, int tid)
	{
		xprefetch<LOGN, sub(), rb()>(twr, tw, tw_len, tid);
	}
	static R8B_HD void run(cd* buf, const cd* twr, const cd* tw, int tw_len, int tid)
	{
		xpass<LOGN, sub(), rb(), false>(buf, twr, tw, tw_len, tid);
	}
};

// ---- backward: DIT, rest passes (sub-length growing), last pass big_pass_bits ------------------

template<int LOGN2, int I>
struct InvPass
{
	static constexpr int BL = big_pass_bits(LOGN2);
	static constexpr int NR = rest_passes(LOGN2); // passes before the last one
	static constexpr int rb() { return I < NR ? rest_bits(LOGN2, NR - 1 - I) : BL; }
	static constexpr int sub()
	{
		int s = 0;
		for (int i = 0; i <= I && i < NR; i++) s += rest_bits(LOGN2, NR - 1 - i);
		if (I >= NR) s += BL;
		return s;
	}
	static R8B_HD void prefetch(cd* twr, const cd* tw, int tw_len, int tid)
	{
		xprefetch<LOGN2, sub(), rb()>(twr, tw, tw_len, tid);
	}
	static R8B_HD void run(cd* buf, const cd* twr, const cd* tw, int tw_len, int tid)
	{
		xpass<LOGN2, sub(), rb(), true>(buf, twr, tw, tw_len, tid);
	}
};

// ---- phases ------------------------------------------------------------------------------------

// two consecutive samples at an even position as one aligned 16-byte load through a selected
// address (the host sets ConvLaunch::vec_ok only when every position used here is even and the
// buffers are 16-byte aligned)
R8B_HD cd src_load2(const SrcView& s, int ch, long long pos)
{
	const double* pr = s.ring + ((long long) ch * s.ring_stride + (pos & s.ring_mask));
	const double* pc = s.cur + ((long long) ch * s.cur_stride + (pos - s.cur_base));
	const cd v = *reinterpret_cast<const cd*>(pos >= s.cur_base ? pc : pr);
	cd r;
	r.re = pos < 0 ? 0.0 : v.re;
	r.im = pos < 0 ? 0.0 : v.im;
	return r;
}

// K1: the block's input goes from global memory straight into the registers of the first
// forward pass (thread b owns butterfly b: complex elements b + p*N/R, i.e. real samples
// 2e, 2e+1 of the circular block).  Issued one block ahead, at the start of the long
// interpolation phase of the previous block, so that HBM latency is off the critical path.
// MODE 3, non-2^k up-sampling (3x): virtual sample t of the zero-stuffed stream is x[t / up] when
// up divides t, else 0 (reference CDSPBlockConvolver.h:414-496, copyUpsample).  base_v = up * B +
// bm is the block's virtual start; for the circular offset rel_v the source index is
// B + floor((bm + rel_v) / up).
R8B_HD double cx_stuffed_sample(const SrcBlock& sb, int up, int bm, int rel_v, int bias)
{
	// bias = up * kb keeps the dividend positive (rel_v >= -bl2)
	const unsigned w = (unsigned) (bm + rel_v + bias);
	const unsigned q = up == 3 ? w / 3u : w / (unsigned) up;
	const bool hit = q * (unsigned) up == w;
	const double v = src_block_load1(sb, hit ? (int) q - bias / up : 0);
	return hit ? v : 0.0;
}

template<int LOGN, int UPLOG, int MODE = 0>
R8B_HD void cx_prefetch(const ConvLaunch& L, ConvxState<LOGN, UPLOG>& st, long long k, int ch, int tid)
{
	typedef ConvxState<LOGN, UPLOG> St;
	constexpr int R = St::RF, N = 1 << LOGN, nb = N / R, NIN = 2 * N;
	if constexpr (MODE == 3)
	{
		if (!L.up_pow2)
		{
			const long long base_v = k * (long long) L.blk_stride + L.blk_offset;
			const long long B = base_v / L.up;
			const int bm = (int) (base_v - B * L.up);
			const int bias = L.up * (NIN / L.up + 2);
			const SrcBlock sb = src_block(L.src, ch, B);
#pragma unroll
			for (int f = 0; f < St::FF; f++)
			{
				const int b = tid + f * kConvxThreads;
				if (b >= nb) continue;
#pragma unroll
				for (int p = 0; p < R; p++)
				{
					const int i = 2 * (b + p * nb);
					st.pre[f][p].re = cx_stuffed_sample(sb, L.up, bm, i < L.in_len ? i : i - NIN, bias);
					st.pre[f][p].im = cx_stuffed_sample(sb, L.up, bm,
						i + 1 < L.in_len ? i + 1 : i + 1 - NIN, bias);
				}
			}
			return;
		}
	}
	const int iln = L.in_len / L.up;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) / L.up;
	const SrcBlock sb = src_block(L.src, ch, base);
#pragma unroll
	for (int f = 0; f < St::FF; f++)
	{
		const int b = tid + f * kConvxThreads;
		if (b >= nb) continue;
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			const int i = 2 * (b + p * nb);
			const long long pos = i < iln ? base + i : base + i - NIN;
			if (L.vec_ok) st.pre[f][p] = src_block_load2(sb, i < iln ? i : i - NIN);
			else
			{
				// odd geometry or unaligned caller buffer: the two samples may even sit on
				// different sides of the fresh/history split
				const long long pos1 = i + 1 < iln ? base + i + 1 : base + i + 1 - NIN;
				st.pre[f][p].re = src_load(L.src, ch, pos);
				st.pre[f][p].im = src_load(L.src, ch, pos1);
			}
		}
	}
}

// first forward pass: inputs from the prefetch registers, results into the padded LDS array
template<int LOGN, int UPLOG>
R8B_HD void cx_first_pass(const ConvLaunch& L, cd* buf, const ConvxState<LOGN, UPLOG>& st, int tid)
{
	typedef ConvxState<LOGN, UPLOG> St;
	constexpr int R = St::RF, N = 1 << LOGN, q = N / R;
#pragma unroll
	for (int f = 0; f < St::FF; f++)
	{
		const int b = tid + f * kConvxThreads;
		if (b >= q) continue;
		double vr[R], vi[R];
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			vr[p] = st.pre[f][p].re;
			vi[p] = st.pre[f][p].im;
		}
		dif_regs<R>(vr, vi);
		if constexpr (LOGN > big_pass_bits(LOGN))
		{
			// one butterfly per thread: twiddles prefetched into st.tw0; more: fetched here
			cd loc[6];
			const cd* twr = st.tw0;
			if constexpr (St::FF > 1)
			{
				tw_fetch<R>(loc, L.tw, L.tw_len, N, b);
				twr = loc;
			}
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
			buf[cpad(b + p * q)] = v;
		}
	}
}

// slot -> forward bin handled by that slot.  Slots 0..N/2-1 take bins in bit-reversed order, so
// that consecutive lanes touch consecutive LDS words of the bit-reversed spectrum instead of
// one bank; slot N/2 takes bin N/2; larger slots are idle (returned bin > N/2).
template<int LOGN>
R8B_HD int cx_spec_bin(int slot)
{
	constexpr int N = 1 << LOGN;
	if (slot < N / 2) return bitrev_n(slot, LOGN - 1);
	return slot == N / 2 ? N / 2 : N;
}

// ---- spectral stage addressed by OUTPUT POSITION (up-sampling 1 or 2) -------------------------------
// Every backward input is Z'[k'] = ca(k') Z[k' mod N] + cb(k') conj(Z[-k' mod N]) (all cases of
// cx_spec_write() below reduce to this form).  With thread <-> position P of the backward array
// (bin k' = bitrev(P)) every LDS access is conflict free: the write is position P itself, the first
// read is forward position P >> UPLOG (pairs of lanes share it when UPLOG = 1), and the second read
// -- the position of the NEGATED bin -- mirrors the first inside its power-of-two range:
// negating k keeps its lowest set bit and inverts the bits above it; in the bit-reversed domain
// that is "keep the highest set bit of p, invert the bits below": 3*2^m - 1 - p, m = floor(log2 p),
// so 64 consecutive lanes read 64 consecutive positions in descending order.  (Addressed by forward
// bin pair, as before, the reads and the four writes scatter bit-reversed: 2-4-way conflicts, half
// of all bank-conflict cycles of the kernel.)  Constants: L.spec2[c * N2 + P], c = 0 (ca), 1 (cb).
#ifndef R8B_CX_SPEC_BY_POSITION
#define R8B_CX_SPEC_BY_POSITION 1
#endif

R8B_HD int cx_negpos(int p)
{
	if (p == 0) return 0;
	const int m = 31 - __builtin_clz((unsigned) p);
	return (3 << m) - 1 - p;
}

template<int LOGN, int UPLOG>
R8B_HD void cx_spec2_compute(const ConvLaunch& L, const cd* buf, ConvxState<LOGN, UPLOG>& st, int tid)
{
	typedef ConvxState<LOGN, UPLOG> St;
	constexpr int N2 = St::N2, SH = UPLOG > 0 ? UPLOG : 0;
#pragma unroll
	for (int s = 0; s < St::SO; s++)
	{
		const int P = tid + s * kConvxThreads;
		if (P >= N2) continue;
		const int pf = P >> SH;
		const cd u = buf[cpad(pf)], v = buf[cpad(cx_negpos(pf))];
		const cd ca = L.spec2[P], cb = L.spec2[N2 + P];
		cd o;
		o.re = ca.re * u.re - ca.im * u.im + cb.re * v.re + cb.im * v.im;
		o.im = ca.re * u.im + ca.im * u.re + cb.im * v.re - cb.re * v.im;
		st.so[s] = o;
	}
}

template<int LOGN, int UPLOG>
R8B_HD void cx_spec2_write(cd* buf, const ConvxState<LOGN, UPLOG>& st, int tid)
{
	typedef ConvxState<LOGN, UPLOG> St;
#pragma unroll
	for (int s = 0; s < St::SO; s++)
	{
		const int P = tid + s * kConvxThreads;
		if (P < St::N2) buf[cpad(P)] = st.so[s];
	}
}

// spectral stage, part 1: forward bins (kf, N-kf) -> registers
template<int LOGN, int UPLOG>
R8B_HD void cx_spec_read(const cd* buf, ConvxState<LOGN, UPLOG>& st, int tid)
{
	constexpr int N = 1 << LOGN;
	if constexpr (UPLOG < 0)
	{
		// decimating: slot -> backward pair (k, N2-k), fed by forward bins k, N-k, N2-k, N-N2+k
		constexpr int LOGN2 = LOGN + UPLOG, N2 = 1 << LOGN2;
#pragma unroll
		for (int s = 0; s < ConvxState<LOGN, UPLOG>::SP; s++)
		{
			const int k = cx_spec_bin<LOGN2>(tid + s * kConvxThreads);
			if (k <= N2 / 2)
			{
				st.sp[s][0] = buf[cpad(bitrev_n(k, LOGN))];
				st.sp[s][1] = buf[cpad(bitrev_n((N - k) & (N - 1), LOGN))];
				st.sp[s][2] = buf[cpad(bitrev_n(N2 - k, LOGN))];
				st.sp[s][3] = buf[cpad(bitrev_n(N - N2 + k, LOGN))];
			}
		}
		return;
	}
#pragma unroll
	for (int s = 0; s < ConvxState<LOGN, UPLOG>::SP; s++)
	{
		const int kf = cx_spec_bin<LOGN>(tid + s * kConvxThreads);
		if (kf <= N / 2)
		{
			st.sp[s][0] = buf[cpad(bitrev_n(kf & (N - 1), LOGN))];
			st.sp[s][1] = buf[cpad(bitrev_n((N - kf) & (N - 1), LOGN))];
		}
	}
}

// spectral stage, part 2: from the held pair to the 2 (1:1) or 4 (2x) backward inputs.
// Everything between the forward bins and the backward inputs -- real-FFT unpacking, spectrum
// replication (K3), multiplication by the zero-phase kernel (K4), packing for the half-length
// backward transform -- is linear in (Z1, conj Z1, Z2, conj Z2), so the host folds it into
// complex constants per slot (Engine::spectral_constants); the kernel is two complex
// multiply-adds per backward input.  Table layout: constant c of slot s at spec[c * slots + s].
template<int LOGN, int UPLOG>
R8B_HD void cx_spec_write(const ConvLaunch& L, cd* buf, const ConvxState<LOGN, UPLOG>& st, int tid)
{
	constexpr int N = 1 << LOGN, LOGN2 = LOGN + UPLOG;
	constexpr int SLOTS = ConvxState<LOGN, UPLOG>::SLOTS;
	// out = a * (p.re, sp * p.im) + b * (q.re, sq * q.im)
#define R8B_CMADD(out, a, p, sp, b, q, sq) \
	{ \
		const cd ca = (a), cb = (b); \
		const double pi_ = (sp) * (p).im, qi_ = (sq) * (q).im; \
		(out).re = ca.re * (p).re - ca.im * pi_ + cb.re * (q).re - cb.im * qi_; \
		(out).im = ca.re * pi_ + ca.im * (p).re + cb.re * qi_ + cb.im * (q).re; \
	}
	if constexpr (UPLOG < 0)
	{
		// Z'[k]    = c0 Z1 + c1 conj Z2 + c2 conj Z3 + c3 Z4
		// Z'[N2-k] = c4 conj Z1 + c5 Z2 + c6 Z3 + c7 conj Z4
		// (slot k = 0 has one output, the sum of both forms: its Nyquist fix-up needs Z[N2] and
		// Z[N-N2] plain and conjugated, reference CDSPBlockConvolver.h:329-342)
		constexpr int N2 = 1 << LOGN2;
#pragma unroll
		for (int s = 0; s < ConvxState<LOGN, UPLOG>::SP; s++)
		{
			const int slot = tid + s * kConvxThreads;
			const int k = cx_spec_bin<LOGN2>(slot);
			if (k > N2 / 2) continue;
			const cd z1 = st.sp[s][0], z2 = st.sp[s][1], z3 = st.sp[s][2], z4 = st.sp[s][3];
			const cd* c = L.spec + slot;
			cd o1, o2, t;
			R8B_CMADD(o1, c[0], z1, 1.0, c[SLOTS], z2, -1.0)
			R8B_CMADD(t, c[2 * SLOTS], z3, -1.0, c[3 * SLOTS], z4, 1.0)
			o1.re += t.re;
			o1.im += t.im;
			R8B_CMADD(o2, c[4 * SLOTS], z1, -1.0, c[5 * SLOTS], z2, 1.0)
			R8B_CMADD(t, c[6 * SLOTS], z3, 1.0, c[7 * SLOTS], z4, -1.0)
			o2.re += t.re;
			o2.im += t.im;
			if (k == 0)
			{
				o1.re += o2.re;
				o1.im += o2.im;
			}
			buf[cpad(bitrev_n(k, LOGN2))] = o1;
			if (k != 0 && k != N2 / 2) buf[cpad(bitrev_n(N2 - k, LOGN2))] = o2;
		}
	}
	else
	{
#pragma unroll
		for (int s = 0; s < ConvxState<LOGN, UPLOG>::SP; s++)
		{
			const int slot = tid + s * kConvxThreads;
			const int kf = cx_spec_bin<LOGN>(slot);
			if (kf > N / 2) continue;
			const cd z1 = st.sp[s][0], z2 = st.sp[s][1];
			const cd* c = L.spec + slot;
			cd o;
			if constexpr (UPLOG == 0)
			{
				R8B_CMADD(o, c[0], z1, 1.0, c[SLOTS], z2, -1.0)            // Z'[kf]
				buf[cpad(bitrev_n(kf, LOGN2))] = o;
				if (kf != 0 && kf != N / 2)
				{
					R8B_CMADD(o, c[2 * SLOTS], z1, -1.0, c[3 * SLOTS], z2, 1.0) // Z'[N-kf]
					buf[cpad(bitrev_n(N - kf, LOGN2))] = o;
				}
			}
			else
			{
				R8B_CMADD(o, c[0], z1, 1.0, c[SLOTS], z2, -1.0)            // Z'[kf]
				buf[cpad(bitrev_n(kf, LOGN2))] = o;
				R8B_CMADD(o, c[4 * SLOTS], z1, -1.0, c[5 * SLOTS], z2, 1.0) // Z'[N-kf]
				buf[cpad(bitrev_n(N - kf, LOGN2))] = o;
				if (kf != 0)
				{
					R8B_CMADD(o, c[2 * SLOTS], z1, -1.0, c[3 * SLOTS], z2, 1.0) // Z'[2N-kf]
					buf[cpad(bitrev_n(2 * N - kf, LOGN2))] = o;
					R8B_CMADD(o, c[6 * SLOTS], z1, 1.0, c[7 * SLOTS], z2, -1.0) // Z'[N+kf]
					buf[cpad(bitrev_n(N + kf, LOGN2))] = o;
				}
			}
		}
	}
#undef R8B_CMADD
}

// last backward pass, part 1: butterflies into registers
template<int LOGN, int UPLOG>
R8B_HD void cx_final_compute(const ConvLaunch& L, const cd* buf, ConvxState<LOGN, UPLOG>& st,
	int tid)
{
	constexpr int LOGN2 = LOGN + UPLOG, RB = big_pass_bits(LOGN2), R = 1 << RB;
	constexpr int nb = (1 << LOGN2) / R;
	typedef ConvxState<LOGN, UPLOG> St;
#pragma unroll
	for (int f = 0; f < St::FIN; f++)
	{
		const int b = tid + f * kConvxThreads;
		if (b >= nb) continue;
		if constexpr (nb > kConvxThreads)
		{
			cd loc[6];
			tw_fetch<R>(loc, L.tw, L.tw_len, 1 << LOGN2, b);
			xdit_regs<R, true>(buf, 1 << LOGN2, b, loc, st.fr[f], st.fi[f]);
		}
		else xdit_regs<R, true>(buf, 1 << LOGN2, b, st.tw, st.fr[f], st.fi[f]);
	}
}

// first time (virtual rate) of block k's valid output run
R8B_HD long long cx_block_t0(const ConvLaunch& L, long long k)
{
	return k * (long long) L.blk_stride + L.blk_offset - L.fl2;
}

// last backward pass, part 2: the block's valid outputs as one linear run y[u], u in [0, in_len),
// y[u] = convolver output at time t0 + u (reals, unpadded, at `y`).  ZERO_NEG clears the outputs
// at negative times (a stage's stream
// starts at t = 0: for the next stage earlier samples do not exist, its history is zero --
// reference CDSPFracInterpolator.h:834-859).
template<int LOGN, int UPLOG, bool ZERO_NEG>
R8B_HD void cx_final_store(const ConvLaunch& L, double* y, const ConvxState<LOGN, UPLOG>& st,
	long long k, int tid)
{
	constexpr int LOGN2 = LOGN + UPLOG, RB = big_pass_bits(LOGN2), R = 1 << RB;
	constexpr int nb = (1 << LOGN2) / R, q = nb;
	typedef ConvxState<LOGN, UPLOG> St;
	static_assert(UPLOG >= 0 || !ZERO_NEG, "the decimating form has no fused interpolator");
	const int mask = (2 << LOGN2) - 1;
	const long long t0 = cx_block_t0(L, k);
	// decimating: the backward transform's real sample e sits at circular time e * down, the run
	// holds the in_len / down outputs of the block and starts fl2 / down samples before time 0
	const int fl2 = UPLOG < 0 ? L.fl2 >> -UPLOG : L.fl2;
	const int in_len = UPLOG < 0 ? L.in_len >> -UPLOG : L.in_len;
	const int nzero = !ZERO_NEG || t0 >= 0 ? 0 : (-t0 > in_len ? in_len : (int) -t0);
	const bool pair = (fl2 & 1) == 0;
#pragma unroll
	for (int f = 0; f < St::FIN; f++)
	{
		const int b = tid + f * kConvxThreads;
		if (b >= nb) continue;
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			const int e = b + p * q; // complex index: reals 2e, 2e+1 at circular positions c
			const int u0 = (2 * e + fl2) & mask, u1 = (2 * e + 1 + fl2) & mask;
			const double v0 = u0 < nzero ? 0.0 : st.fr[f][p], v1 = u1 < nzero ? 0.0 : st.fi[f][p];
			if (pair && u0 + 1 < in_len)
			{
				// even filter half-length: the pair is one aligned 16-byte store (two 8-byte
				// stores at a 16-byte lane stride conflict two ways)
				cd v;
				v.re = v0;
				v.im = v1;
				*reinterpret_cast<cd*>(y + u0) = v;
				continue;
			}
			if (u0 < in_len) y[u0] = v0;
			if (u1 < in_len) y[u1] = v1;
		}
	}
	// zero extension read (times zero taps) by the padded polyphase rows: a window starts at most at
	// in_len - flen and is FLENP <= 32 long, flen >= 6
	if (tid < kConvxRunPad) y[in_len + tid] = 0.0;
}

// MODE 0: K7, write the block's valid outputs that fall into [a, b).  Decimating: output q sits at
// virtual time q * down; the block's first one is (block start) / down - floor(fl2 / down), both
// in_len and the block starts being multiples of down (reference CDSPBlockConvolver.h:150-165).
template<int UPLOG, int MODE = 0>
R8B_HD void cx_store_conv(const ConvLaunch& L, const double* y, long long k, int ch, int tid)
{
	long long q0 = cx_block_t0(L, k);
	int n = L.in_len;
	if constexpr (MODE == 3)
	{
		if (!L.down_pow2 && L.down > 1)
		{
			// strided decimation (3x): output q sits at virtual time q * down (reference
			// CDSPBlockConvolver.h:564-583); y[u] is virtual time t0 + u
			const long long t0 = q0;
			const long long qf = t0 <= 0 ? -((-t0) / L.down) : (t0 + L.down - 1) / L.down;
			const int u0 = (int) (qf * L.down - t0);
			for (int j = tid; u0 + j * L.down < n; j += kConvxThreads)
			{
				const long long q = qf + j;
				if (q >= L.a && q < L.b) dst_store(L.dst, ch, q, y[u0 + j * L.down]);
			}
			return;
		}
	}
	if constexpr (UPLOG < 0)
	{
		q0 = ((k * (long long) L.blk_stride + L.blk_offset) >> -UPLOG) - (L.fl2 >> -UPLOG);
		n >>= -UPLOG;
	}
	for (int u = tid; u < n; u += kConvxThreads)
	{
		const long long q = q0 + u;
		if (q >= L.a && q < L.b) dst_store(L.dst, ch, q, y[u]);
	}
}

R8B_HD long long ceil_div_pos(long long a, long long b) { return a <= 0 ? 0 : (a + b - 1) / b; }

// MODE 1: K8 on the block's linear run.  Thread t owns outputs j = t (mod OutStep) whose tap
// window [r-fll, r+fl2] lies inside [t0, t0+in_len); r = floor(j*In/Out).
template<int FLENP>
R8B_HD void cx_whole_row(const ConvxLaunch& X, double* row, int tid)
{
	// X.wtab is the bank transposed and permuted by the host for this access: tap i of the row
	// thread t uses (phase t*InStep mod OutStep) sits at wtab[i * OutStep + t], so a wave reads
	// 64 consecutive doubles per tap.  Rows shorter than FLENP are padded with zero taps (the
	// run y[] is zero-extended to match).
	const int t = tid < X.out_step ? tid : 0;
#pragma unroll
	for (int i = 0; i < FLENP; i++) row[i] = i < X.flen ? X.wtab[(long) i * X.out_step + t] : 0.0;
}

#ifndef R8B_CX_ACCS
#define R8B_CX_ACCS 2 // independent accumulators of the tap sum (4 or 8: measured no faster)
#endif
R8B_HD void cx_mac8(const double* row, const double* v, double* s)
{
#pragma unroll
	for (int i = 0; i < 8; i++) s[i % R8B_CX_ACCS] += row[i] * v[i];
}

template<int FLEN>
R8B_HD void cx_whole_compute(const ConvxLaunch& X, const double* y, const double* row, long long k,
	int ch, int tid)
{
	const ConvLaunch& L = X.c;
	if (tid >= X.out_step) return;
	const SpanInfo& B = X.blk[k - L.k0];
	// first j >= jlo with j = tid (mod OutStep)
	int d = tid - B.jlo_mod;
	if (d < 0) d += X.out_step;
	long long j = B.jlo + d;
	const long long jhi = B.jhi;
	if (j >= jhi) return;
	// Every tap is one 8-byte LDS read.  R8B_LDS_WINDOW is supplied by the includer: on the GPU
	// inline-asm ds_read_b64 (256 B/clk; left to the compiler, unaligned pairs become
	// ds_read2_b64 at half that rate, and an aligned two-copy layout costs a second 21 KB of LDS,
	// i.e. one resident workgroup per CU), in the host emulation plain loads.
	int u = B.u_lo + (int) ((unsigned) (B.ph_lo + d * X.in_step) / (unsigned) X.out_step);
	for (; j < jhi; j += X.out_step, u += X.in_step)
	{
		double v[FLEN];
		R8B_LDS_WINDOW(FLEN, v, y + u);
		double s[R8B_CX_ACCS] = {};
		// taps in groups of eight, each behind the arrival of its own reads
		// (the fences keep the scheduler from sinking the multiply-adds below the last wait)
		R8B_LDS_ARRIVED(FLEN, v, 0);
		cx_mac8(row, v, s);
		R8B_SCHED_FENCE();
		R8B_LDS_ARRIVED(FLEN, v, 8);
		cx_mac8(row + 8, v + 8, s);
		R8B_SCHED_FENCE();
		R8B_LDS_ARRIVED(FLEN, v, 16);
		cx_mac8(row + 16, v + 16, s);
		if constexpr (FLEN > 24)
		{
			R8B_SCHED_FENCE();
			R8B_LDS_ARRIVED(FLEN, v, 24);
			cx_mac8(row + 24, v + 24, s);
		}
		double sum = s[0];
#pragma unroll
		for (int i = 1; i < R8B_CX_ACCS; i++) sum += s[i];
		dst_store(X.wdst, ch, j, sum);
	}
}

// ---- the kernel body as a sequence of barrier-separated phases ----------------------------------
//
// `Exec::phase(f)` runs f(tid, state) for every thread of the workgroup and ends with a barrier:
// on the GPU it is `f(threadIdx.x, st); __syncthreads();` (r8b_kernels.hip), in the host
// emulation of tests/emul it is a loop over tid.  Writing the sequence once keeps both in step.

// Two consecutive "rest" passes of equal radix with one butterfly per thread are wave-local: the
// sub-blocks a wave's butterflies read in the second pass are exactly the ones it wrote in the
// first (butterfly b works inside sub-block b / (sub-block size / radix) in both), so they run in
// one phase with a wave-level ordering point instead of a workgroup barrier.
template<int LOGTOT, int RBA, int RBB>
constexpr bool cx_wave_local() { return RBA == RBB && ((1 << LOGTOT) >> RBA) <= kConvxThreads; }

template<int LOGN, int UPLOG, int I, class Exec>
R8B_HD void cx_fwd_seq(Exec& ex, const ConvLaunch& L, cd* buf)
{
	typedef ConvxState<LOGN, UPLOG> St;
	constexpr int NP = FwdPass<LOGN, 0>::NP;
	if constexpr (I >= 1 && I + 2 == NP &&
		cx_wave_local<LOGN, FwdPass<LOGN, I>::rb(), FwdPass<LOGN, I + 1>::rb()>())
	{
		// (the last forward pass has no twiddles, so st.tw serves the first of the pair)
		ex.wave_phase2(
			[&](int tid, St& st) { FwdPass<LOGN, I>::run(buf, st.tw, L.tw, L.tw_len, tid); },
			[&](int tid, St& st) { FwdPass<LOGN, I + 1>::run(buf, st.tw, L.tw, L.tw_len, tid); });
	}
	else
	{
		ex.phase([&](int tid, St& st)
		{
			FwdPass<LOGN, I>::run(buf, st.tw, L.tw, L.tw_len, tid);
			if constexpr (I + 1 < NP) FwdPass<LOGN, I + 1>::prefetch(st.tw, L.tw, L.tw_len, tid);
		});
		if constexpr (I + 1 < NP) cx_fwd_seq<LOGN, UPLOG, I + 1>(ex, L, buf);
	}
}

// backward passes 0 .. NR-1 (the last one, NR, is split into compute/store by the caller)
template<int LOGN, int UPLOG>
constexpr bool cx_inv_pair_local()
{
	constexpr int LOGN2 = LOGN + UPLOG;
	if constexpr (InvPass<LOGN2, 0>::NR == 2)
		return cx_wave_local<LOGN2, InvPass<LOGN2, 0>::rb(), InvPass<LOGN2, 1>::rb()>();
	else
		return false;
}

template<int LOGN, int UPLOG, int I, class Exec>
R8B_HD void cx_inv_seq(Exec& ex, const ConvLaunch& L, cd* buf)
{
	typedef ConvxState<LOGN, UPLOG> St;
	constexpr int LOGN2 = LOGN + UPLOG, NR = InvPass<LOGN2, 0>::NR;
	if constexpr (I == 0 && cx_inv_pair_local<LOGN, UPLOG>())
	{
		// pass 0 (sub-length = radix) has no twiddles: st.tw already holds pass 1's (fetched by
		// the caller), and the last pass's are fetched behind the pair
		ex.wave_phase2(
			[&](int tid, St& st) { InvPass<LOGN2, 0>::run(buf, st.tw, L.tw, L.tw_len, tid); },
			[&](int tid, St& st)
			{
				InvPass<LOGN2, 1>::run(buf, st.tw, L.tw, L.tw_len, tid);
				InvPass<LOGN2, 2>::prefetch(st.tw, L.tw, L.tw_len, tid);
			});
	}
	else if constexpr (I < NR)
	{
		ex.phase([&](int tid, St& st)
		{
			InvPass<LOGN2, I>::run(buf, st.tw, L.tw, L.tw_len, tid);
			InvPass<LOGN2, I + 1>::prefetch(st.tw, L.tw, L.tw_len, tid);
		});
		cx_inv_seq<LOGN, UPLOG, I + 1>(ex, L, buf);
	}
}

// One workgroup = one block of one channel.  (A persistent variant that walks several blocks and
// prefetches the next block's input during the output phase was tried: the loop-carried state
// pushes hipcc into heavy SGPR/VGPR spilling, 3x slower.  Latency hiding is left to the 3-4
// workgroups resident per CU.)
template<int LOGN, int UPLOG, int MODE, int FLENP, class Exec>
R8B_HD void convx_body(Exec& ex, const ConvxLaunch& X, double* rbuf, long long k, int ch)
{
	typedef ConvxState<LOGN, UPLOG> St;
	constexpr int LOGN2 = LOGN + UPLOG;
	constexpr int NPF = FwdPass<LOGN, 0>::NP;
	const ConvLaunch& L = X.c;
	cd* const buf = reinterpret_cast<cd*>(rbuf);
	ex.phase([&](int tid, St& st)
	{
		cx_prefetch<LOGN, UPLOG, MODE>(L, st, k, ch, tid);
		if constexpr (St::FF == 1)
		{
			if (tid < (1 << LOGN) / St::RF)
				tw_fetch<St::RF>(st.tw0, L.tw, L.tw_len, 1 << LOGN, tid);
		}
		if constexpr (NPF > 1) FwdPass<LOGN, 1>::prefetch(st.tw, L.tw, L.tw_len, tid);
		cx_first_pass<LOGN, UPLOG>(L, buf, st, tid);
	});
	if constexpr (NPF > 1) cx_fwd_seq<LOGN, UPLOG, 1>(ex, L, buf);
	ex.phase([&](int tid, St& st)
	{
		if constexpr (UPLOG >= 0 && R8B_CX_SPEC_BY_POSITION) cx_spec2_compute<LOGN, UPLOG>(L, buf, st, tid);
		else cx_spec_read<LOGN, UPLOG>(buf, st, tid);
	});
	ex.phase([&](int tid, St& st)
	{
		if constexpr (UPLOG >= 0 && R8B_CX_SPEC_BY_POSITION) cx_spec2_write<LOGN, UPLOG>(buf, st, tid);
		else cx_spec_write<LOGN, UPLOG>(L, buf, st, tid);
		if constexpr (cx_inv_pair_local<LOGN, UPLOG>())
			InvPass<LOGN2, 1>::prefetch(st.tw, L.tw, L.tw_len, tid);
		else
			InvPass<LOGN2, 0>::prefetch(st.tw, L.tw, L.tw_len, tid);
	});
	cx_inv_seq<LOGN, UPLOG, 0>(ex, L, buf);
	ex.phase([&](int tid, St& st) { cx_final_compute<LOGN, UPLOG>(L, buf, st, tid); });
	ex.phase([&](int tid, St& st)
	{
		cx_final_store<LOGN, UPLOG, MODE == 1>(L, rbuf, st, k, tid);
		if constexpr (MODE == 1) cx_whole_row<FLENP>(X, st.row, tid);
	});
	// history for the next call (stage 0 only): the tail of the caller's buffer goes into the other history ring, every
	// workgroup of a channel copying its share (block b of the launch's nblk the b-th slice), eight samples in
	// flight per thread (cf. r8b_convp.h); issued before the output phase, the stores need no wait
	if (L.tail_ring != nullptr)
	{
		// (64-bit products: an unfused launch -- launch_stage -- carries as many blocks as the call holds)
		const unsigned long long tn = (unsigned long long) (L.tail_p1 - L.tail_p0), nb = (unsigned long long) L.nblk,
			bi = (unsigned long long) (k - L.k0);
		const long long s0 = L.tail_p0 + (long long) (tn * bi / nb), s1 = L.tail_p0 + (long long) (tn * (bi + 1u) / nb);
		ex.each([&](int tid, St&)
		{
			constexpr int TB = 8;
			for (long long i0 = s0 + tid; i0 < s1; i0 += (long long) TB * kConvxThreads)
			{
				double v[TB];
#pragma unroll
				for (int j = 0; j < TB; j++)
				{
					const long long i = i0 + (long long) j * kConvxThreads;
					v[j] = 0.0;
					if (i < s1) v[j] = src_load(L.src, ch, i);
				}
#pragma unroll
				for (int j = 0; j < TB; j++)
				{
					const long long i = i0 + (long long) j * kConvxThreads;
					if (i < s1) L.tail_ring[(long long) ch * L.src.ring_stride + (i & L.src.ring_mask)] = v[j];
				}
			}
		});
	}
	ex.phase([&](int tid, St& st)
	{
		if constexpr (MODE == 1) cx_whole_compute<FLENP>(X, rbuf, st.row, k, ch, tid);
		else cx_store_conv<UPLOG, MODE>(L, rbuf, k, ch, tid);
	});
}

} // namespace r8bhip

#endif
