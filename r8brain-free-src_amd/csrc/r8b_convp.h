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
//     destination, nothing goes back to LDS;  MODE 3: the same behind a 3x zero-stuffing load and / or
//     in front of a 3x strided store (ratios 3/1, 1/3, 2/3).
// Per block and channel: half as many barrier-separated phases as the one-channel form, no spectral
// stage, ~45 % less LDS traffic in the transforms.  LDS: one array of N2 complex (64 KB for the
// 4096-point backward transform), two workgroups per CU; the XOR swizzle pswz() makes every pass
// conflict free for 16-byte accesses without pad slots.
//
// Geometries: backward transforms of 64 ... 4096 points, 1:1 or 2x up-sampling (4096: BASELINE configs
// 2-5).  A thread always owns 16 elements of the backward transform, so a block pair takes NT = N2 / 16
// threads and a 256-thread workgroup carries SUB = 4096 / N2 consecutive blocks of its channel pair, each
// in its own N2-element part of the 64 KB array (short filters: the per-workgroup fixed costs and the
// barriers are shared by up to 16 blocks, and below 1024 points a block never leaves its wave).  The
// interpolator at the end works workgroup-wide over the runs of all SUB blocks.  Everything else stays on
// r8b_convx.h.
//
// Reference semantics reproduced: CDSPBlockConvolver.h:252-354, 512-593, 606-629;
// CDSPRealFFT.h:289-385; CDSPFracInterpolator.h:991-1060.
#ifndef R8B_CONVP_H
#define R8B_CONVP_H

#include <cstdlib>
#include "r8b_convx.h"

// R8B_FORCE4: the four values are computed HERE (device: an empty asm statement that reads them)
#ifndef R8B_FORCE4
#define R8B_FORCE4(a, b, c, d)
#endif
// R8B_OUT_STORE16: a 16-byte store of an output pair to the caller's rows (the device build may mark it non-temporal)
// R8B_OPAQUE2: the two values are produced HERE as far as the compiler can tell (device: an empty asm statement that
// rewrites them) -- what is computed from them stays where it is written, e.g. inside a loop
#ifndef R8B_OPAQUE2
#define R8B_OPAQUE2(a, b)
#endif
// R8B_MEM_FENCE: no memory access moves across this point (device: an empty asm statement with a memory clobber -- the
// scheduling fence alone does not keep the LDS reads of a window's later chunks from being hoisted to the loop's top)
#ifndef R8B_MEM_FENCE
#if defined(__HIP_DEVICE_COMPILE__)
#define R8B_MEM_FENCE() asm volatile("" ::: "memory")
#else
#define R8B_MEM_FENCE()
#endif
#endif
#ifndef R8B_OUT_STORE16
#define R8B_OUT_STORE16(ptr, v) { *reinterpret_cast<cd*>(ptr) = (v); }
#endif
// R8B_OUT_STORE16U: the same pair to an address that is only 8-byte aligned (a call whose outputs start at an odd column
// of the caller's rows).  Device: ONE 16-byte store instruction -- global memory takes vector accesses at element
// alignment --; host: two doubles.
#ifndef R8B_OUT_STORE16U
#define R8B_OUT_STORE16U(ptr, v) { (ptr)[0] = (v).re; (ptr)[1] = (v).im; }
#endif
// R8B_IN_LOAD16U: two neighbouring doubles from an address that is only 8-byte aligned, as ONE load on the device
#ifndef R8B_IN_LOAD16U
#define R8B_IN_LOAD16U(ptr, a, b) { (a) = (ptr)[0]; (b) = (ptr)[1]; }
#endif

namespace r8bhip {

// ---- timing ablations (development builds only: tools/variant.sh ... -DR8B_ABL_TABLES=n; results are WRONG, the time says
// what a resource costs; round 5, profiles/r05_ceiling.txt).  The shipped build expands every knob to the plain access.
//   R8B_ABL_TABLES  1: every table fetch (twiddles, kernel constants, interpolator rows) reads one of four entries of its
//                      row whatever the lane -- the same instructions, a handful of cache lines that stay in the CU's L1
//                      (no L1 <- L2 fill traffic for tables);  2: no fetch at all, the value is made up in registers
//                      (what a form that keeps every table in registers could reach at best);  3: the fetch is a
//                      16-byte LDS read at lane-consecutive addresses (what a form that keeps the tables in LDS pays)
//   R8B_ABL_SAMPLES 1: the block's sample loads read eight samples of the row whatever the lane (no HBM reads)
//   R8B_ABL_STORES  1: the interpolator's output stores land in the first 64 outputs of the block's rows (no HBM writes)
#ifndef R8B_ABL_TABLES
#define R8B_ABL_TABLES 0
#endif
#ifndef R8B_ABL_SAMPLES
#define R8B_ABL_SAMPLES 0
#endif
#ifndef R8B_ABL_STORES
#define R8B_ABL_STORES 0
#endif
// (per table class: R8B_ABL_T twiddles, R8B_ABL_H kernel constants, R8B_ABL_R interpolator rows; default R8B_ABL_TABLES)
#ifndef R8B_ABL_T
#define R8B_ABL_T R8B_ABL_TABLES
#endif
#ifndef R8B_ABL_H
#define R8B_ABL_H R8B_ABL_TABLES
#endif
#ifndef R8B_ABL_R
#define R8B_ABL_R R8B_ABL_TABLES
#endif
#if R8B_ABL_T || R8B_ABL_H || R8B_ABL_R
R8B_HD cd abl_made_up(int uoff, int loff)
{
	cd v;
	v.re = 0.7;
	v.im = -0.7;
	(void) uoff;
	(void) loff;
#ifdef __HIP_DEVICE_COMPILE__
	asm volatile("" : "+v"(v.re), "+v"(v.im));
#endif
	return v;
}
R8B_HD cd abl_from_lds(int uoff, int loff)
{
#ifdef __HIP_DEVICE_COMPILE__
	typedef double abl_d2_t __attribute__((ext_vector_type(2)));
	const abl_d2_t t = *(const __attribute__((address_space(3))) abl_d2_t*) (size_t) ((unsigned) (((uoff) + (loff)) & 4095) << 4);
	cd v;
	v.re = t.x;
	v.im = t.y;
	return v;
#else
	cd v;
	v.re = v.im = 0.0;
	(void) uoff;
	(void) loff;
	return v;
#endif
}
#endif
#define R8B_TAB_LD_0(base, uoff, loff) ((base)[(uoff) + (loff)])
#define R8B_TAB_LD_1(base, uoff, loff) ((base)[(uoff) + ((loff) & 3)])
#define R8B_TAB_LD_2(base, uoff, loff) abl_made_up((uoff), (loff))
#define R8B_TAB_LD_3(base, uoff, loff) abl_from_lds((uoff), (loff))
#define R8B_TAB_CAT2(a, b) a##b
#define R8B_TAB_CAT(a, b) R8B_TAB_CAT2(a, b)
#define R8B_TAB_LD_T R8B_TAB_CAT(R8B_TAB_LD_, R8B_ABL_T)
#define R8B_TAB_LD_H R8B_TAB_CAT(R8B_TAB_LD_, R8B_ABL_H)
#define R8B_TAB_LD_R R8B_TAB_CAT(R8B_TAB_LD_, R8B_ABL_R)

static const int kConvpThreads = 256;

// R8B_SPLIT_UP2 (development builds, tools/variant.sh: the occupancy experiment of round 4): the 2x up-sampling pair
// kernel with 2048 -> 4096-point transforms computes its backward transform as TWO 2048-point transforms -- the even
// outputs from Z (H[k] + H[k+N]), the odd ones from Z (H[k] - H[k+N]) e^{+2 pi i k / 2N} -- one after the other in a
// 32 KB array, so that three or four workgroups fit a CU instead of two (cp_split_*, convolver-only modes).
#ifdef R8B_SPLIT_UP2
template<int LN, int UL> constexpr bool kSplit = LN == 11 && UL == 1;
#else
template<int LN, int UL> constexpr bool kSplit = false;
#endif

// physical slot of complex element e: index bits 0-3 XOR bits 4-7.  Every access pattern of the
// passes below (lanes = consecutive elements, lanes = elements 4 / 8 / 16 apart, 8 or 16
// consecutive elements per lane) then touches 16 different 16-byte bank groups per 16-lane
// service group.
R8B_HD int pswz(int e) { return e ^ ((e >> 4) & 15); }

// mode 19: polyphase 3x form (cp_p3_*; geometries <10, 0> ... <12, 0>)
constexpr bool convp_mode_p3(int m) { return m == 19; }
// kernel modes of the long-block forms on the 8192-point geometries (convp_body): split 2x up-sampling form 8 / 9
// (12 / 13 with a complex kernel spectrum), one-channel form 10 / 11 (14 / 15)
constexpr bool convp_mode_sp(int m) { return m == 8 || m == 9 || m == 12 || m == 13; }
// (18: the one-channel form with the whole-step interpolator fused in -- round 5, cp_solo_final_store / cp_whole_compute_solo)
constexpr bool convp_mode_solo(int m) { return m == 10 || m == 11 || m == 14 || m == 15 || m == 18; }

template<int LN, int UL>
struct ConvpGeom
{
	// UL = 0: 1:1; 1: 2x up-sampling; -1 / -2: 2x / 4x decimation in the spectrum
	static constexpr int N = 1 << LN, LN2 = LN + UL, N2 = 1 << LN2;
	static constexpr int DL = UL < 0 ? -UL : 0;
	static_assert(N <= 8192 && N2 <= 8192 && (N >= 64 || N2 >= 64) && N2 >= 16 && UL >= -2 && UL <= 1,
		"pair kernel: transforms of 64 ... 8192 points");
	static constexpr int NA = N > N2 ? N : N2;        // a block pair's part of the array (complex elements)
	static constexpr int NT = NA / 16;                // threads per block pair
	static constexpr int WT = NT > kConvpThreads ? NT : kConvpThreads; // threads per workgroup (512 for 8192 points)
	static constexpr int SUB = WT / NT;               // block pairs per workgroup
	static constexpr int E1 = N / NT, E2 = N2 / NT;    // elements per thread, forward / backward
	static constexpr int EB1 = UL == 1 ? 3 : 4;
	static constexpr int NPRE = (LN - 1) / EB1;   // forward passes before the middle one (radix E1)
	static constexpr int MB = LN - NPRE * EB1;    // log2 radix of the forward butterflies in the middle pass
	static constexpr int RM = 1 << MB, NBF = E1 / RM;
	// backward passes after the middle one, E2 = 16 (sub-length 16): sub-length 256 (radix 16; B1: only
	// when N2 >= 256), then sub-length N2 (none when N2 = 256): radix R2, NB2 butterflies per thread
	static constexpr bool B1 = N2 >= 256;
	static constexpr int R2 = B1 ? N2 / 256 : N2 / 16, NB2 = R2 > 1 ? 16 / R2 : 0;
	static constexpr int NBASE2 = R2 >= 16 ? 6 : (R2 >= 8 ? 4 : (R2 >= 4 ? 3 : 1));
	static constexpr int NTW = NB2 * NBASE2 > 6 ? NB2 * NBASE2 : 6;
	// decimating form (E2 = 8 or 4) and 8192-point backward transforms (E2 = 16), the mirror image of the
	// forward side: the middle pass does the first MB2 bits (NBB butterflies of radix RMB over the thread's
	// E2 consecutive positions), then NPOST passes of radix E2, one butterfly per thread, sub-lengths RMB E2,
	// RMB E2^2, ..., N2
	static constexpr bool POST = DL > 0 || N2 > 4096;
	static constexpr int EB2 = 4 - DL;
	static constexpr int NPOST = POST ? (LN2 - 1) / EB2 : 0;
	static constexpr int MB2 = LN2 - NPOST * EB2, RMB = 1 << MB2, NBB = E2 / RMB;
	// Wave w of a block pair (NW waves each) owns forward positions [w N/NW, (w+1) N/NW) after the first
	// pass and backward positions [w N2/NW, (w+1) N2/NW) up to the last pass: the passes in between
	// never leave that range (their butterflies span at most 64 E1 / 64 E2 consecutive elements), so they
	// need no workgroup barrier -- provided the wave's forward and backward data share one part of the
	// array: with 2x up-sampling forward position p sits at slot fslot(p) = (p / FW) * BW + p mod FW, when
	// decimating backward position p at bslot(p) = (p / BW) * FW + p mod BW  (the identity when N2 = N or
	// when a block pair fits one wave).
	static constexpr int NW = NT >= 64 ? NT / 64 : 1;
	static constexpr int FW = N / NW, BW = N2 / NW;
};

// position of a thread inside its workgroup: block pair `sub`, thread `lt` of that pair
template<int LN, int UL> R8B_HD int convp_sub(int tid)
{
	if constexpr (ConvpGeom<LN, UL>::SUB == 1) return 0;
	else return tid / ConvpGeom<LN, UL>::NT;
}
template<int LN, int UL> R8B_HD int convp_lt(int tid)
{
	if constexpr (ConvpGeom<LN, UL>::SUB == 1) return tid;
	else return tid & (ConvpGeom<LN, UL>::NT - 1);
}

// (HAF: the half-array form -- mode 21, cp_ha_* --: the forward transform alone in an array of N complex elements, identity map)
template<int LN, int UL, bool HAF = false>
R8B_HD int fslot(int p)
{
	typedef ConvpGeom<LN, UL> G;
	if constexpr (UL <= 0 || G::NW == 1 || kSplit<LN, UL> || HAF) return pswz(p);
	else return pswz((p / G::FW) * G::BW + (p & (G::FW - 1)));
}

template<int LN, int UL, bool HAF = false>
R8B_HD int bslot(int p)
{
	typedef ConvpGeom<LN, UL> G;
	if constexpr (UL >= 0 || G::NW == 1 || HAF) return pswz(p);
	else return pswz((p / G::BW) * G::FW + (p & (G::BW - 1)));
}

// The swizzle is linear over XOR and the wave maps only move bits, so for an element e0 | d whose offset d (a
// compile-time constant: p times a power-of-two stride) has no bit in common with e0 the slot is
//   slot(e0 | d) = (slot(e0) ^ xc) + hi,   xc = (m ^ (m >> 4)) & 15, hi = m & ~15, m = map(d):
// one address register per pass (the slot of e0, as a BYTE offset), one XOR per access when xc != 0 and the rest in
// the LDS instruction's immediate offset -- instead of a full swizzle (4 ... 7 integer instructions) per access.
R8B_HD constexpr int sw_xc(int m) { return (m ^ (m >> 4)) & 15; }
R8B_HD constexpr int sw_hi(int m) { return m & ~15; }
template<int LN, int UL, bool HAF = false>
R8B_HD constexpr int fmap_c(int d)
{
	typedef ConvpGeom<LN, UL> G;
	return (UL <= 0 || G::NW == 1 || kSplit<LN, UL> || HAF) ? d : (d / G::FW) * G::BW + (d & (G::FW - 1));
}
template<int LN, int UL, bool HAF = false>
R8B_HD constexpr int bmap_c(int d)
{
	typedef ConvpGeom<LN, UL> G;
	return (UL >= 0 || G::NW == 1 || HAF) ? d : (d / G::BW) * G::FW + (d & (G::BW - 1));
}
// SwBase: the per-pass address register.  On the GPU it is the ABSOLUTE LDS byte address of slot(e0): the XOR
// constants live in address bits 4-7 and every block pair's array starts on a multiple of 256 bytes (the dynamic LDS
// segment starts at 0, an array is NA * 16 bytes), so the XOR may be applied to the address itself -- which also keeps
// the compiler from adding the segment's base (a late constant it does not fold) to every access.  The host emulation
// (tests/emul) has no such alignment and XORs the offset.
#if defined(R8B_LDS_ABS) && defined(__HIP_DEVICE_COMPILE__) // (the host pass of hipcc compiles the other form, never runs it)
typedef __attribute__((address_space(3))) cd lds_cd_t;
struct SwBase { unsigned a; };
R8B_HD SwBase sw_base(const cd* buf, int slot0)
{
	SwBase b;
	b.a = (unsigned) (size_t) (const lds_cd_t*) buf + ((unsigned) slot0 << 4);
	return b;
}
// (the element as ONE 16-byte access -- ds_read_b128 / ds_write_b128 -- by its type: left to the compiler's merging of
// the two 8-byte halves some phases came out as pairs of ds_read_b64, which the swizzle is not conflict free for)
typedef double lds_d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lds_d2_t lds_d2a_t;
#ifdef R8B_LDS_STRUCT_ACCESS // (development A/B: the element accessed as the struct it is, merging left to the compiler)
R8B_HD cd sw_ld(SwBase b, int m)
{
	return *(const lds_cd_t*) (size_t) ((b.a ^ (unsigned) (sw_xc(m) << 4)) + (unsigned) (sw_hi(m) << 4));
}
R8B_HD void sw_st(SwBase b, int m, cd v)
{
	*(lds_cd_t*) (size_t) ((b.a ^ (unsigned) (sw_xc(m) << 4)) + (unsigned) (sw_hi(m) << 4)) = v;
}
#else
R8B_HD cd sw_ld(SwBase b, int m)
{
	const lds_d2_t t = *(const lds_d2a_t*) (size_t) ((b.a ^ (unsigned) (sw_xc(m) << 4)) + (unsigned) (sw_hi(m) << 4));
	cd v;
	v.re = t.x;
	v.im = t.y;
	return v;
}
R8B_HD void sw_st(SwBase b, int m, cd v)
{
	lds_d2_t t;
	t.x = v.re;
	t.y = v.im;
	*(lds_d2a_t*) (size_t) ((b.a ^ (unsigned) (sw_xc(m) << 4)) + (unsigned) (sw_hi(m) << 4)) = t;
}
#endif
#else
struct SwBase { char* p; int bb; };
R8B_HD SwBase sw_base(const cd* buf, int slot0)
{
	SwBase b;
	b.p = reinterpret_cast<char*>(const_cast<cd*>(buf));
	b.bb = slot0 << 4;
	return b;
}
R8B_HD cd sw_ld(SwBase b, int m)
{
	return *reinterpret_cast<const cd*>(b.p + ((b.bb ^ (sw_xc(m) << 4)) + (sw_hi(m) << 4)));
}
R8B_HD void sw_st(SwBase b, int m, cd v)
{
	*reinterpret_cast<cd*>(b.p + ((b.bb ^ (sw_xc(m) << 4)) + (sw_hi(m) << 4))) = v;
}
#endif

template<int LN, int UL>
struct ConvpState
{
	double vr[16], vi[16];
	double pr[16], pi[16]; // the block's input samples (channel A, channel B) of the first pass
	cd tw[ConvpGeom<LN, UL>::NTW];
	cd hp[16]; // (8 pairs of reals; 16 complex values when the kernel spectrum is complex)
	double row[32];
	double rows2[2 * 27]; // modes 4 / 5: the two rows of the thread's phase pair (25 or 27 entries each)
	int pt;               // ... and its entry of X.ptab
	double pk[2];         // ... the thread's element of the previous call's parked outputs and where it goes (cp_park_slice_*)
	double* pka;
	int pf;               // ... and whether the workgroup holds the call's last block and parks what lies beyond the call
	double tk[2];         // the thread's element of the history tail behind the last block's window (cp_tail_slice_*)
	double* tka;
	double er[16], ei[16]; // split 2x up-sampling form (modes 8 / 9 / 12 / 13): the even half's outputs while the odd half is transformed
	double zr[16], zi[16];   // polyphase 3x form (mode 19): the block's spectrum, kept across the three backward transforms
	double p3r[16], p3i[16]; // ... the second component's outputs (the first's wait in er / ei; kP3Keep of the sixteen)
	cd twlv;              // the thread's entry of the wave-local passes' twiddle table on its way to LDS (front -> first pass)
	cd twp[4];            // walk form (convp_walk): the thread's own twiddles, kept across blocks -- [0], [1] first pass (w, w^4), [2], [3] last backward pass
};

// (+ 64 bytes behind the array: one word per wave for the workgroup-wide "channel has a non-zero sample" bits; + 256 bytes:
// the blocks' level words -- cp_level_words --, one per wave or, where a wave carries several blocks, one per block;
// ... and the blocks' shift words -- cp_level_shift, what the end of the body reads)
static const int kConvpLevelWords = 64;
static const int kConvpFlagBytes = 64 + 8 * kConvpLevelWords;
template<int LN, int UL> constexpr int convp_array_bytes()
{
	return kSplit<LN, UL> ? ConvpGeom<LN, UL>::N * 16 : ConvpGeom<LN, UL>::SUB * ConvpGeom<LN, UL>::NA * 16;
}

// Twiddles of the wave-local passes through LDS (round 5).  The passes between the first forward and the last
// backward one have few DISTINCT twiddles -- JM = 4 ... 32 entries per row (ptw_fetch), the same for every wave, block
// and channel pair -- yet every thread fetched its 4 + 4 + 6 of them with vector loads of its own: 14 of the kernel's
// ~50 table loads per thread, each a trip through the address unit and the L1 that the pass then waits for
// (profiles/r05_ceiling.txt: the kernel gains more from a table load NOT ISSUED than from its bytes not moved).  Now
// the workgroup's first NE threads fetch ONE entry each at kernel entry (NE <= 240: the whole table), leave it in LDS
// behind the array and the flags before the first barrier, and a pass reads its base powers from there when it starts
// (16-byte LDS reads of a few consecutive entries: conflict free, and no register is held across phases for them).
// Forward pass I uses slot I (rows of JM = min(n / E1, NT) entries, n = N >> (I EB1)), the backward pass with sub-length
// 256 slot 3 (16 entries per row).  Geometries with the mirrored backward side (decimating, 8192 points) keep their
// global fetches.
#ifndef R8B_TW_LDS
#define R8B_TW_LDS 1
#endif
template<int LN, int UL>
struct ConvpTwLds
{
	typedef ConvpGeom<LN, UL> G;
	static constexpr bool ON = R8B_TW_LDS != 0 && !G::POST && !kSplit<LN, UL>;
	static constexpr int NBF = G::E1 >= 16 ? 6 : (G::E1 >= 8 ? 4 : (G::E1 >= 4 ? 3 : 1)); // base powers of a forward pass
	static constexpr int jm(int i) { return ((G::N >> (i * G::EB1)) / G::E1 < G::NT) ? (G::N >> (i * G::EB1)) / G::E1 : G::NT; }
	static constexpr int JM1 = ON && G::NPRE > 1 ? jm(1) : 0;
	static constexpr int JM2 = ON && G::NPRE > 2 ? jm(2) : 0;
	static constexpr int JM3 = ON && G::B1 ? (16 < G::NT ? 16 : G::NT) : 0;
	static constexpr int O1 = 0, O2 = O1 + NBF * JM1, O3 = O2 + NBF * JM2, NE = O3 + 6 * JM3;
	static_assert(NE <= G::WT, "one entry per thread");
	// entry e of the LDS table -> its index in the global table (rows of NT entries, 6 rows per slot)
	static R8B_HD int src_index(int e)
	{
		int slot = 1, jmv = JM1, r = e;
		if (e >= O3) { slot = 3; jmv = JM3; r = e - O3; }
		else if (e >= O2) { slot = 2; jmv = JM2; r = e - O2; }
		if (jmv == 0) return 0;
		const int c = r / jmv, t = r - c * jmv;
		return (slot * 6 + c) * G::NT + t;
	}
};
template<int LN, int UL> constexpr int convp_lds_bytes()
{
	return convp_array_bytes<LN, UL>() + kConvpFlagBytes + (ConvpTwLds<LN, UL>::ON ? ConvpTwLds<LN, UL>::NE * 16 : 0);
}
// the table as the phases see it: ltw = the workgroup's LDS + convp_array_bytes + kConvpFlagBytes
#if defined(R8B_LDS_ABS) && defined(__HIP_DEVICE_COMPILE__)
R8B_HD cd twl_ld(const cd* ltw, int idx)
{
	const lds_d2_t t = *(const lds_d2a_t*) (size_t) ((unsigned) (size_t) (const lds_cd_t*) ltw + ((unsigned) idx << 4));
	cd v;
	v.re = t.x;
	v.im = t.y;
	return v;
}
R8B_HD void twl_st(cd* ltw, int idx, cd v)
{
	lds_d2_t t;
	t.x = v.re;
	t.y = v.im;
	*(lds_d2a_t*) (size_t) ((unsigned) (size_t) (const lds_cd_t*) ltw + ((unsigned) idx << 4)) = t;
}
#else
R8B_HD cd twl_ld(const cd* ltw, int idx) { return ltw[idx]; }
R8B_HD void twl_st(cd* ltw, int idx, cd v) { ltw[idx] = v; }
#endif
// base powers of a pass from the LDS table: rows of JM entries starting at entry OFF
template<int NB, int JM>
R8B_HD void twl_fetch(cd* twr, const cd* ltw, int off, int lt)
{
	const int j = off + (lt & (JM - 1));
#pragma unroll
	for (int c = 0; c < NB; c++) twr[c] = twl_ld(ltw, j + c * JM);
}

// Silence stays silence.  Two channels share one complex transform, so each picks up rounding residue of the
// order of 1e-16 of its PARTNER's amplitude; for a channel whose samples are all zero that residue would be the
// whole output, where the reference -- one object per channel -- returns exact zeros.  Every thread therefore
// reports whether its samples of channel A / B hold anything but zero (bits 0 / 1; -0.0 counts as zero, NaN does
// not); the bits are combined over the workgroup (Exec::post_bits / collect_bits, across the barrier that ends the
// first pass) and a channel without a non-zero sample in any of the workgroup's blocks gets its results replaced
// by zeros before they are stored or interpolated.
template<int LN, int UL>
R8B_HD unsigned cp_nonzero_bits(const ConvpState<LN, UL>& st)
{
#ifdef R8B_NO_SILENCE
	return 3u; // (development: timing without the detection)
#endif
	// (integer form: a double is +-0 exactly when its low word and its high word without the sign are both zero --
	// three cheap integer instructions per two samples instead of a 64-bit compare and a mask merge per sample)
	unsigned a = 0, b = 0;
#pragma unroll
	for (int p = 0; p < ConvpGeom<LN, UL>::E1; p++)
	{
		unsigned long long ua, ub;
		__builtin_memcpy(&ua, &st.pr[p], 8);
		__builtin_memcpy(&ub, &st.pi[p], 8);
		a |= (unsigned) ua | ((unsigned) (ua >> 32) << 1);
		b |= (unsigned) ub | ((unsigned) (ub >> 32) << 1);
	}
	return (a != 0 ? 1u : 0u) | (b != 0 ? 2u : 0u);
}

template<int LN, int UL>
R8B_HD void cp_silence(ConvpState<LN, UL>& st, unsigned nzbits)
{
	if (nzbits == 3u) return; // (uniform over the workgroup: the usual case costs one scalar branch)
	const bool za = !(nzbits & 1u), zb = !(nzbits & 2u);
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		if (za) st.vr[p] = 0.0;
		if (zb) st.vi[p] = 0.0;
	}
}

// Partners at their own level (round 6).  Two channels share one complex transform, so each carries rounding residue of
// the order of 1e-16 of the LOUDER one: a channel at -120 dBFS beside a full-scale partner came out with an error of 1e-9
// of its own level, where the reference -- one object per channel (README.md:52-55, CDSPResampler.h:559-575) -- keeps
// every channel's error at 1e-16 of that channel.  Each block therefore brings its two channels to the same binary
// order of magnitude before they are packed: every thread reports the largest exponent field among its samples of A and
// of B (cp_level_words), the block's threads combine them (Exec::post_levels / collect_levels, across a workgroup
// barrier in front of the first pass), the QUIETER channel's samples are multiplied by 2^d, d = the difference of the
// two exponents, and its outputs by 2^-d behind the last backward pass.  Multiplying by a power of two is exact (no
// result is subnormal that was not before: the scaled channel only moves UP to its partner's exponent; the way back
// is exact unless the output itself is subnormal), so the quiet channel's result is what a transform with a partner of
// its own level gives, the loud one's is unchanged, and partners of equal level (d = 0: one scalar branch) are
// computed exactly as before.  d is a function of the block's window alone -- blocks are anchored at absolute stream
// positions --, so the output stays bitwise independent of how the stream is cut into calls.  A channel without a
// normal sample (zeros, subnormals) or with an Inf / NaN takes part unscaled.  The one-channel forms have no partner.
// (a thread's levels: the larger of its samples' high words without the sign, per channel -- non-negative doubles order
// like their bit patterns; packed levels: A's exponent field in bits 16-26, B's in bits 0-10)
struct CpLevels { unsigned a, b; };
template<int LN, int UL>
R8B_HD CpLevels cp_level_words(const ConvpState<LN, UL>& st)
{
	CpLevels v;
	v.a = v.b = 0;
#pragma unroll
	for (int p = 0; p < ConvpGeom<LN, UL>::E1; p++)
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
R8B_HD unsigned cp_level_pack(CpLevels v) { return ((v.a >> 20) << 16) | (v.b >> 20); }
R8B_HD unsigned cp_level_max(unsigned x, unsigned y)
{
	const unsigned xa = x & 0xffff0000u, ya = y & 0xffff0000u, xb = x & 0xffffu, yb = y & 0xffffu;
	return (xa > ya ? xa : ya) | (xb > yb ? xb : yb);
}
// d > 0: channel B is the quieter one, by d binary orders of magnitude; d < 0: channel A, by -d
R8B_HD int cp_level_shift(unsigned lv)
{
#ifdef R8B_NO_LEVELS
	return 0; // (development: timing without the equalisation's arithmetic)
#endif
	const int la = (int) (lv >> 16), lb = (int) (lv & 0xffffu);
	if (la == 0 || lb == 0 || la == 2047 || lb == 2047) return 0;
	const int d = la - lb;
	return d > 1000 ? 1000 : (d < -1000 ? -1000 : d);
}
R8B_HD double cp_pow2(int n)
{
	const unsigned long long u = (unsigned long long) (1023 + n) << 52;
	double v;
	__builtin_memcpy(&v, &u, 8);
	return v;
}
template<int LN, int UL>
R8B_HD void cp_scale_in(ConvpState<LN, UL>& st, int d)
{
	if (d == 0) return;
	const double s = cp_pow2(d > 0 ? d : -d);
#pragma unroll
	for (int p = 0; p < ConvpGeom<LN, UL>::E1; p++)
	{
		if (d > 0) st.pi[p] *= s;
		else st.pr[p] *= s;
	}
}
// (vr: channel A's outputs, vi: channel B's)
template<int N>
R8B_HD void cp_scale_out(double* vr, double* vi, int d)
{
	if (d == 0) return;
	const double s = cp_pow2(d > 0 ? -d : d);
#pragma unroll
	for (int p = 0; p < N; p++)
	{
		if (d > 0) vi[p] *= s;
		else vr[p] *= s;
	}
}

// Twiddle base powers of a pass, pre-gathered per thread by the host (pair_twiddles() in
// r8b_engine.cpp): entry (row * NT + t) = w_n^(j(t) m_c), row = 6 slot + c, m = {1, 2, 3, 4, 8, 12}; a
// wave reads consecutive 16-byte entries per load (the strided reads of the shared exp() table touch up
// to 64 cache lines per load).  Slots: 0 first pass, 1 / 2 forward passes 1 / 2, 3 backward pass with
// sub-length 256, 4 + m butterfly m of the last backward pass (sub-length N2, j = t + NT m).
// JM: the butterflies of a sub-transform (j = t mod JM): a row holds only JM distinct entries, and reading them at
// t mod JM instead of t keeps a pass's whole table to JM x 16 bytes per row -- a few cache lines that stay in
// the CU's L1 instead of a kilobyte per wave from L2 (the passes between the first forward and the last backward
// one have 4 ... 32 distinct entries per row)
template<int R, int NT, int JM = NT>
R8B_HD void ptw_fetch(cd* twr, const cd* ptw, int slot, int lt)
{
	constexpr int NB = R >= 16 ? 6 : (R >= 8 ? 4 : (R >= 4 ? 3 : 1));
	if constexpr (JM < NT) lt &= JM - 1;
#pragma unroll
	for (int c = 0; c < NB; c++) twr[c] = R8B_TAB_LD_T(ptw, slot * 6 * NT + c * NT, lt);
}

// First forward / last backward pass: every thread has twiddles of its own (NT distinct rows: 16 KB ... 24 KB per pass and
// block from L2, and the L1's miss queue is what the kernel waits for -- DESIGN.md section 5), so only the powers that
// cannot be had from others are fetched -- w (and w^4 for radix 16) -- and the rest of the base set comes from
// products: w^2 = w w, w^3 = w^2 w, (radix 8: w^4 = w^2 w^2,) w^8 = w^4 w^4, w^12 = w^8 w^4.  A twiddle of a butterfly
// is then the product of at most three rounded products instead of one (tw_get); the stream's distance from the
// reference stays where it was (tests: RMS 3e-16).  R8B_TW_DERIVE = 0: every base power fetched, as before.
#ifndef R8B_TW_DERIVE
#define R8B_TW_DERIVE 1
#endif
template<int R, int NT>
R8B_HD void ptw_fetch_lean(cd* twr, const cd* ptw, int slot, int lt)
{
	constexpr int NB = R >= 16 ? 6 : (R >= 8 ? 4 : (R >= 4 ? 3 : 1));
	if constexpr (!R8B_TW_DERIVE) ptw_fetch<R, NT>(twr, ptw, slot, lt);
	else
	{
		twr[0] = R8B_TAB_LD_T(ptw, slot * 6 * NT, lt);
		if constexpr (NB == 6) twr[3] = R8B_TAB_LD_T(ptw, slot * 6 * NT + 3 * NT, lt);
	}
}
R8B_HD cd tw_sq(cd a)
{
	cd r;
	r.re = a.re * a.re - a.im * a.im;
	r.im = 2.0 * (a.re * a.im);
	return r;
}
R8B_HD cd tw_mul(cd a, cd b)
{
	cd r;
	r.re = a.re * b.re - a.im * b.im;
	r.im = a.re * b.im + a.im * b.re;
	return r;
}
// the base set of tw_get (w, w^2, w^3, w^4, w^8, w^12) completed from what ptw_fetch_lean fetched
template<int R>
R8B_HD void tw_expand(cd* twr)
{
	constexpr int NB = R >= 16 ? 6 : (R >= 8 ? 4 : (R >= 4 ? 3 : 1));
	if constexpr (R8B_TW_DERIVE && NB >= 3)
	{
		twr[1] = tw_sq(twr[0]);
		twr[2] = tw_mul(twr[1], twr[0]);
		if constexpr (NB == 4) twr[3] = tw_sq(twr[1]);
		if constexpr (NB == 6)
		{
			twr[4] = tw_sq(twr[3]);
			twr[5] = tw_mul(twr[4], twr[3]);
		}
	}
}

// ---- passes over the swizzled array ---------------------------------------------------------------

// (the arithmetic of a forward pass on the R values a thread has loaded: the butterflies, then the twiddles)
template<int R, bool TW>
R8B_HD void pdif_arith(const cd* twr, double* vr, double* vi)
{
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
}

template<int LN, int UL, int R, bool TW, bool HAF = false>
R8B_HD void pdif(cd* buf, int n, int b, const cd* twr)
{
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
	const SwBase bb = sw_base(buf, fslot<LN, UL, HAF>(e0));
	double vr[R], vi[R];
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = sw_ld(bb, fmap_c<LN, UL, HAF>(p * q));
		vr[p] = v.re;
		vi[p] = v.im;
	}
	// (written out, not pdif_arith(): through the helper the split 2x up-sampling form -- modes 8 / 9, 256 registers -- came
	// out with 84 bytes per lane of scratch)
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
		sw_st(bb, fmap_c<LN, UL, HAF>(p * q), v);
	}
}

// (the arithmetic of a backward pass on the R values a thread has loaded: twiddles, then the butterflies)
template<int R, bool TW>
R8B_HD void pdit_arith(const cd* twr, double* vr, double* vi)
{
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
R8B_HD void pdit_regs(const cd* buf, int n, int b, const cd* twr, double* vr, double* vi)
{
	const int q = n / R;
	const int blk = b / q, j = b - blk * q;
	const int e0 = blk * n + j;
	const SwBase bb = sw_base(buf, pswz(e0));
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = sw_ld(bb, p * q);
		vr[p] = v.re;
		vi[p] = v.im;
	}
	pdit_arith<R, TW>(twr, vr, vi);
}

// ---- phases -----------------------------------------------------------------------------------------
// (buf = the block pair's own N2-element array, lt = the thread's index inside the pair, 0 .. NT-1)

// K1: thread lt owns the radix-E1 butterfly over elements lt + NT p of the first pass; element i of the
// circular block is sample i of channel A (real part) and of channel B (imaginary part).  A wave
// reads 64 consecutive samples of each channel per load.
// (SPU: the split 2x up-sampling form -- modes 8 / 9 on a 1:1 geometry: the block is loaded as a 2x up-sampling one)
template<int LN, int UL, int MODE = 0, bool SPU = false, bool FAST = false>
R8B_HD void cp_load(const ConvLaunch& L, ConvpState<LN, UL>& st, long long k, int chA, int chB, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
	if constexpr (MODE == 3)
	{
		// 3x zero stuffing folded into the load (cf. cx_prefetch, reference CDSPBlockConvolver.h:414-496):
		// element i of the block is virtual sample base_v + rel, i.e. x[(base_v + rel) / up] when up divides
		// it, else 0
		if (!L.up_pow2)
		{
			const long long base_v = k * (long long) L.blk_stride + L.blk_offset;
			const long long B = base_v / L.up;
			const int bm = (int) (base_v - B * L.up);
			const int bias = L.up * (G::N / L.up + 2);
			const SrcBlock sa = src_block(L.src, chA, B), sb = src_block(L.src, chB, B);
#pragma unroll
			for (int p = 0; p < R; p++)
			{
				const int i = lt + p * q;
				const int rel = i < L.in_len ? i : i - G::N;
				st.pr[p] = cx_stuffed_sample(sa, L.up, bm, rel, bias);
				st.pi[p] = cx_stuffed_sample(sb, L.up, bm, rel, bias);
			}
			return;
		}
	}
	constexpr int US = SPU ? 1 : (UL > 0 ? UL : 0); // (L.up == 1 << US)
	const int iln = L.in_len >> US;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) >> US; // (>= 0, even)
	// Element i of the circular array holds sample base + rel((i + rot) mod N), rel(j) = j < iln ? j : j - N: the N
	// CONSECUTIVE samples base - (N - iln) ... base + iln - 1, sample w of that window at element (w - wr) mod N,
	// wr = (rot + N - iln) mod N.
	const int wr = (L.rot + G::N - iln) & (G::N - 1);
	// most blocks of a call lie entirely inside the caller's buffer: one uniform row pointer per channel
	// and a 32-bit offset per load (the general form selects ring / buffer / zero per sample: ~12
	// vector instructions per load)
	// (FAST: the caller knows that the window lies inside the caller's fp64 buffer -- walk form, convp_walk_range)
	if (FAST || (L.src.cur_fmt == kPcmF64 && base - (G::N - iln) >= L.src.cur_base && base - (G::N - iln) >= 0))
	{
		const long long w0 = base - (G::N - iln) - L.src.cur_base;
		const double* const pa = L.src.cur + ((long long) chA * L.src.cur_stride + w0);
		const double* const pb = L.src.cur + ((long long) chB * L.src.cur_stride + w0);
		const unsigned l0 = (unsigned) (lt + wr);
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			const unsigned w = (l0 + (unsigned) (p * q)) & (unsigned) (G::N - 1);
#if R8B_ABL_SAMPLES
			st.pr[p] = pa[w & 7u];
			st.pi[p] = pb[w & 7u];
#else
			st.pr[p] = pa[w];
			st.pi[p] = pb[w];
#endif
		}
		return;
	}
	const SrcBlock sa = src_block(L.src, chA, base), sb = src_block(L.src, chB, base);
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const int i = (lt + p * q + L.rot) & (G::N - 1);
		const int rel = i < iln ? i : i - G::N;
		st.pr[p] = src_block_load1(sa, rel);
		st.pi[p] = src_block_load1(sb, rel);
	}
}

// ---- mode 20: a half-band decimator in front of the block, taken in the block's load (geometry <12, -1>) -------------
// MEASURED AND NOT THE DEFAULT (engine option fuse_hbconv = 0; profiles/r06_experiments.txt item 10): correct -- the
// fused launch equals k_hbdown + k_convp bit for bit -- and 20 % slower than the two launches.
// The block's window is N = 4096 consecutive CONVOLVER inputs n = w0 .. w0 + N - 1, each the decimator's output
//   y[n] = x[2n] + sum_k f[k] (x[2n + 1 + 2k] + x[2n - 1 - 2k])     (reference CDSPHBDownsampler.h:282-295; hbdown_compute)
// of the raw stream x.  Two rounds of kHbfRound = 2048 outputs: the round's 2 * 2048 + 4 TP - 3 raw samples of both
// channels are staged in LDS over the (still unused) array, de-interleaved -- tap samples (odd raw positions) and centre
// samples (even ones) in rows of their own, one pad slot per eight so that lanes eight outputs apart meet different
// banks --, and thread t computes outputs 8 t .. 8 t + 7 of the round from a sliding window of 2 TP + 7 tap samples in
// registers (5 LDS reads per output instead of 2 TP + 1).  Sums in k_hbdown's order: the fused chain equals the two
// launches bit for bit.  The sixteen values per thread and channel then go through the array once (element = window
// sample rotated as cp_load lays it out) and the first pass reads its strided sixteen as if cp_load had fetched them.
static const int kHbfRow = 2352;
static const int kHbfLdsBytes = 4 * kHbfRow * 8;
static_assert(kHbfRound + 2 * kHbfTapsMax + ((kHbfRound + 2 * kHbfTapsMax) >> 3) < kHbfRow, "staging rows of the half-band front");
R8B_HD constexpr int hbf_pad(int j) { return j + (j >> 3); }
struct SlotHbf
{
	// raw index i of a round (0 = an odd raw position): tap sample i / 2 or centre sample i / 2 of the channel's rows
	R8B_HD int operator()(int i) const { return ((i & 1) ? kHbfRow : 0) + hbf_pad(i >> 1); }
};

// round r of block k: the raw samples of both channels into LDS.  Every load of the round is issued before the first
// LDS store (one trip to memory per round: the workgroup has nothing else to do meanwhile, and as loops of eight loads
// the two channels' staging was six dependent trips -- 15 000 cycles per round on a block of 34 000); a span inside the
// caller's fp64 buffer is read as 16-byte (tap, centre) pairs through a uniform row pointer.
template<int LN, int UL>
R8B_HD void cp_hbf_stage(const ConvLaunch& L, const ConvxLaunch& XM, double* xs, long long k, int r, int chA, int chB, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	static_assert(UL < 0 && G::SUB == 1 && G::N == 2 * kHbfRound && G::WT == 256, "half-band front: the 4096-point decimating geometry");
	const int TP = XM.hbf.p.np;
	const long long base = k * (long long) L.blk_stride + L.blk_offset;
	const long long n_r = base - (G::N - L.in_len) + (long long) r * kHbfRound; // the round's first output
	const long long lo = 2 * n_r - (2 * TP - 1);
	const int len = 2 * kHbfRound - 1 + 2 * (2 * TP - 1);
	const int npair = (len + 1) / 2; // (tap, centre) pairs; the last one is a tap alone
	constexpr int NP = (kHbfRound + 2 * kHbfTapsMax + 255) / 256;
	if (L.src.cur_fmt == kPcmF64 && lo >= L.src.cur_base && lo >= 0 && lo + 2 * (long long) npair <= XM.hbf.p.end)
	{
		const double* const pa = L.src.cur + ((long long) chA * L.src.cur_stride + (lo - L.src.cur_base));
		const double* const pb = L.src.cur + ((long long) chB * L.src.cur_stride + (lo - L.src.cur_base));
		double ae[NP], ac[NP], be[NP], bc[NP];
#pragma unroll
		for (int u = 0; u < NP; u++)
		{
			const int j = tid + 256 * u;
			const unsigned jj = (unsigned) (j < npair ? j : npair - 1) * 2u;
			// (element alignment is enough for a 16-byte global load on this target)
			R8B_IN_LOAD16U(pa + jj, ae[u], ac[u]);
			R8B_IN_LOAD16U(pb + jj, be[u], bc[u]);
		}
#pragma unroll
		for (int u = 0; u < NP; u++)
		{
			const int j = tid + 256 * u;
			if (j < npair)
			{
				const int sl = hbf_pad(j);
				xs[sl] = ae[u]; xs[kHbfRow + sl] = ac[u];
				xs[2 * kHbfRow + sl] = be[u]; xs[3 * kHbfRow + sl] = bc[u];
			}
		}
		return;
	}
	const int end = clamp_rel(XM.hbf.p.end - lo);
	const int lim = end < len ? end : len;
	const SrcBlock sa = src_block(L.src, chA, lo), sb = src_block(L.src, chB, lo);
	constexpr int NS = 2 * NP;
	double va[NS], vb[NS];
#pragma unroll
	for (int u = 0; u < NS; u++)
	{
		const int i = tid + 256 * u;
		const int ii = lim <= 0 ? 0 : (i < lim ? i : lim - 1);
		va[u] = lim <= 0 ? 0.0 : src_block_load1(sa, ii);
		vb[u] = lim <= 0 ? 0.0 : src_block_load1(sb, ii);
	}
#pragma unroll
	for (int u = 0; u < NS; u++)
	{
		const int i = tid + 256 * u;
		if (i < len)
		{
			const int sl = SlotHbf()(i);
			xs[sl] = i < lim ? va[u] : 0.0;
			xs[2 * kHbfRow + sl] = i < lim ? vb[u] : 0.0;
		}
	}
}

// the thread's eight outputs of round r, both channels: st.pr / st.pi [8 r + o] = window sample 2048 r + 8 t + o
template<int TP, int RND, int LN, int UL>
R8B_HD void cp_hbf_compute_t(const ConvLaunch& L, const ConvxLaunch& XM, const double* xs, ConvpState<LN, UL>& st, long long k,
	int tid)
{
	constexpr int r = RND;
	typedef ConvpGeom<LN, UL> G;
	const long long base = k * (long long) L.blk_stride + L.blk_offset;
	const long long n0 = base - (G::N - L.in_len) + (long long) r * kHbfRound + 8 * tid; // the thread's first output
	double f[TP];
#pragma unroll
	for (int i = 0; i < TP; i++) f[i] = XM.hbf.p.taps[i];
#pragma unroll
	for (int c = 0; c < 2; c++)
	{
		const LdsWin e = lds_win(xs + c * 2 * kHbfRow), ctr = lds_win(xs + c * 2 * kHbfRow + kHbfRow);
		double ev[2 * TP + 7];
#pragma unroll
		for (int j = 0; j < 2 * TP + 7; j++) ev[j] = e[9 * tid + hbf_pad(j)]; // (hbf_pad(8 t + j) = 9 t + hbf_pad(j))
		double cv[8];
#pragma unroll
		for (int o = 0; o < 8; o++)
		{
			// (centre of output 8 t + o: centre sample 8 t + o + TP - 1)
			const int j = o + TP - 1;
			cv[o] = ctr[9 * tid + hbf_pad(j)];
		}
#pragma unroll
		for (int o = 0; o < 8; o++)
		{
			double s = cv[o];
#pragma unroll
			for (int kk = 0; kk < TP; kk++) s += f[kk] * (ev[o + TP + kk] + ev[o + TP - 1 - kk]);
			// (convolver inputs in front of the stream's start are zeros, not decimator outputs)
			s = n0 + o < 0 ? 0.0 : s;
			if (c == 0) st.pr[8 * r + o] = s;
			else st.pi[8 * r + o] = s;
		}
	}
}

// (RND, the round, at compile time: it indexes the state's register arrays)
template<int RND, int LN, int UL>
R8B_HD void cp_hbf_compute(const ConvLaunch& L, const ConvxLaunch& XM, const double* xs, ConvpState<LN, UL>& st, long long k,
	int tid)
{
	if (XM.hbf.p.np <= 4) cp_hbf_compute_t<4, RND>(L, XM, xs, st, k, tid);
	else if (XM.hbf.p.np <= 8) cp_hbf_compute_t<8, RND>(L, XM, xs, st, k, tid);
	else cp_hbf_compute_t<kHbfTapsMax, RND>(L, XM, xs, st, k, tid);
}

// the thread's sixteen window samples into the array: element i holds window sample (i + wr) mod N (cp_load)
template<int LN, int UL>
R8B_HD void cp_hbf_scatter(const ConvLaunch& L, cd* buf, const ConvpState<LN, UL>& st, int tid)
{
	typedef ConvpGeom<LN, UL> G;
	const int wr = (L.rot + G::N - L.in_len) & (G::N - 1);
#pragma unroll
	for (int r = 0; r < 2; r++)
#pragma unroll
		for (int o = 0; o < 8; o++)
		{
			const int w = r * kHbfRound + 8 * tid + o;
			cd v;
			v.re = st.pr[8 * r + o];
			v.im = st.pi[8 * r + o];
			buf[fslot<LN, UL>((w - wr) & (G::N - 1))] = v;
		}
}

// ... and the first pass's sixteen back out of it (thread lt: elements lt + NT p)
template<int LN, int UL>
R8B_HD void cp_hbf_gather(const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const cd v = buf[fslot<LN, UL>(lt + p * q)];
		st.pr[p] = v.re;
		st.pi[p] = v.im;
	}
}

// History for the next call out of the registers cp_load() filled (convp_tail_owners): the block's own part of the
// tail -- positions base(k) .. base(k + 1) - 1 of its window, clipped to [tail_c0, tail_c1) -- goes to the other
// history ring as it arrives; no load, no wait.
template<int LN, int UL, bool SPU = false>
R8B_HD void cp_tail_owned(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int chA, int chB, bool bvalid, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
	constexpr int US = SPU ? 1 : (UL > 0 ? UL : 0);
	if (k < L.k0 + L.tail_bf) return;
	const int iln = L.in_len >> US;
	const long long base = (k * (long long) L.blk_stride + L.blk_offset) >> US;
	const long long nbase = ((k + 1) * (long long) L.blk_stride + L.blk_offset) >> US;
	const long long lo = k == L.k0 + L.tail_bf || base < L.tail_c0 ? L.tail_c0 : base;
	const long long hi = k == L.k0 + L.nblk - 1 || nbase > L.tail_c1 ? L.tail_c1 : nbase;
	if (lo >= hi) return;
	const long long w0 = base - (G::N - iln);
	// (0 <= lo - w0 < hi - w0 <= N: both inside the window)
	const unsigned lo_r = (unsigned) (lo - w0), n_r = (unsigned) (hi - lo);
	const unsigned m = (unsigned) L.src.ring_mask, w0m = (unsigned) (w0 & L.src.ring_mask);
	double* const ra = L.tail_ring + (long long) chA * L.src.ring_stride;
	double* const rb = L.tail_ring + (long long) chB * L.src.ring_stride;
	const unsigned l0 = (unsigned) (lt + ((L.rot + G::N - iln) & (G::N - 1)));
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const unsigned w = (l0 + (unsigned) (p * q)) & (unsigned) (G::N - 1);
		if (w - lo_r < n_r)
		{
			const unsigned e = (w0m + w) & m;
			ra[e] = st.pr[p];
			if (bvalid) rb[e] = st.pi[p];
		}
	}
}

// ---- one-channel form (modes 10 / 11, 14 / 15 with a complex spectrum; geometries <13, 0> and <13, -1>: 16384-point blocks
// of ONE channel) --------------------------------------------------------------------------------------------------
// A block whose transforms are 16384 real points does not fit a pair's array twice (256 KB); it runs on the 8192-point
// 1:1 geometry as ONE channel in the classic packing z[n] = x[2n] + i x[2n+1]: element i of the circular array holds
// samples 2i, 2i + 1 of the 16384-sample circular block.  (cp_solo_mid_a / _b: what the spectrum needs for that.)
// K1 of the one-channel form: a wave reads 128 consecutive samples per load
template<int LN, int UL, int MODE = 0>
R8B_HD void cp_load_solo(const ConvLaunch& L, ConvpState<LN, UL>& st, long long k, int ch, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R, NR = 2 * G::N;
	if constexpr (MODE == 3)
	{
		// 3x zero stuffing folded into the load, as in cp_load
		if (!L.up_pow2)
		{
			const long long base_v = k * (long long) L.blk_stride + L.blk_offset;
			const long long B = base_v / L.up;
			const int bm = (int) (base_v - B * L.up);
			const int bias = L.up * (NR / L.up + 2);
			const SrcBlock sa = src_block(L.src, ch, B);
#pragma unroll
			for (int p = 0; p < R; p++)
			{
				const int e = 2 * (lt + p * q);
				st.pr[p] = cx_stuffed_sample(sa, L.up, bm, e < L.in_len ? e : e - NR, bias);
				st.pi[p] = cx_stuffed_sample(sa, L.up, bm, e + 1 < L.in_len ? e + 1 : e + 1 - NR, bias);
			}
			return;
		}
	}
	// (in_len is even -- convp_solo_ok --: a pair of samples never straddles the block's seam)
	const int iln = L.in_len;
	const long long base = k * (long long) L.blk_stride + L.blk_offset;
	// Real element e holds sample base + rel((e + 2 rot) mod 2N), rel(j) = j < iln ? j : j - 2N: the 2N consecutive
	// samples base - (2N - iln) ... base + iln - 1, samples 2w, 2w + 1 of that window at element (w - wr) mod N
	const int wr = (L.rot + G::N - iln / 2) & (G::N - 1);
	if (L.src.cur_fmt == kPcmF64 && base - (NR - iln) >= L.src.cur_base && base - (NR - iln) >= 0)
	{
		const long long w0 = base - (NR - iln) - L.src.cur_base;
		const double* const pa = L.src.cur + ((long long) ch * L.src.cur_stride + w0);
		const unsigned l0 = (unsigned) (lt + wr);
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			const unsigned w = (l0 + (unsigned) (p * q)) & (unsigned) (G::N - 1);
			st.pr[p] = pa[2u * w];
			st.pi[p] = pa[2u * w + 1u];
		}
		return;
	}
	const SrcBlock sa = src_block(L.src, ch, base);
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const int e = 2 * ((lt + p * q + L.rot) & (G::N - 1));
		const int rel = e < iln ? e : e - NR;
		st.pr[p] = src_block_load1(sa, rel);
		st.pi[p] = src_block_load1(sa, rel + 1);
	}
}

// ... and its cp_tail_owned: the window is 2N samples, the thread's registers hold samples 2w, 2w + 1 of it
template<int LN, int UL>
R8B_HD void cp_tail_owned_solo(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int ch, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R, NR = 2 * G::N;
	if (k < L.k0 + L.tail_bf) return;
	const int iln = L.in_len;
	const long long base = k * (long long) L.blk_stride + L.blk_offset;
	const long long nbase = (k + 1) * (long long) L.blk_stride + L.blk_offset;
	const long long lo = k == L.k0 + L.tail_bf || base < L.tail_c0 ? L.tail_c0 : base;
	const long long hi = k == L.k0 + L.nblk - 1 || nbase > L.tail_c1 ? L.tail_c1 : nbase;
	if (lo >= hi) return;
	const long long w0 = base - (NR - iln);
	const unsigned lo_r = (unsigned) (lo - w0), n_r = (unsigned) (hi - lo);
	const unsigned m = (unsigned) L.src.ring_mask, w0m = (unsigned) (w0 & L.src.ring_mask);
	double* const ra = L.tail_ring + (long long) ch * L.src.ring_stride;
	const unsigned l0 = (unsigned) (lt + ((L.rot + G::N - iln / 2) & (G::N - 1)));
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		const unsigned w = 2u * ((l0 + (unsigned) (p * q)) & (unsigned) (G::N - 1));
		if (w - lo_r < n_r) ra[(w0m + w) & m] = st.pr[p];
		if (w + 1u - lo_r < n_r) ra[(w0m + w + 1u) & m] = st.pi[p];
	}
}

// positions [s0, s1) of the caller's fp64 buffer to the other history ring, by the WT threads of a workgroup, eight
// samples per channel in flight per thread
template<int WT>
R8B_HD void cp_tail_rest(const ConvLaunch& L, long long s0, long long s1, int chA, int chB, bool bvalid, int tid)
{
	constexpr int TB = 8;
	const double* const pa = L.src.cur + ((long long) chA * L.src.cur_stride - L.src.cur_base);
	const double* const pb = L.src.cur + ((long long) chB * L.src.cur_stride - L.src.cur_base);
	double* const ra = L.tail_ring + (long long) chA * L.src.ring_stride;
	double* const rb = L.tail_ring + (long long) chB * L.src.ring_stride;
	for (long long i0 = s0 + tid; i0 < s1; i0 += (long long) TB * WT)
	{
		double va[TB], vb[TB];
#pragma unroll
		for (int j = 0; j < TB; j++)
		{
			const long long i = i0 + (long long) j * WT;
			va[j] = vb[j] = 0.0;
			if (i < s1)
			{
				va[j] = pa[i];
				if (bvalid) vb[j] = pb[i];
			}
		}
#pragma unroll
		for (int j = 0; j < TB; j++)
		{
			const long long i = i0 + (long long) j * WT;
			if (i < s1)
			{
				ra[i & L.src.ring_mask] = va[j];
				if (bvalid) rb[i & L.src.ring_mask] = vb[j];
			}
		}
	}
}

// Parked outputs of the previous call (ConvxLaunch::park_src) to the caller's rows: outputs [park_j0, park_j0 + park_n) of
// both channels by the WT threads of the launch's first workgroup of the pair, eight per channel in flight per thread,
// issued beside the block's sample loads (one wait for both)
template<int WT>
R8B_HD void cp_park_back(const ConvxLaunch& XM, const DstView& wd, int chA, int chB, bool bvalid, int tid)
{
	constexpr int TB = 8;
	const double* const sa = XM.park_src + (long long) chA * XM.park_stride;
	const double* const sb = XM.park_src + (long long) chB * XM.park_stride;
	const int n = XM.park_n;
	const long long j0 = XM.park_j0;
	for (int i0 = tid; i0 < n; i0 += TB * WT)
	{
		double va[TB], vb[TB];
#pragma unroll
		for (int j = 0; j < TB; j++)
		{
			const int i = i0 + j * WT;
			va[j] = vb[j] = 0.0;
			if (i < n)
			{
				va[j] = sa[i];
				if (bvalid) vb[j] = sb[i];
			}
		}
#pragma unroll
		for (int j = 0; j < TB; j++)
		{
			const int i = i0 + j * WT;
			if (i < n)
			{
				dst_store(wd, chA, j0 + i, va[j]);
				if (bvalid) dst_store(wd, chB, j0 + i, vb[j]);
			}
		}
	}
}

// The same copy shared by the workgroups of the pair's blocks (ConvxLaunch::park_slices): workgroup bgi of the launch
// takes elements [bgi WT, (bgi + 1) WT), one per thread and channel -- requested at entry behind the block's samples (the
// kernel arguments it needs arrive with all the others), stored when the interpolator starts: no workgroup waits for
// it and no phase reads an argument from memory for it (the destination is the caller's fp64 rows: Engine::launch_fused)
template<int WT, class St>
R8B_HD void cp_park_slice_load(const ConvxLaunch& XM, const DstView& wd, St& st, int bgi, int chA, int chB, int tid)
{
	const int i = bgi * WT + tid;
	st.pk[0] = st.pk[1] = 0.0;
	st.pka = nullptr;
	if (i < XM.park_n)
	{
		st.pk[0] = XM.park_src[(long long) chA * XM.park_stride + i];
		st.pk[1] = XM.park_src[(long long) chB * XM.park_stride + i];
		st.pka = wd.p + ((long long) chA * wd.stride + (XM.park_j0 + i + wd.off));
	}
}
template<class St>
R8B_HD void cp_park_slice_store(const DstView& wd, const St& st, int chA, int chB, bool bvalid)
{
	if (st.pka != nullptr)
	{
		st.pka[0] = st.pk[0];
		if (bvalid) st.pka[(long long) (chB - chA) * wd.stride] = st.pk[1];
	}
}

// The part of the history tail no block of the call holds in registers -- [tail_c1, tail_p1), the input behind the last
// block's window -- shared by the workgroups of the pair's blocks (tail_flags & 8, convp_prepare) like the parked
// outputs: one element per thread and channel requested at entry behind the samples, stored in the workgroup's last
// phase.  (Fetched by the last block alone -- cp_tail_rest -- it made that block 5 000 cycles longer than the others.)
template<int WT, class St>
R8B_HD void cp_tail_slice_load(const ConvLaunch& L, St& st, int bgi, int chA, int chB, int tid)
{
	const long long i = L.tail_c1 + (long long) bgi * WT + tid;
	st.tk[0] = st.tk[1] = 0.0;
	st.tka = nullptr;
	if (i < L.tail_p1)
	{
		st.tk[0] = L.src.cur[(long long) chA * L.src.cur_stride + (i - L.src.cur_base)];
		st.tk[1] = L.src.cur[(long long) chB * L.src.cur_stride + (i - L.src.cur_base)];
		st.tka = L.tail_ring + ((long long) chA * L.src.ring_stride + (i & L.src.ring_mask));
	}
}
template<class St>
R8B_HD void cp_tail_slice_store(const ConvLaunch& L, const St& st, int chA, int chB, bool bvalid)
{
	if (st.tka != nullptr)
	{
		st.tka[0] = st.tk[0];
		if (bvalid) st.tka[(long long) (chB - chA) * L.src.ring_stride] = st.tk[1];
	}
}

// (the first pass's arithmetic: the butterfly over the thread's samples, then the twiddles -- loc: the expanded base set)
template<int LN, int UL>
R8B_HD void cp_first_arith(const ConvpState<LN, UL>& st, const cd* loc, double* vr, double* vi)
{
	constexpr int R = ConvpGeom<LN, UL>::E1;
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
}

// first forward pass, from the registers cp_load() filled
template<int LN, int UL, bool HAF = false>
R8B_HD void cp_first(const ConvLaunch& L, cd* buf, const ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
	// (fetched by the caller ahead of the samples -- ptw_fetch_lean --, completed here)
	constexpr int NBW = R >= 16 ? 6 : (R >= 8 ? 4 : (R >= 4 ? 3 : 1));
	cd loc[NBW];
#pragma unroll
	for (int c = 0; c < NBW; c++) loc[c] = st.tw[c];
	tw_expand<R>(loc);
	double vr[R], vi[R];
	cp_first_arith<LN, UL>(st, loc, vr, vi);
	const SwBase bb = sw_base(buf, fslot<LN, UL, HAF>(lt));
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		cd v;
		v.re = vr[p];
		v.im = vi[p];
		sw_st(bb, fmap_c<LN, UL, HAF>(p * q), v);
	}
}

// forward pass I (1 <= I < NPRE): sub-length N >> (I * EB1), radix E1, one butterfly per thread
template<int LN, int UL, int I>
struct ConvpPre
{
	typedef ConvpGeom<LN, UL> G;
	static constexpr int n = G::N >> (I * G::EB1);
	static R8B_HD void prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int lt)
	{
		if constexpr (!ConvpTwLds<LN, UL>::ON)
			ptw_fetch<G::E1, G::NT, (n / G::E1 < G::NT ? n / G::E1 : G::NT)>(st.tw, L.ptw, I, lt);
	}
	// (ltw: the workgroup's twiddle table in LDS, ConvpTwLds)
	template<bool HAF = false>
	static R8B_HD void run(cd* buf, const ConvpState<LN, UL>& st, int lt, const cd* ltw)
	{
		typedef ConvpTwLds<LN, UL> TL;
		if constexpr (TL::ON)
		{
			cd twr[TL::NBF];
			twl_fetch<TL::NBF, (I == 1 ? TL::JM1 : TL::JM2)>(twr, ltw, I == 1 ? TL::O1 : TL::O2, lt);
			pdif<LN, UL, G::E1, true, HAF>(buf, n, lt, twr);
		}
		else pdif<LN, UL, G::E1, true, HAF>(buf, n, lt, st.tw);
	}
};

// kernel constants of the middle pass, hp[c * NT + t] (a wave reads consecutive 16-byte entries):
//   2x up: (Hs, Hd) of forward position 8 t + c;   1:1: H of backward positions 16 t + 2 c, + 1
//   complex kernel spectrum (CX: minimum phase, or an alignment moved by inherited latency; reference
//   CDSPRealFFT.h:186-274 multiplyBlocks): one complex value per entry -- 1:1: H of backward position 16 t + c;
//   2x up: Hs (c < 8) / Hd (c >= 8) of forward position 8 t + (c & 7); decimating: H of kept position c
template<int LN, int UL, bool CX = false>
R8B_HD void cp_hp_prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int lt)
{
	// (decimating form: H of the thread's kept positions 2c, 2c + 1)
	constexpr int NHP = CX ? (UL < 0 ? ConvpGeom<LN, UL>::E2 : 16) : (UL < 0 ? ConvpGeom<LN, UL>::E2 / 2 : 8);
#pragma unroll
	for (int c = 0; c < NHP; c++)
	{
		st.hp[c] = R8B_TAB_LD_H(L.hp, (c * ConvpGeom<LN, UL>::NT), lt);
	}
}

// middle pass, compute part: results (the backward array's positions 16 t + p after the first
// backward pass) stay in st.vr / st.vi
template<int LN, int UL, bool CX = false, bool HAF = false>
R8B_HD void cp_middle_compute(const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	double zr[G::E1], zi[G::E1];
	const SwBase bbf = sw_base(buf, fslot<LN, UL, HAF>(G::E1 * lt));
#pragma unroll
	for (int c = 0; c < G::E1; c++)
	{
		const cd v = sw_ld(bbf, fmap_c<LN, UL, HAF>(c));
		zr[c] = v.re;
		zi[c] = v.im;
	}
#pragma unroll
	for (int f = 0; f < G::NBF; f++) dif_regs<G::RM>(zr + G::RM * f, zi + G::RM * f);
	if constexpr (CX)
	{
		// complex kernel spectrum: a complex multiplication per backward position
#pragma unroll
		for (int p = 0; p < 16; p++)
		{
			const int src = UL > 0 ? p >> 1 : p;
			const cd h = st.hp[UL > 0 ? (p >> 1) + 8 * (p & 1) : p];
			st.vr[p] = zr[src] * h.re - zi[src] * h.im;
			st.vi[p] = zr[src] * h.im + zi[src] * h.re;
		}
		if constexpr (G::POST)
		{
			if constexpr (UL == 0)
			{
#pragma unroll
				for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
			}
		}
		else if constexpr (UL > 0) DitSt<16, 2>::run(st.vr, st.vi);
		else dit_regs<16>(st.vr, st.vi);
	}
	else if constexpr (UL > 0)
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
		if constexpr (!G::POST) DitSt<16, 2>::run(st.vr, st.vi);
		else static_assert(!G::POST || G::RMB == 2, "8192 points: the folded radix-2 stage is the whole middle part");
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
		if constexpr (!G::POST) dit_regs<16>(st.vr, st.vi);
		else
		{
#pragma unroll
			for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
		}
	}
}

template<int LN, int UL>
R8B_HD void cp_middle_write(cd* buf, const ConvpState<LN, UL>& st, int lt)
{
	const SwBase bb = sw_base(buf, pswz(16 * lt));
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		cd v;
		v.re = st.vr[p];
		v.im = st.vi[p];
		sw_st(bb, p, v);
	}
}

// backward pass with sub-length 256 (radix 16); in place, or -- when it is the last one (N2 = 256: its
// elements lt + 16 p are the thread's elements lt + NT p of the result) -- into st.vr / st.vi
template<int LN, int UL>
R8B_HD void cp_back1(cd* buf, ConvpState<LN, UL>& st, int lt, const cd* ltw)
{
	typedef ConvpTwLds<LN, UL> TL;
	cd twl[6];
	const cd* twr = st.tw;
	if constexpr (TL::ON && ConvpGeom<LN, UL>::B1)
	{
		twl_fetch<6, TL::JM3>(twl, ltw, TL::O3, lt);
		twr = twl;
	}
	(void) ltw;
	if constexpr (!ConvpGeom<LN, UL>::B1) return; // (no such pass below 256 points)
	else if constexpr (ConvpGeom<LN, UL>::R2 == 1) pdit_regs<16, true>(buf, 256, lt, twr, st.vr, st.vi);
	else
	{
		double vr[16], vi[16];
		pdit_regs<16, true>(buf, 256, lt, twr, vr, vi);
		const int e0 = (lt >> 4) * 256 + (lt & 15);
		const SwBase bb = sw_base(buf, pswz(e0));
#pragma unroll
		for (int p = 0; p < 16; p++)
		{
			cd v;
			v.re = vr[p];
			v.im = vi[p];
			sw_st(bb, p * 16, v);
		}
	}
}

// twiddles of the last backward pass: NB2 butterflies per thread, NBASE2 base powers each
template<int LN, int UL>
R8B_HD void cp_back2_prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
#pragma unroll
	for (int m = 0; m < G::NB2; m++) ptw_fetch_lean<G::R2, G::NT>(st.tw + m * G::NBASE2, L.ptw, 4 + m, lt);
}

// last backward pass (sub-length N2, radix R2 = N2 / 256): the thread's elements lt + NT i, i = 0 .. 15,
// = (y_A, y_B) at circular time lt + NT i; butterfly m works on i = m + NB2 p
template<int LN, int UL>
R8B_HD void cp_back2(const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	if constexpr (G::R2 > 1)
	{
#pragma unroll
		for (int m = 0; m < G::NB2; m++) tw_expand<G::R2>(st.tw + m * G::NBASE2);
	}
	if constexpr (G::R2 == 16) pdit_regs<16, true>(buf, G::N2, lt, st.tw, st.vr, st.vi);
	else if constexpr (G::R2 > 1)
	{
		constexpr int R = G::R2, NB = G::NB2;
		const SwBase bb = sw_base(buf, pswz(lt));
#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			const cd v = sw_ld(bb, G::NT * i);
			st.vr[i] = v.re;
			st.vi[i] = v.im;
		}
#pragma unroll
		for (int m = 0; m < NB; m++)
		{
			double ar[R], ai[R];
			ar[0] = st.vr[m];
			ai[0] = st.vi[m];
#pragma unroll
			for (int p = 1; p < R; p++)
			{
				const cd w = tw_get(st.tw + m * G::NBASE2, bitrev_c<R>(p));
				const double xr = st.vr[m + NB * p], xi = st.vi[m + NB * p];
				ar[p] = xr * w.re + xi * w.im;
				ai[p] = xi * w.re - xr * w.im;
			}
			dit_regs<R>(ar, ai);
#pragma unroll
			for (int p = 0; p < R; p++)
			{
				st.vr[m + NB * p] = ar[p];
				st.vi[m + NB * p] = ai[p];
			}
		}
	}
}

// ---- half-array form (round 6; MODE 21 = mode 0 of the 2048 -> 4096-point 2x up-sampling geometry; engine option "half") ----
// Why.  The convolver-only kernel k_convp<11, 1, 0, 24> needs 115 registers -- four waves per SIMD -- but its 64 KB array
// lets only two workgroups share a CU; with the array truncated so that four fit (wrong results, same instruction stream)
// it measured 20 % faster (profiles/r06_experiments.txt item 2), and the eight-elements-per-thread form that gets the four
// waves pays for them with three more passes through LDS (item 3).  This form gets them WITHOUT another pass: the
// array is N2 DOUBLES (32 KB), not N2 complex values.
//   * The forward transform has N = N2 / 2 complex points: it fits as it is (identity slot map, fslot<.., HAF>).
//   * The two exchanges of the backward side -- middle pass -> radix-16 pass of sub-length 256 (inside 16 lanes of a
//     wave), that pass -> last pass (across the workgroup) -- move the REAL parts of the 4096 elements through the
//     array, then the IMAGINARY parts: the same bytes through LDS in twice as many 8-byte accesses (the LDS serves
//     ds_read_b64 / ds_write_b64 at the rate of the 16-byte forms), the butterflies between them unchanged.
// The exchange inside a wave costs nothing but program order (LDS serves a wave's accesses in issue order); the one
// across the workgroup takes three barriers instead of one (real parts written | read | imaginary parts written | read).
// Arithmetic, constants and their order are those of mode 0: results are BITWISE those of k_convp<11, 1, 0, 24> under host
// emulation (tests); on the device the compiler contracts multiply-adds differently in the two kernels -- they agree to
// rounding, RMS 4e-17 --, so an object stays with one form (Engine::half_worth decides per object, never per call).
// Slots: element e of the array of doubles at 8-byte slot dswz(e) = e ^ (bits 4-7 into bits 0-3) ^ (bit 8 into bit 4) --
// every access pattern below meets 16 different slots mod 16 in each 16 consecutive lanes (ds_write_b64) and 32
// different slots mod 32 in each 32 (ds_read_b64); linear over XOR like pswz(): one address per thread and pass, XOR
// constants in address bits 3-7 plus the instruction's immediate offset.
// (MODE 23 = mode 4 -- the whole-step interpolator fused in, two phases per thread -- in this form: the array is what
// the interpolator's run of (A, B) pairs needs, kHaFusedElems complex values instead of 4096 -- three workgroups per
// CU --, its first 32 KB carry the transforms as in mode 21; launch bound 256 x 3 = 168 registers: the rows of the
// thread's phase pair are fetched when the last pass's results have gone to the run, not beside its butterflies)
// (MODE 25: mode 5 -- adjacent windows up to three samples apart, In > Out -- likewise)
// (MODE 22: mode 3 -- the 3x strided store behind the block -- in this form)
// (MODES 27 / 28: modes 0 / 3 of the 4096 -> 2048-point DECIMATING geometry -- there it is the FORWARD transform that has
// the 4096 points: its two exchanges go by parts through 4096 doubles, the backward side's 2048 complex values fit as
// they are; three workgroups per CU)
// (MODES 29 / 30 = 23 / 25, 31 / 32 = 21 / 22 with a COMPLEX kernel spectrum -- minimum-phase chains: modes 16 / 17 / 6 / 7)
// (MODE 33 = mode 5 of the 4096 -> 4096-point 1:1 geometry -- BASELINE's cfg3 --: BOTH transforms have the 4096 points there,
// all four exchanges go by parts -- two of them across the workgroup --; the array is the interpolator's run as in mode 25)
constexpr bool convp_mode_ha(int m) { return m == 21 || m == 22 || m == 23 || m == 25 || m == 27 || m == 28 || (m >= 29 && m <= 33); }
constexpr bool convp_mode_ha_down(int m) { return m == 27 || m == 28; }
constexpr bool convp_mode_ha_fused(int m) { return m == 23 || m == 25 || m == 29 || m == 30 || m == 33; }
// (what leaves the workgroup at 52 KB with the flag words and the twiddle table: three of 53.1 KB -- 163 008 of a CU's
// 163 840 bytes -- were NOT resident together on MI355X, the allocation is rounded up; BASELINE's cfg2 needs 3051)
static const int kHaFusedElems = 3052;
// (the launch bound's second number is WAVES PER SIMD: four for two workgroups of 512 threads as for four of 256)
#ifndef R8B_HA_DOWN_WAVES
#define R8B_HA_DOWN_WAVES 3 // (development builds: the decimating half-array form's register budget, 3 -> 168, 4 -> 128)
#endif
constexpr int convp_ha_minblocks(int m, int wt)
{
	return wt > 256 ? 4 : (convp_mode_ha_fused(m) ? 3 : (convp_mode_ha_down(m) ? R8B_HA_DOWN_WAVES : 4));
}
// (the 2048 -> 4096-point geometry; and, convolver-only modes, the 4096 -> 8192-point one: 512 threads, a middle pass that
// is the folded radix-2 stage alone and three radix-16 passes behind it -- 64 KB instead of 128, TWO workgroups per CU)
template<int LN, int UL> constexpr bool convp_ha_ok()
{
	typedef ConvpGeom<LN, UL> G;
	return (UL == 1 && G::SUB == 1 && G::NPRE == 3 &&
		((!G::POST && G::B1 && G::R2 == 16 && ConvpTwLds<LN, UL>::ON) || (G::POST && G::NPOST == 3 && G::RMB == 2 && G::E2 == 16))) ||
		(UL == -1 && LN == 12 && G::SUB == 1 && G::E1 == 16 && G::NPRE == 2 && G::NBF == 1 && G::NT == 256 && G::NPOST == 3 &&
		!ConvpTwLds<LN, UL>::ON) ||
		(UL == 0 && LN == 12 && G::SUB == 1 && G::E1 == 16 && G::NPRE == 2 && G::NBF == 1 && G::NT == 256 && !G::POST && G::B1 &&
		G::R2 == 16 && ConvpTwLds<LN, UL>::ON);
}
template<int LN, int UL, int MODE = 21> constexpr int convp_ha_array_bytes()
{
	return convp_mode_ha_fused(MODE) ? kHaFusedElems * 16 : (UL < 0 ? ConvpGeom<LN, UL>::N * 8 : ConvpGeom<LN, UL>::N2 * 8);
}
template<int LN, int UL, int MODE = 21> constexpr int convp_ha_lds_bytes()
{
	return convp_ha_array_bytes<LN, UL, MODE>() + kConvpFlagBytes + (ConvpTwLds<LN, UL>::ON ? ConvpTwLds<LN, UL>::NE * 16 : 0);
}
// the array of kernel mode MODE (what lies behind it -- flag words, twiddle table -- starts there)
template<int LN, int UL, int MODE> constexpr int convp_mode_array_bytes()
{
	return convp_mode_ha(MODE) ? convp_ha_array_bytes<LN, UL, MODE>() : convp_array_bytes<LN, UL>();
}
// does the interpolator's run of a fused launch fit the half-array form's array?  (host: the launcher's choice)
// (the full-array kernels allow in_step + 48 slots past the run for the windows of a block's last, partly masked output
// group -- Engine::use_pair_two; here a lane whose outputs are both masked reads nothing -- cp_whole2_compute SKIPM --, and
// a stored output's window -- 24 taps inside the run, one or three padded ones behind it -- ends inside the run's zero
// extension, kConvxRunPad slots, which cp_final_store writes: that is all the array has to hold)
inline bool convp_ha_fused_fits(int run_off, int in_len, int in_step)
{
	(void) in_step;
	return run_off + in_len + kConvxRunPad <= kHaFusedElems;
}
R8B_HD constexpr int dswz(int e) { return e ^ ((e >> 4) & 15) ^ (((e >> 8) & 1) << 4); }
R8B_HD constexpr int dsw_xc(int m) { return dswz(m) & 31; }
R8B_HD constexpr int dsw_hi(int m) { return m & ~31; }
#if defined(R8B_LDS_ABS) && defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double lds_d_t;
R8B_HD SwBase dw_base(const cd* buf, int slot0)
{
	SwBase b;
	b.a = (unsigned) (size_t) (const lds_cd_t*) buf + ((unsigned) slot0 << 3);
	return b;
}
// (volatile: every read ONE ds_read_b64 -- 256 B per clock and CU; left alone the compiler pairs the reads of a pass into
// ds_read2_b64 / ds_read2st64_b64, which move half as many bytes per LDS cycle: MI355X_MICROARCH.md, LDS table)
R8B_HD double dw_ld(SwBase b, int m)
{
	return *(const volatile lds_d_t*) (size_t) ((b.a ^ (unsigned) (dsw_xc(m) << 3)) + (unsigned) (dsw_hi(m) << 3));
}
R8B_HD void dw_st(SwBase b, int m, double v)
{
	*(lds_d_t*) (size_t) ((b.a ^ (unsigned) (dsw_xc(m) << 3)) + (unsigned) (dsw_hi(m) << 3)) = v;
}
#else
R8B_HD SwBase dw_base(const cd* buf, int slot0)
{
	SwBase b;
	b.p = reinterpret_cast<char*>(const_cast<cd*>(buf));
	b.bb = slot0 << 3;
	return b;
}
R8B_HD double dw_ld(SwBase b, int m)
{
	return *reinterpret_cast<const double*>(b.p + ((b.bb ^ (dsw_xc(m) << 3)) + (dsw_hi(m) << 3)));
}
R8B_HD void dw_st(SwBase b, int m, double v)
{
	*reinterpret_cast<double*>(b.p + ((b.bb ^ (dsw_xc(m) << 3)) + (dsw_hi(m) << 3))) = v;
}
#endif
// one part (real or imaginary) of the middle pass's results: the thread's backward positions 16 lt + p
R8B_HD void cp_ha_st_mid(cd* buf, const double* v, int lt)
{
	const SwBase bb = dw_base(buf, dswz(16 * lt));
#pragma unroll
	for (int p = 0; p < 16; p++) dw_st(bb, p, v[p]);
}
// ... of the radix-16 pass of sub-length 256: elements e0 + 16 p, e0 = (lt / 16) 256 + lt mod 16 (read, and written back)
R8B_HD void cp_ha_ld_b1(const cd* buf, double* v, int lt)
{
	const SwBase bb = dw_base(buf, dswz((lt >> 4) * 256 + (lt & 15)));
#pragma unroll
	for (int p = 0; p < 16; p++) v[p] = dw_ld(bb, 16 * p);
}
R8B_HD void cp_ha_st_b1(cd* buf, const double* v, int lt)
{
	const SwBase bb = dw_base(buf, dswz((lt >> 4) * 256 + (lt & 15)));
#pragma unroll
	for (int p = 0; p < 16; p++) dw_st(bb, 16 * p, v[p]);
}
// the general pattern -- a radix-16 pass of sub-length NSUB: butterfly lt works on elements e0 + (NSUB / 16) p,
// e0 = (lt / q) NSUB + lt mod q, q = NSUB / 16 (NSUB = 16: the thread's sixteen consecutive positions)
template<int NSUB>
R8B_HD void cp_ha_ld(const cd* buf, double* v, int lt)
{
	constexpr int q = NSUB / 16;
	const SwBase bb = dw_base(buf, dswz((lt / q) * NSUB + (lt & (q - 1))));
#pragma unroll
	for (int p = 0; p < 16; p++) v[p] = dw_ld(bb, q * p);
}
template<int NSUB>
R8B_HD void cp_ha_st(cd* buf, const double* v, int lt)
{
	constexpr int q = NSUB / 16;
	const SwBase bb = dw_base(buf, dswz((lt / q) * NSUB + (lt & (q - 1))));
#pragma unroll
	for (int p = 0; p < 16; p++) dw_st(bb, q * p, v[p]);
}
// ... of the last pass: elements lt + NT i
template<int NT>
R8B_HD void cp_ha_ld_b2(const cd* buf, double* v, int lt)
{
	static_assert(NT == 256, "half-array form: 4096-point backward transforms");
	const SwBase bb = dw_base(buf, dswz(lt));
#pragma unroll
	for (int i = 0; i < 16; i++) v[i] = dw_ld(bb, NT * i);
}

// ---- split 2x up-sampling form (modes 8 / 9, 12 / 13 with a complex spectrum; geometry <13, 0>: 8192 -> 16384-point blocks)
// A 2x up-sampling block whose backward transform would be 16384 points -- 256 KB as a pair, more than a CU's LDS --
// runs on the 8192-point 1:1 geometry: forward transform of the N = 8192 input samples as there, then the backward
// transform as TWO N-point transforms, one after the other in the same array: the even outputs y[2m] =
// IDFT_N(Z (H[k] + H[k+N]))[m] and the odd ones y[2m+1] = IDFT_N(Z (H[k] - H[k+N]) th^k)[m], th = e^{+2 pi i / 2N}
// (the zero-stuffed spectrum is the forward spectrum repeated, reference CDSPBlockConvolver.h:606-629; the first
// radix-2 stage of the 2N-point transform separates the output parities).  Thread lt owns forward positions 16 lt + c
// (bit-reversed: bin k = bitrev4(c) 512 + bitrev9(lt)), so th^k = conj(tw[bitrev9(lt)]) e^{+2 pi i bitrev4(c) / 32}: one
// entry of the exp(-2 pi i e / 16384) table per thread and the 32nd roots of unity as constants.  hp[c * NT + lt] =
// (H[k] + H[k+N], H[k] - H[k+N]) of position 16 lt + c (Engine: pair_constants_split).  Each half then takes the
// backward passes of the 1:1 geometry (ConvpPost).  Replaces the one-channel kernel k_convx for these blocks (filters
// with a transition band of about 1 % and below): two channels per workgroup instead of one.
template<int LN, int UL>
R8B_HD void cp_sp_hp_prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int lt)
{
#pragma unroll
	for (int c = 0; c < 16; c++) st.hp[c] = L.hp[c * ConvpGeom<LN, UL>::NT + lt];
}
// middle: the last forward butterflies, the two half spectra; the even one goes on (first backward butterflies in
// st.vr / st.vi), the odd one waits in st.er / st.ei
// (CX: complex kernel spectrum -- minimum phase, or an alignment moved by inherited latency: hp[c * NT + lt] = H[k] +
// H[k+N], hp[(16 + c) * NT + lt] = (H[k] - H[k+N]) th^k, both complex, the twiddle folded in by the host --
// pair_constants_split_complex; the second sixteen are fetched into the first sixteen's registers once those are spent)
template<int LN, int UL, bool CX = false>
R8B_HD void cp_sp_middle(const ConvLaunch& L, const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	static_assert(UL == 0 && G::E1 == 16 && G::POST && G::NT == 512, "split 2x up-sampling form: the 8192-point 1:1 geometry");
	double zr[16], zi[16];
	const SwBase bbf = sw_base(buf, fslot<LN, UL>(16 * lt));
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const cd v = sw_ld(bbf, fmap_c<LN, UL>(c));
		zr[c] = v.re;
		zi[c] = v.im;
	}
#pragma unroll
	for (int f = 0; f < G::NBF; f++) dif_regs<G::RM>(zr + G::RM * f, zi + G::RM * f);
	if constexpr (CX)
	{
#pragma unroll
		for (int c = 0; c < 16; c++)
		{
			const cd h = st.hp[c];
			st.vr[c] = zr[c] * h.re - zi[c] * h.im;
			st.vi[c] = zr[c] * h.im + zi[c] * h.re;
		}
#pragma unroll
		for (int c = 0; c < 16; c++) st.hp[c] = L.hp[(16 + c) * G::NT + lt];
#pragma unroll
		for (int c = 0; c < 16; c++)
		{
			const cd h = st.hp[c];
			st.er[c] = zr[c] * h.re - zi[c] * h.im;
			st.ei[c] = zr[c] * h.im + zi[c] * h.re;
		}
#pragma unroll
		for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
		return;
	}
	// th^(bitrev9(lt)): conj of the table's entry (the table has 2N = 16384 entries: tw_len / 16384 = 1)
	unsigned r = (unsigned) lt;
	r = ((r & 0xaaaau) >> 1) | ((r & 0x5555u) << 1);
	r = ((r & 0xccccu) >> 2) | ((r & 0x3333u) << 2);
	r = ((r & 0xf0f0u) >> 4) | ((r & 0x0f0fu) << 4);
	r = ((r & 0xff00u) >> 8) | ((r & 0x00ffu) << 8);
	r >>= 7; // (a 16-bit reversal shifted down: the reversal of the 9 bits of lt)
	const cd wt = L.tw[(L.tw_len >> 14) * (int) r];
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		// e^{+2 pi i j / 32}, j = bitrev4(c)
		constexpr double kC[9] = { 1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
			0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785, 0.0 };
		const int j = ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3);
		// cos(2 pi j / 32), sin(2 pi j / 32) for j = 0 .. 15 from the first-octant values
		const double cr = j <= 8 ? kC[j] : -kC[16 - j];
		const double ci = j <= 8 ? kC[8 - j] : kC[j - 8];
		const double hs = st.hp[c].re, hd = st.hp[c].im;
		st.vr[c] = zr[c] * hs;
		st.vi[c] = zi[c] * hs;
		// conj(wt) * (cr + i ci)
		const double tr = wt.re * cr + wt.im * ci, ti = wt.re * ci - wt.im * cr;
		const double dr = zr[c] * hd, di = zi[c] * hd;
		st.er[c] = dr * tr - di * ti;
		st.ei[c] = dr * ti + di * tr;
	}
#pragma unroll
	for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
}
// between the halves: the even half's outputs (st.vr / st.vi after its last pass) change places with the odd half's
// spectrum, whose first backward butterflies follow
template<int LN, int UL>
R8B_HD void cp_sp_swap(ConvpState<LN, UL>& st)
{
	typedef ConvpGeom<LN, UL> G;
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const double a = st.vr[c], b = st.vi[c];
		st.vr[c] = st.er[c];
		st.vi[c] = st.ei[c];
		st.er[c] = a;
		st.ei[c] = b;
	}
#pragma unroll
	for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
}

// ---- one-channel form, the spectrum (modes 10 / 11; 14 / 15: cp_solo_mid_b<.., CX>) ---------------------------------
// With z[n] = x[2n] + i x[2n+1], Z = DFT_N(z), N = 8192: the spectrum of the even samples is E = (Z[k] + conj Z[N-k]) / 2,
// that of the odd ones O = (Z[k] - conj Z[N-k]) / 2i, the block's 2N-point spectrum X[k] = E + w^k O, X[k+N] = E - w^k O,
// w = e^{-i pi / N}.  Multiplying by the (real, symmetric) kernel spectrum H and packing the result the same way --
// y[2m] + i y[2m+1] = IDFT_N(Z'), Z' = (Y[k] + Y[k+N]) + i w^-k (Y[k] - Y[k+N]) for the unnormalised 2N-point transform --
// collapses into
//     Z'[k] = a[k] Z[k] + i b[k] conj(Z[N-k]),
//     a = (H[k] + H[k+N]) - (H[k] - H[k+N]) sin(pi k / N),   b = (H[k] - H[k+N]) cos(pi k / N):
// two REAL constants per bin (hp[c * NT + lt] = (a, b) of forward position 16 lt + c; Engine: pair_constants_solo) and
// the partner bin N - k.  Thread lt owns forward positions 16 lt + c, bins k = bitrev4(c) 512 + bitrev9(lt); bin N - k
// sits at position 16 lt' + 15 - c, lt' = bitrev9(512 - bitrev9(lt)) -- thread 0 is its own partner with the positions
// permuted --, in another wave: the spectrum goes through the array once (cp_solo_mid_a writes it back, a workgroup
// barrier, cp_solo_mid_b reads the partners' values; another barrier before the backward side overwrites them).
// The rest -- passes, rotation, stores -- is the pair kernel's, with the two "channels" of an element being the even
// and the odd sample.  Replaces the one-channel kernel k_convx (12 barrier phases, 3.4x the time per block).
template<int LN, int UL>
R8B_HD void cp_solo_mid_a(cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	static_assert((UL == 0 || UL == -1 || UL == -2) && G::E1 == 16 && G::POST && G::NT == 512,
		"one-channel form: the 8192-point 1:1 geometry and its decimating ones");
	const SwBase bbf = sw_base(buf, fslot<LN, UL>(16 * lt));
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const cd v = sw_ld(bbf, fmap_c<LN, UL>(c));
		st.vr[c] = v.re;
		st.vi[c] = v.im;
	}
#pragma unroll
	for (int f = 0; f < G::NBF; f++) dif_regs<G::RM>(st.vr + G::RM * f, st.vi + G::RM * f);
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		cd v;
		v.re = st.vr[c];
		v.im = st.vi[c];
		sw_st(bbf, fmap_c<LN, UL>(c), v);
	}
}
// (CX: complex kernel spectrum: Z'[k] = A[k] Z[k] + B[k] conj(Z[N-k]) with COMPLEX A = (H[k] + H[k+N]) - (H[k] - H[k+N])
// sin(pi k / N), B = i (H[k] - H[k+N]) cos(pi k / N); hp[c * NT + lt] = A, hp[(16 + c) * NT + lt] = B --
// pair_constants_solo_complex; B is fetched into A's registers once A is spent)
template<int LN, int UL, bool CX = false>
R8B_HD void cp_solo_mid_b(const ConvLaunch& L, const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	const int lp = bitrev_n((512 - bitrev_n(lt, 9)) & 511, 9);
	const SwBase bp = sw_base(const_cast<cd*>(buf), fslot<LN, UL>(16 * lp));
	double qr[16], qi[16];
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const cd v = sw_ld(bp, fmap_c<LN, UL>(15 - c));
		qr[c] = v.re;
		qi[c] = v.im;
	}
	if (lt == 0)
	{
		// (bins b 512: partner (16 - b) 512, among the thread's own values)
#pragma unroll
		for (int c = 0; c < 16; c++)
		{
			constexpr int kRev[16] = { 0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15 };
			const int cp = kRev[(16 - kRev[c]) & 15];
			qr[c] = st.vr[cp];
			qi[c] = st.vi[cp];
		}
	}
	if constexpr (CX)
	{
#pragma unroll
		for (int c = 0; c < 16; c++)
		{
			const cd a = st.hp[c];
			const double zr = st.vr[c], zi = st.vi[c];
			st.vr[c] = a.re * zr - a.im * zi;
			st.vi[c] = a.re * zi + a.im * zr;
		}
#pragma unroll
		for (int c = 0; c < 16; c++) st.hp[c] = L.hp[(16 + c) * G::NT + lt];
#pragma unroll
		for (int c = 0; c < 16; c++)
		{
			// + B conj(Q)
			const cd b = st.hp[c];
			st.vr[c] += b.re * qr[c] + b.im * qi[c];
			st.vi[c] += b.im * qr[c] - b.re * qi[c];
		}
	}
	else
	{
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const double a = st.hp[c].re, b = st.hp[c].im;
		const double zr = st.vr[c], zi = st.vi[c];
		// a Z + i b conj(Q)
		st.vr[c] = a * zr + b * qi[c];
		st.vi[c] = a * zi + b * qr[c];
	}
	}
#pragma unroll
	for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
}

// ... decimating by 2 in the spectrum (geometry <13, -1>: 16384 -> 8192 real points; reference CDSPBlockConvolver.h:329-344):
// the output block's spectrum is Y[k] = H[k] X[k] for k < N2 = 4096 and the reference's real fix-up value at the new
// Nyquist bin, Y[N2] = H[N2] (Re X[N2] + Im X[N2]); packed for the N2-point backward transform,
//     Z'[k] = Y[k] (1 + g) + conj(Y[j]) (1 - g),   j = N2 - k,   g = i e^{+2 pi i k / 8192}.
// X[k] takes Z[k] and Z[N - k], X[j] takes Z[j] and Z[N - j] = Z[N2 + k]: of the thread's sixteen forward positions the
// even ones c are the kept bins k (backward position 8 lt + c / 2), c + 1 holds Z[N2 + k], and the partner thread's
// positions 15 - c and 14 - c hold Z[N - k] and Z[N2 - k].  hp[c * NT + lt], c even: (H[k], H[j]); c + 1: (cos, sin) of
// 2 pi k / 16384 (Engine: pair_constants_solo_down).
// (CX: complex kernel spectrum: hp[c * NT + lt] = H[k] for even c, hp[(16 + c / 2) * NT + lt] = H[j], both complex; the
// new Nyquist bin is Re(H[N2] X[N2]) then, reference CDSPBlockConvolver.h:340 behind multiplyBlocks --
// pair_constants_solo_down_complex)
template<int LN, int UL, bool CX = false>
R8B_HD void cp_solo_mid_b_down(const ConvLaunch& L, const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	// (D = 2 or 4: the kept bins are the thread's positions c = D e, e < 16 / D; Z[N - k] at the partner's 15 - c, Z[N2 - k]
	// at its 16 - D - c, Z[N2 + k] at the thread's own c + D - 1; w^j = e^{-i pi / D} conj(w^k), g = i (cs + i sn)^D)
	constexpr int D = 1 << G::DL, NE = 16 / D;
	static_assert((UL == -1 || (UL == -2 && !CX)) && G::E1 == 16 && G::E2 == NE && G::POST && G::NT == 512,
		"one-channel form, decimating: the 8192 -> 4096 / 2048-point geometries");
	const int lp = bitrev_n((512 - bitrev_n(lt, 9)) & 511, 9);
	const SwBase bp = sw_base(const_cast<cd*>(buf), pswz(16 * lp));
	cd h2[8];
	if constexpr (CX)
	{
#pragma unroll
		for (int e = 0; e < 8; e++) h2[e] = L.hp[(16 + e) * G::NT + lt];
	}
	double qr[16], qi[16];
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const cd v = sw_ld(bp, c);
		qr[c] = v.re;
		qi[c] = v.im;
	}
	double yr[NE], yi[NE];
#pragma unroll
	for (int e = 0; e < NE; e++)
	{
		constexpr int kRev[16] = { 0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15 };
		const int c = D * e;
		const double pr = st.vr[c], pi = st.vi[c], sr = st.vr[c + D - 1], si = st.vi[c + D - 1];
		double ar = qr[15 - c], ai = qi[15 - c], rr = qr[16 - D - c], ri = qi[16 - D - c]; // Z[N - k], Z[N2 - k]
		if (lt == 0)
		{
			// (bins b 512: the partners are among the thread's own values)
			ar = st.vr[kRev[(16 - kRev[c]) & 15]];
			ai = st.vi[kRev[(16 - kRev[c]) & 15]];
			rr = st.vr[kRev[(16 / D - kRev[c]) & 15]];
			ri = st.vi[kRev[(16 / D - kRev[c]) & 15]];
		}
		const double cs = st.hp[c + 1].re, sn = st.hp[c + 1].im;
		// X[k] = E + w^k O: E = (P + conj A) / 2, O = -i (P - conj A) / 2, w^k = cs - i sn
		const double er = 0.5 * (pr + ar), ei = 0.5 * (pi - ai), o_r = 0.5 * (pi + ai), oi = -0.5 * (pr - ar);
		const double xr = er + cs * o_r + sn * oi, xi = ei + cs * oi - sn * o_r;
		// X[j]: E = (R + conj S) / 2, O = -i (R - conj S) / 2, w^j = e^{-i pi / D} (cs + i sn)  (D = 2: sn - i cs)
		const double fr = 0.5 * (rr + sr), fi = 0.5 * (ri - si), p_r = 0.5 * (ri + si), p_i = -0.5 * (rr - sr);
		constexpr double kCD = D == 2 ? 0.0 : 0.70710678118654752440, kSD = D == 2 ? 1.0 : 0.70710678118654752440;
		const double wjr = D == 2 ? sn : kCD * cs + kSD * sn, wji = D == 2 ? -cs : kCD * sn - kSD * cs;
		const double ur = fr + wjr * p_r - wji * p_i, ui = fi + wjr * p_i + wji * p_r;
		double ykr, yki, yjr, yji;
		if constexpr (CX)
		{
			const cd hk = st.hp[c], hj = h2[e];
			ykr = hk.re * xr - hk.im * xi;
			yki = hk.re * xi + hk.im * xr;
			yjr = hj.re * ur - hj.im * ui;
			yji = hj.re * ui + hj.im * ur;
			// the new Nyquist bin: Re(H X), the reference's fix-up value behind multiplyBlocks
			if (lt == 0 && e == 0) yji = 0.0;
		}
		else
		{
			const double hk = st.hp[c].re, hj = st.hp[c].im;
			ykr = hk * xr;
			yki = hk * xi;
			yjr = hj * ur;
			yji = hj * ui;
			if (lt == 0 && e == 0)
			{
				// the new Nyquist bin: the reference's real fix-up value
				yjr = hj * (ur + ui);
				yji = 0.0;
			}
		}
		// g = i (cs + i sn)^D; Z' = Y[k] (1 + g) + conj(Y[j]) (1 - g)
		const double q2r = cs * cs - sn * sn, q2i = 2.0 * cs * sn; // (cs + i sn)^2
		const double gr = D == 2 ? -q2i : -2.0 * q2r * q2i, gi = D == 2 ? q2r : q2r * q2r - q2i * q2i;
		const double m_r = 1.0 + gr, n_r = 1.0 - gr;
		yr[e] = ykr * m_r - yki * gi + yjr * n_r - yji * gi;
		yi[e] = ykr * gi + yki * m_r - yjr * gi - yji * n_r;
	}
#pragma unroll
	for (int e = 0; e < NE; e++)
	{
		st.vr[e] = yr[e];
		st.vi[e] = yi[e];
	}
#pragma unroll
	for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
}

// ---- decimating form (UL < 0) -------------------------------------------------------------------------
// The reference decimates by 2^d in the spectrum (CDSPBlockConvolver.h:329-344): the backward transform has
// N2 = N / D points and keeps the bins below the new Nyquist frequency, k < N2/2 and k > N - N2/2 -- in the
// bit-reversed order the forward transform leaves them in, the positions p with p mod 2D = 0 or 2D - 1,
// which land at backward position p >> d.  A thread therefore keeps E2 = 16 / D of its 16 consecutive
// forward positions, as its E2 consecutive backward positions.  The new Nyquist bin (backward position 1,
// thread 0) is the reference's fix-up: per channel the REAL value H[m] (Re X[m] + Im X[m]), m = N2 / 2,
// X = that channel's spectrum, from the forward bins m and N - m (positions D and 2D - 1).
// hp[c * NT + t] = H of the thread's kept positions 2c, 2c + 1.
template<int LN, int UL, bool CX = false>
R8B_HD void cp_middle_down_arith(ConvpState<LN, UL>& st, double* zr, double* zi, int lt);

template<int LN, int UL, bool CX = false>
R8B_HD void cp_middle_compute_down(const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	double zr[16], zi[16];
	const SwBase bbf = sw_base(buf, pswz(16 * lt));
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const cd v = sw_ld(bbf, c);
		zr[c] = v.re;
		zi[c] = v.im;
	}
	cp_middle_down_arith<LN, UL, CX>(st, zr, zi, lt);
}

// (the decimating middle pass on the thread's sixteen consecutive forward values: the last forward butterflies, the kept
// bins times the kernel, the new Nyquist bin's fix-up, the first backward butterflies)
template<int LN, int UL, bool CX>
R8B_HD void cp_middle_down_arith(ConvpState<LN, UL>& st, double* zr, double* zi, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int D = 1 << G::DL;
#pragma unroll
	for (int f = 0; f < G::NBF; f++) dif_regs<G::RM>(zr + G::RM * f, zi + G::RM * f);
	if constexpr (CX)
	{
#pragma unroll
		for (int c = 0; c < G::E2; c++)
		{
			const int src = 2 * D * (c >> 1) + ((c & 1) ? 2 * D - 1 : 0);
			const cd h = st.hp[c];
			st.vr[c] = zr[src] * h.re - zi[src] * h.im;
			st.vi[c] = zr[src] * h.im + zi[src] * h.re;
		}
		if (lt == 0)
		{
			// complex kernel: the new Nyquist bin is Re(H[m] X[m]) per channel (reference
			// CDSPBlockConvolver.h:340 after multiplyBlocks); the table entry of kept position 1 is
			// H[N - m] = conj H[m].  X_A = (P + conj Q) / 2, X_B = (P - conj Q) / 2i
			const double pr = zr[D], pi = zi[D], qr = zr[2 * D - 1], qi = zi[2 * D - 1];
			const double hr = st.hp[1].re, hi = -st.hp[1].im;
			const double ar = 0.5 * (pr + qr), ai = 0.5 * (pi - qi); // X_A
			const double br = 0.5 * (pi + qi), bi = -0.5 * (pr - qr); // X_B
			st.vr[1] = hr * ar - hi * ai;
			st.vi[1] = hr * br - hi * bi;
		}
	}
	else
	{
#pragma unroll
	for (int c = 0; c < G::E2; c++)
	{
		const int src = 2 * D * (c >> 1) + ((c & 1) ? 2 * D - 1 : 0);
		const double h = (c & 1) ? st.hp[c >> 1].im : st.hp[c >> 1].re;
		st.vr[c] = zr[src] * h;
		st.vi[c] = zi[src] * h;
	}
	if (lt == 0)
	{
		// P = Z[m], Q = Z[N - m]: X_A = (P + conj Q) / 2, X_B = (P - conj Q) / 2i
		const double pr = zr[D], pi = zi[D], qr = zr[2 * D - 1], qi = zi[2 * D - 1];
		const double h = st.hp[0].im;
		st.vr[1] = h * (0.5 * ((pr + qr) + (pi - qi)));
		st.vi[1] = h * (0.5 * ((pi + qi) - (pr - qr)));
	}
	}
#pragma unroll
	for (int f = 0; f < G::NBB; f++) dit_regs<G::RMB>(st.vr + G::RMB * f, st.vi + G::RMB * f);
}

template<int LN, int UL, bool HAF = false>
R8B_HD void cp_middle_write_down(cd* buf, const ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	const SwBase bb = sw_base(buf, bslot<LN, UL, HAF>(G::E2 * lt));
#pragma unroll
	for (int p = 0; p < G::E2; p++)
	{
		cd v;
		v.re = st.vr[p];
		v.im = st.vi[p];
		sw_st(bb, bmap_c<LN, UL, HAF>(p), v);
	}
}

// backward pass I (1 <= I <= NPOST) of the decimating form: sub-length RMB E2^I, radix E2, one butterfly
// per thread; twiddles in slot 2 + I.  The last one (sub-length N2: the thread's elements lt + NT p = (y_A,
// y_B) at circular time lt + NT p of the decimated block) keeps its results in st.vr / st.vi.
template<int LN, int UL, int I>
struct ConvpPost
{
	typedef ConvpGeom<LN, UL> G;
	static constexpr int n = G::RMB << (I * G::EB2);
	static R8B_HD void prefetch(const ConvLaunch& L, ConvpState<LN, UL>& st, int lt)
	{
		ptw_fetch<G::E2, G::NT, (n / G::E2 < G::NT ? n / G::E2 : G::NT)>(st.tw, L.ptw, 2 + I, lt);
	}
	template<bool HAF = false>
	static R8B_HD void run(cd* buf, ConvpState<LN, UL>& st, int lt)
	{
		constexpr int R = G::E2, q = n / R;
		const int blk = lt / q, j = lt - blk * q;
		const int e0 = blk * n + j;
		const SwBase bb = sw_base(buf, bslot<LN, UL, HAF>(e0));
		double vr[R], vi[R];
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			const cd v = sw_ld(bb, bmap_c<LN, UL, HAF>(p * q));
			vr[p] = v.re;
			vi[p] = v.im;
		}
#pragma unroll
		for (int p = 1; p < R; p++)
		{
			const cd w = tw_get(st.tw, bitrev_c<R>(p));
			const double tr = vr[p] * w.re + vi[p] * w.im;
			const double ti = vi[p] * w.re - vr[p] * w.im;
			vr[p] = tr;
			vi[p] = ti;
		}
		dit_regs<R>(vr, vi);
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			if constexpr (I == G::NPOST)
			{
				st.vr[p] = vr[p];
				st.vi[p] = vi[p];
			}
			else
			{
				cd v;
				v.re = vr[p];
				v.im = vi[p];
				sw_st(bb, bmap_c<LN, UL, HAF>(p * q), v);
			}
		}
	}
};

// Convolver outputs straight from the registers, lean form: everything that is the same for the whole block is folded
// once on the scalar unit -- the two row pointers, the low 32 bits of (first output + view offset), the block's slice of
// the wanted output range as [ulo, uhi) in the block's own index space -- and an element is left with a 32-bit index,
// one range test and the view's mask (rings are far below 2^32 elements, linear views are indexed from the call's first
// output).  (Through dst_store every element repeats ~12 instructions of 64-bit position and address arithmetic: with
// 16 elements per thread and two channels that was as many integer instructions as the last pass has fp64 ones.)
struct CpStoreView
{
	double* pa;
	double* pb;
	unsigned qoff, m;   // element index = (qoff + i) & m, i = the output's index counted from the block's first one
	unsigned ulo, uhi;  // outputs i in [ulo, uhi) are wanted (0 <= ulo <= uhi < 2^31)
	int fmt;            // PcmFormat of the rows (builds with planar PCM views: encoded by the store)
};
// first: the block's first output position; [a, b): the wanted outputs; nmax: outputs a block can hold (< 2^30)
R8B_HD CpStoreView cp_store_view(const DstView& d, int chA, int chB, long long first, long long a, long long b, int nmax)
{
	CpStoreView v;
	v.pa = d.p + (long long) chA * d.stride;
	v.pb = d.p + (long long) chB * d.stride;
#ifndef R8B_NO_PCM_FUSE
	if (d.fmt != kPcmF64)
	{
		// (row starts in BYTES: the view's stride counts samples of its own format)
		const long long bs = pcm_bytes(d.fmt);
		v.pa = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(d.p) + (long long) chA * d.stride * bs);
		v.pb = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(d.p) + (long long) chB * d.stride * bs);
	}
#endif
	v.qoff = (unsigned) (unsigned long long) (first + d.off);
	v.m = (unsigned) (unsigned long long) d.mask;
	v.fmt = d.fmt;
	const long long lo = a - first, hi = b - first;
	v.ulo = (unsigned) (lo < 0 ? 0 : (lo > nmax ? nmax : lo));
	v.uhi = (unsigned) (hi < 0 ? 0 : (hi > nmax ? nmax : hi));
	if (v.uhi < v.ulo) v.uhi = v.ulo;
	return v;
}
R8B_HD void cp_store1(const CpStoreView& v, unsigned i, double ya, double yb, bool bvalid)
{
	if (i - v.ulo < v.uhi - v.ulo)
	{
		const unsigned e = (v.qoff + i) & v.m;
#ifndef R8B_NO_PCM_FUSE
		if (v.fmt != kPcmF64)
		{
			// (planar PCM rows: v.pa / v.pb were formed with the row stride in SAMPLES, element sizes differ)
			const int bs = pcm_bytes(v.fmt);
			pcm_encode(reinterpret_cast<unsigned char*>(v.pa) + (size_t) e * bs, v.fmt, ya);
			if (bvalid) pcm_encode(reinterpret_cast<unsigned char*>(v.pb) + (size_t) e * bs, v.fmt, yb);
			return;
		}
#endif
		v.pa[e] = ya;
		if (bvalid) v.pb[e] = yb;
	}
}

// K7 of the decimating form, from the registers: output q sits at virtual time q * down; the block's first
// one is (block start) / down - floor(fl2 / down), in_len and the block starts being multiples of down
// (reference CDSPBlockConvolver.h:150-165; cf. cx_store_conv)
// (pd / pend: outputs [L.b, pend) of the call's last block belong to the next call and go to the park view pd --
// ConvxLaunch::park_dst; pend = L.b for every other block)
template<int LN, int UL>
R8B_HD void cp_store_conv_down(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int chA,
	int chB, bool bvalid, int lt, const DstView& pd, long long pend)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int mask = G::N2 - 1;
	const int fl2 = L.fl2 >> G::DL, n = L.in_len >> G::DL;
	const long long q0 = ((k * (long long) L.blk_stride + L.blk_offset) >> G::DL) - fl2;
	const CpStoreView v = cp_store_view(L.dst, chA, chB, q0, L.a, L.b, n);
#pragma unroll
	for (int p = 0; p < G::E2; p++)
		cp_store1(v, (unsigned) ((lt + G::NT * p + fl2) & mask), st.vr[p], st.vi[p], bvalid);
	if (pend > L.b)
	{
		// (the call's last block: what lies behind the call's range goes to the park view)
		const CpStoreView w = cp_store_view(pd, chA, chB, q0, L.b, pend, n);
#pragma unroll
		for (int p = 0; p < G::E2; p++)
			cp_store1(w, (unsigned) ((lt + G::NT * p + fl2) & mask), st.vr[p], st.vi[p], bvalid);
	}
}

// ... of the one-channel form: the thread's element p is (y[2 i], y[2 i + 1]) of the decimated block, i = lt + NT p
template<int LN, int UL>
R8B_HD void cp_solo_store_down(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int ch, int lt,
	const DstView& pd, long long pend)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int mask = 2 * G::N2 - 1;
	const int fl2 = L.fl2 >> G::DL, n = L.in_len >> G::DL;
	const long long q0 = ((k * (long long) L.blk_stride + L.blk_offset) >> G::DL) - fl2;
	auto run = [&](const CpStoreView& v)
	{
		const bool al = ((v.qoff + (unsigned) fl2) & 1u) == 0 && v.fmt == kPcmF64 && (v.m & 1u) != 0 &&
			(reinterpret_cast<unsigned long long>(v.pa) & 15ull) == 0;
#pragma unroll
		for (int p = 0; p < G::E2; p++)
		{
			const unsigned iE = (unsigned) ((2 * (lt + G::NT * p) + fl2) & mask), iO = (unsigned) ((iE + 1u) & mask);
			if (al && iO == iE + 1 && iE - v.ulo < v.uhi - v.ulo && iO < v.uhi)
			{
				cd va;
				va.re = st.vr[p];
				va.im = st.vi[p];
				R8B_OUT_STORE16(v.pa + ((v.qoff + iE) & v.m), va);
			}
			else
			{
				cp_store1(v, iE, st.vr[p], 0.0, false);
				cp_store1(v, iO, st.vi[p], 0.0, false);
			}
		}
	};
	run(cp_store_view(L.dst, ch, ch, q0, L.a, L.b, n));
	if (pend > L.b) run(cp_store_view(pd, ch, ch, q0, L.b, pend, n));
}

// MODE 1: the block's valid outputs as one linear run of (A, B) pairs, y[u] = outputs at time t0 + u
// (HAF: the half-array form -- the array ends behind the run's zero extension, nothing is stored past in_len)
template<int LN, int UL, bool HAF = false>
R8B_HD void cp_final_store(const ConvLaunch& L, cd* ybase, cd* y, const ConvpState<LN, UL>& st, long long k, int lt)
{
	// (ybase: the pair's array; y: the run inside it, ybase + run_off)
	typedef ConvpGeom<LN, UL> G;
	constexpr int mask = G::N2 - 1;
	const long long t0 = cx_block_t0(L, k) - L.t_zero;
	// the interpolator's stream starts at this stage's output t_zero (0 but in chains with a fractional latency): earlier
	// outputs do not exist for it (reference CDSPFracInterpolator.h:834-859)
	const int nzero = t0 >= 0 ? 0 : (-t0 > L.in_len ? L.in_len : (int) -t0);
	const int in_len = L.in_len, u0 = (lt + L.fl2r) & mask;
	if (t0 <= 0)
	{
		// (the block that holds the stream's start: the windows of the first outputs begin in front of the run -- by the
		// left half of the interpolator's filter less this stage's fl2, which is a few samples for a minimum-phase
		// filter -- where the stream has zeros too: the slots between the array's start and the run)
		for (int i = lt; i < (int) (y - ybase); i += G::NT)
		{
			cd z;
			z.re = z.im = 0.0;
			ybase[i] = z;
		}
	}
	if (!HAF && nzero == 0 && L.fl2r + (int) (y - ybase) <= G::NT)
	{
		// (every block but the first ones of a stream, in the rotated layout of convp_prepare: fl2r = 0 or 1)
		// The thread's element p is y[lt + fl2r + NT p]: one address register, no index arithmetic, and nothing
		// to mask for p < 15 -- slots in_len ... of the run are only ever multiplied by the zero taps of the
		// padded rows or belong to masked outputs, and what lands there is finite transform data; the slots up to
		// fl2r + 16 NT - 1 lie inside the array.  The last element may wrap or leave the array: the general form.
		cd* const yl = y + (lt + L.fl2r);
#pragma unroll
		for (int p = 0; p < 15; p++)
		{
			cd v;
			v.re = st.vr[p];
			v.im = st.vi[p];
			yl[G::NT * p] = v;
		}
		const int u = (u0 + G::NT * 15) & mask;
		if (u < in_len)
		{
			cd v;
			v.re = st.vr[15];
			v.im = st.vi[15];
			y[u] = v;
		}
		return;
	}
	if (nzero == 0)
	{
		// (every block but the first ones of a stream)
#pragma unroll
		for (int p = 0; p < 16; p++)
		{
			const int u = (u0 + G::NT * p) & mask;
			if (u < in_len)
			{
				cd v;
				v.re = st.vr[p];
				v.im = st.vi[p];
				y[u] = v;
			}
		}
	}
	else
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		const int u = (lt + G::NT * p + L.fl2r) & mask;
		if (u < L.in_len)
		{
			cd v;
			v.re = u < nzero ? 0.0 : st.vr[p];
			v.im = u < nzero ? 0.0 : st.vi[p];
			y[u] = v;
		}
	}
	// zero extension read (times zero taps) by the padded polyphase rows
	for (int i = lt; i < kConvxRunPad; i += G::NT)
	{
		cd z;
		z.re = z.im = 0.0;
		y[L.in_len + i] = z;
	}
}

// MODE 0 / 3: K7 straight from the registers
template<int LN, int UL, int MODE = 0>
R8B_HD void cp_store_conv(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int chA,
	int chB, bool bvalid, int lt, const DstView& pd, long long pend)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int mask = G::N2 - 1;
	const long long t0 = cx_block_t0(L, k);
	if constexpr (MODE == 3)
	{
		// strided decimation (3x): output q sits at virtual time q * down (reference
		// CDSPBlockConvolver.h:564-583); the thread's element p is virtual time t0 + u
		if (!L.down_pow2 && L.down > 1)
		{
			const unsigned down = (unsigned) L.down;
			const long long qf = t0 >= 0 ? t0 / (long long) down : -((-t0 + down - 1) / (long long) down); // floor
			const unsigned r0 = (unsigned) (t0 - qf * (long long) down);
			// (the block's outputs counted from qf: output i = (r0 + u) / down where down divides r0 + u, u < in_len)
			const int nmax = (int) ((r0 + (unsigned) L.in_len) / down) + 1;
			auto run = [&](const CpStoreView& v)
			{
#pragma unroll
				for (int p = 0; p < 16; p++)
				{
					const unsigned u = (unsigned) ((lt + G::NT * p + L.fl2r) & mask);
					const unsigned w = r0 + u;
					const unsigned wq = down == 3u ? w / 3u : w / down;
					if (u < (unsigned) L.in_len && wq * down == w) cp_store1(v, wq, st.vr[p], st.vi[p], bvalid);
				}
			};
			run(cp_store_view(L.dst, chA, chB, qf, L.a, L.b, nmax));
			if (pend > L.b) run(cp_store_view(pd, chA, chB, qf, L.b, pend, nmax));
			return;
		}
	}
	// (valid outputs: u < in_len)
	auto run = [&](const CpStoreView& v)
	{
#pragma unroll
		for (int p = 0; p < 16; p++)
			cp_store1(v, (unsigned) ((lt + G::NT * p + L.fl2r) & mask), st.vr[p], st.vi[p], bvalid);
	};
	run(cp_store_view(L.dst, chA, chB, t0, L.a, L.b, L.in_len));
	// (the call's last block: what lies behind the call's range goes to the park view)
	if (pend > L.b) run(cp_store_view(pd, chA, chB, t0, L.b, pend, L.in_len));
}

// split 2x up-sampling form: the even half's outputs sit in st.er / st.ei (E[i] = y at circular position 2 (lt + NT i)),
// the odd half's in st.vr / st.vi (that position + 1); MODE 3: with the 3x strided store (output q at virtual time 3 q)
// (ea / oa, eb / ob: the even and odd outputs of channels A and B -- the one-channel form passes its single channel's)
template<int LN, int UL, int MODE>
R8B_HD void cp_sp_store(const ConvLaunch& L, const double* ea, const double* oa, const double* eb, const double* ob,
	long long k, int chA, int chB, bool bvalid, int lt, const DstView& pd, long long pend)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int mask = 2 * G::N - 1;
	const long long t0 = cx_block_t0(L, k);
	if constexpr (MODE == 3)
	{
		if (!L.down_pow2 && L.down > 1)
		{
			const unsigned down = (unsigned) L.down;
			const long long qf = t0 >= 0 ? t0 / (long long) down : -((-t0 + down - 1) / (long long) down); // floor
			const unsigned r0 = (unsigned) (t0 - qf * (long long) down);
			const int nmax = (int) ((r0 + (unsigned) L.in_len) / down) + 1;
			auto run = [&](const CpStoreView& v)
			{
#pragma unroll
				for (int i = 0; i < 16; i++)
				{
#pragma unroll
					for (int h = 0; h < 2; h++)
					{
						const unsigned u = (unsigned) ((2 * (lt + G::NT * i) + h + L.fl2r) & mask);
						const unsigned w = r0 + u;
						const unsigned wq = down == 3u ? w / 3u : w / down;
						if (u < (unsigned) L.in_len && wq * down == w)
							cp_store1(v, wq, h ? oa[i] : ea[i], h ? ob[i] : eb[i], bvalid);
					}
				}
			};
			run(cp_store_view(L.dst, chA, chB, qf, L.a, L.b, nmax));
			if (pend > L.b) run(cp_store_view(pd, chA, chB, qf, L.b, pend, nmax));
			return;
		}
	}
	auto run = [&](const CpStoreView& v)
	{
		// (E[i], O[i]) are neighbours in the row: one 16-byte store per channel where the pair starts on an even element
		// of 16-byte aligned fp64 rows (the launch's property: the rotation leaves fl2r = 0 or 1, blocks start in_len --
		// even -- apart); the two 8-byte stores otherwise.  (Separately they are two half-written 32-byte pieces per lane
		// pair on the way to the L2s: measured 403 MB written for 268 MB of outputs.)
		const bool al = ((v.qoff + (unsigned) L.fl2r) & 1u) == 0 && v.fmt == kPcmF64 && (v.m & 1u) != 0 &&
			((reinterpret_cast<unsigned long long>(v.pa) | reinterpret_cast<unsigned long long>(v.pb)) & 15ull) == 0;
#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			const int c0 = 2 * (lt + G::NT * i);
			const unsigned iE = (unsigned) ((c0 + L.fl2r) & mask), iO = (unsigned) ((c0 + 1 + L.fl2r) & mask);
			if (al && iO == iE + 1 && iE - v.ulo < v.uhi - v.ulo && iO < v.uhi)
			{
				const unsigned e = (v.qoff + iE) & v.m;
				cd va, vb;
				va.re = ea[i]; va.im = oa[i];
				vb.re = eb[i]; vb.im = ob[i];
				R8B_OUT_STORE16(v.pa + e, va);
				if (bvalid) R8B_OUT_STORE16(v.pb + e, vb);
			}
			else
			{
				cp_store1(v, iE, ea[i], eb[i], bvalid);
				cp_store1(v, iO, oa[i], ob[i], bvalid);
			}
		}
	};
	run(cp_store_view(L.dst, chA, chB, t0, L.a, L.b, L.in_len));
	if (pend > L.b) run(cp_store_view(pd, chA, chB, t0, L.b, pend, L.in_len));
}

#ifdef R8B_SPLIT_UP2
} // namespace r8bhip
#include "r8b_convp_split.h" // (development builds only: the occupancy experiment of round 4)
namespace r8bhip {
#endif

// MODE 1: K8 on the pair run (cf. cx_whole_compute): one 16-byte LDS read per tap feeds both channels
// Thread -> (phase, group set).  Fewer phases than threads (88200 -> 48000: 40): nsets = wt / out_step lanes
// share a phase and take its output groups in turn (set, set + nsets, ...).  More phases than threads
// (32000 -> 44100: 441): a thread takes phases tid, tid + wt, ...; the row of the first one was fetched ahead
// (st.row), the others are fetched here.
R8B_HD int cp_whole_phase(const ConvxLaunch& X, int tid) { return tid % X.out_step; }

template<int FLEN>
R8B_HD void cp_whole_compute(const ConvxLaunch& X, const SpanInfo& B, const cd* y, double* row, int* row_t,
	int chA, int chB, bool bvalid, int tid, int wt)
{
	// *row_t: the phase whose row `row` holds
	const long long jhi = B.jhi;
	const int nsets = wt >= X.out_step ? wt / X.out_step : 1;
	const int set = wt >= X.out_step ? tid / X.out_step : 0;
	if (set >= nsets) return;
	const int jstep = nsets * X.out_step, ustep = nsets * X.in_step;
	for (int t = wt >= X.out_step ? tid - set * X.out_step : tid; t < X.out_step; t += wt)
	{
		if (t != *row_t)
		{
			cx_whole_row<FLEN>(X, row, t);
			*row_t = t;
		}
		int d = t - B.jlo_mod;
		if (d < 0) d += X.out_step;
		long long j = B.jlo + d + (long long) set * X.out_step;
		if (j >= jhi) continue;
		int u = B.u_lo + (int) ((unsigned) (B.ph_lo + d * X.in_step) / (unsigned) X.out_step) + set * X.in_step;
		for (; j < jhi; j += jstep, u += ustep)
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
}

// ---- one-channel form with the interpolator fused in (MODE 18; round 5) ------------------------------------------
// The long blocks (16384 real points: transition bands of 0.5 ... 0.6 %) ran the interpolator as a launch of its own
// (k_whole) behind the one-channel form, i.e. the convolver's whole 1x stream went to HBM and came back: 268 of the
// 632 MB the 96000 -> 44100 path at a 0.5 % band moved per call.  Here the block's valid outputs
// go to LDS as a linear run of REAL samples -- the run overlays the block's own array, as in the pair form -- and the
// workgroup's 512 threads interpolate from it, one phase per thread, lanes sharing a phase in group sets
// (reference CDSPFracInterpolator.h:861-922, 991-1060 behind CDSPBlockConvolver.h:252-354).
// The thread's backward element i holds real samples 2 (lt + NT i) and + 1 of the circular block (st.vr / st.vi); sample c
// is output u = (c + fl2r) mod 2N of the run when u < in_len (cf. cp_sp_store).
template<int LN, int UL>
R8B_HD void cp_solo_final_store(const ConvLaunch& L, double* y, const ConvpState<LN, UL>& st, long long k, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int mask = 2 * G::N - 1;
	const long long t0 = cx_block_t0(L, k) - L.t_zero;
	// (the interpolator's stream starts at this stage's output t_zero: earlier outputs do not exist for it)
	const int nzero = t0 >= 0 ? 0 : (-t0 > L.in_len ? L.in_len : (int) -t0);
	const int in_len = L.in_len;
#pragma unroll
	for (int i = 0; i < 16; i++)
	{
		const int c0 = 2 * (lt + G::NT * i);
		const int uE = (c0 + L.fl2r) & mask, uO = (c0 + 1 + L.fl2r) & mask;
		if (uE < in_len) y[uE] = uE < nzero ? 0.0 : st.vr[i];
		if (uO < in_len) y[uO] = uO < nzero ? 0.0 : st.vi[i];
	}
	// zero extension read (times zero taps) by the padded polyphase rows
	for (int i = lt; i < kConvxRunPad; i += G::NT) y[in_len + i] = 0.0;
}

// (cp_whole_compute on a run of real samples: every tap one 8-byte LDS read)
template<int FLEN>
R8B_HD void cp_whole_compute_solo(const ConvxLaunch& X, const DstView& wd, const SpanInfo& B, const double* y, double* row,
	int* row_t, int ch, int tid, int wt)
{
	const long long jhi = B.jhi;
	const int nsets = wt >= X.out_step ? wt / X.out_step : 1;
	const int set = wt >= X.out_step ? tid / X.out_step : 0;
	if (set >= nsets) return;
	const int jstep = nsets * X.out_step, ustep = nsets * X.in_step;
	for (int t = wt >= X.out_step ? tid - set * X.out_step : tid; t < X.out_step; t += wt)
	{
		if (t != *row_t)
		{
			cx_whole_row<FLEN>(X, row, t);
			*row_t = t;
		}
		int d = t - B.jlo_mod;
		if (d < 0) d += X.out_step;
		long long j = B.jlo + d + (long long) set * X.out_step;
		if (j >= jhi) continue;
		int u = B.u_lo + (int) ((unsigned) (B.ph_lo + d * X.in_step) / (unsigned) X.out_step) + set * X.in_step;
		for (; j < jhi; j += jstep, u += ustep)
		{
			double v[FLEN];
			R8B_LDS_WINDOW(FLEN, v, y + u);
			double s[2] = { 0.0, 0.0 };
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
			dst_store(wd, ch, j, s[0] + s[1]);
		}
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
// MODE 5: the same with In > Out (down-sampling interpolators, cfg3: 320 / 147): the windows of adjacent
// phases start 2 or 3 samples apart, the rows have 27 entries, 27 reads feed four outputs.
R8B_HD int cp_ptab_fetch(const ConvxLaunch& X, int tid)
{
	// (the tables are laid out for 256 lanes: in a 512-thread workgroup the upper half sits the interpolator out)
	return tid < kConvpThreads ? X.ptab[tid] : -1;
}

template<int T2>
R8B_HD void cp_rows2_fetch(const ConvxLaunch& X, double* rows, int pt)
{
	// X.ctab holds the 2 T2 values of a phase pair as T2 pairs, pair i of phase pair q at [(i * ctp + q) * 2], ctp = the
	// number of phase pairs rounded up to whole quads: a lane quad reads 64 consecutive bytes, and the lanes of other
	// sets with the same phase pairs find them in the CU's cache (idle lanes read pair 0)
	const int q = pt < 0 ? 0 : pt & 0xff;
	const int ctp = (((X.out_step + 1) >> 1) + 3) & ~3;
	const cd* ct = reinterpret_cast<const cd*>(X.ctab);
#pragma unroll
	for (int i = 0; i < T2; i++)
	{
		const cd v = R8B_TAB_LD_R(ct, i * ctp, q);
		rows[2 * i] = v.re;
		rows[2 * i + 1] = v.im;
	}
}

// (wd: where the outputs go -- the launch's X.wdst, or the park buffer for the part of the call's last block that
// belongs to the next call, ConvxLaunch::park_dst)
// (NBUF: chunks of the window in flight -- 2: a chunk's reads issued one chunk ahead of its multiply-adds; 1: the
// half-array form, whose register budget is 168: a chunk's reads, then its multiply-adds, the CU's other workgroups in between)
// (SKIPM: lanes none of whose two outputs is stored in this round read nothing -- the half-array form, whose array ends
// a few slots behind the run: the window of a MASKED output of a block's last group may begin up to in_step slots further)
template<int T2, bool ALIGNED_ONLY = false, int NBUF = 2, int NBUFG = NBUF, bool SKIPM = false>
R8B_HD void cp_whole2_compute(const ConvxLaunch& X, const DstView& wd, const SpanInfo& Bm, const cd* y, const double* rows,
	int pt, int chA, int chB, bool bvalid)
{
	// pt: phase pair q (bits 0-7), group set (8-11), window start floor(2 q In / Out) (12-)
	if (pt < 0) return;
	const int q = pt & 0xff, set = (pt >> 8) & 15, rq = pt >> 12;
	// (block constants into registers once: the kernel arguments live in memory)
	const int hi_mod = Bm.pad, gmax = Bm.ph_lo, lo_mod = Bm.jlo_mod, u_lo = Bm.u_lo;
	if (hi_mod == 0) return;
	const int in_step = X.in_step, out_step = X.out_step, nsets = X.nsets;
	const long long jg0 = Bm.jlo - lo_mod, jg = jg0 + 2 * q;
	// group 0 starts at the block's first output (phase lo_mod), the last group ends before phase hi_mod
	const bool f0 = 2 * q >= lo_mod, f1 = 2 * q + 1 >= lo_mod && 2 * q + 1 < out_step;
	const bool l0 = 2 * q < hi_mod, l1 = 2 * q + 1 < hi_mod && 2 * q + 1 < out_step;
	const bool linear = wd.mask == -1 && wd.fmt == kPcmF64;
	// (row pointers of the block's first group: uniform over the workgroup, so is the alignment test)
	double* const pa0 = wd.p + ((long long) chA * wd.stride + (jg0 + wd.off));
	double* const pb0 = wd.p + ((long long) chB * wd.stride + (jg0 + wd.off));
	double* const pa = pa0 + 2 * q;
	double* const pb = pb0 + 2 * q;
	// (A thread's two outputs are neighbours in the row; pair16: the pairs of every group start at 16-byte aligned
	// addresses.  Without that -- a call whose outputs start at an odd column of the caller's rows, or an odd number of
	// phases -- the pair is still ONE store instruction: R8B_OUT_STORE16U.  As two 8-byte stores the odd calls of a
	// column-0 caller cost 4.4 % more than the even ones: tools/placement_probe.py, profiles/r05_experiments.txt.)
	// (An odd number of phases -- cfg3's 147: the thread's two outputs are neighbours all the same, at an alignment that
	// alternates from group to group, and only the LAST pair has no second phase: p1ok.)
	const bool p1ok = 2 * q + 1 < out_step;
	const bool pair16 = (((size_t) pa0 | (size_t) pb0) & 15) == 0 && (out_step & 1) == 0;
	constexpr int CH = T2 == 25 ? 5 : 3, NCH = T2 / CH;
	static_assert(CH * NCH == T2, "chunks");
	if (ALIGNED_ONLY || (lo_mod == 0 && hi_mod == out_step && linear))
	{
		// Whole groups only (every block of a call but those cut by its ends, when the blocks are aligned to
		// groups -- Engine::launch_fused): nothing to mask, every output pair is one 16-byte store.
		for (int gl = set; gl <= gmax; gl += nsets)
		{
			const cd* w = y + (u_lo + in_step * gl + rq);
			double a0[2] = { 0.0, 0.0 }, b0[2] = { 0.0, 0.0 }, a1[2] = { 0.0, 0.0 }, b1[2] = { 0.0, 0.0 };
			cd v[NBUF][CH];
#pragma unroll
			for (int i = 0; i < CH; i++) v[0][i] = w[i];
#pragma unroll
			for (int c = 0; c < NCH; c++)
			{
				R8B_SCHED_FENCE();
				if (NBUF == 2 && c + 1 < NCH)
				{
#pragma unroll
					for (int i = 0; i < CH; i++) v[(c + 1) & 1][i] = w[CH * (c + 1) + i];
				}
				// (last pair of the chunk first: LDS returns in order, so the wait in front of its multiply-adds covers
				// the whole chunk -- one wait instruction per chunk instead of one per pair)
#pragma unroll
				for (int i = CH - 1; i >= 0; i--)
				{
					const int t = CH * c + i;
					a0[t & 1] += rows[t] * v[c & (NBUF - 1)][i].re;
					b0[t & 1] += rows[t] * v[c & (NBUF - 1)][i].im;
					a1[t & 1] += rows[T2 + t] * v[c & (NBUF - 1)][i].re;
					b1[t & 1] += rows[T2 + t] * v[c & (NBUF - 1)][i].im;
				}
				if (NBUF == 1 && c + 1 < NCH)
				{
					R8B_SCHED_FENCE();
					R8B_MEM_FENCE();
#pragma unroll
					for (int i = 0; i < CH; i++) v[0][i] = w[CH * (c + 1) + i];
				}
			}
			// (channel B's sums are only stored under a uniform condition: without this the compiler moves their
			// multiply-adds behind it, away from the LDS reads they should overlap, and keeps the whole window live)
			R8B_FORCE4(b0[0], b0[1], b1[0], b1[1]);
			// (uniform row pointer + one 32-bit index: no 64-bit address arithmetic per store)
#if R8B_ABL_STORES
			const unsigned o = (unsigned) (out_step * gl + 2 * q) & 63u;
#else
			const unsigned o = (unsigned) (out_step * gl + 2 * q);
#endif
			cd va, vb;
			va.re = a0[0] + a0[1];
			va.im = a1[0] + a1[1];
			vb.re = b0[0] + b0[1];
			vb.im = b1[0] + b1[1];
			if (pair16)
			{
				R8B_OUT_STORE16(pa0 + o, va);
				if (bvalid) R8B_OUT_STORE16(pb0 + o, vb);
			}
			else if (ALIGNED_ONLY || p1ok)
			{
				// (a call whose outputs start at an odd column of the rows: the same values, the same single instruction)
				R8B_OUT_STORE16U(pa0 + o, va);
				if (bvalid) R8B_OUT_STORE16U(pb0 + o, vb);
			}
			else
			{
				// (the last phase of an odd number of them)
				pa0[o] = va.re;
				if (bvalid) pb0[o] = vb.re;
			}
		}
		return;
	}
	if constexpr (!ALIGNED_ONLY)
	for (int gl = set; gl <= gmax; gl += nsets)
	{
		if constexpr (SKIPM)
		{
			if (!((gl > 0 || f0) && (gl < gmax || l0)) && !((gl > 0 ? 2 * q + 1 < out_step : f1) && (gl < gmax || l1))) continue;
		}
		const cd* w = y + (u_lo + in_step * gl + rq);
		double a0[2] = { 0.0, 0.0 }, b0[2] = { 0.0, 0.0 }, a1[2] = { 0.0, 0.0 }, b1[2] = { 0.0, 0.0 };
		// the window in chunks of five taps, each chunk's reads issued one chunk ahead of its
		// multiply-adds (two chunks of 16-byte values are live: the scheduler, left alone, reads all 25
		// first -- 100 registers the prefetched samples of the next block then have to leave for)
		cd v[NBUFG][CH];
#pragma unroll
		for (int i = 0; i < CH; i++) v[0][i] = w[i];
#pragma unroll
		for (int c = 0; c < NCH; c++)
		{
			R8B_SCHED_FENCE();
			if (NBUFG == 2 && c + 1 < NCH)
			{
#pragma unroll
				for (int i = 0; i < CH; i++) v[(c + 1) & 1][i] = w[CH * (c + 1) + i];
			}
			// (the same order of additions as the loop above: an output must not depend on which loop produced it)
#pragma unroll
			for (int i = CH - 1; i >= 0; i--)
			{
				const int t = CH * c + i;
				a0[t & 1] += rows[t] * v[c & (NBUFG - 1)][i].re;
				b0[t & 1] += rows[t] * v[c & (NBUFG - 1)][i].im;
				a1[t & 1] += rows[T2 + t] * v[c & (NBUFG - 1)][i].re;
				b1[t & 1] += rows[T2 + t] * v[c & (NBUFG - 1)][i].im;
			}
			if (NBUFG == 1 && c + 1 < NCH)
			{
				// (the sums are only stored under conditions: left alone the compiler moves their multiply-adds behind
				// those, away from the reads, and keeps the whole window live -- 100 registers)
				R8B_FORCE4(a0[0], a0[1], a1[0], a1[1]);
				R8B_FORCE4(b0[0], b0[1], b1[0], b1[1]);
				R8B_SCHED_FENCE();
				R8B_MEM_FENCE();
#pragma unroll
				for (int i = 0; i < CH; i++) v[0][i] = w[CH * (c + 1) + i];
			}
		}
		const bool v0 = (gl > 0 || f0) && (gl < gmax || l0);
		const bool v1 = (gl > 0 ? 2 * q + 1 < out_step : f1) && (gl < gmax || l1);
		if (linear)
		{
			// caller's buffer: row pointers once, a 32-bit index per output; the two phases of a
			// channel as one 16-byte store when the pair is aligned (the same for every thread: pairs
			// start at even output indices of a group)
			const int o = out_step * gl;
			if (v0 && v1)
			{
				cd va, vb;
				va.re = a0[0] + a0[1];
				va.im = a1[0] + a1[1];
				vb.re = b0[0] + b0[1];
				vb.im = b1[0] + b1[1];
				if (pair16)
				{
					R8B_OUT_STORE16(pa + o, va);
					if (bvalid) R8B_OUT_STORE16(pb + o, vb);
				}
				else
				{
					R8B_OUT_STORE16U(pa + o, va);
					if (bvalid) R8B_OUT_STORE16U(pb + o, vb);
				}
				continue;
			}
			if (v0)
			{
				pa[o] = a0[0] + a0[1];
				if (bvalid) pb[o] = b0[0] + b0[1];
			}
			if (v1)
			{
				pa[o + 1] = a1[0] + a1[1];
				if (bvalid) pb[o + 1] = b1[0] + b1[1];
			}
			continue;
		}
		const long long j = jg + (long long) out_step * gl;
		if (v0)
		{
			dst_store(wd, chA, j, a0[0] + a0[1]);
			if (bvalid) dst_store(wd, chB, j, b0[0] + b0[1]);
		}
		if (v1)
		{
			dst_store(wd, chA, j + 1, a1[0] + a1[1]);
			if (bvalid) dst_store(wd, chB, j + 1, b1[0] + b1[1]);
		}
	}
}

// convolver-only modes: the park view of the call's LAST block (st.pf: this workgroup holds it; in a workgroup that
// carries several blocks only the last of them parks) -- outputs [L.b, park_blk.jhi) at index q - L.b of park_dst
template<class Exec, class St, class Item>
R8B_HD void cp_park_view(const Exec& ex, const ConvxLaunch& XM, const St& st, long long k, const Item& cur, DstView& pd,
	long long& pend)
{
	if (ex.uniform(st.pf) != 0 && k == XM.c.k0 + XM.c.nblk - 1)
	{
		pd.p = XM.park_dst;
		pd.stride = XM.park_stride;
		pd.mask = -1;
		pd.off = -XM.c.b;
		pd.fmt = kPcmF64;
		pend = XM.park_blk.jhi;
	}
	(void) cur;
}

// ---- polyphase 3x form (MODE 19; round 5) ---------------------------------------------------------------------------
// A 3x up-sampling convolver on the zero-stuffing block transforms N points of which two thirds are zeros (reference
// CDSPBlockConvolver.h:414-496 copyUpsample in front of :283-350).  The same outputs from the INPUT stream alone:
// y[3 m + r] = sum_d x[m - d] g_r[d], g_r[d] = h[3 d + r + fl2] (ConvGeom::p3) -- three FIR filters over one window of
// N input samples: ONE forward transform Z, then y_r = IDFT(Z G_r) for r = 0, 1, 2, one after the other in the same
// array (1:1 geometry, complex products per position: G_r is the spectrum of a one-sided piece of h).  The three
// components' outputs meet in registers and leave together (cp_p3_store).
// Block k: window = input samples [t0 / 3 - rot, + N), t0 = k blk_stride + blk_offset - fl2 (a multiple of 3; rot = the
// components' reach into the past), valid outputs = virtual times t0 ... t0 + in_len - 1 at circular positions 0 ...
// in_len / 3 - 1 of each component (Engine::pair_constants_poly3 lays the components out for that).
template<int LN, int UL>
R8B_HD void cp_p3_load(const ConvLaunch& L, ConvpState<LN, UL>& st, long long k, int chA, int chB, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int R = G::E1, q = G::N / R;
	const long long t0 = k * (long long) L.blk_stride + L.blk_offset - L.fl2;
	// (exact: t0 is a multiple of 3, negative for the stream's first block)
	const long long w0 = t0 / 3 - L.rot;
	if (L.src.cur_fmt == kPcmF64 && w0 >= L.src.cur_base && w0 >= 0)
	{
		const double* const pa = L.src.cur + ((long long) chA * L.src.cur_stride + (w0 - L.src.cur_base));
		const double* const pb = L.src.cur + ((long long) chB * L.src.cur_stride + (w0 - L.src.cur_base));
#pragma unroll
		for (int p = 0; p < R; p++)
		{
			st.pr[p] = pa[lt + p * q];
			st.pi[p] = pb[lt + p * q];
		}
		return;
	}
	const SrcBlock sa = src_block(L.src, chA, w0), sb = src_block(L.src, chB, w0);
#pragma unroll
	for (int p = 0; p < R; p++)
	{
		st.pr[p] = src_block_load1(sa, lt + p * q);
		st.pi[p] = src_block_load1(sb, lt + p * q);
	}
}
// the block's spectrum: the last forward butterflies over the thread's 16 consecutive positions
template<int LN, int UL>
R8B_HD void cp_p3_spectrum(const cd* buf, ConvpState<LN, UL>& st, int lt)
{
	typedef ConvpGeom<LN, UL> G;
	static_assert(UL == 0 && G::E1 == 16 && !G::POST, "polyphase 3x form: 1:1 geometries up to 4096 points");
	const SwBase bbf = sw_base(buf, fslot<LN, UL>(G::E1 * lt));
#pragma unroll
	for (int c = 0; c < 16; c++)
	{
		const cd v = sw_ld(bbf, fmap_c<LN, UL>(c));
		st.zr[c] = v.re;
		st.zi[c] = v.im;
	}
#pragma unroll
	for (int f = 0; f < G::NBF; f++) dif_regs<G::RM>(st.zr + G::RM * f, st.zi + G::RM * f);
}
// component r: Z G_r and the first backward butterflies (results in st.vr / st.vi, as cp_middle_compute leaves them)
// (the component's sixteen constants are fetched HERE, one product at a time: fetched a component ahead -- 64 registers
// more across the backward passes -- the kernel measured slower, 0.335 against 0.313 ms; profiles/r05_experiments.txt)
template<int LN, int UL>
R8B_HD void cp_p3_product(const ConvLaunch& L, ConvpState<LN, UL>& st, int lt, int r)
{
	typedef ConvpGeom<LN, UL> G;
	const cd* const hr = L.hp + (size_t) r * 16 * G::NT + lt;
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		const cd h = hr[p * G::NT];
		st.vr[p] = st.zr[p] * h.re - st.zi[p] * h.im;
		st.vi[p] = st.zr[p] * h.im + st.zi[p] * h.re;
	}
	dit_regs<16>(st.vr, st.vi);
}
// The three components' outputs leave TOGETHER -- position i of the block is virtual times t0 + 3 i, + 1, + 2: three
// consecutive doubles per channel, a wave's stores cover 1.5 KB of consecutive bytes back to back -- so the first two wait
// in registers (st.er / st.ei, st.p3r / st.p3i) while the next one is transformed.  Measured against storing each component
// straight from its last pass (8-byte stores 24 bytes apart, the three parts of a cache line microseconds apart; 221
// registers instead of 256 + spills): 0.271 against 0.313 ms per call for 16000 -> 48000 (profiles/r05_experiments.txt).
// Only the thread's first kP3Keep positions wait: a block's valid positions are i < in_len / 3 <= N - (the components'
// reach), i.e. 10.5 of a thread's 16 for the 24-bit filter at 2 % -- the ones beyond 11 (short filters in a long window)
// leave component by component (cp_p3_store_rest), and 40 registers are not held for values that are mostly never stored.
// (The first two of the three as one 16-byte store at element alignment -- R8B_OUT_STORE16U -- measured slower: 0.251
// against 0.227 ms; a wave's three 8-byte stores cover its 1.5 KB evenly, a 16- and an 8-byte one do not.)
static const int kP3Keep = 11;
template<int LN, int UL>
R8B_HD void cp_p3_store(const ConvLaunch& L, const ConvpState<LN, UL>& st, long long k, int chA, int chB, bool bvalid,
	int lt, const DstView& pd, long long pend)
{
	typedef ConvpGeom<LN, UL> G;
	const long long t0 = k * (long long) L.blk_stride + L.blk_offset - L.fl2;
	const int m3 = L.in_len / 3;
	auto run = [&](const CpStoreView& v)
	{
#pragma unroll
		for (int p = 0; p < kP3Keep; p++)
		{
			const int i = lt + G::NT * p;
			if (i < m3)
			{
				cp_store1(v, (unsigned) (3 * i), st.er[p], st.ei[p], bvalid);
				cp_store1(v, (unsigned) (3 * i + 1), st.p3r[p], st.p3i[p], bvalid);
				cp_store1(v, (unsigned) (3 * i + 2), st.vr[p], st.vi[p], bvalid);
			}
		}
	};
	run(cp_store_view(L.dst, chA, chB, t0, L.a, L.b, L.in_len));
	if (pend > L.b) run(cp_store_view(pd, chA, chB, t0, L.b, pend, L.in_len));
}
// (component r of the positions that do not wait, straight from st.vr / st.vi)
template<int LN, int UL>
R8B_HD void cp_p3_store_rest(const ConvLaunch& L, const ConvpState<LN, UL>& st, int r, long long k, int chA, int chB,
	bool bvalid, int lt, const DstView& pd, long long pend)
{
	typedef ConvpGeom<LN, UL> G;
	const int m3 = L.in_len / 3;
	if (m3 <= G::NT * kP3Keep) return; // (uniform: the usual case)
	const long long t0 = k * (long long) L.blk_stride + L.blk_offset - L.fl2;
	auto run = [&](const CpStoreView& v)
	{
#pragma unroll
		for (int p = kP3Keep; p < 16; p++)
		{
			const int i = lt + G::NT * p;
			if (i < m3) cp_store1(v, (unsigned) (3 * i + r), st.vr[p], st.vi[p], bvalid);
		}
	};
	run(cp_store_view(L.dst, chA, chB, t0, L.a, L.b, L.in_len));
	if (pend > L.b) run(cp_store_view(pd, chA, chB, t0, L.b, pend, L.in_len));
}

// ---- the kernel body ---------------------------------------------------------------------------------

// one workgroup's work: blocks k0 .. k0 + nvalid - 1 (nvalid <= SUB) of the channel pair (chA, chB);
// bvalid: chB is a real channel (an odd channel count leaves the last one without a partner: its block
// rides alone, chB = chA)
struct ConvpItem
{
	long long k;
	int nvalid;
	int chA, chB;
	bool bvalid;
};

// Walk form (round 5; convp_walk): a workgroup takes SEVERAL consecutive blocks of its channel pair, one after the other,
// and keeps across them what is the same for every block -- the interpolator's two rows and lane-table entry and the
// thread's own twiddles in registers, the wave-local passes' twiddle table in LDS (ConvpTwLds) -- and requests block
// k + 1's samples while block k is interpolated.  more: another block follows this one.
struct ConvpWalk
{
	bool more;
};
template<int LN, int UL, int MODE> constexpr bool convp_walk_ok()
{
	// (two phases per thread, one block pair per workgroup, the plain backward side, two lean twiddles per pass at most.
	// UL >= 1: the 2x up-sampling convolver in front of the interpolator -- every up-sampling ratio's first stage.  The 1:1
	// geometry <12, 0> qualifies by structure, but its walk body keeps 280 bytes per lane in scratch and no ratio of the
	// rate table takes it: not instantiated)
	return (MODE == 4 || MODE == 5 || MODE == 16 || MODE == 17) && UL >= 1 && ConvpGeom<LN, UL>::SUB == 1 &&
		!ConvpGeom<LN, UL>::POST && ConvpGeom<LN, UL>::NB2 == 1 && ConvpTwLds<LN, UL>::ON;
}


// X: the launch descriptor as the phases read it -- on the GPU a copy whose hot scalars k_convp has pinned in
// scalar registers (see there); XM: the descriptor in kernel-argument memory, for its per-block array only.
template<int LN, int UL, int MODE, int FLENP, bool WALK = false, class Exec>
R8B_HD void convp_body(Exec& ex, const ConvxLaunch& X, const ConvxLaunch& XM, cd* buf, const ConvpItem& cur,
	const ConvpWalk walk = ConvpWalk())
{
	typedef ConvpGeom<LN, UL> G;
	typedef ConvpState<LN, UL> St;
	static_assert(!WALK || convp_walk_ok<LN, UL, MODE>(), "walk form: fused two-phase modes of the 4096-point geometries");
	(void) walk;
	// modes 6 / 7: modes 0 / 3 with a complex kernel spectrum
	// modes 16 / 17: modes 4 / 5 (fused interpolator, two phases per thread) with a complex kernel spectrum
	constexpr bool CX = MODE == 6 || MODE == 7 || MODE == 16 || MODE == 17 || (MODE >= 29 && MODE <= 32);
	// modes 8 / 9: modes 0 / 3 of the split 2x up-sampling form (cp_sp_*: geometry <13, 0> only); 12 / 13: the same with a
	// complex kernel spectrum
	constexpr bool SP = convp_mode_sp(MODE);
	// modes 10 / 11: modes 0 / 3 of the one-channel form (cp_solo_*: geometries <13, 0> and <13, -1>; cur.chA is the
	// channel, cur.bvalid false); 14 / 15: the same with a complex kernel spectrum
	constexpr bool SOLO = convp_mode_solo(MODE);
	constexpr bool CXL = MODE >= 12 && MODE <= 15;

	constexpr int BM = MODE == 6 || MODE == 8 || MODE == 10 || MODE == 12 || MODE == 14 ? 0 :
		(MODE == 7 || MODE == 9 || MODE == 11 || MODE == 13 || MODE == 15 ? 3 : (MODE == 16 ? 4 : (MODE == 17 ? 5 :
		(MODE == 18 ? 1 : (MODE == 19 || MODE == 20 || MODE == 21 || MODE == 27 || MODE == 31 ? 0 : (MODE == 22 || MODE == 28 || MODE == 32 ? 3 :
		(MODE == 23 || MODE == 29 ? 4 : (MODE == 25 || MODE == 30 || MODE == 33 ? 5 : MODE))))))));
	// mode 20: mode 0 of the decimating form behind a half-band decimator taken in the load (cp_hbf_*)
	constexpr bool HBF = MODE == 20;
	// mode 21: mode 0 in the half-array form (cp_ha_*: the backward side's exchanges by parts through an array of doubles)
	constexpr bool HA = convp_mode_ha(MODE);
	static_assert(!HA || convp_ha_ok<LN, UL>(), "half-array form: the 2048 -> 4096-point and 4096 -> 8192-point 2x up-sampling geometries");
	static_assert(!HA || !ConvpGeom<LN, UL>::POST || BM == 0 || BM == 3, "half-array form on 8192 points: convolver-only modes");
	// (the forward transform's exchanges by parts too: the decimating form, and the 1:1 form where both sides go that way)
	constexpr bool HAF = HA && UL <= 0;
	// mode 19: polyphase 3x form (cp_p3_*): a convolver-only mode with its own load, middle and store
	constexpr bool P3 = convp_mode_p3(MODE);
	// (development builds, R8B_SPLIT_UP2: the other modes of the geometry are compiled as before and must not be launched)
	constexpr bool SPLIT = kSplit<LN, UL> && !CX && (BM == 0 || BM == 3);
	(void) SPLIT;
	// the pair's two channels are brought to one binary order of magnitude per block (cp_level_words; the one-channel
	// forms have no partner)
	constexpr bool LEVELS = !SOLO;
	const ConvLaunch& L = X.c;
	const int chA = cur.chA, chB = cur.chB;
	const bool bvalid = cur.bvalid;
	// a thread's block pair: its own part of the array; slots past the launch's last block redo that block
	// (they take part in every barrier) and store nothing
	auto hp_prefetch = [&](St& st, int lt)
	{
		if constexpr (P3) { (void) st; (void) lt; } // (fetched where they are multiplied: cp_p3_product)
		else
		if constexpr (SOLO) { (void) st; (void) lt; } // (fetched behind the spectrum's write: cp_solo_mid_a)
		else if constexpr (SP) cp_sp_hp_prefetch<LN, UL>(L, st, lt);
		else cp_hp_prefetch<LN, UL, CX>(L, st, lt);
	};
	auto sub_of = [&](int tid) { return convp_sub<LN, UL>(tid); };
	auto lt_of = [&](int tid) { return convp_lt<LN, UL>(tid); };
	auto buf_of = [&](int tid) { return buf + sub_of(tid) * G::NA; };
	// (the workgroup's twiddle table in LDS, behind the array and the flag words: ConvpTwLds)
	typedef ConvpTwLds<LN, UL> TL;
	constexpr int ABYTES = convp_mode_array_bytes<LN, UL, MODE>();
	cd* const ltw = reinterpret_cast<cd*>(reinterpret_cast<unsigned char*>(buf) + ABYTES + kConvpFlagBytes);
	auto k_of = [&](int tid)
	{
		if constexpr (G::SUB == 1) return cur.k;
		else
		{
			const int sb = sub_of(tid);
			return cur.k + (sb < cur.nvalid ? sb : cur.nvalid - 1);
		}
	};
	auto live = [&](int tid)
	{
		if constexpr (G::SUB == 1) return true;
		else return sub_of(tid) < cur.nvalid;
	};
	// The shared one-element copies (cp_park_slice_*, cp_tail_slice_*): requested at entry and stored in the last phase --
	// or, on the 512-thread geometries (8192-point arrays: 255 registers and spills, blocks of 80 000+ cycles) and in the
	// polyphase 3x form (three backward transforms per block, spills), requested AND stored in the last phase: twelve
	// registers less held across the block for a load whose wait ends the workgroup (modes 8 / 9: 92 -> 0 bytes of scratch
	// per lane, 12 / 13: 76 -> 32, 14 / 15: 28 -> 0)
	// (... and in the half-array form, whose budget is 128 registers)
	constexpr bool LATE = G::WT > 256 || P3 || HA;
	auto slices_out = [&](int tid, St& st)
	{
		if constexpr (LATE)
		{
			if constexpr (BM != 1)
			{
				st.tka = nullptr;
				if ((L.tail_flags & 8) != 0) cp_tail_slice_load<G::WT>(L, st, (int) (cur.k - L.k0) / G::SUB, chA, chB, tid);
			}
			if constexpr (BM != 1 || SOLO)
			{
				st.pka = nullptr;
				if (X.park_n > 0 && X.park_slices != 0)
					cp_park_slice_load<G::WT>(XM, X.wdst, st, (int) (cur.k - L.k0) / G::SUB, chA, chB, tid);
			}
		}
		(void) tid;
		cp_park_slice_store(X.wdst, st, chA, chB, bvalid);
		cp_tail_slice_store(L, st, chA, chB, bvalid);
	};
	auto front = [&](int tid, St& st, cd& twl_v)
	{
		const int lt = lt_of(tid);
		twl_v.re = twl_v.im = 0.0;
		if constexpr (WALK)
		{
			// (walk form: twiddles, table and this block's samples are there already -- convp_walk, or the previous
			// block's interpolator phase)
			// (opaque copies: the powers the pass derives from them are loop invariant, and hoisted out of the block loop
			// they would stay live -- or spilled -- across all phases)
			st.tw[0] = st.twp[0];
			R8B_OPAQUE2(st.tw[0].re, st.tw[0].im);
			if constexpr (G::E1 >= 16)
			{
				st.tw[3] = st.twp[1];
				R8B_OPAQUE2(st.tw[3].re, st.tw[3].im);
			}
			// (an interior block: its window lies inside the caller's fp64 buffer)
			cp_load<LN, UL, BM, SP, true>(L, st, k_of(tid), chA, chB, lt);
		}
		else
		{
		// (the first pass's twiddles -- L2 -- requested ahead of the samples -- HBM --, not behind their wait)
		ptw_fetch_lean<G::E1, G::NT>(st.tw, L.ptw, 0, lt);
		// (... and this thread's entry of the wave-local passes' twiddle table)
		if constexpr (TL::ON)
		{
			if (tid < TL::NE) twl_v = L.ptw[TL::src_index(tid)];
		}
		ex.stamp2();
		if constexpr (HBF) cp_hbf_gather<LN, UL>(buf_of(tid), st, lt);
		else if constexpr (P3) cp_p3_load<LN, UL>(L, st, k_of(tid), chA, chB, lt);
		else if constexpr (SOLO) cp_load_solo<LN, UL, BM>(L, st, k_of(tid), chA, lt);
		else cp_load<LN, UL, BM, SP>(L, st, k_of(tid), chA, chB, lt);
		}
		// (walk form: interior blocks own no part of the history tail and never hold the call's last output -- the shared
		// one-element copies are all they take part in)
		if constexpr (!WALK)
		if ((L.tail_flags & 2) != 0 && cur.k + (G::SUB == 1 ? 1 : cur.nvalid) > L.k0 + L.tail_bf)
		{
			// (the launch's last block: the samples behind its window -- read by no block of this call -- requested
			// beside its own, one wait for both; all of them in the caller's fp64 buffer: convp_tail_owners)
			if ((L.tail_flags & 8) == 0 && cur.k + (G::SUB == 1 ? 1 : cur.nvalid) == L.k0 + L.nblk)
				cp_tail_rest<G::WT>(L, L.tail_c1, L.tail_p1, chA, chB, bvalid, tid);
			ex.stamp2();
			if constexpr (SOLO) cp_tail_owned_solo<LN, UL>(L, st, k_of(tid), chA, lt);
			else if (live(tid)) cp_tail_owned<LN, UL, SP>(L, st, k_of(tid), chA, chB, bvalid, lt);
		}
		if constexpr (BM != 1)
		{
			st.tka = nullptr;
			if constexpr (!LATE)
				if ((L.tail_flags & 8) != 0) cp_tail_slice_load<G::WT>(L, st, (int) (cur.k - L.k0) / G::SUB, chA, chB, tid);
		}
		if constexpr (BM != 1 || SOLO)
		{
			// Parked outputs (ConvxLaunch::park_*): does this workgroup hold the call's last block (whose outputs behind
			// the call's range are parked, not computed again by the next call)?  The previous call's parked outputs:
			// this thread's element requested here, behind the samples, and stored in the workgroup's last phase --
			// or all of them by the launch's first workgroup of the pair (short calls).
			st.pf = !WALK && X.park_out != 0 && cur.k + (G::SUB == 1 ? 1 : cur.nvalid) == L.k0 + L.nblk ? 1 : 0;
			st.pka = nullptr;
			if (X.park_n > 0)
			{
				if (X.park_slices != 0)
				{
					if constexpr (!LATE) cp_park_slice_load<G::WT>(XM, X.wdst, st, (int) (cur.k - L.k0) / G::SUB, chA, chB, tid);
				}
				else if (!WALK && cur.k == L.k0) cp_park_back<G::WT>(XM, X.wdst, chA, chB, bvalid, tid);
			}
		}
		ex.stamp2();
		ex.post_bits(tid, cp_nonzero_bits<LN, UL>(st));
		if constexpr (LEVELS) ex.post_levels(tid, sub_of(tid), cp_level_words<LN, UL>(st));
		ex.stamp2();
	};
	// (the block's level shift as the end of the body sees it: read back from the block's shift word -- one LDS word, left
	// there by the first pass --, not kept in registers across the phases)
	auto level_shift = [&](int tid)
	{
		if constexpr (LEVELS) return ex.collect_shift(sub_of(tid));
		else { (void) tid; return 0; }
	};
	auto first_pass = [&](int tid, St& st, const cd& twl_v)
	{
		const int lt = lt_of(tid);
		if constexpr (LEVELS)
		{
			const int lsh = cp_level_shift(ex.collect_levels(sub_of(tid)));
			ex.post_shift(tid, sub_of(tid), lt, lsh);
			cp_scale_in<LN, UL>(st, lsh);
		}
		if constexpr (HAF)
		{
			// (half-array form of the forward side: the first pass's results stay in registers; their real parts go to the array)
			constexpr int NBW = 6;
			cd loc[NBW];
#pragma unroll
			for (int c = 0; c < NBW; c++) loc[c] = st.tw[c];
			tw_expand<G::E1>(loc);
			cp_first_arith<LN, UL>(st, loc, st.vr, st.vi);
			cp_ha_st<G::N>(buf_of(tid), st.vr, lt);
		}
		else
		cp_first<LN, UL, HA>(L, buf_of(tid), st, lt);
		if constexpr (TL::ON && !WALK)
		{
			if (tid < TL::NE) twl_st(ltw, tid, twl_v);
		}
		(void) twl_v;
		// (modes 4 / 5: the thread's entry of the interpolator's lane table, long before its rows are addressed with it)
		if constexpr ((BM == 4 || BM == 5) && !WALK) st.pt = cp_ptab_fetch(X, tid);
		if constexpr (G::NPRE > 1) ConvpPre<LN, UL, 1>::prefetch(L, st, lt);
		else hp_prefetch(st, lt);
	};
	if constexpr (HBF)
	{
		// (the decimator's outputs for the block's window: two rounds of staging + sliding-window sums, then through
		// the array into the first pass's layout -- a barrier behind each step: they all work on the same LDS)
		double* const xs = reinterpret_cast<double*>(buf);
		ex.phase([&](int tid, St&) { cp_hbf_stage<LN, UL>(L, XM, xs, cur.k, 0, chA, chB, tid); });
		ex.phase([&](int tid, St& st) { cp_hbf_compute<0, LN, UL>(L, XM, xs, st, cur.k, tid); });
		ex.phase([&](int tid, St&) { cp_hbf_stage<LN, UL>(L, XM, xs, cur.k, 1, chA, chB, tid); });
		ex.phase([&](int tid, St& st) { cp_hbf_compute<1, LN, UL>(L, XM, xs, st, cur.k, tid); });
		ex.phase([&](int tid, St& st) { cp_hbf_scatter<LN, UL>(L, buf, st, tid); });
	}
	if constexpr (WALK)
	{
		// (a barrier between the two: the slowest wave of the PREVIOUS block still reads its run from the array the first
		// pass overwrites)
		ex.phase([&](int tid, St& st)
		{
			cd twl_v;
			front(tid, st, twl_v);
		});
		ex.phase([&](int tid, St& st)
		{
			cd twl_v;
			twl_v.re = twl_v.im = 0.0;
			first_pass(tid, st, twl_v);
		});
	}
	else if constexpr (LEVELS)
	{
		// (a barrier between the two: the block's levels come from all of its threads)
		ex.phase([&](int tid, St& st) { front(tid, st, st.twlv); });
		ex.phase([&](int tid, St& st) { first_pass(tid, st, st.twlv); });
	}
	else
	ex.phase([&](int tid, St& st)
	{
		cd twl_v;
		front(tid, st, twl_v);
		first_pass(tid, st, twl_v);
	});
	if constexpr (HAF)
	{
		// (half-array form of the forward side, the exchange behind the first pass -- across the workgroup: real parts written |
		// read | imaginary parts written | read, the last in the wave-local steps' first one)
		ex.phase([&](int tid, St& st) { cp_ha_ld<256>(buf_of(tid), st.er, lt_of(tid)); });
		ex.phase([&](int tid, St& st) { cp_ha_st<G::N>(buf_of(tid), st.vi, lt_of(tid)); });
	}
	// forward passes 1 .., the middle pass and the first backward pass stay inside each wave's own range
	// of the array (ConvpGeom): wave-level ordering points instead of workgroup barriers between them
	auto s_pre1 = [&](int tid, St& st)
	{
		const int lt = lt_of(tid);
		ConvpPre<LN, UL, 1>::template run<HA>(buf_of(tid), st, lt, ltw);
		if constexpr (G::NPRE > 2) ConvpPre<LN, UL, 2>::prefetch(L, st, lt);
		else hp_prefetch(st, lt);
	};
	auto s_pre2 = [&](int tid, St& st)
	{
		const int lt = lt_of(tid);
		ConvpPre<LN, UL, 2>::template run<HA>(buf_of(tid), st, lt, ltw);
		hp_prefetch(st, lt);
	};
	// (two steps: every lane has read its forward data before any lane's backward data overwrites it --
	// on the GPU program order alone guarantees that, LDS serves a wave's accesses in issue order)
	if constexpr (G::POST)
	{
		// decimating form / 8192 points: steps a geometry does not have are empty
		auto d_pre1 = [&](int tid, St& st)
		{
			if constexpr (G::NPRE > 1) s_pre1(tid, st);
		};
		auto d_pre2 = [&](int tid, St& st)
		{
			if constexpr (G::NPRE > 2) s_pre2(tid, st);
		};
		auto d_midc = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			if constexpr (SP) cp_sp_middle<LN, UL, CXL>(L, buf_of(tid), st, lt);
			else if constexpr (UL < 0) cp_middle_compute_down<LN, UL, CX>(buf_of(tid), st, lt);
			else cp_middle_compute<LN, UL, CX, HA>(buf_of(tid), st, lt);
			ConvpPost<LN, UL, 1>::prefetch(L, st, lt);
		};
		auto d_midw = [&](int tid, St& st)
		{
			if constexpr (UL < 0) cp_middle_write_down<LN, UL, HA>(buf_of(tid), st, lt_of(tid));
			else cp_middle_write<LN, UL>(buf_of(tid), st, lt_of(tid));
		};
		auto d_post1 = [&](int tid, St& st)
		{
			if constexpr (G::NPOST > 1)
			{
				const int lt = lt_of(tid);
				ConvpPost<LN, UL, 1>::template run<HA>(buf_of(tid), st, lt);
				ConvpPost<LN, UL, 2>::prefetch(L, st, lt);
			}
		};
		auto d_post2 = [&](int tid, St& st)
		{
			if constexpr (G::NPOST > 2)
			{
				const int lt = lt_of(tid);
				ConvpPost<LN, UL, 2>::template run<HA>(buf_of(tid), st, lt);
				ConvpPost<LN, UL, 3>::prefetch(L, st, lt);
			}
		};
		auto d_post3 = [&](int tid, St& st)
		{
			if constexpr (G::NPOST > 3)
			{
				const int lt = lt_of(tid);
				ConvpPost<LN, UL, 3>::run(buf_of(tid), st, lt);
				ConvpPost<LN, UL, 4>::prefetch(L, st, lt);
			}
		};
		auto d_post4 = [&](int tid, St& st)
		{
			if constexpr (G::NPOST > 4)
			{
				const int lt = lt_of(tid);
				ConvpPost<LN, UL, 4>::run(buf_of(tid), st, lt);
				ConvpPost<LN, UL, 5>::prefetch(L, st, lt);
			}
		};
		static_assert(G::NPRE <= 3 && G::NPOST >= 1 && G::NPOST <= 5, "pair kernel, decimating / 8192 points: pass plan");
		// (every pass but the last stays inside a wave's range when its sub-length N2 / E2 <= 64 E2; the one
		// geometry where the last but one does not -- 8192 points decimated by 4 -- takes a barrier more)
		static_assert(G::NPOST < 5 || G::N2 / G::E2 > 64 * G::E2, "pass plan");
		static_assert(G::NPOST == 5 || G::NW == 1 || G::N2 / G::E2 <= 64 * G::E2, "pass plan");
		if constexpr (SOLO)
		{
			// one-channel form: the spectrum through the array for the partner bins, a barrier either side of their reads
			static_assert(G::NPOST == 3 || G::NPOST == 5, "one-channel form: pass plans of the 8192-point geometries");
			auto q_mida = [&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				cp_solo_mid_a<LN, UL>(buf_of(tid), st, lt);
				cp_sp_hp_prefetch<LN, UL>(L, st, lt);
			};
			ex.wave_steps(d_pre1, d_pre2, q_mida);
			ex.phase([&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				if constexpr (UL < 0) cp_solo_mid_b_down<LN, UL, CXL>(L, buf_of(tid), st, lt);
				else cp_solo_mid_b<LN, UL, CXL>(L, buf_of(tid), st, lt);
				ConvpPost<LN, UL, 1>::prefetch(L, st, lt);
			});
			ex.wave_steps(d_midw, d_post1, d_post2, d_post3);
		}
		else if constexpr (HA && UL < 0)
		{
			// decimating half-array form: the forward side's second exchange (sub-length 256 -> the thread's sixteen
			// consecutive positions) by parts inside the wave, then the backward side as it is -- its 2048 complex values
			// in the wave's own part of the same 32 KB (identity slot maps)
			static_assert(G::NPRE == 2 && G::NPOST == 3, "decimating half-array form: pass plan 16 x 16 x 16 | 4 x 8 x 8 x 8");
			auto f1 = [&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				cp_ha_ld<256>(buf_of(tid), st.vi, lt);
				pdif_arith<16, true>(st.tw, st.er, st.vi);
				cp_ha_st<256>(buf_of(tid), st.er, lt);
				hp_prefetch(st, lt);
			};
			auto f2 = [&](int tid, St& st) { cp_ha_ld<16>(buf_of(tid), st.er, lt_of(tid)); };
			auto f3 = [&](int tid, St& st) { cp_ha_st<256>(buf_of(tid), st.vi, lt_of(tid)); };
			auto f4 = [&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				double zi[16];
				cp_ha_ld<16>(buf_of(tid), zi, lt);
				cp_middle_down_arith<LN, UL, CX>(st, st.er, zi, lt);
				ConvpPost<LN, UL, 1>::prefetch(L, st, lt);
			};
			ex.wave_steps(f1, f2, f3, f4, d_midw, d_post1, d_post2);
		}
		else if constexpr (HA)
		{
			// half-array form on 8192 points: the three exchanges of the backward side by parts through an array of doubles.
			// Middle -> sub-length 32 -> sub-length 512 inside a wave (program order); sub-length 512 -> 8192 across the
			// workgroup: written | barrier | read | barrier | written | barrier | read (the body's last phase).
			constexpr int N1 = ConvpPost<LN, UL, 1>::n, N2P = ConvpPost<LN, UL, 2>::n;
			static_assert(N1 == 32 && N2P == 512 && ConvpPost<LN, UL, 3>::n == G::N2, "half-array form: pass plan 2 x 16 x 16 x 16");
			auto g1 = [&](int tid, St& st) { cp_ha_st<16>(buf_of(tid), st.vr, lt_of(tid)); };
			auto g2 = [&](int tid, St& st) { cp_ha_ld<N1>(buf_of(tid), st.er, lt_of(tid)); };
			auto g3 = [&](int tid, St& st) { cp_ha_st<16>(buf_of(tid), st.vi, lt_of(tid)); };
			auto g4 = [&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				cp_ha_ld<N1>(buf_of(tid), st.vi, lt);
				pdit_arith<16, true>(st.tw, st.er, st.vi);
				cp_ha_st<N1>(buf_of(tid), st.er, lt);
				ConvpPost<LN, UL, 2>::prefetch(L, st, lt);
			};
			auto g5 = [&](int tid, St& st) { cp_ha_ld<N2P>(buf_of(tid), st.vr, lt_of(tid)); };
			auto g6 = [&](int tid, St& st) { cp_ha_st<N1>(buf_of(tid), st.vi, lt_of(tid)); };
			auto g7 = [&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				cp_ha_ld<N2P>(buf_of(tid), st.vi, lt);
				pdit_arith<16, true>(st.tw, st.vr, st.vi);
				cp_ha_st<N2P>(buf_of(tid), st.vr, lt);
			};
			ex.wave_steps(d_pre1, d_pre2, d_midc, g1, g2, g3, g4, g5, g6, g7);
			ex.phase([&](int tid, St& st) { cp_ha_ld<G::N2>(buf_of(tid), st.vr, lt_of(tid)); });
			// (the last pass's twiddles requested here, behind the imaginary parts whose registers they take: held from the
			// wave-local steps on, across three barriers, they were spilled -- the budget is 128 registers)
			ex.phase([&](int tid, St& st)
			{
				cp_ha_st<N2P>(buf_of(tid), st.vi, lt_of(tid));
				ConvpPost<LN, UL, 3>::prefetch(L, st, lt_of(tid));
			});
		}
		else
		ex.wave_steps(d_pre1, d_pre2, d_midc, d_midw, d_post1, d_post2, d_post3);
		if constexpr (G::NPOST > 4) ex.wave_steps(d_post4);
		if constexpr (SP)
		{
			// split 2x up-sampling form: the even half's last pass (across the waves), then the odd half through the same
			// passes; the even outputs wait in st.er / st.ei
			static_assert(G::NPOST == 3, "split 2x up-sampling form: pass plan of the 8192-point geometry");
			ex.phase([&](int tid, St& st)
			{
				const int lt = lt_of(tid);
				ConvpPost<LN, UL, G::NPOST>::run(buf_of(tid), st, lt);
				ConvpPost<LN, UL, 1>::prefetch(L, st, lt);
			});
			auto o_mid = [&](int, St& st) { cp_sp_swap<LN, UL>(st); };
			ex.wave_steps(o_mid, d_midw, d_post1, d_post2);
		}
	}
	else
	{
	auto s_midc = [&](int tid, St& st)
	{
		const int lt = lt_of(tid);
		cp_middle_compute<LN, UL, CX, HA>(buf_of(tid), st, lt);
		if constexpr (G::B1 && TL::ON) {} // (the pass fetches its twiddles from LDS itself)
		else if constexpr (G::B1) ptw_fetch<16, G::NT, (16 < G::NT ? 16 : G::NT)>(st.tw, L.ptw, 3, lt);
		else cp_back2_prefetch<LN, UL>(L, st, lt);
	};
	auto s_midw = [&](int tid, St& st) { cp_middle_write<LN, UL>(buf_of(tid), st, lt_of(tid)); };
	auto s_b1 = [&](int tid, St& st)
	{
		const int lt = lt_of(tid);
		cp_back1<LN, UL>(buf_of(tid), st, lt, ltw);
		if constexpr (WALK)
		{
			st.tw[0] = st.twp[2];
			R8B_OPAQUE2(st.tw[0].re, st.tw[0].im);
			if constexpr (G::R2 >= 16)
			{
				st.tw[3] = st.twp[3];
				R8B_OPAQUE2(st.tw[3].re, st.tw[3].im);
			}
		}
		else cp_back2_prefetch<LN, UL>(L, st, lt);
	};
	static_assert(G::NPRE >= 1 && G::NPRE <= 3, "pair kernel: one to three forward passes before the middle");
#ifdef R8B_SPLIT_UP2
	if constexpr (SPLIT)
	{
		static_assert(G::NPRE == 3, "split form: 2048 -> 4096 points");
		// even half: middle + radix 8 in registers, sub-lengths 64 and 512 inside the wave, last pass across the waves
		auto p_mid = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			cp_split_middle<LN, UL>(L, buf_of(tid), st, lt);
			cp_split_first<LN, UL>(buf_of(tid), st.vr, st.vi, lt);
			tw_fetch<8>(st.tw, L.tw, L.tw_len, 64, lt & 7);
		};
		auto p_b = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			cp_split_pass<64>(buf_of(tid), st.tw, lt);
			tw_fetch<8>(st.tw, L.tw, L.tw_len, 512, lt & 63);
		};
		auto p_c = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			cp_split_pass<512>(buf_of(tid), st.tw, lt);
			tw_fetch<4>(st.tw, L.tw, L.tw_len, 2048, lt);
			tw_fetch<4>(st.tw + 3, L.tw, L.tw_len, 2048, lt + 256);
		};
		ex.wave_steps(s_pre1, s_pre2, p_mid, p_b, p_c);
		ex.phase([&](int tid, St& st) { cp_split_last<LN, UL>(buf_of(tid), st.tw, st.vr, st.vi, lt_of(tid)); });
		// odd half
		auto q_first = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			cp_split_first<LN, UL>(buf_of(tid), st.vr + 8, st.vi + 8, lt);
			tw_fetch<8>(st.tw, L.tw, L.tw_len, 64, lt & 7);
		};
		ex.wave_steps(q_first, p_b, p_c);
	}
	else
#endif
	if constexpr (P3)
	{
		static_assert(!P3 || (G::NPRE == 2 && G::B1 && G::R2 > 1), "polyphase 3x form: 1024 ... 4096-point 1:1 geometries");
		// the spectrum once, then the three components one after the other through the same backward passes; the first
		// two components' outputs wait in registers (st.er / st.ei, st.p3r / st.p3i)
		auto p_mid0 = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			cp_p3_spectrum<LN, UL>(buf_of(tid), st, lt);
			cp_p3_product<LN, UL>(L, st, lt, 0);
		};
		auto p_mid1 = [&](int tid, St& st) { cp_p3_product<LN, UL>(L, st, lt_of(tid), 1); };
		auto p_mid2 = [&](int tid, St& st) { cp_p3_product<LN, UL>(L, st, lt_of(tid), 2); };
		ex.wave_steps(s_pre1, p_mid0, s_midw, s_b1);
		auto p_rest = [&](int tid, St& st, int r)
		{
			// (the positions that do not wait for the other components: rare -- cp_p3_store_rest)
			cp_silence<LN, UL>(st, ex.collect_bits());
			if (live(tid))
			{
				DstView pd = L.dst;
				long long pend = L.b;
				cp_park_view(ex, XM, st, k_of(tid), cur, pd, pend);
				cp_p3_store_rest<LN, UL>(L, st, r, k_of(tid), chA, chB, bvalid, lt_of(tid), pd, pend);
			}
		};
		ex.phase([&](int tid, St& st)
		{
			cp_back2<LN, UL>(buf_of(tid), st, lt_of(tid));
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid));
			if (L.in_len / 3 > G::NT * kP3Keep) p_rest(tid, st, 0);
#pragma unroll
			for (int p = 0; p < kP3Keep; p++)
			{
				st.er[p] = st.vr[p];
				st.ei[p] = st.vi[p];
			}
		});
		ex.wave_steps(p_mid1, s_midw, s_b1);
		ex.phase([&](int tid, St& st)
		{
			cp_back2<LN, UL>(buf_of(tid), st, lt_of(tid));
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid));
			if (L.in_len / 3 > G::NT * kP3Keep) p_rest(tid, st, 1);
#pragma unroll
			for (int p = 0; p < kP3Keep; p++)
			{
				st.p3r[p] = st.vr[p];
				st.p3i[p] = st.vi[p];
			}
		});
		ex.wave_steps(p_mid2, s_midw, s_b1);
		ex.each([&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			slices_out(tid, st);
			cp_back2<LN, UL>(buf_of(tid), st, lt);
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid)); // (the first two components' outputs: where they were computed)
			const unsigned nzb = ex.collect_bits();
			cp_silence<LN, UL>(st, nzb);
			if (nzb != 3u)
			{
				// (the first two components' outputs too)
#pragma unroll
				for (int p = 0; p < kP3Keep; p++)
				{
					if (!(nzb & 1u)) st.er[p] = st.p3r[p] = 0.0;
					if (!(nzb & 2u)) st.ei[p] = st.p3i[p] = 0.0;
				}
			}
			if (live(tid))
			{
				DstView pd = L.dst;
				long long pend = L.b;
				cp_park_view(ex, XM, st, k_of(tid), cur, pd, pend);
				cp_p3_store<LN, UL>(L, st, k_of(tid), chA, chB, bvalid, lt, pd, pend);
				cp_p3_store_rest<LN, UL>(L, st, 2, k_of(tid), chA, chB, bvalid, lt, pd, pend);
			}
		});
	}
	else
	if constexpr (HA)
	{
		// (1:1 geometry: the forward side's second exchange by parts in front of them -- sub-length 256 -> the thread's sixteen
		// consecutive positions, inside the wave --, then the middle pass's arithmetic on the values in registers)
		auto f1 = [&](int tid, St& st)
		{
			if constexpr (HAF)
			{
				const int lt = lt_of(tid);
				cd twr[TL::NBF];
				twl_fetch<TL::NBF, TL::JM1>(twr, ltw, TL::O1, lt);
				cp_ha_ld<256>(buf_of(tid), st.vi, lt);
				pdif_arith<16, true>(twr, st.er, st.vi);
				cp_ha_st<256>(buf_of(tid), st.er, lt);
				hp_prefetch(st, lt);
			}
		};
		auto f2 = [&](int tid, St& st) { if constexpr (HAF) cp_ha_ld<16>(buf_of(tid), st.er, lt_of(tid)); };
		auto f3 = [&](int tid, St& st) { if constexpr (HAF) cp_ha_st<256>(buf_of(tid), st.vi, lt_of(tid)); };
		auto f4 = [&](int tid, St& st)
		{
			if constexpr (HAF)
			{
				static_assert(!HAF || (UL == 0 && !CX && G::RM == 16), "1:1 half-array form: a real kernel spectrum, radix 16 in the middle");
				double zi[16];
				cp_ha_ld<16>(buf_of(tid), zi, lt_of(tid));
				dif_regs<16>(st.er, zi);
#pragma unroll
				for (int c = 0; c < 8; c++)
				{
					st.vr[2 * c] = st.er[2 * c] * st.hp[c].re;
					st.vi[2 * c] = zi[2 * c] * st.hp[c].re;
					st.vr[2 * c + 1] = st.er[2 * c + 1] * st.hp[c].im;
					st.vi[2 * c + 1] = zi[2 * c + 1] * st.hp[c].im;
				}
				dit_regs<16>(st.vr, st.vi);
			}
		};
		// half-array form: the two exchanges of the backward side by parts -- real parts, then imaginary parts -- through
		// the array of doubles.  The first stays inside sixteen lanes of a wave: program order (and the steps' ordering
		// points) is all it needs; the second crosses the workgroup: written | barrier | read | barrier | written |
		// barrier | read (the last read opens the body's last phase, below).
		auto h_a1 = [&](int tid, St& st) { cp_ha_st_mid(buf_of(tid), st.vr, lt_of(tid)); };
		auto h_a2 = [&](int tid, St& st) { cp_ha_ld_b1(buf_of(tid), st.er, lt_of(tid)); };
		auto h_a3 = [&](int tid, St& st) { cp_ha_st_mid(buf_of(tid), st.vi, lt_of(tid)); };
		auto h_a4 = [&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			cd twl[6];
			twl_fetch<6, TL::JM3>(twl, ltw, TL::O3, lt);
			cp_ha_ld_b1(buf_of(tid), st.vi, lt);
			pdit_arith<16, true>(twl, st.er, st.vi);
			cp_ha_st_b1(buf_of(tid), st.er, lt);
			cp_back2_prefetch<LN, UL>(L, st, lt);
		};
		if constexpr (HAF) ex.wave_steps(f1, f2, f3, f4, h_a1, h_a2, h_a3, h_a4);
		else
		ex.wave_steps(s_pre1, s_pre2, s_midc, h_a1, h_a2, h_a3, h_a4);
		ex.phase([&](int tid, St& st) { cp_ha_ld_b2<G::NT>(buf_of(tid), st.vr, lt_of(tid)); });
		ex.phase([&](int tid, St& st) { cp_ha_st_b1(buf_of(tid), st.vi, lt_of(tid)); });
	}
	else
	if constexpr (G::NPRE == 3) ex.wave_steps(s_pre1, s_pre2, s_midc, s_midw, s_b1);
	else if constexpr (G::NPRE == 2 && G::B1) ex.wave_steps(s_pre1, s_midc, s_midw, s_b1);
	else if constexpr (G::NPRE == 2) ex.wave_steps(s_pre1, s_midc, s_midw);
	else if constexpr (G::B1) ex.wave_steps(s_midc, s_midw, s_b1);
	else ex.wave_steps(s_midc, s_midw);
	}
	// History for the next call (stage 0 only): the tail of the caller's buffers goes into the other history ring.
	// Calls whose blocks read the caller's fp64 buffer (tail_flags & 2, convp_tail_owners): the blocks that hold the tail
	// in registers stored it in the first phase (cp_tail_owned: no load at all), the launch's last block fetched what lies
	// behind its window beside its samples (cp_tail_rest); left for this place is only history older than the caller's
	// buffer -- short calls --, with the first block.  Otherwise (tail_flags & 1: PCM input, 3x zero stuffing, calls
	// shorter than a window) every workgroup of a channel pair copies its share of the tail here, block b of the
	// launch's nblk the b-th slice, eight samples per channel in flight per thread.  (History of this code, all
	// measured, DESIGN.md section 5: the first block copying the whole tail with one load in flight per thread took
	// 40 000 cycles on top of its 31 000 -- tools/timeline_probe.py --; slices for every block, here: cfg3 -18 %, but
	// a round trip of 3 500-5 500 cycles per workgroup wherever the loads were put; from the registers: cfg2 -3.5 %,
	// cfg3 -6 % again, the last block 5 000 cycles longer than the others -- the samples it fetches for the ring miss
	// the caches like its own, and a CU's L1 keeps only so many misses in flight.)
	if (!WALK && (L.tail_flags & 5) != 0)
	{
		const unsigned long long tn = (unsigned long long) (L.tail_p1 - L.tail_p0), nb = (unsigned long long) L.nblk;
		const unsigned long long bi = (unsigned long long) (cur.k - L.k0), be = bi + (unsigned) (G::SUB == 1 ? 1 : cur.nvalid);
		// (64-bit products: a fused launch has at most kConvxMaxBlocks blocks, an unfused one -- launch_stage -- as many
		// as the call holds)
		long long s0 = L.tail_p0 + (long long) (tn * bi / nb), s1 = L.tail_p0 + (long long) (tn * be / nb);
		if ((L.tail_flags & 2) != 0)
		{
			// (what the blocks' registers did not hold -- cp_tail_owned -- and the last block did not fetch beside its
			// samples: history older than the caller's buffer, with the launch's first block)
			const bool first = bi == 0;
			s0 = L.tail_p0;
			s1 = first ? L.tail_c0 : L.tail_p0;
		}
		ex.each([&](int tid, St&)
		{
			constexpr int TB = 8;
			// (positions s0 + rel through the block form of the view -- uniform row pointers, a select per load, no
			// branch: through src_load every load sat behind its own ring / buffer / zero branches and the sixteen of a
			// thread went out one round trip after the other, 46 000 cycles for a call of one block per channel pair)
			const SrcBlock sba = src_block(L.src, chA, s0), sbb = src_block(L.src, chB, s0);
			for (long long i0 = s0 + tid; i0 < s1; i0 += (long long) TB * G::WT)
			{
				double va[TB], vb[TB];
#pragma unroll
				for (int j = 0; j < TB; j++)
				{
					const long long i = i0 + (long long) j * G::WT;
					// (lanes past the end load a clamped, valid position and store nothing)
					const int rel = (int) ((i < s1 ? i : s1 - 1) - s0);
					va[j] = src_block_load1(sba, rel);
					vb[j] = src_block_load1(sbb, rel);
				}
#pragma unroll
				for (int j = 0; j < TB; j++)
				{
					const long long i = i0 + (long long) j * G::WT;
					if (i < s1)
					{
						L.tail_ring[(long long) chA * L.src.ring_stride + (i & L.src.ring_mask)] = va[j];
						if (bvalid) L.tail_ring[(long long) chB * L.src.ring_stride + (i & L.src.ring_mask)] = vb[j];
					}
				}
			}
		});
	}
#ifdef R8B_SPLIT_UP2
	if constexpr (SPLIT)
	{
		ex.each([&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			slices_out(tid, st);
			cp_split_last<LN, UL>(buf_of(tid), st.tw, st.vr + 8, st.vi + 8, lt);
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid));
			cp_silence<LN, UL>(st, ex.collect_bits());
			if (live(tid))
			{
				DstView pd = L.dst;
				long long pend = L.b;
				cp_park_view(ex, XM, st, k_of(tid), cur, pd, pend);
				cp_split_store<LN, UL>(L, st, k_of(tid), chA, chB, bvalid, lt, pd, pend);
			}
		});
	}
	else
#endif
	if constexpr (G::POST && (BM == 0 || BM == 3))
	{
		ex.each([&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			slices_out(tid, st);
			if constexpr (HA && UL > 0)
			{
				// (the imaginary parts of the last pass's elements; the real parts wait in st.vr)
				cp_ha_ld<G::N2>(buf_of(tid), st.vi, lt);
				pdit_arith<16, true>(st.tw, st.vr, st.vi);
			}
			else
			ConvpPost<LN, UL, G::NPOST>::template run<HA>(buf_of(tid), st, lt);
			if constexpr (LEVELS)
			{
				const int lsh = level_shift(tid);
				cp_scale_out<16>(st.vr, st.vi, lsh);
				if constexpr (SP) cp_scale_out<16>(st.er, st.ei, lsh); // (the even half's outputs too)
			}
			unsigned nzb = ex.collect_bits();
			// (one-channel form: the element's two parts are one channel's samples)
			if constexpr (SOLO) nzb = nzb != 0 ? 3u : 0u;
			cp_silence<LN, UL>(st, nzb);
			if constexpr (SP)
			{
				// (the even half's outputs too)
				if (nzb != 3u)
				{
#pragma unroll
					for (int p = 0; p < 16; p++)
					{
						if (!(nzb & 1u)) st.er[p] = 0.0;
						if (!(nzb & 2u)) st.ei[p] = 0.0;
					}
				}
			}
			if (live(tid))
			{
				DstView pd = L.dst;
				long long pend = L.b;
				cp_park_view(ex, XM, st, k_of(tid), cur, pd, pend);
				if constexpr (SOLO && UL < 0) cp_solo_store_down<LN, UL>(L, st, k_of(tid), chA, lt, pd, pend);
				else if constexpr (SOLO)
					cp_sp_store<LN, UL, BM>(L, st.vr, st.vi, st.vr, st.vi, k_of(tid), chA, chB, false, lt, pd, pend);
				else if constexpr (SP)
					cp_sp_store<LN, UL, BM>(L, st.er, st.vr, st.ei, st.vi, k_of(tid), chA, chB, bvalid, lt, pd, pend);
				else if constexpr (UL < 0) cp_store_conv_down<LN, UL>(L, st, k_of(tid), chA, chB, bvalid, lt, pd, pend);
				else cp_store_conv<LN, UL, BM>(L, st, k_of(tid), chA, chB, bvalid, lt, pd, pend);
			}
		});
	}
	else if constexpr (P3) {} // (stored component by component, above)
	else if constexpr (BM == 0 || BM == 3)
	{
		ex.each([&](int tid, St& st)
		{
			const int lt = lt_of(tid);
			slices_out(tid, st);
			if constexpr (HA)
			{
				// (the imaginary parts of the last pass's elements; the real parts wait in st.vr)
				cp_ha_ld_b2<G::NT>(buf_of(tid), st.vi, lt);
				tw_expand<16>(st.tw);
				pdit_arith<16, true>(st.tw, st.vr, st.vi);
			}
			else
			cp_back2<LN, UL>(buf_of(tid), st, lt);
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid));
			cp_silence<LN, UL>(st, ex.collect_bits());
			if (live(tid))
			{
				DstView pd = L.dst;
				long long pend = L.b;
				cp_park_view(ex, XM, st, k_of(tid), cur, pd, pend);
				cp_store_conv<LN, UL, BM>(L, st, k_of(tid), chA, chB, bvalid, lt, pd, pend);
			}
		});
	}
	else if constexpr (BM == 4 || BM == 5)
	{
		constexpr int T2 = BM == 4 ? 25 : 27;
		static_assert(UL >= 0, "the decimating form has no fused interpolator");
		ex.phase([&](int tid, St& st)
		{
			if constexpr (HA)
			{
				// (the imaginary parts of the last pass's elements; the real parts wait in st.vr)
				cp_ha_ld_b2<G::NT>(buf_of(tid), st.vi, lt_of(tid));
				tw_expand<16>(st.tw);
				pdit_arith<16, true>(st.tw, st.vr, st.vi);
			}
			else
			if constexpr (G::POST) ConvpPost<LN, UL, G::NPOST>::run(buf_of(tid), st, lt_of(tid));
			else cp_back2<LN, UL>(buf_of(tid), st, lt_of(tid));
			if constexpr (!WALK && !HA) cp_rows2_fetch<T2>(X, st.rows2, st.pt);
		});
		ex.phase([&](int tid, St& st)
		{
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid));
			cp_silence<LN, UL>(st, ex.collect_bits());
			cp_final_store<LN, UL, HA>(L, buf_of(tid), buf_of(tid) + X.run_off, st, k_of(tid), lt_of(tid));
			// (half-array form: the rows behind the results, whose registers they take)
			if constexpr (HA) cp_rows2_fetch<T2>(X, st.rows2, st.pt);
		});
		// the interpolator: all 256 threads over the run of one block pair after the other
		ex.each([&](int tid, St& st)
		{
			(void) tid;
			slices_out(tid, st);
			if constexpr (WALK)
			{
				// (an interior block: whole output groups only -- convp_walk_range)
				cp_whole2_compute<T2, true>(X, X.wdst, XM.blk[cur.k - L.k0], buf, st.rows2, st.pt, chA, chB, bvalid);
				return;
			}
			const int nv = G::SUB == 1 ? 1 : cur.nvalid;
#ifndef R8B_HA_NBUF
#define R8B_HA_NBUF 2 // (two chunks of the window in flight in the whole-group loop: -0.9 % on cfg2, 167 registers; 1: 152)
#endif
			constexpr int NBUF = HA ? R8B_HA_NBUF : 2;
			for (int sb = 0; sb < nv; sb++)
				cp_whole2_compute<T2, false, NBUF, (HA ? 1 : 2), HA>(X, X.wdst, XM.blk[cur.k + sb - L.k0], buf + sb * G::NA, st.rows2, st.pt, chA, chB, bvalid);
			if (ex.uniform(st.pf) != 0)
			{
				// (the call's last block: its outputs behind the call's range belong to the next call -- parked, not
				// computed again there)
				DstView pd;
				pd.p = XM.park_dst;
				pd.stride = XM.park_stride;
				pd.mask = -1;
				pd.off = -XM.wb;
				pd.fmt = kPcmF64;
				cp_whole2_compute<T2, false, NBUF, (HA ? 1 : 2), HA>(X, pd, XM.park_blk, buf + (nv - 1) * G::NA, st.rows2, st.pt, chA, chB, bvalid);
			}
		});
	}
	else
	{
		static_assert(UL >= 0, "the decimating form has no fused interpolator");
		ex.phase([&](int tid, St& st)
		{
			if constexpr (G::POST) ConvpPost<LN, UL, G::NPOST>::run(buf_of(tid), st, lt_of(tid));
			else cp_back2<LN, UL>(buf_of(tid), st, lt_of(tid));
			cx_whole_row<FLENP>(X, st.row, cp_whole_phase(X, tid));
		});
		ex.phase([&](int tid, St& st)
		{
			if constexpr (SOLO)
			{
				// (one-channel form: the element's two parts are one channel's samples)
				cp_silence<LN, UL>(st, ex.collect_bits() != 0 ? 3u : 0u);
				cp_solo_final_store<LN, UL>(L, reinterpret_cast<double*>(buf), st, k_of(tid), lt_of(tid));
			}
			else
			{
			cp_scale_out<16>(st.vr, st.vi, level_shift(tid));
			cp_silence<LN, UL>(st, ex.collect_bits());
			cp_final_store<LN, UL>(L, buf_of(tid), buf_of(tid), st, k_of(tid), lt_of(tid));
			}
		});
		ex.each([&](int tid, St& st)
		{
			int row_t = cp_whole_phase(X, tid); // (what cx_whole_row fetched ahead)
			if constexpr (SOLO)
			{
				cp_whole_compute_solo<FLENP>(X, X.wdst, XM.blk[cur.k - L.k0], reinterpret_cast<const double*>(buf), st.row, &row_t, chA, tid, G::WT);
				if (ex.uniform(st.pf) != 0)
				{
					// (the call's last block: its outputs behind the call's range belong to the next call -- parked, not
					// computed again there; cf. modes 4 / 5)
					DstView pd;
					pd.p = XM.park_dst;
					pd.stride = XM.park_stride;
					pd.mask = -1;
					pd.off = -XM.wb;
					pd.fmt = kPcmF64;
					cp_whole_compute_solo<FLENP>(X, pd, XM.park_blk, reinterpret_cast<const double*>(buf), st.row, &row_t, chA, tid, G::WT);
				}
			}
			else
			for (int sb = 0; sb < (G::SUB == 1 ? 1 : cur.nvalid); sb++)
				cp_whole_compute<FLENP>(X, XM.blk[cur.k + sb - L.k0], buf + sb * G::NA, st.row, &row_t, chA, chB, bvalid, tid, G::WT);
		});
	}
}

template<int LN, int UL, int MODE, int FLENP, class Exec>
R8B_HD void convp_body(Exec& ex, const ConvxLaunch& X, cd* buf, const ConvpItem& cur)
{
	convp_body<LN, UL, MODE, FLENP>(ex, X, X, buf, cur);
}

// Walk form: blocks cur.k ... cur.k + nit - 1 of the channel pair by one workgroup (ConvpWalk).  The prologue fetches
// what every block needs once; results are those of the one-block form bit for bit (same arithmetic on the same values,
// tests: option walk = 0 / 1).
template<int LN, int UL, int MODE, int FLENP, class Exec>
R8B_HD void convp_walk(Exec& ex, const ConvxLaunch& X, const ConvxLaunch& XM, cd* buf, ConvpItem cur, int nit)
{
	typedef ConvpGeom<LN, UL> G;
	typedef ConvpState<LN, UL> St;
	typedef ConvpTwLds<LN, UL> TL;
	constexpr int BM = MODE == 16 ? 4 : (MODE == 17 ? 5 : MODE);
	constexpr int T2 = BM == 4 ? 25 : 27;
	const ConvLaunch& L = X.c;
	cd* const ltw = reinterpret_cast<cd*>(reinterpret_cast<unsigned char*>(buf) + convp_array_bytes<LN, UL>() + kConvpFlagBytes);
	ex.each([&](int tid, St& st)
	{
		const int lt = convp_lt<LN, UL>(tid);
		cd t0[6], t4[6];
		ptw_fetch_lean<G::E1, G::NT>(t0, L.ptw, 0, lt);
		ptw_fetch_lean<G::R2, G::NT>(t4, L.ptw, 4, lt);
		cd twl_v;
		twl_v.re = twl_v.im = 0.0;
		if (tid < TL::NE) twl_v = L.ptw[TL::src_index(tid)];
		st.pt = cp_ptab_fetch(X, tid);
		cp_rows2_fetch<T2>(X, st.rows2, st.pt);
		st.twp[0] = t0[0];
		st.twp[1] = t0[3];
		st.twp[2] = t4[0];
		st.twp[3] = t4[3];
		if (tid < TL::NE) twl_st(ltw, tid, twl_v);
	});
	for (int it = 0; it < nit; it++)
	{
		ConvpWalk w;
		w.more = it + 1 < nit;
		ex.next_block();
		convp_body<LN, UL, MODE, FLENP, true>(ex, X, XM, buf, cur, w);
		cur.k++;
	}
}

// What a launcher (r8b_kernels.hip, tests/emul) sets in its copy of the descriptor before the kernel runs: the
// rotation of the block inside the circular array (ConvLaunch::rot / fl2r).  Rotating the input by rot samples
// rotates the circular convolution's output by up * rot, so with rot = -(fl2 / up) mod N the valid outputs start at
// circular position fl2 mod up instead of fl2.  Power-of-two zero stuffing only (the 3x forms and the decimating
// form keep rot = 0, fl2r = fl2).
// History tail from the registers (cp_tail_owned): which blocks of the launch put which part of [tail_p0, tail_p1)
// into the ring.  Block k's window is the N samples ending at base(k) + iln (cp_load); blocks kf .. read it from the
// caller's fp64 buffer or the history ring in front of it, and consecutive windows start at most iln apart, so position i of
// [max(tail_p0, window start of kf), min(tail_p1, window end of the last block)) lies in the window of the block with
// base(k) <= i < base(k + 1) (kf below its base, the last block above the next base).
template<int N, int US>
inline void convp_tail_owners(ConvLaunch& L)
{
	if (L.tail_ring == nullptr || L.src.cur_fmt != kPcmF64 || L.blk_stride > L.in_len || L.tail_p1 <= L.tail_p0) return;
	const long long iln = L.in_len >> US, hist = N - iln;
	auto base = [&](long long k) { return (k * (long long) L.blk_stride + L.blk_offset) >> US; };
	const long long klast = L.k0 + L.nblk - 1;
	// (a block's registers hold its whole window wherever the samples came from -- the caller's buffer or, in front of
	// it, the history ring: the launch's first block may own its part like any other.  Until round 4 the first owner
	// was the first block whose window lay inside the caller's buffer, and a call of ONE block per channel pair copied
	// its whole tail through the loop at the end of the kernel: 34 000 - 46 000 cycles on a block of 45 000.)
	long long kf = L.k0;
	// (what lies behind the last window is read from the caller's buffer: cp_tail_rest, cp_tail_slice_load)
	if (base(klast) + iln < (L.src.cur_base > 0 ? L.src.cur_base : 0)) return;
	long long c0 = base(kf) - hist, c1 = base(klast) + iln;
	c0 = c0 < L.tail_p0 ? L.tail_p0 : (c0 > L.tail_p1 ? L.tail_p1 : c0);
	c1 = c1 > L.tail_p1 ? L.tail_p1 : (c1 < c0 ? c0 : c1);
	// (the first block that has something to do: the one whose part of the stream reaches past c0, at the latest the
	// last one -- it fetches what lies behind c1)
	while (kf < klast && base(kf + 1) <= c0) kf++;
	L.tail_flags = 2 | (L.tail_p0 < c0 ? 4 : 0);
	L.tail_bf = (int) (kf - L.k0);
	L.tail_c0 = c0;
	L.tail_c1 = c1;
}

// (slices: the kernel shares the parked outputs' copy-back and the history tail's rest among the pair's workgroups --
// every mode but 1)
// (spu: the split 2x up-sampling form on a 1:1 geometry -- modes 8 / 9)
// (solo: the one-channel form on that geometry -- modes 10 / 11: the window is 2N samples, rotated by pairs)
template<int LN, int UL>
inline void convp_prepare(ConvxLaunch& X, bool slices = true, bool spu = false, bool solo = false, bool p3 = false)
{
	// (p3 -- polyphase 3x form, mode 19: rot carries the components' reach into the past, set by the engine; the
	// history tail is copied in slices)
	if (!p3) X.c.rot = 0;
	X.c.fl2r = X.c.fl2;
	X.c.tail_flags = X.c.tail_ring != nullptr ? 1 : 0;
	X.c.tail_bf = 0;
	X.c.tail_c0 = X.c.tail_c1 = 0;
	if (p3) {}
	else if (solo && UL < 0)
	{
		// (decimation in the spectrum: no rotation)
		if (X.c.up == 1 && (X.c.in_len & 1) == 0) convp_tail_owners<2 * ConvpGeom<LN, UL>::N, 0>(X.c);
	}
	else if (solo)
	{
		if (UL == 0 && X.c.up_pow2 && X.c.up == 1 && (X.c.in_len & 1) == 0)
		{
			constexpr int N = ConvpGeom<LN, UL>::N;
			X.c.rot = (N - ((X.c.fl2 / 2) & (N - 1))) & (N - 1);
			X.c.fl2r = X.c.fl2 % 2;
			convp_tail_owners<2 * N, 0>(X.c);
		}
	}
	else if (spu)
	{
		if (UL == 0 && X.c.up_pow2 && X.c.up == 2)
		{
			constexpr int N = ConvpGeom<LN, UL>::N;
			X.c.rot = (N - ((X.c.fl2 / 2) & (N - 1))) & (N - 1);
			X.c.fl2r = X.c.fl2 % 2;
			convp_tail_owners<N, 1>(X.c);
		}
	}
	else if constexpr (UL >= 0)
	{
		if (X.c.up_pow2 && X.c.up == (1 << (UL > 0 ? UL : 0)))
		{
			constexpr int N = ConvpGeom<LN, UL>::N;
			X.c.rot = (N - ((X.c.fl2 / X.c.up) & (N - 1))) & (N - 1);
			X.c.fl2r = X.c.fl2 % X.c.up;
			convp_tail_owners<N, (UL > 0 ? UL : 0)>(X.c);
		}
	}
	else
	{
		// (decimation in the spectrum: the block is loaded as in the 1:1 form)
		if (X.c.up == 1) convp_tail_owners<ConvpGeom<LN, UL>::N, 0>(X.c);
	}
	{
		// what the workgroups of a pair's blocks can share, one element per thread and channel (cp_park_slice_*,
		// cp_tail_slice_*; mode 1 -- one phase per thread -- keeps the forms of one workgroup)
		typedef ConvpGeom<LN, UL> G;
		const long long nwg = (X.c.nblk + G::SUB - 1) / G::SUB;
		X.park_slices = slices && X.park_n > 0 && X.park_n <= nwg * G::WT ? 1 : 0;
#ifndef R8B_NO_TAIL_SLICES // (development A/B: the last block fetches the rest alone)
		if (slices && (X.c.tail_flags & 2) != 0 && X.c.tail_p1 > X.c.tail_c1 && X.c.tail_p1 - X.c.tail_c1 <= nwg * G::WT &&
			X.c.tail_c1 >= X.c.src.cur_base)
			X.c.tail_flags |= 8;
#endif
	}
}

// Walk form: which of the launch's blocks are INTERIOR ones, i.e. may run on the lean walk body (convp_body<.., WALK>)?
// Block i of the launch (k = k0 + i) qualifies when its window lies inside the caller's fp64 buffer (the fast load
// path), it owns no part of the history tail and is not the call's last block (no parked outputs), and it holds whole
// output groups only (SpanInfo of the two-phase interpolator: first phase 0, last group complete), the destination being
// linear fp64 rows.  Returns the longest such run [*i0, *i1) (called after convp_prepare; false: no walk).
template<int LN, int UL>
inline bool convp_walk_range(const ConvxLaunch& X, int* i0, int* i1)
{
	typedef ConvpGeom<LN, UL> G;
	constexpr int US = UL > 0 ? UL : 0;
	const ConvLaunch& L = X.c;
	if (L.src.cur_fmt != kPcmF64 || X.wdst.fmt != kPcmF64 || X.wdst.mask != -1 || !L.up_pow2 || L.up != (1 << US)) return false;
	if ((L.tail_flags & 5) != 0) return false; // (history copied in slices by every block: short calls)
	if ((X.out_step & 1) != 0) return false;   // (phase pairs straddle the groups)
	const long long iln = L.in_len >> US;
	int best0 = 0, best1 = 0, run0 = -1;
	for (int i = 0; i <= L.nblk; i++)
	{
		bool ok = i < L.nblk;
		if (ok)
		{
			const long long k = L.k0 + i;
			const long long bs = (k * (long long) L.blk_stride + L.blk_offset) >> US;
			const long long w0 = bs - (G::N - iln);
			ok = w0 >= L.src.cur_base && w0 >= 0;
			if ((L.tail_flags & 2) != 0 && i >= L.tail_bf) ok = false;
			if (i + 1 == L.nblk) ok = false;
			const SpanInfo& B = X.blk[i];
			if (B.jhi <= B.jlo || B.jlo_mod != 0 || B.pad != X.out_step) ok = false;
			// (the stream's first blocks: zeros in front of the run, cp_final_store)
			if (k * (long long) L.blk_stride + L.blk_offset - L.fl2 - L.t_zero <= 0) ok = false;
		}
		if (ok && run0 < 0) run0 = i;
		if (!ok && run0 >= 0)
		{
			if (i - run0 > best1 - best0) { best0 = run0; best1 = i; }
			run0 = -1;
		}
	}
	*i0 = best0;
	*i1 = best1;
	return best1 > best0;
}

// workgroup i of a launch, pair major: the block groups of one channel pair are consecutive
// (solo: the one-channel form -- item pr is channel pr)
template<int SUB>
R8B_HD ConvpItem convp_item(const ConvLaunch& L, long long i, bool solo = false)
{
	ConvpItem it;
	const int nbg = (L.nblk + SUB - 1) / SUB;
	const int pr = (int) (i / nbg);
	const int b0 = (int) (i - (long long) pr * nbg) * SUB;
	it.k = L.k0 + b0;
	it.nvalid = L.nblk - b0 < SUB ? L.nblk - b0 : SUB;
	it.chA = solo ? pr : 2 * pr;
	it.bvalid = !solo && it.chA + 1 < L.nch;
	it.chB = it.bvalid ? it.chA + 1 : it.chA;
	return it;
}

// floor(i / d) for i * d < 2^32 by a multiplication: magic = floor(2^32 / d) + 1 (set by the launcher)
R8B_HD unsigned convp_div(unsigned i, unsigned magic)
{
	return (unsigned) (((unsigned long long) i * magic) >> 32);
}

} // namespace r8bhip

#endif
