// r8b_launch.h -- the narrow interface between the host engine (r8b_engine.cpp, plain C++) and the
// device side (r8b_kernels.hip): launch descriptors, launchers and a handful of memory helpers.
// Pointers inside the descriptors are DEVICE pointers.  Positions are absolute stream positions
// (samples since construction/clear) of the respective stage's input or output stream.
#ifndef R8B_LAUNCH_H
#define R8B_LAUNCH_H

#include <cstddef>

namespace r8bhip {

// PCM sample formats a caller buffer may hold (r8b_pcm_codec.h, r8b_pcm.h); values are part of the C ABI
// (include/r8bsrc.h, enum r8b_pcm_format)
enum PcmFormat { kPcmF64 = 0, kPcmF32 = 1, kPcmS16 = 2, kPcmS24 = 3, kPcmS32 = 4 };

// Where a stage reads its input stream x[pos] for channel ch:
//   pos <  0          -> 0.0 (the stream starts at 0)
//   pos >= cur_base   -> cur[ch*cur_stride + pos - cur_base]     (this call's fresh samples)
//   otherwise         -> ring[ch*ring_stride + (pos & ring_mask)] (history kept from earlier calls)
struct SrcView
{
	const double* ring;
	long long ring_stride;
	long long ring_mask;
	const double* cur;
	long long cur_stride;
	long long cur_base;
	int cur_fmt; // PcmFormat of `cur` (planar; strides in samples); the ring is always fp64
};

// Where a stage writes output sample q of channel ch: p[ch*stride + ((q + off) & mask)].
// Ring: off = 0, mask = size-1.  Linear (user buffer): mask = -1, off = -first_q.
struct DstView
{
	double* p;
	long long stride;
	long long mask;
	long long off;
	int fmt; // PcmFormat of the buffer (caller's planar PCM buffer at the last stage), else fp64
};

struct cd;

static const int kMaxPasses = 16;

struct ConvLaunch
{
	// geometry (r8b_plan.h ConvGeom)
	int up, down, fl2, bl2, in_len, n_in, n_out;
	// pair form (r8b_convp.h convp_prepare): the block sits in the transform's circular array rotated by `rot`
	// input samples, chosen so that the valid outputs come out at circular positions fl2r, fl2r + 1, ... with
	// fl2r = fl2 mod up (0 or 1) instead of fl2: the run needs no index rotation when it is written out.
	// (0 / fl2 elsewhere.)
	int rot, fl2r;
	// virtual samples between the starts of consecutive blocks: in_len (the reference's own block
	// anchoring, reference CDSPBlockConvolver.h:283-305) except in the fused fast path
	int blk_stride;
	int blk_offset; // virtual position of block 0's first fresh sample (0 except fused MFMA mode)
	// fast path: the source admits aligned 16-byte loads of sample pairs (even positions)
	int vec_ok;
	int inplace; // generic kernel: backward transform in the forward array (n_in == n_out only)
	// generic kernel, blocks of the reference's longest filters (32768 points in front of a decimation in the spectrum:
	// k_conv_big): the forward transform runs in two halves through LDS, the packed backward spectrum of a block passes
	// through global memory -- one array of n_out doubles per workgroup SLOT (the launch's workgroups walk the (block,
	// channel) items); null: the ordinary generic kernel, both arrays in LDS.  fwd_radix[0] is then the radix-2 stage
	// taken in the load, the rest the passes of a sub-block (n_in / 4 complex)
	double* work;
	int work_slots;
	// fast path at stage 0: the workgroups of the first block also copy stream positions
	// [tail_p0, tail_p1) into the history ring the NEXT call reads (null: nothing to copy)
	double* tail_ring;
	long long tail_p0, tail_p1;
	// pair form (r8b_convp.h convp_prepare / cp_tail_owned), tail_flags: 1 -- every block copies a slice of the tail;
	// 2 -- positions [tail_c0, tail_c1) of the tail go to the ring from the registers of the blocks that load them
	// anyway (blocks k0 + tail_bf ..: the ones that read the caller's buffer alone and hold a part of it), the launch's
	// last block fetches [tail_c1, tail_p1) beside its samples; 4 (with 2) -- the first block copies [tail_p0, tail_c0).
	// 8 (with 2) -- [tail_c1, tail_p1) is shared by the workgroups of the pair's blocks, one element per thread and channel
	// (cp_tail_slice_*), instead of being fetched by the last block alone.
	// 0: nothing to do (tail_ring == nullptr).  (Two integers the kernel keeps in scalar registers: a block with
	// nothing to do finds that out without a load.)
	int tail_flags, tail_bf;
	long long tail_c0, tail_c1;
	int up_pow2, down_pow2;
	// transform plan: radices of the forward passes in execution order (sub-length N, N/r0, ...)
	// and of the backward passes in execution order (sub-length grows to N2)
	int n_fwd, n_inv;
	int fwd_radix[kMaxPasses];
	int inv_radix[kMaxPasses];
	const double* H; // bl2/2+1 reals: zero-phase kernel spectrum / bl2
	const cd* Hc;    // generic kernel only: bl2/2+1 complex when the spectrum is not real (minimum phase), else null
	const cd* tw;    // tw_len complex: exp(-2 pi i e / tw_len)
	const cd* spec;  // fast path only: per-slot spectral-stage constants (r8b_convx.h)
	const cd* spec2; // fast path, up 1 or 2: (ca, cb) per backward POSITION, [c * N2 + P]
	const cd* hp;    // pair form (r8b_convp.h): kernel constants of the middle pass, [c * 256 + thread]
	const cd* ptw;   // pair form: twiddle base powers per pass and thread, [(slot * 6 + c) * 256 + thread]
	int t_zero;      // fused with an interpolator: its stream starts at this stage's output t_zero (0; a chain with a
	                 // fractional latency skips outputs -- StagePlan::out_skip): earlier outputs read as zeros
	int tw_len;
	// work: blocks [k0, k0+nblk) x channels [0, nch); outputs clipped to [a, b)
	long long k0;
	int nblk;
	long long a, b;
	int nch;
	int threads;
	SrcView src;
	DstView dst;
};

struct WholeLaunch
{
	int in_step, out_step, flen, fl2, fll;
	int pos0;            // output j sits at position j*in_step + pos0 (reference InitFracPosW; 0 for linear phase)
	const double* table; // out_step rows x flen
	const double* wtab;  // the same rows transposed and in output order, [tap][k] = table[(k * in_step) % out_step][tap]
	                     // (null: not available); consecutive outputs then read consecutive doubles per tap
	int inv_in;          // in_step^-1 mod out_step: output with phase ph is row k = (ph * inv_in) % out_step of wtab
	long long a, b;      // outputs to produce
	int tile;            // outputs per workgroup
	int span_max;        // LDS doubles per workgroup
	int nch;
	SrcView src;
	DstView dst;
};

struct PolyLaunch
{
	int flen, fl2, fll, fracs;
	const double* table; // (fracs+1) rows x flen x 3
	double ssr, dsr;
	// counter state at the start of the call (r8b_plan.h PolyState)
	long long rpos0;
	double fpos0;
	long long counter0, pos_int0;
	double shift;
	long long a, b;
	int nch;
	int span_max; // tiled kernel: max input span of an output tile (0: one output per thread)
	int pitch;    // tiled kernel: LDS doubles per channel row (>= span_max; residue mod 32 chosen for the step)
	int front;    // tiled kernel: samples and bank entries fetched in one phase (spans up to 16 * kPolyNV)
	SrcView src;
	DstView dst;
};

struct TailLaunch
{
	SrcView src;        // where the stream is read (history ring and/or the caller's buffer)
	long long p0, p1;   // positions [p0, p1) to copy into the ring
	double* ring;
	long long ring_stride;
	long long ring_mask;
	int nch;
};

struct HBLaunch
{
	int ntaps;
	double taps[16];
	long long a, b; // outputs to produce
	int tile;       // up: input indices per workgroup; down: outputs per workgroup
	int nch;
	SrcView src;
	DstView dst;
	// a history copy carried by this launch (Engine::process: the stream's tail for the next call, when no convolver
	// keeps it): with carry_tail set the grid has a second z layer whose workgroups copy `tail` instead of computing a
	// tile (the layer is known from the workgroup id alone: the tiles' workgroups do not look at these fields)
	int carry_tail;
	TailLaunch tail;
};

// A run of consecutive 2x half-band up-samplers executed by one kernel: every intermediate stream
// stays in LDS, only the first stage's input is read and the last stage's output written.
static const int kMaxCascade = 8;

struct HBCascadeLaunch
{
	int nst;                       // stages in the run
	int ntaps[kMaxCascade];
	double taps[kMaxCascade][14];
	long long a, b;                // outputs of the LAST stage to produce
	int tile;                      // last-stage outputs per workgroup (multiple of 2^nst)
	int buf, buf2;                 // doubles of the two LDS buffers: tile/2 + slack, tile/4 + slack
	int pair_ok;                   // the destination admits aligned 16-byte stores of output pairs
	// up-sampling run only: the output range of stage s for the tile [q0, q1) in closed form,
	//   lo_s = (q0 - rlo[s]) >> (nst-1-s),  hi_s = ((q1 - 1 + rhi[s]) >> (nst-1-s)) + 1,
	// and the input span of stage 0 with rlo[nst] / rhi[nst] and a shift of nst (hbc_fill_ranges); equal to
	// walking the stages back from the tile (floors of halves compose), without a chain of dependent loads
	long long rlo[kMaxCascade + 1], rhi[kMaxCascade + 1];
	// chains with a fractional latency: stage s emits its stream from output skip[s] on (StagePlan::out_skip), i.e. input n
	// of stage s + 1 is output n + skip[s] of stage s, and outputs below skip[s] read as zeros there (up-sampling run;
	// the last stage's skip is folded into a, b and the destination's offset by the launcher: 0 here)
	int skip[kMaxCascade];
	int has_skip;                  // any of them non-zero
	long long in_end;              // input positions >= in_end have not arrived: the zero-padded
	                               // taps reach past the real filter, those loads must not happen
	int nch;
	SrcView src;                   // input stream of the first stage
	DstView dst;
	int carry_tail;                // a carried history copy, as in HBLaunch
	TailLaunch tail;
};

inline void hbc_fill_ranges(HBCascadeLaunch& L)
{
	long long cl = 0, ch = 0;
	for (int s = L.nst - 1; s >= 0; s--)
	{
		const int k = L.nst - 1 - s;
		L.rlo[s] = cl;
		L.rhi[s] = ch;
		// (stage s - 1 has to deliver the input range of stage s, skip[s - 1] outputs later)
		const long long sk = s > 0 ? L.skip[s - 1] : 0;
		cl += ((long long) (L.ntaps[s] - 1) - sk) << (k + 1);
		ch += ((long long) L.ntaps[s] + sk) << (k + 1);
	}
	L.rlo[L.nst] = cl;
	L.rhi[L.nst] = ch;
}


struct PcmLaunch
{
	void* pcm;              // PCM buffer (read by the ingest kernel, written by the egress kernel)
	int fmt;                // PcmFormat
	int interleaved;        // 1: frame-major, sample (f, c) at element f * pcm_stride + c
	                        // 0: planar, at element c * pcm_stride + f
	long long pcm_stride;   // in samples
	double* planar;         // the resampler's rows: (c, f) at planar[c * planar_stride + f]
	long long planar_stride;
	int nch;
	long long n;            // frames
};

// fast path (r8b_convx.h): power-of-two block convolver, optionally fused with the whole-step
// interpolator that follows it
// Interpolator outputs one block owns, precomputed by the host so that the kernel needs no 64-bit
// divisions: outputs [jlo, jhi); for j = jlo + d the tap window starts at index
// u_lo + (ph_lo + d*in_step) / out_step of the block's linear run.
struct SpanInfo
{
	long long jlo, jhi;
	int jlo_mod; // jlo mod out_step
	int u_lo;
	int ph_lo;   // (jlo * in_step) mod out_step
	int pad;
};

#ifndef R8B_CONVX_MAX_BLOCKS
#define R8B_CONVX_MAX_BLOCKS 104
#endif
// blocks per fused launch (longer calls are split).  104: the launch descriptor is a kernel argument, ~ 700 + 32 x 104
// of the 4096 bytes a launch can carry; 44100 -> 96000 at a 10 % transition band has 99 - 104 blocks per 16384-sample
// call (0.253 -> 0.242 ms as one launch instead of two, profiles/r05_experiments.txt; with 99 blocks per launch that
// configuration fell back to two launches in round 6: 0.272)
static const int kConvxMaxBlocks = R8B_CONVX_MAX_BLOCKS;

// pair form, mode 20 (r8b_convp.h cp_hbf_*): convolver inputs per staging round of the half-band front (two rounds per
// 4096-point window), the longest half-band filter it takes
static const int kHbfRound = 2048;
static const int kHbfTapsMax = 14;

// pair form, mode 20 (r8b_convp.h cp_hbf_*): a half-band decimator in front of the convolver taken in the block's load --
// c.src is the DECIMATOR's input stream (caller's buffer + history ring), the block's window of N convolver inputs is
// computed from 2 N + 4 np raw samples in LDS (reference CDSPHBDownsampler.h:137-239 in front of
// CDSPBlockConvolver.h:283-350).  n taps, rounded up to np (4 / 8 / 14, the extra taps zero); raw positions >= end have
// not arrived and read as zeros (only the zero taps ever reach them).
struct HbFront
{
	int n, np;
	long long end;
	double taps[kHbfTapsMax];
};
static_assert(sizeof(HbFront) <= 4 * sizeof(SpanInfo), "ConvxLaunch::hbf lies over the last four block entries");

struct ConvxLaunch
{
	ConvLaunch c;        // geometry, tables, block range, conv output range [c.a, c.b), src, dst
	// MODE 1 (fused whole-step interpolator); c.dst is unused then
	int in_step, out_step, flen, fl2w, fllw;
	const double* table; // out_step rows x flen (boundary kernel)
	const double* wtab;  // flen x out_step: wtab[i*out_step + t] = table[(t*in_step % out_step)*flen + i]
	long long wa, wb;    // interpolator outputs to produce
	DstView wdst;
	// per block c.k0 + i (mode 1) -- or, in the convolver-only mode 20 (which has no per-block spans), the half-band
	// front's parameters in the array's last four entries (the descriptor has no room for both)
	union
	{
		SpanInfo blk[kConvxMaxBlocks];
		struct
		{
			SpanInfo unused_[kConvxMaxBlocks - 4];
			HbFront p;
		} hbf;
	};
	// pair form, mode 4 (two adjacent phases per thread, r8b_convp.h): the run of (A, B) pairs starts
	// at LDS slot run_off; per block blk[].u_lo = run slot of the window of phase 0 of the block's first
	// output group, .ph_lo = groups - 1, .pad = the phase the block's last group ends before (1 ..
	// out_step; 0: the block has no output); per thread its phase pair, group set and window start
	// (ptab[t] = q | set << 8 | floor(2 q in_step / out_step) << 12, -1: idle), the two 25-tap rows as
	// 25 pairs (ctab[(i * 256 + t) * 2]), the number of group sets
	int run_off;
	const int* ptab;
	const double* ctab;
	int nsets;
	unsigned nblk_magic; // floor(2^32 / c.nblk) + 1 (filled in by the launcher; r8b_convp.h convp_div)
	// Parked outputs (pair form, modes 4 / 5 at the chain's last stage; r8b_engine.cpp launch_fused): the block that
	// holds a call's last output usually holds outputs of the NEXT call too (the reference answers one block late,
	// reference CDSPBlockConvolver.h:94-101, 283-305, so a call's outputs end in the middle of a block).  Instead of
	// computing that block again in the next call, the launch's last block leaves its outputs [wb, park_blk.jhi) in a
	// small per-channel buffer (park_dst: row ch at park_dst + ch * park_stride, output j at index j - wb), and the next
	// call's first workgroup of each channel pair copies them to the caller's rows beside its sample loads (park_src:
	// outputs [park_j0, park_j0 + park_n) at indices 0 ..).  park_n = 0 / park_out = 0: nothing to do.
	// park_slices (set by convp_prepare): the copy is shared by the workgroups of the pair's blocks in this launch, one
	// element per thread and channel, requested a phase ahead of the interpolator and stored at its start (no wait of
	// its own); 0: the launch's first workgroup of the pair copies everything beside its sample loads (short calls)
	int park_n, park_out, park_slices;
	// walk form of the fused two-phase modes (r8b_convp.h convp_walk): blocks per workgroup (a workgroup takes `walk`
	// consecutive blocks of its channel pair and keeps rows and twiddles across them); 0: one workgroup per block
	int walk;
	// ... set by the launcher (convp_walk_range): blocks [walk_i0, walk_i1) of the launch are interior ones and run on
	// the walk body, walk_len of them per workgroup; the others -- a call's first and last blocks -- on the general body,
	// a workgroup each, in the same launch
	int walk_i0 = 0, walk_i1 = 0, walk_len = 0;
	// eight elements per thread (r8b_convq.h: the 2048 -> 4096-point block pair on 512 threads, four waves per SIMD);
	// the engine sets it where that form exists (option "quad") and the launcher takes it
	int quad = 0;
	// half-array form of the same block pair (r8b_convp.h cp_ha_*, kernel mode 21: the backward side's exchanges by parts
	// through 32 KB of LDS, four workgroups per CU); engine option "half", taken by the launcher where the form exists
	int half = 0;
	// ... and of the fused two-phase block pair (kernel mode 23: the interpolator's run in 49 KB, three workgroups per CU;
	// taken in place of mode 4 and its walk form); engine option "half_fused"
	int half_fused = 0;
	long long park_j0, park_stride;
	const double* park_src;
	double* park_dst;
	SpanInfo park_blk;
};
static_assert(sizeof(ConvxLaunch) <= 4096, "ConvxLaunch is passed by value: 4096 bytes of kernel arguments");

// geometries the fast path is instantiated for: (log2 of the forward complex length, up shift), and
// for the 2x-decimating convolver (log2 of the forward complex length, down shift)
// (R8B_DEV_GEOMS: a development build that instantiates only the geometries named on the compiler's command line --
// tools/variant.sh; a one-geometry build takes a minute instead of ten)
#ifndef R8B_DEV_GEOMS
#define R8B_CONVX_GEOMS(M) M(8, 1) M(9, 0) M(9, 1) M(10, 0) M(10, 1) M(11, 0) M(11, 1) M(12, 0) \
	M(12, 1) M(13, 0)
#define R8B_CONVX_GEOMS_DOWN(M) M(8, 1) M(9, 1) M(10, 1) M(11, 1) M(12, 1) M(13, 1) \
	M(10, 2) M(11, 2) M(12, 2) M(13, 2)
#endif

inline bool convx_geometry_ok(int n_in, int n_out, int up, int down, bool up_pow2)
{
	if (!up_pow2 || (down != 1 && down != 2 && down != 4) || (up != 1 && up != 2)) return false;
	if (down > 1 ? (up != 1 || n_out * down != n_in) : n_out != n_in * up) return false;
	int logn = 0;
	while ((2 << logn) < n_in) logn++;
	if ((2 << logn) != n_in) return false;
	if (down > 1)
	{
#define R8B_CONVX_CHECK(LN, DL) if (logn == LN && down == (1 << DL)) return true;
		R8B_CONVX_GEOMS_DOWN(R8B_CONVX_CHECK)
#undef R8B_CONVX_CHECK
		return false;
	}
#define R8B_CONVX_CHECK(LN, UL) if (logn == LN && up == (1 << UL)) return true;
	R8B_CONVX_GEOMS(R8B_CONVX_CHECK)
#undef R8B_CONVX_CHECK
	return false;
}

// MODE 3 of the fast path: the same transforms behind a zero-stuffing load (3x up-sampling) and/or
// in front of a strided store (3x decimation) -- ratios 3/1, 1/3, 2/3, 3/2, 3/4
inline bool convx_mode3_ok(int n_in, int n_out, int up, int down, bool up_pow2, bool down_pow2)
{
	if (!((!up_pow2 && up == 3) || (!down_pow2 && down == 3))) return false;
	return convx_geometry_ok(n_in, n_out, up_pow2 ? up : 1, down_pow2 ? down : 1, true);
}

// pair form of the fast path (r8b_convp.h): two channels as one complex transform; backward transforms
// of 64 ... 8192 points, 1:1 or 2x up-sampled (below 4096 points a workgroup carries several blocks)
inline bool convp_geometry_ok(int n_in, int n_out, int up, int down, bool up_pow2)
{
	if (!up_pow2 || (up != 1 && up != 2)) return false;
	if (down == 2 || down == 4)
	{
		// 2x / 4x decimation in the spectrum (the caller passes down_pow2 geometries only)
		if (up != 1 || n_out * down != n_in) return false;
		return n_in <= 8192 && n_in >= 64 && (n_in & (n_in - 1)) == 0;
	}
	if (down != 1 || n_out != n_in * up) return false;
	return n_out == 64 || n_out == 128 || n_out == 256 || n_out == 512 || n_out == 1024 || n_out == 2048 ||
		n_out == 4096 || n_out == 8192;
}
// split 2x up-sampling form of the pair kernel (r8b_convp.h cp_sp_*, modes 8 / 9 -- 12 / 13 with a complex spectrum -- on the 8192-point 1:1 geometry): 8192 ->
// 16384-point blocks -- a 2x up-sampling filter with a transition band of about 1 % and below --, optionally in front of
// the 3x strided store
inline bool convp_split_ok(int n_in, int n_out, int up, int down, bool up_pow2, bool down_pow2)
{
	return up_pow2 && up == 2 && n_in == 8192 && n_out == 16384 && (down == 1 || (!down_pow2 && down == 3));
}
// one-channel form of the pair kernel (r8b_convp.h cp_solo_*, modes 10 / 11 -- 14 / 15 with a complex spectrum -- on the 8192-point 1:1 geometry): 16384-point
// blocks 1:1 (an even number of new samples per block: the samples travel in pairs), optionally behind the 3x zero
// stuffing load or in front of the 3x strided store
inline bool convp_solo_ok(int n_in, int n_out, int up, int down, bool up_pow2, bool down_pow2, int in_len)
{
	if (n_in != 16384 || n_out != 16384) return false;
	if (up_pow2 && up == 1) return (in_len & 1) == 0 && (down == 1 || (!down_pow2 && down == 3));
	return !up_pow2 && up == 3 && down == 1;
}
// ... decimating by 2 or 4 in the spectrum (geometries <13, -1> / <13, -2>: 16384 -> 8192 / 4096 points; by 4: real spectra
// only), optionally behind the 3x zero stuffing load
inline bool convp_solo_down_ok(int n_in, int n_out, int up, int down, bool up_pow2, bool down_pow2, int in_len)
{
	if (n_in != 16384 || !down_pow2 || !((down == 2 && n_out == 8192) || (down == 4 && n_out == 4096))) return false;
	return up_pow2 ? (up == 1 && (in_len & 1) == 0) : up == 3;
}
// ... with the whole-step interpolator fused in (modes 1 and 4)
inline bool convp_fused_ok(int n_in, int n_out, int up, int down, bool up_pow2)
{
	return down == 1 && convp_geometry_ok(n_in, n_out, up, down, up_pow2);
}
// MODE 3 of the pair form: 3x zero stuffing in the load and / or 3x strided decimation in the store
// around a 1:1 or 2x-up transform pair -- ratios 3/1, 1/3, 2/3 (3/2 and 3/4 decimate by a power of two
// in the spectrum as well: the decimating form of the pair transforms)
inline bool convp_mode3_ok(int n_in, int n_out, int up, int down, bool up_pow2, bool down_pow2)
{
	if (!((!up_pow2 && up == 3) || (!down_pow2 && down == 3))) return false;
	if (up_pow2 ? (up != 1 && up != 2) : up != 3) return false;
	// (down_pow2 is false for down = 1)
	if (down != 1 && !(down_pow2 ? (down == 2 || down == 4) : down == 3)) return false;
	return convp_geometry_ok(n_in, n_out, up_pow2 ? up : 1, down_pow2 ? down : 1, true);
}
#ifndef R8B_DEV_GEOMS
#define R8B_CONVP_GEOMS(M) M(11, 1) M(12, 0) M(10, 1) M(11, 0) M(9, 1) M(10, 0) M(8, 1) M(9, 0) M(7, 1) M(8, 0) \
	M(6, 1) M(7, 0) M(5, 1) M(6, 0)
// 8192-point blocks (512-thread workgroups)
#define R8B_CONVP_GEOMS_BIG(M) M(13, 0) M(12, 1)
// decimating form: (log2 forward length, log2 decimation)
#define R8B_CONVP_GEOMS_DOWN(M) M(13, 1) M(12, 1) M(11, 1) M(10, 1) M(9, 1) M(8, 1) M(7, 1) M(6, 1) \
	M(13, 2) M(12, 2) M(11, 2) M(10, 2) M(9, 2) M(8, 2) M(7, 2) M(6, 2)
#endif

// launchers (asynchronous on `stream`, a hipStream_t)
void launch_conv(const ConvLaunch& L, void* stream);
void launch_whole(const WholeLaunch& L, void* stream);
void launch_poly(const PolyLaunch& L, void* stream);
void launch_hbup(const HBLaunch& L, void* stream);
void launch_hbdown(const HBLaunch& L, void* stream);
void launch_hbcascade(const HBCascadeLaunch& L, void* stream);
// a run of half-band decimators (same descriptor; buf / buf2 = even / odd stage inputs)
void launch_hbdcascade(const HBCascadeLaunch& L, void* stream);
void launch_tail(const TailLaunch& L, void* stream);
void launch_pcm_in(const PcmLaunch& L, void* stream);  // PCM -> planar fp64
void launch_pcm_out(const PcmLaunch& L, void* stream); // planar fp64 -> PCM
// mode 0: convolver output to X.c.dst; mode 1: fused interpolator output to X.wdst; mode 3: radix-3 edges
void launch_convx(const ConvxLaunch& X, int mode, void* stream);
// the same work in pair form (modes 0, 1, 3, 4 ... 9; needs X.c.hp)
void launch_convp(const ConvxLaunch& X, int mode, void* stream);
// blocks the calling THREAD's launches have put on the walk body so far (the launcher, not the engine, decides per
// launch -- convp_walk_range; an engine counts its own as the difference around its launches: Engine::stat("walk_blocks"))
long long launch_walk_blocks();
void launch_walk_blocks_add(long long n);
// the device symbol of the calling thread's most recent kernel launch, as rocprofv3 prints it without namespace and
// argument list ("k_convp_walk<11, 1, 4, 24>"); nullptr before the first launch.  The launchers note it, the engine
// reads it behind a stage's launch when option "timing" is on (Engine::stage_symbol, r8b_batch_stage_symbol)
void launch_symbol_note(const char* symbol);
const char* launch_symbol_last();

// memory helpers; all throw std::runtime_error with the HIP error text on failure
// Device selection.  dev_resolve: the ordinal an object created with `device` lives on (-1: the
// thread's current device).  DevGuard makes `device` current for its lifetime and restores the
// caller's device afterwards, so that an object on device 1 never changes what the calling thread
// (or torch) considers current.
int dev_resolve(int device);
int dev_swap(int device);              // makes `device` current, returns the previous one
void dev_restore(int device) noexcept; // makes `device` current again; never throws (destructors, unwinding)
struct DevGuard
{
	int prev;
	explicit DevGuard(int device) : prev(dev_swap(device)) {}
	~DevGuard() { if (prev >= 0) dev_restore(prev); }
	DevGuard(const DevGuard&) = delete;
	DevGuard& operator=(const DevGuard&) = delete;
};
void* dev_alloc(size_t bytes);        // zero-initialised
void dev_free(void* p);
void dev_zero(void* p, size_t bytes, void* stream);
void dev_upload(void* dst, const void* src, size_t bytes);   // synchronous
void dev_download(void* dst, const void* src, size_t bytes, void* stream); // synchronous on stream
void dev_upload_async(void* dst, const void* src, size_t bytes, void* stream);
void dev_sync(void* stream);
void dev_check_last(const char* what);
// events for the optional per-stage timing (hipEvent_t on the launching stream)
void* dev_event_create();
void dev_event_destroy(void* ev);
void dev_event_record(void* ev, void* stream);
float dev_event_elapsed_ms(void* start, void* stop); // waits for `stop`

} // namespace r8bhip

#endif
