// r8b_plan.h -- host-side schedule of a resampler chain.
//
// All control flow on the resampling path is data independent: how many
// samples each stage emits per process() call, and which absolute stream positions they are,
// depends only on the rates, the filter geometry and the number of samples fed so far.  The plan
// keeps that integer state for ONE stream; every channel of a batch follows it, and the kernels
// are addressed by absolute stream position.
//
// Per stage the plan answers: after M input samples in total, how many output samples has the
// reference's stage emitted (`total`)?  The closed forms restate
//   CDSPBlockConvolver::process      reference CDSPBlockConvolver.h:252-354 (+ geometry :62-185)
//   CDSPFracInterpolator::process    reference CDSPFracInterpolator.h:861-922, 991-1060, 1069-1179
//   CDSPHBUpsampler::process         reference CDSPHBUpsampler.h:674-732
//   CDSPHBDownsampler::process       reference CDSPHBDownsampler.h:137-239
// and were checked call by call against the compiled reference (tests/test_plan.py).
#ifndef R8B_PLAN_H
#define R8B_PLAN_H

#include <string>
#include <vector>

#include "r8b_design.h"

namespace r8bhip {

// Geometry of one overlap-save block convolver (reference CDSPBlockConvolver.h:62-185).
struct ConvGeom
{
	int up = 1, down = 1;
	int fl2 = 0;        // alignment of outputs against inputs, virtual samples: output q*down uses inputs up to
	                    // q*down + fl2 (linear phase: the filter's half length == its latency)
	int bl2 = 0;        // BlockLen2: length of the circular block (virtual, upsampled rate)
	int in_len = 0;     // InputLen: new virtual samples per block
	int prev_len = 0;   // PrevInputLen (input-rate samples when up is 2^k, virtual otherwise)
	int latency = 0;    // samples (virtual rate, before decimation) the stage swallows
	int n_in = 0;       // length of the forward real FFT
	int n_out = 0;      // length of the inverse real FFT
	// The reference's own block (describe(), latency and per-call counts follow it).  When its
	// transforms exceed what one workgroup can hold in LDS (16384 points), the kernels run the SAME
	// filter on shorter blocks bl2 / in_len (overlap-save is exact for any block longer than the
	// filter, and a shorter block never needs input the reference's latency has not already waited
	// for), anchored at multiples of the shorter in_len: still bitwise chunk invariant.
	int ref_bl2 = 0, ref_in_len = 0, ref_n_in = 0, ref_n_out = 0;
	// the kernel spectrum is complex: minimum-phase filter, or a linear-phase one whose alignment moved by
	// whole samples of latency inherited from the previous stage (fl2 != the filter's half length)
	bool complex_h = false;
	bool up_pow2 = true;   // up-sampling by spectrum replication (else explicit zero stuffing)
	bool down_pow2 = false; // decimation by spectrum truncation (else strided pick)
	// 3x up-sampling WITHOUT transforming the stuffed zeros (round 5; r8b_convp.h mode 19, reference
	// CDSPBlockConvolver.h:414-496 copyUpsample + :283-350): output 3 m + r of the stage is the r-th polyphase component
	// of the filter applied to the INPUT stream, y[3 m + r] = sum_d x[m - d] g_r[d], g_r[d] = h[3 d + r + fl2], d in
	// [-p3_a, p3_b]; a block is a window of p3_n input samples -- ONE forward transform, three backward ones -- with
	// p3_m = p3_n - p3_a - p3_b valid outputs per component, block k holding stage outputs [k 3 p3_m + p3_off - fl2,
	// + 3 p3_m), a multiple of 3 at its start.  p3: the form applies (3x, no decimation, the filter's components short
	// enough for a window of 1024 ... 4096 points that the reference's latency covers to be at least half valid); the fields
	// above keep describing the zero-stuffing
	// block, which the engine falls back to (option up3_poly = 0).
	bool p3 = false;
	int p3_n = 0, p3_a = 0, p3_b = 0, p3_m = 0, p3_off = 0;
	// (engine, Engine::eff_geom: the geometry above replaced by the polyphase block -- in_len = 3 p3_m, bl2 = 3 p3_n,
	// n_in = n_out = p3_n -- and the offset of block 0's first fresh virtual sample)
	bool poly3 = false;
	int blk_off = 0;
};

// State of a CDSPFracInterpolator's non-whole-stepping position counter
// (reference CDSPFracInterpolator.h:907-919, 1153-1168).
struct PolyState
{
	long long rpos = 0;     // absolute integer input position of the next output
	double pos_frac = 0.0;  // InPosFrac
	int in_counter = 0;     // InCounter
	int in_pos_int = 0;     // InPosInt
	double pos_shift = 0.0; // InPosShift
};

struct StagePlan
{
	StageDesc desc;
	// conv
	ConvGeom cg;
	LpFilterRef lp;      // (shared with the designer's bounded cache: alive as long as this plan)
	// frac
	bool whole = false;
	int in_step = 0, out_step = 0;
	FracBankRef bank;
	int flen = 0, fl2 = 0, fll = 0;
	double ssr = 0, dsr = 0;
	PolyState poly;      // state at the start of the next call
	// half-band
	const double* hb_taps = nullptr;
	int hb_n = 0;
	double hb_att = 0;

	// Fractional latency bookkeeping (reference: every stage takes the previous stage's left-over
	// fractional latency `PrevLatency` and reports its own, CDSPResampler.h:688).  Linear-phase chains
	// have all of this zero.
	//   out_skip  output samples of the underlying stream function that are dropped at the start of
	//             the stream (reference: LatencyLeft of the half-band stages, CDSPHBUpsampler.h:715-727;
	//             plus the input samples a following interpolator swallows, CDSPFracInterpolator.h:866-877);
	//             emitted sample q is underlying sample q + out_skip
	//   pos0      whole-step interpolator: InitFracPosW (CDSPFracInterpolator.h:741-743)
	//   frac0     polynomial interpolator: InitFracPos (:721-723)
	//   lat_frac  what this stage hands to the next one
	long long out_skip = 0;
	int pos0 = 0;
	double frac0 = 0.0;
	double lat_frac = 0.0;

	// stream counters
	long long m = 0;     // input samples received so far
	long long done = 0;  // output samples emitted so far

	void clear();
	// Feeds l more input samples; returns the absolute range [*a, *b) of output samples the
	// reference's stage returns from this call.  For the polynomial interpolator *ps receives
	// the counter state the call starts from.
	void step(int l, long long* a, long long* b, PolyState* ps);
	int max_out_len(int maxin) const;
	int in_len_before_out_pos(int pos) const;
	// number of input samples preceding position m that the next call may still read
	int history() const;
	std::string describe() const;

private:
	long long total(long long mm) const;
};

// prev_lat: the fractional latency the previous stage left (0 for the first stage)
StagePlan make_stage_plan(const StageDesc& d, double prev_lat = 0.0);

struct ChainPlan
{
	std::vector<StagePlan> stages;
	int max_in = 0;
	int max_out_len = 0;             // CDSPResampler::getMaxOutLen(0)
	std::vector<int> stage_max_in;   // max new samples per call entering each stage

	void init(const std::vector<StageDesc>& descs, int maxin);
	void clear();
	int in_len_before_out_pos(int pos) const; // reference CDSPResampler.h:406-421
	int input_required(int nout) const;        // reference CDSPResampler.h:476-484
	std::string describe() const;
};

} // namespace r8bhip

#endif
