// r8b_design.h -- host-side designer of the read-only coefficient sets and of the stage chain.
//
// Everything here runs once per parameter set, on the host, in fp64; its outputs (low-pass
// spectrum, fractional-delay bank, half-band taps, chain topology) are uploaded to HBM and are
// the only inputs the kernels take besides samples.  The numbers must equal the reference's to
// rounding level or the resampled stream differs, so every routine cites what it reproduces:
//   low-pass designer        reference CDSPFIRFilter.h:220-537, CDSPSincFilterGen.h:114-123,
//                            230-241, 312-338, 586-605
//   fractional-delay bank    reference CDSPFracInterpolator.h:61-189, 279-341,
//                            CDSPSincFilterGen.h:168-177, 452-552
//   half-band tap selection  reference CDSPHBUpsampler.h:47-316, 331-552
//   chain topology           reference CDSPResampler.h:135-394
#ifndef R8B_DESIGN_H
#define R8B_DESIGN_H

#include <memory>
#include <string>
#include <vector>

namespace r8bhip {

struct LpFilter
{
	// linear phase: h[-fl2..fl2] stored at [0..2*fl2]; minimum phase: causal h[0..kernel_len-1].
	// DC gain == gain either way.
	std::vector<double> taps;
	// Latency in samples (integer part).  Linear phase: the one-sided length.  Minimum phase: the integer
	// part of the group delay at DC (reference CDSPFIRFilter.h:476-484); lat_frac holds the rest.
	int fl2 = 0;
	double lat_frac = 0.0;
	bool zero_phase = true;
	int kernel_len = 0;       // 2*fl2+1 for linear phase
	int block_len_bits = 0;   // CDSPFIRFilter::getBlockLenBits()
};

#ifdef R8B_TEST_HOOKS
// Parity-test hook, compiled into TEST builds only (tests/emul/Makefile, `make testhooks` in csrc/: -DR8B_TEST_HOOKS);
// the shipped libr8bsrc_hip.so has neither the symbol nor the code.  A provider may supply a low-pass filter's taps
// instead of the designer (e.g. the REFERENCE's own minimum-phase taps, so that the kernels' parity can be checked
// apart from the conditioning of the cepstral transform, whose result depends on the rounding noise of the FFT that
// computes it).  Called on a cache miss, with the cache unlocked; returns the number of taps written (<= cap), 0 to
// decline.
typedef int (*LpProvider)(double norm_freq, double trans_band, double atten, double gain, int phase,
	double* taps, int cap, int* latency, double* lat_frac, int* block_len_bits);
void set_lp_provider(LpProvider p);
#endif

// reference EDSPFilterPhaseResponse (CDSPFIRFilter.h:28-45)
enum FilterPhase { kLinearPhase = 0, kMinPhase = 1 };

// Kaiser-power-windowed sinc low-pass; min_phase: followed by the cepstral minimum-phase transform
// (reference CDSPRealFFT.h:681-785 calcMinPhaseTransform, 16x oversampled).
// The result is shared with the designer's cache, which is bounded like the reference's (CDSPFIRFilter.h:598-694,
// r8bconf.h:90 R8B_FILTER_CACHE_MAX 96 / :103 R8B_FRACBANK_CACHE_MAX 12): least recently used entries nobody holds
// any more are dropped once the cache is full; what a live object holds stays alive through its shared_ptr.
typedef std::shared_ptr<const LpFilter> LpFilterRef;
LpFilterRef design_lp(double norm_freq, double trans_band, double atten, double gain,
	bool min_phase = false);
static const int kFilterCacheMax = 96, kFracBankCacheMax = 12, kLaneDealCacheMax = 96;
// entries the caches hold right now: [0] low-pass filters, [1] fractional-delay banks, [2] lane tables of the fused
// interpolator (Engine, one per ratio)
void design_cache_counts(int counts[3]);
void lane_deal_cache_count(int delta_or_zero, int* count); // (kept by r8b_engine.cpp; read through design_cache_counts)

struct FracBank
{
	int filter_len = 0;
	int fracs = 0;        // FilterFracs
	int element_size = 0; // 1 (plain taps) or 3 (c0,c1,c2 per tap)
	double atten = 0.0;   // rounded attenuation of the chosen row
	std::vector<double> table; // (fracs+1) rows x filter_len*element_size
};

typedef std::shared_ptr<const FracBank> FracBankRef;
FracBankRef design_frac_bank(int fracs, int element_size, int interp_points, double atten,
	bool third);

// Tap selection over the generated half-band tables.  Returns tap count, *taps points to
// static storage.
int select_hb_filter(double atten, int steep, bool third, const double** taps, double* att);

bool whole_stepping(double ssr, double dsr, int* in_step, int* out_step);

inline int bit_occupancy(long long v)
{
	int b = 0;
	while (v > 0) { b++; v >>= 1; }
	return b < 1 ? 1 : b;
}

enum StageKind { kConv = 0, kFrac = 1, kHBUp = 2, kHBDown = 3 };

struct StageDesc
{
	StageKind kind;
	// conv: a=norm_freq b=trans_band c=atten d=gain i0=up i1=down
	// frac: a=src b=dst c=atten i0=third
	// hbup/hbdown: a=atten i0=steep i1=third
	double a = 0, b = 0, c = 0, d = 0;
	int i0 = 0, i1 = 0;
	int phase = kLinearPhase; // conv only
};

// Stage chain a CDSPResampler(src, dst, ., tb, atten, phase) is made of.
std::vector<StageDesc> build_topology(double src, double dst, double tb, double atten,
	int phase = kLinearPhase);

} // namespace r8bhip

#endif
