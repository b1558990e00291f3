// r8b_engine.h -- N-channel batch resampler: host schedule (r8b_plan.h) + device state + launches.
//
// One Engine = `nch` independent streams that share one ChainPlan.  process() mirrors
// r8b::CDSPResampler::process (reference CDSPResampler.h:559-575): it walks the stage chain, but
// instead of ping-ponging host buffers it enqueues one position-addressed kernel per stage on a
// HIP stream.  Stage-to-stage streams live in per-channel rings in HBM; the first stage reads the
// caller's device buffer directly (plus a short history ring), the last stage writes the caller's
// device buffer directly.
#ifndef R8B_ENGINE_H
#define R8B_ENGINE_H

#include <map>
#include <utility>
#include <string>
#include <vector>

#include "r8b_launch.h"
#include "r8b_plan.h"

namespace r8bhip {

class Engine
{
public:
	Engine(const std::vector<StageDesc>& descs, int maxin, int nch, int device);
	~Engine();
	Engine(const Engine&) = delete;
	Engine& operator=(const Engine&) = delete;

	// returns output samples per channel produced by this call
	int process(const double* d_in, long long in_stride, int l, double* d_out,
		long long out_stride, void* stream);
	// the same with PLANAR buffers of PCM samples (PcmFormat; strides in samples), converted by
	// the first stage's loads and the last stage's stores; needs at least one stage (Src != Dst)
	// can the first / last stage of the chain take planar PCM caller buffers itself?
	bool pcm_fused_in() const;
	bool pcm_fused_out() const;
	int process_planar(const void* d_in, int in_fmt, long long in_stride, int l, void* d_out,
		int out_fmt, long long out_stride, void* stream);
	void clear();
	bool set_option(const std::string& name, int value);
	// counters since creation ("conv_blocks", "park_calls", "park_only_calls"); -1: unknown name
	long long stat(const std::string& name) const;
	void bump(const std::string& name) { stat_[name]++; } // (counters kept for the layers above: "pcm_staged_sides")

	// per-stage kernel time accumulated since the last call (only while option "timing" is 1):
	// resolves pending events, returns total milliseconds and the number of launches
	bool stage_timing(size_t stage, double* ms_sum, int* launches, std::string* kernel,
		long long* in_samples, long long* out_samples);

	// device symbol of the stage's most recent launch made while option "timing" was 1, as rocprofv3 prints it without
	// namespace and argument list ("k_convp_walk<11, 1, 4, 24>"); empty before the first one.  stage_timing's name is
	// the engine's LABEL for the stage's form ("k_convp_whole": convolver + interpolator in one launch)
	std::string stage_symbol(size_t stage) const;

	// Checkpoint of the streaming state of all channels: the plan's counters
	// and the contents of every history ring, as one host blob.  load_state() accepts only a blob
	// saved by an object of the same configuration (rates, filter parameters, MaxInLen, channel
	// count, engine options); a stream resumed from it continues bit-identically.  Both wait for
	// `stream`, the stream the process() calls were enqueued on.  state_size() is a constant of the
	// object; load_state() checks the whole blob before it changes anything.
	size_t state_size() const;
	size_t save_state(void* buf, size_t cap, void* stream);
	void load_state(const void* buf, size_t size, void* stream);

	const ChainPlan& plan() const { return plan_; }
	int channels() const { return nch_; }
	int device() const { return device_; }

private:
	struct StageDev
	{
		double* ring = nullptr; // input ring of this stage, nch x ring_size
		double* ring_alt = nullptr; // stage 0 only: second history ring (calls alternate)
		long long ring_size = 0;
		double* H = nullptr;
		cd* Hc = nullptr;    // complex kernel spectrum (minimum phase; generic kernel only)
		cd* tw = nullptr;
		cd* spec = nullptr; // fast-path spectral constants
		cd* spec2 = nullptr; // the same per backward position (up 1 or 2)
		cd* hp = nullptr;    // pair kernel: kernel constants of the middle pass (r8b_convp.h)
		cd* ptw = nullptr;   // pair kernel: twiddle base powers per pass and thread
		cd* hp3 = nullptr;   // polyphase 3x form (ConvGeom::p3, r8b_convp.h mode 19): spectra of the three components
		cd* ptw3 = nullptr;  // ... and the twiddles of its 4096-point 1:1 geometry
		int tw_len = 0;
		double* table = nullptr;
		double* wtab = nullptr; // whole-step bank, transposed per residue class (fused kernel)
		// pair kernel, two adjacent phases per thread (mode 4): thread table and 25-tap row pairs
		int* ptab = nullptr;
		double* ctab = nullptr;
		int nsets = 0;
		int taps2 = 25; // entries per row: 25 (In <= Out) or 27
		// parked outputs of a fused pair at the end of the chain (launch_fused, ConvxLaunch::park_*): two buffers of
		// nch x park_stride doubles used in turn (a call reads the one the previous call filled while its own last block
		// fills the other); outputs [park_base, park_end) of the stream sit at indices 0 .. of buffer park_cur
		double* park[2] = { nullptr, nullptr };
		long long park_stride = 0;
		long long park_base = 0, park_end = 0;
		int park_cur = 0;
		// generic convolver on the reference's 32768-point blocks (k_conv_big): the packed backward spectra on their way
		// between the two forward halves and the backward transform, one array of n_out doubles per workgroup of the launch
		double* work = nullptr;
		int work_slots = 0;
		std::vector<int> fwd_radix, inv_radix;
		std::vector<std::pair<void*, void*>> pending; // (start, stop) events not yet read
		std::vector<void*> free_events;
		double ms_sum = 0.0;
		int launches = 0;
		long long t_in = 0, t_out = 0; // per-channel samples in/out over the timed launches
		std::string symbol; // device symbol of the stage's most recent timed launch (launch_symbol_last)
	};
	void* get_event(StageDev& d);
	// Convolver + whole-step interpolator of a chain with a fractional latency (minimum phase) as ONE launch: the shifts
	// that map the interpolator's emitted outputs onto the canonical stream the fused kernels compute (launch_fused)
	struct FusedShift
	{
		long long js; // canonical output J = emitted j + js
		long long d;  // convolver output time of J's window start = floor(J In / Out) + d
		int t_zero;   // the interpolator's stream starts at this convolver output
	};
	FusedShift fused_shift(size_t s) const;
	bool fuse_latency_ok(size_t s) const;
	void release();
	unsigned long long config_hash() const;
	bool stage_owns_ring(size_t s) const;

	void plan_transforms();
	void ensure_ring(size_t s);
	void ensure_work(size_t s, int slots, void* stream);
	void take_carried_tail(TailLaunch& T, int* carry);
	// half-band decimator s + convolver s + 1 as ONE launch (r8b_convp.h mode 20: the decimator taken in the block's load)
	bool fuse_hbconv(size_t s) const;
	bool hbconv_possible(size_t s) const;
	long long hbconv_history(size_t s) const;
	void launch_hbconv(size_t s, long long wa, long long wb, const SrcView& src, const DstView& dst, void* stream);
	bool fuse_with_next(size_t s) const;
	bool use_solo_fused(size_t s) const;
	bool use_pair(const ConvGeom& g) const;
	bool use_pair_fused(const ConvGeom& g) const;
	bool fast_geometry(const ConvGeom& g) const;
	enum { kPathGeneric, kPathConvx, kPathConvx3, kPathPair, kPathPair3, kPathPairP3 };
	// the geometry stage s runs with: the plan's, or its polyphase 3x block (ConvGeom::p3, option up3_poly)
	ConvGeom eff_geom(size_t s) const;
	int conv_path(const ConvGeom& g) const;
	bool latency_chain() const; // some stage carries fractional-latency state (minimum phase): no fusing
	bool use_pair_two(size_t s, int* run_off) const;
	bool half_worth(size_t s) const;
	void fused_blocking(size_t s, long long* S, long long* off) const;
	bool stage_parks(size_t s) const;
	int conv_once(size_t s, const DstView& dst) const;
	long long park_row_len(size_t s) const;
	long long park_len_of(size_t s, bool end_of_chain) const;
	void ensure_park(size_t s);
	void prepare_two_phase(size_t s);
	int group_len(size_t s) const;
	void launch_cascade(size_t s, int glen, long long fa, long long fb, const SrcView& src,
		const DstView& dst, void* stream);
	void launch_dcascade(size_t s, int glen, long long fa, long long fb, const SrcView& src,
		const DstView& dst, void* stream);
	long long stage_history(size_t s) const;
	void fill_conv(size_t s, ConvLaunch& L, const SrcView& src) const;
	void launch_fused(size_t s, long long wa, long long wb, const SrcView& src,
		const DstView& dst, void* stream);
	void launch_stage(size_t s, long long m_prev, long long a, long long b, const PolyState& ps,
		const SrcView& src, const DstView& dst, void* stream);

	ChainPlan plan_;
	int nch_;
	// channel window of the launches being issued (process() walks a convolver + polynomial-interpolator pair in
	// channel groups so that the stream between them stays in the 256 MB Infinity Cache; [0, nch_) otherwise)
	int ch0_ = 0, nchw_ = 0;
	int device_;
	std::vector<StageDev> dev_;
	std::map<std::string, int> opt_;
	std::map<std::string, long long> stat_;
	int io_in_fmt_ = kPcmF64, io_out_fmt_ = kPcmF64; // formats of the current call's buffers
	bool tail_done_ = false; // stage-0 history already written by the convolver kernel
	// the call's history copy while it waits for a launch to carry it (Engine::process, take_carried_tail)
	TailLaunch carry_tail_{};
	// launch_hbconv -> launch_stage: the half-band stage whose filter the convolver's launch takes in its load (-1: none),
	// and back: where the next call's first block starts reading the RAW stream (the history the call has to leave)
	long long hb_front_ = -1;
	long long hb_next_raw_ = 0;
	bool carry_ = false;
};

// complex twiddle table exp(-2 pi i e / len), exact on the axes; interleaved (re, im)
std::vector<double> make_twiddles(int len);
// zero-phase kernel spectrum H[m] = sum_t h[t] cos(2 pi m t / bl2), m = 0..bl2/2, times `scale`
std::vector<double> kernel_spectrum(const LpFilter& f, int bl2, double scale);
// general form: H[m] = scale * sum_n taps[n] exp(-2 pi i m (n - align) / bl2), m = 0..bl2/2 (interleaved re, im);
// align = ConvGeom::fl2
std::vector<double> kernel_spectrum_complex(const LpFilter& f, int bl2, int align, double scale);
// constants of the fast path's spectral stage (r8b_convx.h cx_spec_write) for a block convolver
// with forward complex length N = n_in/2 and backward length N2 = N*up (up in {1,2}): per slot
// s (bin kf = bitrev(s) for s < N/2, kf = N/2 for s == N/2) 4 (up 1) or 8 (up 2) complex values,
// constant c of slot s at [c*(N/2+1) + s]; interleaved (re, im).  H is the scaled kernel
// spectrum (kernel_spectrum), tw the exp(-2 pi i e / bl2) table.
std::vector<double> spectral_constants(const std::vector<double>& H, const std::vector<double>& tw,
	int bl2, int n_in, int up);
// kernel constants of the pair kernel's middle pass (r8b_convp.h): 8 x 256 pairs, entry c * 256 + t;
// 2x up (n_out = 2 n_in): (H[k] + H[k+N], H[k] - H[k+N]) for forward position 8 t + c, bin k =
// bitrev(position), N = n_in; 1:1: H of backward positions 16 t + 2 c and 16 t + 2 c + 1.  H is the
// scaled kernel spectrum (bl2/2 + 1 reals), mirrored for bins above bl2/2.
std::vector<double> pair_constants(const std::vector<double>& H, int n_in, int n_out);
// ... of the split 2x up-sampling form (r8b_convp.h cp_sp_middle): 16 x (n_in / 16) pairs, entry c * NT + t = (H[k] +
// H[k + n_in], H[k] - H[k + n_in]) for forward position 16 t + c, bin k = bitrev(position); H over 2 n_in points
std::vector<double> pair_constants_split(const std::vector<double>& H, int n_in);
// ... of the one-channel form (r8b_convp.h cp_solo_mid_b): 16 x (n / 16) pairs (a, b) per forward position 16 t + c of the
// n-point complex transform, bin k = bitrev(position): a = (H[k] + H[k + n]) - (H[k] - H[k + n]) sin(pi k / n),
// b = (H[k] - H[k + n]) cos(pi k / n); H over 2 n points
std::vector<double> pair_constants_solo(const std::vector<double>& H, int n);
// ... decimating by `down` = 2 or 4 (cp_solo_mid_b_down): per forward position 16 t + c, c a multiple of down -- kept bin k --:
// (H[k], H[n / down - k]);
// c + 1: (cos, sin) of pi k / n
std::vector<double> pair_constants_solo_down(const std::vector<double>& H, int n, int down);
// ... of the split form and of the one-channel form (1:1) with a complex kernel spectrum Hc (n + 1 complex bins, Hermitian
// beyond): 32 x (n / 16) complex entries -- split: H[k] + H[k+n], then (H[k] - H[k+n]) e^{+i pi k / n}; one-channel:
// A = (H[k] + H[k+n]) - (H[k] - H[k+n]) sin(pi k / n), then B = i (H[k] - H[k+n]) cos(pi k / n)
std::vector<double> pair_constants_split_complex(const std::vector<double>& Hc, int n);
// ... one-channel form decimating by 2: 24 x (n / 16) entries -- row c even: H[k]; c + 1: (cos, sin) of pi k / n; row
// 16 + c / 2: H[n / 2 - k]
std::vector<double> pair_constants_solo_down_complex(const std::vector<double>& Hc, int n);
std::vector<double> pair_constants_solo_complex(const std::vector<double>& Hc, int n);
// twiddle base powers of the pair kernel's passes per thread (r8b_convp.h ptw_fetch): 5 slots x 6 x 256
// complex; tw = exp(-2 pi i e / tw_len) table (interleaved), n_in = forward length (2048 or 4096)
std::vector<double> pair_twiddles(const std::vector<double>& tw, int tw_len, int n_in);
// radices (each in {2,4,8,16}, <= max_radix) whose product is N, largest first
std::vector<int> plan_radices(int N, int max_radix);

} // namespace r8bhip

#endif
