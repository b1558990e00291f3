/*
 * include/r8bsrc.h -- C ABI of libr8bsrc_hip.so, the MI355X-native batched sample-rate converter.
 *
 * Part 1 re-declares, symbol for symbol, what the reference's DLL exports
 * (reference DLL/r8bsrc.h:31-132, implemented there by DLL/r8bsrc.cpp:67-107), so that a program
 * linked against the reference's r8bsrc library links against this one unchanged.
 * Part 2 is additive: an N-channel batch object whose process() takes DEVICE (HBM) pointers --
 * this is the entry point the throughput numbers are quoted on.  Every signature uses plain
 * pointers and sizes only (no torch types, no C++ types).
 *
 * All functions are thread-compatible the same way the reference is (DLL/r8bsrc.h, README
 * "one object per stream"): distinct objects may be used from distinct threads, one object must
 * not be used concurrently.
 */
#ifndef R8BSRC_HIP_INCLUDED
#define R8BSRC_HIP_INCLUDED

#ifndef R8BSRC_DECL
	#define R8BSRC_DECL __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Part 1: drop-in symbols (reference DLL/r8bsrc.h)
 * ------------------------------------------------------------------------------------------- */

/* reference DLL/r8bsrc.h:31 */
typedef void* CR8BResampler;

/* reference DLL/r8bsrc.h:37-43; ReqAtten 136.45 / 109.56 / 180.15 dB
 * (reference CDSPResampler.h:746,777,807 via DLL/r8bsrc.cpp:67-85). */
enum ER8BResamplerRes
{
	r8brr16 = 0,
	r8brr16IR = 1,
	r8brr24 = 2
};

/* replaces reference DLL/r8bsrc.h:68-70 (r8b_create).  Returns NULL (after printing the reason to
 * stderr) when no MI355X/HIP device is usable: there is no CPU fallback. */
R8BSRC_DECL CR8BResampler r8b_create(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, enum ER8BResamplerRes Res);

/* replaces reference DLL/r8bsrc.h:79 */
R8BSRC_DECL void r8b_delete(CR8BResampler rs);

/* replaces reference DLL/r8bsrc.h:92 (CDSPResampler::getInputRequiredForOutput,
 * CDSPResampler.h:476-484) */
R8BSRC_DECL int r8b_inlen(CR8BResampler rs, int ReqOutSamples);

/* replaces reference DLL/r8bsrc.h:102 */
R8BSRC_DECL void r8b_clear(CR8BResampler rs);

/* replaces reference DLL/r8bsrc.h:131-132.  The reference declares `double*& op0`; at ABI level
 * that is a `double**` (DLL/r8bsrc.pas:31-32 binds it as `var op0: PR8BDouble`).  `ip0` and `*op0`
 * are HOST pointers, exactly like the reference: `*op0` points into a buffer owned by the object,
 * valid until the next call on it (CDSPResampler.h:546-552); if Src == Dst, `*op0 = ip0`. */
#ifdef __cplusplus
R8BSRC_DECL int r8b_process(CR8BResampler rs, double* ip0, int l, double*& op0);
#else
R8BSRC_DECL int r8b_process(CR8BResampler rs, double* ip0, int l, double** op0);
#endif

/* ---------------------------------------------------------------------------------------------
 * Part 2: N-channel batch object (additive; no counterpart in the reference, whose callers loop
 * "for each channel: process(same length)" -- example.cpp:63-67, bench/r8bfreesrc.cpp:120-124)
 * ------------------------------------------------------------------------------------------- */

typedef void* CR8BBatch;

/* Creates `nch` independent linear-phase resamplers that share one schedule, on HIP device
 * `device` (-1 = current).  ReqAtten is the reference constructor's ReqAtten in dB
 * (CDSPResampler.h:117-120).  NULL + message in r8b_last_error() on failure.
 *
 * Channel independence and accuracy.  The output of every channel equals the reference's for that channel's samples
 * within RMS 1e-15 / peak 1e-13 OF THAT CHANNEL'S OWN LEVEL (for full-scale +-1.0 noise: the absolute figures; measured
 * RMS 2-4e-16, the reference's own noise between two of its builds), and is bitwise independent of how the stream is
 * cut into calls.  Channels 2c and 2c+1 are convolved as the real and imaginary part of one complex transform (the
 * filter kernels are real); per transform block the quieter of the two is brought to its partner's binary order of
 * magnitude by an exact power of two and taken back afterwards, so a channel at -240 dBFS beside a full-scale partner
 * keeps its error at 1e-16 of its own level, as in the reference's one-object-per-channel use (README.md:52-55).  A
 * channel whose samples are all zero comes out as exact zeros whatever its partner carries (silence is detected per
 * transform block); an Inf / NaN in one channel of a pair reaches its partner's samples of that block (the reference
 * keeps it to its own object).  "pair_conv" = 0 (r8b_batch_set_option, before the first sample) selects the one-channel
 * kernels (slower).
 *
 * Block lengths.  Conversions whose REFERENCE block is 32768 points (a radix-3 ratio with a transition band of
 * 0.5 ... 0.6 %: 8 507 - 13 633 taps) run on 16384-point blocks of the same filter where overlap-save does not depend
 * on the block length (ratios 3/1, 1/3, 2/3), and on the REFERENCE'S OWN 32768-point block where the convolver also
 * decimates by 2 or 4 in the spectrum (ratios 3/2, 3/4: the residue of truncating a block's spectrum, -219 dB of the
 * signal, depends on the block length -- CDSPBlockConvolver.h:329-344); all of them meet the tolerance above (until
 * round 6 the two decimating ratios ran on 16384-point blocks and met 1e-13 / 1e-10 only).  Per-call output counts,
 * latency queries and chunk invariance are the reference's in every case (tests/cases.py REBLOCK_CASES, DESIGN.md
 * section 6).  Minimum phase (r8b_batch_create_ex): see DESIGN.md section 6 -- the bound there is the reference's own
 * run-to-run noise of its cepstral designer, the kernels meet 1e-15 on the reference's taps.
 * One sample the reference leaves undefined (a ONE-tap half-band up-sampler, >= 32x up-sampling below 55 dB: the
 * stream's first odd output reads an unwritten ring slot, CDSPHBUpsampler.h:606-693) is the filter's value here. */
R8BSRC_DECL CR8BBatch r8b_batch_create(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, double ReqAtten, int nch, int device);

R8BSRC_DECL void r8b_batch_delete(CR8BBatch b);
R8BSRC_DECL void r8b_batch_clear(CR8BBatch b);
R8BSRC_DECL int r8b_batch_channels(CR8BBatch b);
/* ordinal of the HIP device the object lives on (device = -1 at creation: the device that was
   current then).  Every entry point runs on that device and leaves the caller's current device as
   it found it. */
R8BSRC_DECL int r8b_batch_device(CR8BBatch b);

/* CDSPResampler::getMaxOutLen(0) for the MaxInLen given at creation (CDSPResampler.h:502-519):
 * the minimum per-channel capacity of the output buffer. */
R8BSRC_DECL int r8b_batch_max_out_len(CR8BBatch b);
/* The chain's residual fractional latency in OUTPUT samples: r8b::CDSPResampler::getLatencyFrac() (reference
 * CDSPResampler.h:491-494: the LatencyFrac the last stage reports, :688).  0.0 for linear-phase objects; for
 * fprMinPhase (r8b_batch_create_ex) what a caller that compensates latency has to add to getLatency() = 0. */
R8BSRC_DECL double r8b_batch_latency_frac(CR8BBatch b);
/* CDSPResampler::getInputRequiredForOutput / getInLenBeforeOutPos (CDSPResampler.h:476-484,406) */
R8BSRC_DECL int r8b_batch_inlen(CR8BBatch b, int ReqOutSamples);
R8BSRC_DECL int r8b_batch_inlen_before_outpos(CR8BBatch b, int OutPos);

/* One process() step for all channels.  d_in / d_out are DEVICE pointers, channel-major:
 * channel c's samples start at d_in + c*in_stride (doubles), l <= MaxInLen samples each;
 * channel c's output is written at d_out + c*out_stride (out_stride >= r8b_batch_max_out_len).
 * `stream` is a hipStream_t (NULL = default stream); the call only enqueues work on it.
 * Any 8-byte aligned rows work; rows that start on 16-byte boundaries (even strides, better multiples of 8
 * samples = 64 bytes) let the last stage store pairs of outputs as 16 bytes.  Fastest of all (optional): rows on a
 * 64-byte pitch and d_out advanced by (outputs this object has produced so far) mod 8 samples, so that output j of
 * the stream always lands at a column congruent to j mod 8 -- the fused kernels' 64-byte store pieces are then
 * whole aligned segments in every call (worth 3-4 % on 44100 -> 96000).
 * Returns the number of output samples produced per channel (identical for all channels and equal
 * to what the reference's process() returns for the same call sequence), or -1 on error. */
R8BSRC_DECL int r8b_batch_process(CR8BBatch b, const double* d_in, long long in_stride, int l,
	double* d_out, long long out_stride, void* stream);

/* Convenience: same with HOST buffers (copies through the device, synchronous). */
R8BSRC_DECL int r8b_batch_process_host(CR8BBatch b, const double* in, long long in_stride, int l,
	double* out, long long out_stride);

/* PCM boundary next to the hot path (the reference's command-line tool does
 * this on the host through CWaveFile, call sites bench/r8bfreesrc.cpp:103,134): one process()
 * step whose input and output are DEVICE buffers of PCM samples, converted by ingest / egress
 * kernels on `stream`, so that 2-4 bytes per sample cross PCIe and the HBM edge instead of 8.
 *   interleaved != 0: frame-major, sample (frame f, channel c) is element f*stride + c
 *                     (stride >= channel count; the usual WAV layout has stride == channels)
 *   interleaved == 0: planar, element c*stride + f
 * Strides are in samples.  Integer formats decode as value / 2^(bits-1) and encode as
 * round-to-nearest-even of v * 2^(bits-1), saturated, without dither; R8B_PCM_S24 is packed
 * 3-byte little-endian.  THIS CONVENTION IS THIS LIBRARY'S DEFINITION: the reference's converter (CWaveFile, from the
 * author's libvox) is not part of the reference sources, so its scale (2^(bits-1) vs 2^(bits-1) - 1), rounding and
 * dither cannot be pinned to it; the codec is bit-exact against its own specification (tests/test_pcm.py) and the fp64
 * core between the two conversions is pinned to the reference.  Returns the output frames produced, or -1 on error. */
enum r8b_pcm_format
{
	R8B_PCM_F64 = 0,
	R8B_PCM_F32 = 1,
	R8B_PCM_S16 = 2,
	R8B_PCM_S24 = 3,
	R8B_PCM_S32 = 4
};
R8BSRC_DECL int r8b_batch_process_pcm(CR8BBatch b, const void* d_in, int in_format,
	int in_interleaved, long long in_stride, int l, void* d_out, int out_format,
	int out_interleaved, long long out_stride, void* stream);
R8BSRC_DECL int r8b_pcm_sample_bytes(int format);

/* Checkpoint / resume of the streaming state of all channels (the reference
 * keeps this state inside each CDSPProcessor and offers only clear()).  The blob is host memory:
 * the schedule's counters plus every history ring and the park buffer (outputs a call's last block
 * has computed for the next call).  r8b_batch_state_size() is a constant of the object; save returns
 * the bytes written; load accepts only a blob saved by an object created with the same parameters and
 * options (checked as a whole before anything changes), after which the stream continues
 * bit-identically.  Blobs are not canonical: ring and buffer positions that no later call reads keep
 * whatever earlier calls left there, so two blobs of the same stream position taken after different ways
 * of cutting the stream into calls may differ in bytes (never in the stream that follows).  Both wait
 * for `stream` (the stream the process calls were enqueued on).  Return -1 on error. */
R8BSRC_DECL long long r8b_batch_state_size(CR8BBatch b);
R8BSRC_DECL long long r8b_batch_state_save(CR8BBatch b, void* buf, long long cap, void* stream);
R8BSRC_DECL int r8b_batch_state_load(CR8BBatch b, const void* buf, long long size, void* stream);

/* The same with the phase response of the low-pass filters selectable (reference CDSPResampler.h:117-120,
 * EDSPFilterPhaseResponse CDSPFIRFilter.h:28-45): ReqPhase 0 = fprLinearPhase, 1 = fprMinPhase.  A
 * minimum-phase chain carries fractional latencies from stage to stage like the reference does; its convolvers
 * run on the pair kernel with a complex kernel spectrum, the interpolator behind them unfused.  Samples agree
 * with the reference's to 1e-15 when both use the same taps (tests feed the reference's through
 * r8b_design_set_lp_provider of a test build); with the taps of THIS library's designer the streams differ by 3e-7 ... 6e-5 RMS
 * (-48 dB for 1/3-band filters at 180 dB), because the cepstral transform's result depends on the rounding noise
 * of the FFT that computes it. */
R8BSRC_DECL CR8BBatch r8b_batch_create_ex(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, double ReqAtten, int ReqPhase, int nch, int device);

/* Single DSP stage as a batch object (the reference's CDSPProcessor boundary,
 * CDSPProcessor.h:64-127), used by the stage-level parity tests:
 *   kind 0: CDSPBlockConvolver(getLPFilter(a=ReqNormFreq, b=ReqTransBand, c=ReqAtten, linear,
 *           d=ReqGain), i0=UpFactor, i1=DownFactor)           (CDSPBlockConvolver.h:62)
 *   kind 1: CDSPFracInterpolator(a=SrcRate, b=DstRate, c=ReqAtten, i0=IsThird)
 *                                                              (CDSPFracInterpolator.h:713)
 *   kind 2: CDSPHBUpsampler(a=ReqAtten, i0=SteepIndex, i1=IsThird)   (CDSPHBUpsampler.h:572)
 *   kind 3: CDSPHBDownsampler(a=ReqAtten, i0=SteepIndex, i1=IsThird) (CDSPHBDownsampler.h:47) */
R8BSRC_DECL CR8BBatch r8b_batch_create_stage(int kind, double a, double b, double c, double d,
	int i0, int i1, int MaxInLen, int nch, int device);

/* Number of stages and a printable description of the chain (what the reference prints through
 * R8BCONSOLE, r8bconf.h:31-42).  Returns the full length; writes at most cap-1 chars. */
R8BSRC_DECL int r8b_batch_describe(CR8BBatch b, char* buf, int cap);

/* Kernel tuning/instrumentation knob: name/value pairs understood by the engine
 * ("fuse", "conv_threads", ...).  Returns 0 if the knob exists.  Knobs that change where a stream's state lives
 * ("fuse", "fuse_hb", "fuse_hbd", "fold_tail", "fast_conv", "pair_conv", "pair_two", "pair_split", "pair_solo", "align_groups", "park", "fuse_latency")
 * or how a stream is rounded ("solo_fuse", "up3_poly", "half", "half_fused", "quad") are
 * refused (-1) once samples have been processed, until r8b_batch_clear().  "half" / "half_fused" (default 1: objects whose
 * largest call holds at least 512 workgroups of the stage -- channel pairs x overlap-save blocks --; 2: every object; 0: never): the half-array forms of the 2x up-sampling, 2x decimating and fused
 * 2048 -> 4096-point pair kernels -- the full-array kernels' arithmetic with three or four workgroups
 * per CU instead of two; results agree with theirs to rounding (RMS 4e-17), the choice is a constant of the object.  "park" (default 1): every overlap-save
 * block is computed once -- the block that holds a call's last output keeps what it holds of the next call in a
 * per-channel park buffer (or writes it ahead into the next stage's ring) instead of being computed again by the next
 * call; 0 restores the recomputation.  The output stream is bit for bit the same either way. */
R8BSRC_DECL int r8b_batch_set_option(CR8BBatch b, const char* name, int value);

/* Instrumentation: with option "timing" = 1 every stage launch is bracketed by HIP events on the
 * caller's stream.  r8b_batch_stage_timing waits for the pending events of `stage` and returns
 * the accumulated kernel milliseconds and launch count since the previous query (then resets),
 * the per-channel input/output sample counts those launches covered, and the kernel's name.
 * 0 on success. */
R8BSRC_DECL int r8b_batch_stage_count(CR8BBatch b);
/* Instrumentation: a counter of the engine since creation (-1: unknown name).  "conv_blocks": overlap-save blocks
 * per channel the compile-time-sized convolver kernels were launched for (the reference computes every block once,
 * CDSPBlockConvolver.h:283-350; so does this library where a call's last block parks what it holds of the next
 * call); "park_calls": calls that took outputs from the park buffer; "park_only_calls": calls served from it alone;
 * "pcm_staged_sides": planar PCM sides of r8b_batch_process_pcm calls that had to go through the staging rows (the
 * first / last stage's kernel is built for fp64 rows only); "walk_blocks": blocks per channel that ran on the walk
 * form of the fused pair kernel; "tail_launches": calls whose history copy (the input's tail for the next call,
 * CDSPBlockConvolver.h:296-305) needed a launch of its own -- no convolver kept it and no half-band launch carried it. */
R8BSRC_DECL long long r8b_batch_stat(CR8BBatch b, const char* name);
R8BSRC_DECL int r8b_batch_stage_timing(CR8BBatch b, int stage, double* ms_sum, int* launches,
	long long* in_samples, long long* out_samples, char* kernel, int cap);
/* The device symbol of `stage`'s most recent launch made with "timing" = 1, as rocprofv3 prints it without namespace
 * and argument list ("k_convp_walk<11, 1, 4, 24>"; r8b_batch_stage_timing's `kernel` is the engine's label for the
 * stage's form, e.g. "k_convp_whole" = convolver + interpolator in one launch).  "" before the first such launch.
 * 0 on success. */
R8BSRC_DECL int r8b_batch_stage_symbol(CR8BBatch b, int stage, char* symbol, int cap);

/* Last error message of the calling thread ("" if none). */
R8BSRC_DECL const char* r8b_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Part 3: host-side designer/plan queries (no device needed).  These expose the read-only
 * coefficient sets the kernels consume so that they can be compared with the reference's
 * (CDSPFIRFilter.h:220-537, CDSPFracInterpolator.h:61-189, CDSPHBUpsampler.h:47-552).
 * ------------------------------------------------------------------------------------------- */

/* Low-pass taps h[0..KernelLen) (zero-phase filter, centre at KernelLen/2, DC gain ReqGain).
 * Returns KernelLen; fills at most cap taps; *BlockLenBits and *Latency like
 * CDSPFIRFilter::getBlockLenBits/getLatency (CDSPFIRFilter.h:139-164). */
R8BSRC_DECL int r8b_design_lpfilter(double ReqNormFreq, double ReqTransBand, double ReqAtten,
	double ReqGain, int* BlockLenBits, int* Latency, double* taps, int cap);

/* The same with the phase response selectable (0 = fprLinearPhase, 1 = fprMinPhase; reference
 * CDSPFIRFilter.h:28-45).  Minimum phase: causal taps h[0..KernelLen-1]; *Latency / *LatencyFrac =
 * integer and fractional part of the group delay at DC (CDSPFIRFilter.h:476-484). */
R8BSRC_DECL int r8b_design_lpfilter_ex(double ReqNormFreq, double ReqTransBand, double ReqAtten,
	double ReqGain, int ReqPhase, int* BlockLenBits, int* Latency, double* LatencyFrac, double* taps,
	int cap);

#ifdef R8B_TEST_HOOKS
/* PARITY-TEST HOOK -- NOT part of the shipped library: libr8bsrc_hip.so neither exports this symbol nor contains its
 * code; only builds made with -DR8B_TEST_HOOKS do (tests/emul, tests/_build/libr8bsrc_hip_testhooks.so).  A
 * provider may supply the taps of a low-pass filter in place of the designer: it is asked on every designer cache miss
 * (cache unlocked) with the filter's parameters and returns the number of taps it wrote (<= cap; 0 = "not mine", the
 * designer runs), their group-delay split *Latency / *LatencyFrac and *BlockLenBits.  tests/ use it to feed the
 * REFERENCE's own minimum-phase taps (recovered from CDSPFIRFilter::getKernelBlock) through the kernels, which
 * separates kernel parity (1e-15) from the conditioning of the cepstral minimum-phase transform
 * (CDSPRealFFT.h:681-785), whose output depends on the rounding noise of the particular FFT used.  Filters obtained
 * under a provider are cached apart from designed ones; NULL removes it. */
typedef int (*r8b_lp_provider)(double ReqNormFreq, double ReqTransBand, double ReqAtten, double ReqGain,
	int ReqPhase, double* taps, int cap, int* Latency, double* LatencyFrac, int* BlockLenBits);
R8BSRC_DECL void r8b_design_set_lp_provider(r8b_lp_provider provider);
#endif

/* Fractional-delay bank (CDSPFracDelayFilterBank): rows 0..FilterFracs, FilterLen*ElementSize
 * doubles each, natural (unshuffled) element order.  FilterFracs = -1 selects the default
 * (CDSPFracInterpolator.h:79-84).  Returns the total number of doubles. */
R8BSRC_DECL int r8b_design_fracbank(int FilterFracs, int ElementSize, int InterpPoints,
	double ReqAtten, int IsThird, int* FilterLen, int* Fracs, double* table, int cap);

/* Half-band taps (CDSPHBUpsampler::getHBFilter / getHBFilterThird).  Returns the tap count. */
R8BSRC_DECL int r8b_design_hbfilter(double ReqAtten, int SteepIndex, int IsThird, double* taps,
	double* att);

/* getWholeStepping (CDSPFracInterpolator.h:644-673). */
R8BSRC_DECL int r8b_design_whole_stepping(double SSampleRate, double DSampleRate, int* InStep,
	int* OutStep);

/* The designer's caches are bounded like the reference's (r8bconf.h:90 R8B_FILTER_CACHE_MAX 96, :103
 * R8B_FRACBANK_CACHE_MAX 12; CDSPFIRFilter.h:598-694, CDSPFracInterpolator.h:516): least recently used entries that no
 * live object holds are dropped once a cache is full, so a host that sweeps ratios does not grow without limit; what a
 * live resampler uses stays alive with it.  Writes the number of entries held right now: low-pass filters,
 * fractional-delay banks, lane tables of the fused interpolator (one per ratio, at most 96). */
R8BSRC_DECL void r8b_design_cache_counts(int* Filters, int* FracBanks, int* LaneTables);

/* Host-only schedule object: the integer bookkeeping of a resampler without any device work.
 * r8b_plan_step feeds l input samples and returns how many output samples the reference's
 * process() returns for that call. */
typedef void* CR8BPlan;
R8BSRC_DECL CR8BPlan r8b_plan_create(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, double ReqAtten);
R8BSRC_DECL void r8b_plan_delete(CR8BPlan p);
R8BSRC_DECL void r8b_plan_clear(CR8BPlan p);
R8BSRC_DECL int r8b_plan_step(CR8BPlan p, int l);
R8BSRC_DECL int r8b_plan_max_out_len(CR8BPlan p);
R8BSRC_DECL int r8b_plan_inlen(CR8BPlan p, int ReqOutSamples);
R8BSRC_DECL int r8b_plan_inlen_before_outpos(CR8BPlan p, int OutPos);
R8BSRC_DECL int r8b_plan_describe(CR8BPlan p, char* buf, int cap);

R8BSRC_DECL const char* r8b_version(void);

#ifdef __cplusplus
}
#endif

#endif /* R8BSRC_HIP_INCLUDED */
