// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI shim around the *unmodified* reference headers, compiled where they
// lie under /root/reference by oracle/Makefile into oracle/_ref/libr8bref.so.
// Nothing of the reference is copied into this repository: this file only
// #includes the reference's public headers at build time and forwards calls.
//
// What it exposes (all `refx_*`), and the reference interface each wraps:
//   refx_create/process/...   r8b::CDSPResampler ctor/process/clear/
//                             getInputRequiredForOutput/getInLenBeforeOutPos/
//                             getMaxOutLen            (CDSPResampler.h:117,559,521,476,406,502)
//   refx_topology             the R8BCONSOLE lines the constructors print
//                             (r8bconf.h:31-42), captured into a string
//   refx_lpfilter             CDSPFIRFilterCache::getLPFilter + getKernelBlock
//                             (CDSPFIRFilter.h:598,171)
//   refx_fracbank             CDSPFracDelayFilterBank table (CDSPFracInterpolator.h:61-189)
//   refx_hbfilter             CDSPHBUpsampler::getHBFilter[Third] (CDSPHBUpsampler.h:47,331)
//   refx_stage_*              one CDSPProcessor stage on its own (BlockConvolver /
//                             FracInterpolator / HBUpsampler / HBDownsampler)
//   refx_bench                N-channel CPU baseline: one CDSPResampler per channel,
//                             channels statically split over std::threads, timing only
//                             the process() loop like bench/r8bfreesrc.cpp:118-126.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.

#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include <mutex>

static thread_local std::string* g_console = nullptr;

static void refx_console(const char* fmt, ...)
{
	if (g_console == nullptr) return;
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_console->append(buf);
}

#define R8BCONSOLE(...) refx_console(__VA_ARGS__)

#include "CDSPResampler.h"

using namespace r8b;

#define REFX_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------- front-end

REFX_API void* refx_create(double src, double dst, int maxin, double tb, double atten)
{
	return new CDSPResampler(src, dst, maxin, tb, atten, fprLinearPhase);
}

REFX_API void refx_delete(void* h) { delete (CDSPResampler*) h; }
REFX_API void refx_clear(void* h) { ((CDSPResampler*) h)->clear(); }

REFX_API int refx_process(void* h, const double* in, int l, double* out, int outcap)
{
	double* op = nullptr;
	const int n = ((CDSPResampler*) h)->process(const_cast<double*>(in), l, op);
	const int c = n < outcap ? n : outcap;
	if (c > 0 && out != nullptr) memcpy(out, op, (size_t) c * sizeof(double));
	return n;
}

REFX_API int refx_input_required(void* h, int nout)
{
	return ((CDSPResampler*) h)->getInputRequiredForOutput(nout);
}

REFX_API int refx_inlen_before_outpos(void* h, int pos)
{
	return ((CDSPResampler*) h)->getInLenBeforeOutPos(pos);
}

REFX_API int refx_inlen_before_outstart(void* h, int pos)
{
	return ((CDSPResampler*) h)->getInLenBeforeOutStart(pos);
}

REFX_API int refx_maxoutlen(void* h) { return ((CDSPResampler*) h)->getMaxOutLen(0); }
// CDSPResampler::getLatencyFrac (CDSPResampler.h:491-494)
REFX_API double refx_latency_frac(void* h) { return ((CDSPResampler*) h)->getLatencyFrac(); }

REFX_API int refx_topology(double src, double dst, int maxin, double tb, double atten,
	char* buf, int cap)
{
	std::string s;
	g_console = &s;
	{
		CDSPResampler r(src, dst, maxin, tb, atten, fprLinearPhase);
	}
	g_console = nullptr;
	const int n = (int) s.size() < cap - 1 ? (int) s.size() : cap - 1;
	if (cap > 0) { memcpy(buf, s.data(), (size_t) n); buf[n] = 0; }
	return (int) s.size();
}

// ---------------------------------------------------------------- tables

REFX_API int refx_lpfilter(double normfreq, double tb, double atten, double gain,
	int* kernellen, int* blocklenbits, int* latency, double* kernelblock, int cap)
{
	CDSPFIRFilter& f = CDSPFIRFilterCache::getLPFilter(normfreq, tb, atten,
		fprLinearPhase, gain);
	*kernellen = f.getKernelLen();
	*blocklenbits = f.getBlockLenBits();
	*latency = f.getLatency();
	const int n = 2 << f.getBlockLenBits();
	if (kernelblock != nullptr)
		memcpy(kernelblock, f.getKernelBlock(), (size_t) (n < cap ? n : cap) * sizeof(double));
	f.unref();
	return n;
}

// the same with the phase response selectable (0 linear, 1 minimum phase); kernelblock = the filter's
// spectrum block as the reference keeps it (CDSPRealFFT::forward of the normalised taps: [0] = DC,
// [1] = Nyquist, then (re, im) pairs, Ooura's e^{+i} sign); *latfrac = getLatencyFrac()
REFX_API int refx_lpfilter_ex(double normfreq, double tb, double atten, double gain, int phase,
	int* kernellen, int* blocklenbits, int* latency, double* latfrac, double* kernelblock, int cap)
{
	CDSPFIRFilter& f = CDSPFIRFilterCache::getLPFilter(normfreq, tb, atten,
		phase ? fprMinPhase : fprLinearPhase, gain);
	*kernellen = f.getKernelLen();
	*blocklenbits = f.getBlockLenBits();
	*latency = f.getLatency();
	*latfrac = f.getLatencyFrac();
	const int n = 2 << f.getBlockLenBits();
	if (kernelblock != nullptr)
		memcpy(kernelblock, f.getKernelBlock(), (size_t) (n < cap ? n : cap) * sizeof(double));
	f.unref();
	return n;
}

// front-end object with the phase response selectable
REFX_API void* refx_create_ex(double src, double dst, int maxin, double tb, double atten, int phase)
{
	return new CDSPResampler(src, dst, maxin, tb, atten, phase ? fprMinPhase : fprLinearPhase);
}

REFX_API int refx_fracbank(int fracs, int elsize, int interppoints, double atten,
	int third, int* fltlen, int* nfracs, double* table, int cap)
{
	CDSPFracDelayFilterBank fb(fracs, elsize, interppoints, atten, third != 0);
	*fltlen = fb.getFilterLen();
	*nfracs = fb.getFilterFracs();
	// rows 0..FilterFracs inclusive are addressable through operator[]
	const int n = fb.getFilterLen() * elsize * (fb.getFilterFracs() + 1);
	if (table != nullptr)
		memcpy(table, &fb[0], (size_t) (n < cap ? n : cap) * sizeof(double));
	return n;
}

REFX_API double refx_fracbank_round_atten(double atten, int third)
{
	CDSPFracDelayFilterBank::roundReqAtten(atten, third != 0);
	return atten;
}

REFX_API int refx_hbfilter(double atten, int steep, int third, double* taps, double* att)
{
	const double* flt;
	int fltt;
	double a;
	if (third) CDSPHBUpsampler::getHBFilterThird(atten, steep, flt, fltt, a);
	else CDSPHBUpsampler::getHBFilter(atten, steep, flt, fltt, a);
	if (taps != nullptr) memcpy(taps, flt, (size_t) fltt * sizeof(double));
	if (att != nullptr) *att = a;
	return fltt;
}

REFX_API int refx_whole_stepping(double ssr, double dsr, int* instep, int* outstep)
{
	return getWholeStepping(ssr, dsr, *instep, *outstep) ? 1 : 0;
}

// ---------------------------------------------------------------- single stages

struct RefxStage
{
	CDSPProcessor* p;
	std::vector<double> buf;
};

// kind 0: BlockConvolver(normfreq, tb, atten, gain, up, down)
// kind 1: FracInterpolator(src, dst, atten, third)
// kind 2: HBUpsampler(atten, steep, third)
// kind 3: HBDownsampler(atten, steep, third)
REFX_API void* refx_stage_create(int kind, double a, double b, double c, double d,
	int i0, int i1)
{
	RefxStage* s = new RefxStage();
	if (kind == 0)
		s->p = new CDSPBlockConvolver(CDSPFIRFilterCache::getLPFilter(a, b, c,
			fprLinearPhase, d), i0, i1, 0.0);
	else if (kind == 1)
		s->p = new CDSPFracInterpolator(a, b, c, i0 != 0, 0.0);
	else if (kind == 2)
		s->p = new CDSPHBUpsampler(a, i0, i1 != 0, 0.0);
	else
		s->p = new CDSPHBDownsampler(a, i0, i1 != 0, 0.0);
	return s;
}

REFX_API void refx_stage_delete(void* h)
{
	RefxStage* s = (RefxStage*) h;
	delete s->p;
	delete s;
}

REFX_API void refx_stage_clear(void* h) { ((RefxStage*) h)->p->clear(); }

REFX_API int refx_stage_maxoutlen(void* h, int maxin)
{
	return ((RefxStage*) h)->p->getMaxOutLen(maxin);
}

REFX_API int refx_stage_inlen_before_outpos(void* h, int pos)
{
	return ((RefxStage*) h)->p->getInLenBeforeOutPos(pos);
}

REFX_API int refx_stage_process(void* h, const double* in, int l, double* out, int outcap)
{
	RefxStage* s = (RefxStage*) h;
	const int mo = s->p->getMaxOutLen(l) + 64;
	if ((int) s->buf.size() < mo) s->buf.resize((size_t) mo);
	std::vector<double> tmp(in, in + l); // stages may write into their input
	double* op = s->buf.data();
	const int n = s->p->process(tmp.data(), l, op);
	const int cnt = n < outcap ? n : outcap;
	if (cnt > 0 && out != nullptr) memcpy(out, op, (size_t) cnt * sizeof(double));
	return n;
}

// ---------------------------------------------------------------- CPU baseline

static inline double splitmix_next(uint64_t& s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return (double) (z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

// Runs `nch` independent resamplers for `warm`+`calls` process() calls of L samples
// on `nthreads` threads; returns seconds spent in the timed process() loops (max over
// threads) and the total number of output samples through *outsamples.
REFX_API double refx_bench(double src, double dst, int L, double tb, double atten,
	int nch, int warm, int calls, int nthreads, long long* outsamples)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nch) nthreads = nch;
	std::vector<double> secs((size_t) nthreads, 0.0);
	std::vector<long long> outs((size_t) nthreads, 0);
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; t++)
	{
		th.emplace_back([&, t]()
		{
			const int c0 = (int) ((long long) nch * t / nthreads);
			const int c1 = (int) ((long long) nch * (t + 1) / nthreads);
			const int n = c1 - c0;
			std::vector<CDSPResampler*> rs((size_t) n);
			std::vector<uint64_t> seeds((size_t) n);
			std::vector<double> in((size_t) n * (size_t) L);
			for (int c = 0; c < n; c++)
			{
				rs[(size_t) c] = new CDSPResampler(src, dst, L, tb, atten, fprLinearPhase);
				seeds[(size_t) c] = (uint64_t) (1 + c0 + c);
			}
			double tsum = 0.0;
			long long osum = 0;
			for (int k = 0; k < warm + calls; k++)
			{
				for (int c = 0; c < n; c++)
					for (int i = 0; i < L; i++)
						in[(size_t) c * (size_t) L + (size_t) i] = splitmix_next(seeds[(size_t) c]);
				const auto t0 = std::chrono::steady_clock::now();
				long long o = 0;
				for (int c = 0; c < n; c++)
				{
					double* op;
					o += rs[(size_t) c]->process(&in[(size_t) c * (size_t) L], L, op);
				}
				const auto t1 = std::chrono::steady_clock::now();
				if (k >= warm)
				{
					tsum += std::chrono::duration<double>(t1 - t0).count();
					osum += o;
				}
			}
			for (int c = 0; c < n; c++) delete rs[(size_t) c];
			secs[(size_t) t] = tsum;
			outs[(size_t) t] = osum;
		});
	}
	for (auto& x : th) x.join();
	double mx = 0.0;
	long long os = 0;
	for (int t = 0; t < nthreads; t++)
	{
		if (secs[(size_t) t] > mx) mx = secs[(size_t) t];
		os += outs[(size_t) t];
	}
	if (outsamples != nullptr) *outsamples = os;
	return mx;
}

// Batch checker for the full-size parity tests: one reference resampler per channel walks `calls`
// process() calls of lens[k] samples (rows of x, xstride doubles apart, calls back to back) on
// `nthreads` threads and compares each call's output with the same call's segment of y (rows ystride
// apart, calls back to back; counts[k] = the count the device path returned for call k).  Per
// channel: sum of squared differences and peak difference.  Returns 0, or 1 + k for the first call
// whose output count differs from counts[k] (in any channel).
static int g_batch_phase = 0; // set by refx_batch_check_ex around the call (tests are single-threaded)

REFX_API int refx_batch_check(double src, double dst, int maxin, double tb, double atten, int nch,
	int calls, const int* lens, const double* x, long long xstride, const double* y, long long ystride,
	const int* counts, int nthreads, double* sqerr, double* peak, long long* total)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nch) nthreads = nch;
	std::vector<int> bad((size_t) nthreads, 0);
	std::vector<std::thread> th;
	long long tot = 0;
	for (int k = 0; k < calls; k++) tot += counts[k];
	for (int t = 0; t < nthreads; t++)
	{
		th.emplace_back([&, t]()
		{
			const int c0 = (int) ((long long) nch * t / nthreads);
			const int c1 = (int) ((long long) nch * (t + 1) / nthreads);
			for (int c = c0; c < c1; c++)
			{
				CDSPResampler rs(src, dst, maxin, tb, atten, g_batch_phase ? fprMinPhase : fprLinearPhase);
				const double* xi = x + (long long) c * xstride;
				const double* yi = y + (long long) c * ystride;
				double sq = 0.0, pk = 0.0;
				for (int k = 0; k < calls; k++)
				{
					double* op;
					// (process() may write into its input buffer's working copy only: pass a copy)
					std::vector<double> in(xi, xi + lens[k]);
					const int n = rs.process(in.data(), lens[k], op);
					if (n != counts[k])
					{
						if (bad[(size_t) t] == 0) bad[(size_t) t] = 1 + k;
						break;
					}
					for (int i = 0; i < n; i++)
					{
						const double d = yi[i] - op[i];
						sq += d * d;
						const double a = d < 0 ? -d : d;
						if (a > pk) pk = a;
					}
					xi += lens[k];
					yi += n;
				}
				sqerr[c] = sq;
				peak[c] = pk;
			}
		});
	}
	for (auto& v : th) v.join();
	if (total != nullptr) *total = tot;
	for (int t = 0; t < nthreads; t++)
		if (bad[(size_t) t] != 0) return bad[(size_t) t];
	return 0;
}

REFX_API int refx_batch_check_ex(double src, double dst, int maxin, double tb, double atten, int phase,
	int nch, int calls, const int* lens, const double* x, long long xstride, const double* y,
	long long ystride, const int* counts, int nthreads, double* sqerr, double* peak, long long* total)
{
	g_batch_phase = phase;
	const int rc = refx_batch_check(src, dst, maxin, tb, atten, nch, calls, lens, x, xstride, y, ystride,
		counts, nthreads, sqerr, peak, total);
	g_batch_phase = 0;
	return rc;
}

REFX_API const char* refx_version() { return R8B_VERSION; }
