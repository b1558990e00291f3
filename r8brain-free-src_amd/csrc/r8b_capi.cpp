// r8b_capi.cpp -- the C ABI of include/r8bsrc.h over r8bhip::Engine.
#include "../../include/r8bsrc.h"

#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "r8b_engine.h"

using namespace r8bhip;

namespace {

thread_local std::string g_err;

void set_err(const char* where, const std::exception& e)
{
	g_err = std::string(where) + ": " + e.what();
}

struct Batch
{
	std::unique_ptr<Engine> eng;
	// staging for the host-pointer entry points
	double* d_in = nullptr;
	double* d_out = nullptr;
	long long in_cap = 0, out_cap = 0; // per-channel capacities
	~Batch()
	{
		dev_free(d_in);
		dev_free(d_out);
	}
	void need_staging()
	{
		if (d_in) return;
		in_cap = eng->plan().max_in;
		out_cap = eng->plan().max_out_len > 0 ? eng->plan().max_out_len : 1;
		d_in = (double*) dev_alloc((size_t) in_cap * eng->channels() * sizeof(double));
		d_out = (double*) dev_alloc((size_t) out_cap * eng->channels() * sizeof(double));
	}
};

// the reference-shaped single-stream object (r8b_create ... r8b_process)
struct Single
{
	Batch b;
	std::vector<double> out;
};

double res_atten(int res)
{
	// reference DLL/r8bsrc.cpp:67-85 -> CDSPResampler16 / 16IR / 24 (CDSPResampler.h:746,777,807)
	switch (res)
	{
	case r8brr16: return 136.45;
	case r8brr16IR: return 109.56;
	default: return 180.15;
	}
}

int copy_text(const std::string& s, char* buf, int cap)
{
	if (buf != nullptr && cap > 0)
	{
		const int n = (int) s.size() < cap - 1 ? (int) s.size() : cap - 1;
		memcpy(buf, s.data(), (size_t) n);
		buf[n] = 0;
	}
	return (int) s.size();
}

int batch_process_host(Batch* h, const double* in, long long in_stride, int l, double* out,
	long long out_stride)
{
	h->need_staging();
	Engine& e = *h->eng;
	const int nch = e.channels();
	if (l > h->in_cap) throw std::runtime_error("input length exceeds MaxInLen");
	std::vector<double> tmp;
	if (l > 0)
	{
		if (in_stride == l && h->in_cap == l)
			dev_upload(h->d_in, in, (size_t) l * nch * sizeof(double));
		else
			for (int c = 0; c < nch; c++)
				dev_upload(h->d_in + (long long) c * h->in_cap, in + (long long) c * in_stride,
					(size_t) l * sizeof(double));
	}
	const int n = e.process(h->d_in, h->in_cap, l, h->d_out, h->out_cap, nullptr);
	dev_sync(nullptr);
	dev_check_last("r8b process");
	for (int c = 0; c < nch && n > 0; c++)
		dev_download(out + (long long) c * out_stride, h->d_out + (long long) c * h->out_cap,
			(size_t) n * sizeof(double), nullptr);
	return n;
}

// PCM boundary.  Planar buffers are decoded by the first stage's loads and encoded by the last
// stage's stores (Engine::process_planar); an interleaved side goes through a transposing kernel
// and the staging rows.  Everything is enqueued on `stream`, nothing synchronises.
int batch_process_pcm(Batch* h, const void* d_in, int in_fmt, int in_interleaved,
	long long in_stride, int l, void* d_out, int out_fmt, int out_interleaved, long long out_stride,
	void* stream)
{
	Engine& e = *h->eng;
	auto valid = [](int f) { return f >= kPcmF64 && f <= kPcmS32; };
	if (!valid(in_fmt) || !valid(out_fmt)) throw std::runtime_error("unknown PCM sample format");
	if (l < 0 || l > e.plan().max_in) throw std::runtime_error("input length exceeds MaxInLen");
	if (l == 0) return 0;
	// Src == Dst has no stage to fuse into: both sides staged
	const bool pass = e.plan().stages.empty();
	// ... and so has a planar PCM side whose first / last stage is a compile-time-sized convolver (fp64 views only)
	const bool stage_in = in_interleaved || pass || (in_fmt != kPcmF64 && !e.pcm_fused_in());
	const bool stage_out = out_interleaved || pass || (out_fmt != kPcmF64 && !e.pcm_fused_out());
	if (stage_in || stage_out) h->need_staging();
	if (stage_in && !in_interleaved && !pass) e.bump("pcm_staged_sides");
	if (stage_out && !out_interleaved && !pass) e.bump("pcm_staged_sides");
	PcmLaunch P;
	P.nch = e.channels();
	if (stage_in)
	{
		P.pcm = const_cast<void*>(d_in);
		P.fmt = in_fmt;
		P.interleaved = in_interleaved ? 1 : 0;
		P.pcm_stride = in_stride;
		P.planar = h->d_in;
		P.planar_stride = h->in_cap;
		P.n = l;
		launch_pcm_in(P, stream);
		d_in = h->d_in;
		in_fmt = kPcmF64;
		in_stride = h->in_cap;
	}
	void* const out = stage_out ? h->d_out : d_out;
	const int n = pass ? e.process(static_cast<const double*>(d_in), in_stride, l, h->d_out,
		h->out_cap, stream) : e.process_planar(d_in, in_fmt, in_stride, l, out,
		stage_out ? (int) kPcmF64 : out_fmt, stage_out ? (long long) h->out_cap : out_stride, stream);
	if (stage_out && n > 0)
	{
		P.pcm = d_out;
		P.fmt = out_fmt;
		P.interleaved = out_interleaved ? 1 : 0;
		P.pcm_stride = out_stride;
		P.planar = h->d_out;
		P.planar_stride = h->out_cap;
		P.n = n;
		launch_pcm_out(P, stream);
	}
	return n;
}

} // namespace

extern "C" {

R8BSRC_DECL const char* r8b_last_error(void) { return g_err.c_str(); }
R8BSRC_DECL const char* r8b_version(void) { return "r8bsrc-hip 0.1 (gfx950; tracks r8brain-free-src 7.1 semantics)"; }

// ---------------------------------------------------------------- batch object

R8BSRC_DECL CR8BBatch r8b_batch_create(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, double ReqAtten, int nch, int device)
{
	try
	{
		std::unique_ptr<Batch> b(new Batch());
		b->eng.reset(new Engine(build_topology(SrcSampleRate, DstSampleRate, ReqTransBand,
			ReqAtten), MaxInLen, nch, device));
		g_err.clear();
		return b.release();
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_create", e);
		return nullptr;
	}
}

R8BSRC_DECL CR8BBatch r8b_batch_create_ex(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, double ReqAtten, int ReqPhase, int nch, int device)
{
	try
	{
		if (ReqPhase != kLinearPhase && ReqPhase != kMinPhase)
			throw std::runtime_error("ReqPhase must be 0 (linear phase) or 1 (minimum phase)");
		std::unique_ptr<Batch> b(new Batch());
		b->eng.reset(new Engine(build_topology(SrcSampleRate, DstSampleRate, ReqTransBand,
			ReqAtten, ReqPhase), MaxInLen, nch, device));
		g_err.clear();
		return b.release();
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_create_ex", e);
		return nullptr;
	}
}

R8BSRC_DECL CR8BBatch r8b_batch_create_stage(int kind, double a, double b, double c, double d,
	int i0, int i1, int MaxInLen, int nch, int device)
{
	try
	{
		if (kind < 0 || kind > 3) throw std::runtime_error("stage kind must be 0 (convolver), 1 "
			"(interpolator), 2 (half-band up) or 3 (half-band down)");
		StageDesc sd;
		sd.kind = (StageKind) kind;
		sd.a = a; sd.b = b; sd.c = c; sd.d = d; sd.i0 = i0; sd.i1 = i1;
		std::unique_ptr<Batch> bo(new Batch());
		bo->eng.reset(new Engine(std::vector<StageDesc>(1, sd), MaxInLen, nch, device));
		g_err.clear();
		return bo.release();
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_create_stage", e);
		return nullptr;
	}
}

// (the entries that can report an error refuse a NULL handle; the plain getters, like the reference's, do not test it)
static Batch* need(CR8BBatch b)
{
	if (b == nullptr) throw std::runtime_error("null handle");
	return (Batch*) b;
}

R8BSRC_DECL void r8b_batch_delete(CR8BBatch b) { delete (Batch*) b; }

R8BSRC_DECL void r8b_batch_clear(CR8BBatch b)
{
	if (b) ((Batch*) b)->eng->clear();
}

R8BSRC_DECL int r8b_batch_channels(CR8BBatch b) { return ((Batch*) b)->eng->channels(); }

R8BSRC_DECL int r8b_batch_device(CR8BBatch b) { return ((Batch*) b)->eng->device(); }

R8BSRC_DECL int r8b_batch_max_out_len(CR8BBatch b) { return ((Batch*) b)->eng->plan().max_out_len; }

R8BSRC_DECL int r8b_batch_inlen(CR8BBatch b, int ReqOutSamples)
{
	return ((Batch*) b)->eng->plan().input_required(ReqOutSamples);
}

R8BSRC_DECL int r8b_batch_inlen_before_outpos(CR8BBatch b, int OutPos)
{
	return ((Batch*) b)->eng->plan().in_len_before_out_pos(OutPos);
}

R8BSRC_DECL int r8b_batch_process(CR8BBatch b, const double* d_in, long long in_stride, int l,
	double* d_out, long long out_stride, void* stream)
{
	try
	{
		return need(b)->eng->process(d_in, in_stride, l, d_out, out_stride, stream);
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_process", e);
		return -1;
	}
}

R8BSRC_DECL int r8b_batch_process_host(CR8BBatch b, const double* in, long long in_stride, int l,
	double* out, long long out_stride)
{
	try
	{
		return batch_process_host(need(b), in, in_stride, l, out, out_stride);
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_process_host", e);
		return -1;
	}
}

R8BSRC_DECL int r8b_batch_process_pcm(CR8BBatch b, const void* d_in, int in_format,
	int in_interleaved, long long in_stride, int l, void* d_out, int out_format,
	int out_interleaved, long long out_stride, void* stream)
{
	try
	{
		return batch_process_pcm(need(b), d_in, in_format, in_interleaved, in_stride, l, d_out,
			out_format, out_interleaved, out_stride, stream);
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_process_pcm", e);
		return -1;
	}
}

R8BSRC_DECL int r8b_pcm_sample_bytes(int format)
{
	switch (format)
	{
	case kPcmF64: return 8;
	case kPcmF32: return 4;
	case kPcmS16: return 2;
	case kPcmS24: return 3;
	case kPcmS32: return 4;
	}
	return 0;
}

R8BSRC_DECL long long r8b_batch_state_size(CR8BBatch b)
{
	return (long long) ((Batch*) b)->eng->state_size();
}

R8BSRC_DECL long long r8b_batch_state_save(CR8BBatch b, void* buf, long long cap, void* stream)
{
	try
	{
		return (long long) need(b)->eng->save_state(buf, cap < 0 ? 0 : (size_t) cap, stream);
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_state_save", e);
		return -1;
	}
}

R8BSRC_DECL int r8b_batch_state_load(CR8BBatch b, const void* buf, long long size, void* stream)
{
	try
	{
		need(b)->eng->load_state(buf, size < 0 ? 0 : (size_t) size, stream);
		return 0;
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_state_load", e);
		return -1;
	}
}

R8BSRC_DECL int r8b_batch_describe(CR8BBatch b, char* buf, int cap)
{
	return copy_text(((Batch*) b)->eng->plan().describe(), buf, cap);
}

R8BSRC_DECL int r8b_batch_set_option(CR8BBatch b, const char* name, int value)
{
	return ((Batch*) b)->eng->set_option(name, value) ? 0 : -1;
}

R8BSRC_DECL double r8b_batch_latency_frac(CR8BBatch b)
{
	// (what the last stage hands on: reference CDSPResampler.h:688, addProcessor)
	const r8bhip::ChainPlan& pl = ((Batch*) b)->eng->plan();
	return pl.stages.empty() ? 0.0 : pl.stages.back().lat_frac;
}

R8BSRC_DECL long long r8b_batch_stat(CR8BBatch b, const char* name)
{
	return name != nullptr ? ((Batch*) b)->eng->stat(name) : -1;
}

R8BSRC_DECL int r8b_batch_stage_count(CR8BBatch b)
{
	return (int) ((Batch*) b)->eng->plan().stages.size();
}

R8BSRC_DECL int r8b_batch_stage_timing(CR8BBatch b, int stage, double* ms_sum, int* launches,
	long long* in_samples, long long* out_samples, char* kernel, int cap)
{
	try
	{
		std::string name;
		if (!need(b)->eng->stage_timing((size_t) stage, ms_sum, launches, &name,
			in_samples, out_samples)) return -1;
		copy_text(name, kernel, cap);
		return 0;
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_stage_timing", e);
		return -1;
	}
}

R8BSRC_DECL int r8b_batch_stage_symbol(CR8BBatch b, int stage, char* symbol, int cap)
{
	try
	{
		Batch* B = need(b);
		if (stage < 0 || stage >= (int) B->eng->plan().stages.size()) return -1;
		copy_text(B->eng->stage_symbol((size_t) stage), symbol, cap);
		return 0;
	}
	catch (const std::exception& e)
	{
		set_err("r8b_batch_stage_symbol", e);
		return -1;
	}
}

// ---------------------------------------------------------------- drop-in single-stream ABI

R8BSRC_DECL CR8BResampler r8b_create(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, enum ER8BResamplerRes Res)
{
	try
	{
		std::unique_ptr<Single> s(new Single());
		s->b.eng.reset(new Engine(build_topology(SrcSampleRate, DstSampleRate, ReqTransBand,
			res_atten((int) Res)), MaxInLen, 1, -1));
		s->out.resize((size_t) (s->b.eng->plan().max_out_len > 0 ?
			s->b.eng->plan().max_out_len : 1));
		g_err.clear();
		return s.release();
	}
	catch (const std::exception& e)
	{
		// the reference has no error path; a missing GPU must not go unnoticed
		set_err("r8b_create", e);
		fprintf(stderr, "r8bsrc-hip: %s\n", g_err.c_str());
		return nullptr;
	}
}

R8BSRC_DECL void r8b_delete(CR8BResampler rs) { delete (Single*) rs; }

R8BSRC_DECL int r8b_inlen(CR8BResampler rs, int ReqOutSamples)
{
	return ((Single*) rs)->b.eng->plan().input_required(ReqOutSamples);
}

R8BSRC_DECL void r8b_clear(CR8BResampler rs) { ((Single*) rs)->b.eng->clear(); }

R8BSRC_DECL int r8b_process(CR8BResampler rs, double* ip0, int l, double*& op0)
{
	Single* s = (Single*) rs;
	try
	{
		if (s->b.eng->plan().stages.empty())
		{
			op0 = ip0; // reference CDSPResampler.h:534-535
			return l;
		}
		op0 = s->out.data();
		return batch_process_host(&s->b, ip0, l, l, s->out.data(), (long long) s->out.size());
	}
	catch (const std::exception& e)
	{
		set_err("r8b_process", e);
		fprintf(stderr, "r8bsrc-hip: %s\n", g_err.c_str());
		op0 = s->out.data();
		return 0;
	}
}

// ---------------------------------------------------------------- designer / plan queries

R8BSRC_DECL int r8b_design_lpfilter(double ReqNormFreq, double ReqTransBand, double ReqAtten,
	double ReqGain, int* BlockLenBits, int* Latency, double* taps, int cap)
{
	const LpFilterRef fr = design_lp(ReqNormFreq, ReqTransBand, ReqAtten, ReqGain);
	const LpFilter& f = *fr;
	if (BlockLenBits) *BlockLenBits = f.block_len_bits;
	if (Latency) *Latency = f.fl2;
	if (taps)
		memcpy(taps, f.taps.data(), sizeof(double) * (size_t) (f.kernel_len < cap ? f.kernel_len : cap));
	return f.kernel_len;
}

R8BSRC_DECL int r8b_design_lpfilter_ex(double ReqNormFreq, double ReqTransBand, double ReqAtten,
	double ReqGain, int ReqPhase, int* BlockLenBits, int* Latency, double* LatencyFrac, double* taps,
	int cap)
{
	const LpFilterRef fr = design_lp(ReqNormFreq, ReqTransBand, ReqAtten, ReqGain, ReqPhase != 0);
	const LpFilter& f = *fr;
	if (BlockLenBits) *BlockLenBits = f.block_len_bits;
	if (Latency) *Latency = f.fl2;
	if (LatencyFrac) *LatencyFrac = f.lat_frac;
	if (taps)
		memcpy(taps, f.taps.data(), sizeof(double) * (size_t) (f.kernel_len < cap ? f.kernel_len : cap));
	return f.kernel_len;
}

#ifdef R8B_TEST_HOOKS
R8BSRC_DECL void r8b_design_set_lp_provider(r8b_lp_provider provider)
{
	set_lp_provider(reinterpret_cast<LpProvider>(provider));
}
#endif

R8BSRC_DECL int r8b_design_fracbank(int FilterFracs, int ElementSize, int InterpPoints,
	double ReqAtten, int IsThird, int* FilterLen, int* Fracs, double* table, int cap)
{
	if (!((ElementSize == 1 && InterpPoints == 2) || (ElementSize == 3 && InterpPoints == 8)))
		return -1;
	const FracBankRef br = design_frac_bank(FilterFracs, ElementSize, InterpPoints, ReqAtten,
		IsThird != 0);
	const FracBank& b = *br;
	if (FilterLen) *FilterLen = b.filter_len;
	if (Fracs) *Fracs = b.fracs;
	const int n = (int) b.table.size();
	if (table) memcpy(table, b.table.data(), sizeof(double) * (size_t) (n < cap ? n : cap));
	return n;
}

R8BSRC_DECL void r8b_design_cache_counts(int* Filters, int* FracBanks, int* LaneTables)
{
	int c[3] = { 0, 0, 0 };
	design_cache_counts(c);
	if (Filters) *Filters = c[0];
	if (FracBanks) *FracBanks = c[1];
	if (LaneTables) *LaneTables = c[2];
}

R8BSRC_DECL int r8b_design_hbfilter(double ReqAtten, int SteepIndex, int IsThird, double* taps,
	double* att)
{
	const double* t;
	double a;
	const int n = select_hb_filter(ReqAtten, SteepIndex, IsThird != 0, &t, &a);
	if (taps) memcpy(taps, t, sizeof(double) * (size_t) n);
	if (att) *att = a;
	return n;
}

R8BSRC_DECL int r8b_design_whole_stepping(double SSampleRate, double DSampleRate, int* InStep,
	int* OutStep)
{
	int i = 0, o = 0;
	const bool ok = whole_stepping(SSampleRate, DSampleRate, &i, &o);
	if (InStep) *InStep = i;
	if (OutStep) *OutStep = o;
	return ok ? 1 : 0;
}

R8BSRC_DECL CR8BPlan r8b_plan_create(double SrcSampleRate, double DstSampleRate, int MaxInLen,
	double ReqTransBand, double ReqAtten)
{
	try
	{
		ChainPlan* p = new ChainPlan();
		p->init(build_topology(SrcSampleRate, DstSampleRate, ReqTransBand, ReqAtten), MaxInLen);
		return p;
	}
	catch (const std::exception& e)
	{
		set_err("r8b_plan_create", e);
		return nullptr;
	}
}

R8BSRC_DECL void r8b_plan_delete(CR8BPlan p) { delete (ChainPlan*) p; }
R8BSRC_DECL void r8b_plan_clear(CR8BPlan p) { ((ChainPlan*) p)->clear(); }

R8BSRC_DECL int r8b_plan_step(CR8BPlan p, int l)
{
	ChainPlan* c = (ChainPlan*) p;
	int n = l;
	for (StagePlan& s : c->stages)
	{
		long long a, b;
		s.step(n, &a, &b, nullptr);
		n = (int) (b - a);
	}
	return n;
}

R8BSRC_DECL int r8b_plan_max_out_len(CR8BPlan p)
{
	ChainPlan* c = (ChainPlan*) p;
	return c->stages.empty() ? c->max_in : c->max_out_len;
}

R8BSRC_DECL int r8b_plan_inlen(CR8BPlan p, int n) { return ((ChainPlan*) p)->input_required(n); }

R8BSRC_DECL int r8b_plan_inlen_before_outpos(CR8BPlan p, int pos)
{
	return ((ChainPlan*) p)->in_len_before_out_pos(pos);
}

R8BSRC_DECL int r8b_plan_describe(CR8BPlan p, char* buf, int cap)
{
	return copy_text(((ChainPlan*) p)->describe(), buf, cap);
}

} // extern "C"
