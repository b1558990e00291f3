// r8b_engine.cpp -- see r8b_engine.h.  Host C++ only; every device operation goes through
// r8b_launch.h.
#include "r8b_engine.h"

#include <climits>
#include <cmath>
#include <stdexcept>

namespace r8bhip {

std::vector<double> make_twiddles(int len)
{
	std::vector<double> t((size_t) len * 2);
	const long double two_pi = 6.283185307179586476925286766559L;
	const int quarter = len / 4;
	for (int e = 0; e < len; e++)
	{
		double c, s;
		if (quarter > 0 && len % 4 == 0)
		{
			const int quad = e / quarter, rem = e - quad * quarter;
			long double c0 = 1.0L, s0 = 0.0L;
			if (rem != 0)
			{
				// evaluate in the first octant pair and reflect for accuracy/symmetry
				if (2 * rem <= quarter)
				{
					c0 = cosl(two_pi * rem / len);
					s0 = sinl(two_pi * rem / len);
				}
				else
				{
					const int r2 = quarter - rem;
					c0 = sinl(two_pi * r2 / len);
					s0 = cosl(two_pi * r2 / len);
				}
			}
			switch (quad)
			{
			case 0: c = (double) c0; s = (double) s0; break;
			case 1: c = (double) -s0; s = (double) c0; break;
			case 2: c = (double) -c0; s = (double) -s0; break;
			default: c = (double) s0; s = (double) -c0; break;
			}
		}
		else
		{
			c = (double) cosl(two_pi * e / len);
			s = (double) sinl(two_pi * e / len);
		}
		t[(size_t) e * 2] = c;
		t[(size_t) e * 2 + 1] = -s;
	}
	return t;
}

std::vector<double> kernel_spectrum(const LpFilter& f, int bl2, double scale)
{
	const std::vector<double> tw = make_twiddles(bl2);
	std::vector<double> H((size_t) bl2 / 2 + 1);
	const double* c = &f.taps[(size_t) f.fl2];
	for (int m = 0; m <= bl2 / 2; m++)
	{
		long double acc = c[0];
		for (int t = 1; t <= f.fl2; t++)
		{
			const int e = (int) (((long long) m * t) & (bl2 - 1));
			acc += 2.0L * c[t] * tw[(size_t) e * 2];
		}
		H[(size_t) m] = (double) (acc * scale);
	}
	return H;
}

std::vector<int> plan_radices(int N, int max_radix)
{
	std::vector<int> r;
	if (max_radix < 2) max_radix = 2;
	if (max_radix > 16) max_radix = 16;
	int bits = 0;
	while ((1 << bits) < N) bits++;
	int mb = 0;
	while ((2 << mb) <= max_radix) mb++;
	// spread the bits as evenly as possible over the fewest passes
	const int np = (bits + mb - 1) / mb;
	for (int i = 0; i < np; i++)
	{
		const int b = (bits + np - 1 - i) / np;
		r.push_back(1 << b);
	}
	return r;
}

static long long pow2_at_least(long long v)
{
	long long p = 1;
	while (p < v) p <<= 1;
	return p;
}

Engine::Engine(const std::vector<StageDesc>& descs, int maxin, int nch, int device)
	: nch_(nch), device_(device)
{
	if (nch < 1) throw std::runtime_error("channel count must be >= 1");
	if (maxin < 1) throw std::runtime_error("MaxInLen must be >= 1");
	dev_select(device);
	plan_.init(descs, maxin);
	opt_["conv_radix"] = 8;
	opt_["conv_threads"] = 256;
	opt_["whole_tile"] = 1024;
	opt_["hb_tile"] = 1024;
	opt_["timing"] = 0;
	dev_.resize(plan_.stages.size());
	for (size_t s = 0; s < plan_.stages.size(); s++)
	{
		const StagePlan& sp = plan_.stages[s];
		StageDev& d = dev_[s];
		const long long hist = sp.history();
		d.ring_size = pow2_at_least(s == 0 ? hist : hist + plan_.stage_max_in[s]);
		d.ring = (double*) dev_alloc((size_t) d.ring_size * (size_t) nch * sizeof(double));
		if (sp.desc.kind == kConv)
		{
			const ConvGeom& g = sp.cg;
			if (g.n_in < 32 || g.n_out < 32)
				throw std::runtime_error("block convolver transform too short");
			if ((size_t) (g.n_in + g.n_out) * sizeof(double) > 160 * 1024)
				throw std::runtime_error("low-pass filter too long for the LDS-resident "
					"block convolver (transition band too narrow)");
			const std::vector<double> H = kernel_spectrum(*sp.lp, g.bl2, 1.0 / g.bl2);
			d.H = (double*) dev_alloc(H.size() * sizeof(double));
			dev_upload(d.H, H.data(), H.size() * sizeof(double));
			const std::vector<double> tw = make_twiddles(g.bl2);
			d.tw_len = g.bl2;
			d.tw = (cd*) dev_alloc(tw.size() * sizeof(double));
			dev_upload(d.tw, tw.data(), tw.size() * sizeof(double));
		}
		else if (sp.desc.kind == kFrac)
		{
			const std::vector<double>& t = sp.bank->table;
			d.table = (double*) dev_alloc(t.size() * sizeof(double));
			dev_upload(d.table, t.data(), t.size() * sizeof(double));
		}
	}
	plan_transforms();
}

Engine::~Engine()
{
	for (StageDev& d : dev_)
	{
		for (auto& pr : d.pending)
		{
			dev_event_destroy(pr.first);
			dev_event_destroy(pr.second);
		}
		for (void* e : d.free_events) dev_event_destroy(e);
		dev_free(d.ring);
		dev_free(d.H);
		dev_free(d.tw);
		dev_free(d.table);
	}
}

void Engine::plan_transforms()
{
	for (size_t s = 0; s < plan_.stages.size(); s++)
	{
		const StagePlan& sp = plan_.stages[s];
		if (sp.desc.kind != kConv) continue;
		dev_[s].fwd_radix = plan_radices(sp.cg.n_in / 2, opt_["conv_radix"]);
		std::vector<int> inv = plan_radices(sp.cg.n_out / 2, opt_["conv_radix"]);
		// backward passes run with growing sub-transform length: smallest radix group first
		dev_[s].inv_radix.assign(inv.rbegin(), inv.rend());
	}
}

bool Engine::set_option(const std::string& name, int value)
{
	auto it = opt_.find(name);
	if (it == opt_.end()) return false;
	it->second = value;
	plan_transforms();
	return true;
}

void* Engine::get_event(StageDev& d)
{
	if (!d.free_events.empty())
	{
		void* e = d.free_events.back();
		d.free_events.pop_back();
		return e;
	}
	return dev_event_create();
}

bool Engine::stage_timing(size_t stage, double* ms_sum, int* launches, std::string* kernel,
	long long* in_samples, long long* out_samples)
{
	if (stage >= dev_.size()) return false;
	StageDev& d = dev_[stage];
	for (auto& pr : d.pending)
	{
		d.ms_sum += dev_event_elapsed_ms(pr.first, pr.second);
		d.launches++;
		d.free_events.push_back(pr.first);
		d.free_events.push_back(pr.second);
	}
	d.pending.clear();
	if (ms_sum) *ms_sum = d.ms_sum;
	if (launches) *launches = d.launches;
	if (in_samples) *in_samples = d.t_in;
	if (out_samples) *out_samples = d.t_out;
	if (kernel)
	{
		const StagePlan& sp = plan_.stages[stage];
		switch (sp.desc.kind)
		{
		case kConv: *kernel = "k_conv"; break;
		case kFrac: *kernel = sp.whole ? "k_whole" : "k_poly"; break;
		case kHBUp: *kernel = "k_hbup"; break;
		case kHBDown: *kernel = "k_hbdown"; break;
		}
	}
	d.ms_sum = 0.0;
	d.launches = 0;
	d.t_in = d.t_out = 0;
	return true;
}

void Engine::clear()
{
	// ring contents need no reset: positions restart at 0 and every position >= 0 is rewritten
	// before it is read again, positions < 0 read as zero by construction
	plan_.clear();
}

void Engine::launch_stage(size_t s, long long m_prev, long long a, long long b,
	const PolyState& ps, const SrcView& src, const DstView& dst, void* stream)
{
	const StagePlan& sp = plan_.stages[s];
	const StageDev& d = dev_[s];
	(void) m_prev;
	switch (sp.desc.kind)
	{
	case kConv:
	{
		const ConvGeom& g = sp.cg;
		ConvLaunch L;
		L.up = g.up; L.down = g.down; L.fl2 = g.fl2; L.bl2 = g.bl2; L.in_len = g.in_len;
		L.n_in = g.n_in; L.n_out = g.n_out;
		L.up_pow2 = g.up_pow2 ? 1 : 0;
		L.down_pow2 = g.down_pow2 ? 1 : 0;
		L.n_fwd = (int) d.fwd_radix.size();
		L.n_inv = (int) d.inv_radix.size();
		if (L.n_fwd > kMaxPasses || L.n_inv > kMaxPasses)
			throw std::runtime_error("transform plan too deep");
		for (int i = 0; i < L.n_fwd; i++) L.fwd_radix[i] = d.fwd_radix[(size_t) i];
		for (int i = 0; i < L.n_inv; i++) L.inv_radix[i] = d.inv_radix[(size_t) i];
		L.H = d.H; L.tw = d.tw; L.tw_len = d.tw_len;
		L.k0 = ((long long) g.down * a + g.fl2) / g.in_len;
		const long long k1 = ((long long) g.down * (b - 1) + g.fl2) / g.in_len;
		L.nblk = (int) (k1 - L.k0 + 1);
		L.a = a; L.b = b; L.nch = nch_;
		L.threads = opt_.at("conv_threads");
		L.src = src; L.dst = dst;
		launch_conv(L, stream);
		break;
	}
	case kFrac:
		if (sp.whole)
		{
			WholeLaunch L;
			L.in_step = sp.in_step; L.out_step = sp.out_step; L.flen = sp.flen;
			L.fl2 = sp.fl2; L.fll = sp.fll;
			L.table = d.table;
			L.a = a; L.b = b;
			L.tile = opt_.at("whole_tile");
			L.span_max = (int) ((long long) L.tile * sp.in_step / sp.out_step) + sp.flen + 4;
			while (L.span_max > 12288 && L.tile > 64)
			{
				L.tile /= 2;
				L.span_max = (int) ((long long) L.tile * sp.in_step / sp.out_step) + sp.flen + 4;
			}
			L.nch = nch_;
			L.src = src; L.dst = dst;
			launch_whole(L, stream);
		}
		else
		{
			PolyLaunch L;
			L.flen = sp.flen; L.fl2 = sp.fl2; L.fll = sp.fll; L.fracs = sp.bank->fracs;
			L.table = d.table;
			L.ssr = sp.ssr; L.dsr = sp.dsr;
			L.rpos0 = ps.rpos; L.fpos0 = ps.pos_frac;
			L.counter0 = ps.in_counter; L.pos_int0 = ps.in_pos_int; L.shift = ps.pos_shift;
			L.a = a; L.b = b; L.nch = nch_;
			L.src = src; L.dst = dst;
			launch_poly(L, stream);
		}
		break;
	case kHBUp:
	case kHBDown:
	{
		HBLaunch L;
		L.ntaps = sp.hb_n;
		if (sp.hb_n > 16) throw std::runtime_error("half-band filter too long");
		for (int i = 0; i < 16; i++) L.taps[i] = i < sp.hb_n ? sp.hb_taps[i] : 0.0;
		L.a = a; L.b = b;
		L.tile = opt_.at("hb_tile");
		L.nch = nch_;
		L.src = src; L.dst = dst;
		if (sp.desc.kind == kHBUp) launch_hbup(L, stream);
		else launch_hbdown(L, stream);
		break;
	}
	}
}

int Engine::process(const double* d_in, long long in_stride, int l, double* d_out,
	long long out_stride, void* stream)
{
	if (l < 0 || l > plan_.max_in) throw std::runtime_error("input length exceeds MaxInLen");
	if (l == 0) return 0;
	dev_select(device_);
	const size_t ns = plan_.stages.size();
	if (ns == 0)
	{
		// Src == Dst: the reference hands the input back (reference CDSPResampler.h:534-535);
		// the batch entry copies it into the caller's output buffer
		TailLaunch T;
		T.cur = d_in; T.cur_stride = in_stride; T.cur_base = 0;
		T.p0 = 0; T.p1 = l;
		T.ring = d_out; T.ring_stride = out_stride; T.ring_mask = -1;
		T.nch = nch_;
		launch_tail(T, stream);
		return l;
	}
	int n = l;
	for (size_t s = 0; s < ns; s++)
	{
		StagePlan& sp = plan_.stages[s];
		const long long m_prev = sp.m;
		long long a, b;
		PolyState ps;
		sp.step(n, &a, &b, &ps);
		SrcView src;
		src.ring = dev_[s].ring;
		src.ring_stride = dev_[s].ring_size;
		src.ring_mask = dev_[s].ring_size - 1;
		if (s == 0)
		{
			src.cur = d_in;
			src.cur_stride = in_stride;
			src.cur_base = m_prev;
		}
		else
		{
			src.cur = nullptr;
			src.cur_stride = 0;
			src.cur_base = LLONG_MAX;
		}
		DstView dst;
		if (s + 1 == ns)
		{
			dst.p = d_out;
			dst.stride = out_stride;
			dst.mask = -1;
			dst.off = -a;
		}
		else
		{
			dst.p = dev_[s + 1].ring;
			dst.stride = dev_[s + 1].ring_size;
			dst.mask = dev_[s + 1].ring_size - 1;
			dst.off = 0;
		}
		if (b > a)
		{
			if (opt_.at("timing"))
			{
				void* e0 = get_event(dev_[s]);
				void* e1 = get_event(dev_[s]);
				dev_event_record(e0, stream);
				launch_stage(s, m_prev, a, b, ps, src, dst, stream);
				dev_event_record(e1, stream);
				dev_[s].pending.emplace_back(e0, e1);
				dev_[s].t_in += n;
				dev_[s].t_out += b - a;
			}
			else launch_stage(s, m_prev, a, b, ps, src, dst, stream);
		}
		if (s == 0)
		{
			// keep the tail of the caller's buffer as history for the next call
			TailLaunch T;
			T.cur = d_in; T.cur_stride = in_stride; T.cur_base = m_prev;
			T.p1 = sp.m;
			T.p0 = sp.m - dev_[0].ring_size;
			if (T.p0 < m_prev) T.p0 = m_prev;
			T.ring = dev_[0].ring;
			T.ring_stride = dev_[0].ring_size;
			T.ring_mask = dev_[0].ring_size - 1;
			T.nch = nch_;
			launch_tail(T, stream);
		}
		n = (int) (b - a);
	}
	return n;
}

} // namespace r8bhip
