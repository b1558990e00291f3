// r8b_plan.cpp -- see r8b_plan.h.
#include "r8b_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace r8bhip {

StagePlan make_stage_plan(const StageDesc& d, double prev_lat)
{
	StagePlan s;
	s.desc = d;
	if (d.kind == kConv)
	{
		// geometry, reference CDSPBlockConvolver.h:62-185 (linear phase, PrevLatency 0,
		// DoConsumeLatency)
		s.lp = design_lp(d.a, d.b, d.c, d.d, d.phase == kMinPhase);
		const LpFilter& f = *s.lp;
		ConvGeom& g = s.cg;
		g.up = d.i0;
		g.down = d.i1;
		g.fl2 = f.fl2;
		g.bl2 = 2 << f.block_len_bits;
		const int ups = bit_occupancy(g.up) - 1;
		g.up_pow2 = (1 << ups) == g.up;
		const int dsh = bit_occupancy(g.down) - 1;
		g.down_pow2 = ((1 << dsh) == g.down) && g.down > 1;
		// block geometry for a circular block of bl2 virtual samples
		auto shape = [&](int bl2)
		{
			g.bl2 = bl2;
			if (g.up_pow2)
			{
				g.prev_len = (f.kernel_len - 1 + g.up - 1) / g.up;
				g.in_len = g.bl2 - g.prev_len * g.up;
				g.n_in = g.bl2 / g.up;
			}
			else
			{
				g.prev_len = f.kernel_len - 1;
				g.in_len = g.bl2 - g.prev_len;
				g.n_in = g.bl2;
			}
			g.n_out = g.bl2;
			int ilc = 0;
			if (g.down_pow2)
			{
				ilc = g.in_len & (g.down - 1);
				g.prev_len += ilc;
				g.in_len -= ilc;
				g.n_out = g.bl2 / g.down;
			}
			return ilc;
		};
		const int ilc = shape(g.bl2);
		(void) ilc;
		// reference CDSPBlockConvolver.h:94-101: the filter's own fractional latency plus what the
		// previous stage left (at this stage's virtual rate); the integer part is consumed here -- for
		// the kernels it only moves the alignment of outputs against inputs, exactly like the half
		// length of a linear-phase filter does, so it is folded into fl2 -- the rest goes on
		{
			double lf = f.lat_frac + prev_lat * g.up;
			const int extra = (int) lf;
			lf -= extra;
			g.fl2 = f.fl2 + extra;
			g.complex_h = !f.zero_phase || extra != 0;
			s.lat_frac = lf / g.down;
		}
		if (!g.up_pow2 && g.up == 3 && g.down == 1)
		{
			// polyphase form of the 3x zero-stuffing convolver (ConvGeom::p3): reach of the three components into the
			// future (a) and the past (b) of the input stream
			auto fdiv = [](long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
			int a3 = 0, b3 = 0;
			for (int r = 0; r < 3; r++)
			{
				const long long dmin = -fdiv((long long) r + g.fl2, 3);                    // ceil((-r - fl2) / 3)
				const long long dmax = fdiv((long long) f.kernel_len - 1 - r - g.fl2, 3);
				if (dmax < dmin) continue;
				a3 = std::max(a3, (int) std::max(-dmin, 0LL));
				b3 = std::max(b3, (int) std::max(dmax, 0LL));
			}
			g.p3_a = a3;
			g.p3_b = b3;
			g.p3_off = -((3 - g.fl2 % 3) % 3);
			// The window: the largest power of two whose block has its COMPLETE input when the reference owes the block's
			// first output -- the reference answers in_len + fl2 virtual samples late (CDSPBlockConvolver.h:94-101), so a
			// block may reach p3_m + p3_a input samples past its first output's position only while that stays below the
			// latency --, at least half of it valid.
			const int lat_in = (g.in_len + g.fl2) / 3;
			for (int n = 4096; n >= 1024 && !g.p3; n /= 2)
			{
				const int m3 = (n - a3 - b3) & ~1;
				if (g.fl2 >= 0 && m3 >= n / 2 && m3 + a3 + 2 <= lat_in)
				{
					g.p3 = true;
					g.p3_n = n;
					g.p3_m = m3;
				}
			}
		}
		g.latency = g.in_len + g.fl2; // reference: InputLen (after the divisibility adjustment) + latency
		g.ref_bl2 = g.bl2; g.ref_in_len = g.in_len; g.ref_n_in = g.n_in; g.ref_n_out = g.n_out;
		// transforms longer than 16384 points do not fit a workgroup's LDS: shorter blocks, same filter -- exact for plain
		// overlap-save and strided decimation.  NOT where the reference decimates by truncating the BLOCK's spectrum
		// (2^k down factors, CDSPBlockConvolver.h:329-344): the truncation residue (-219 dB) depends on the block length,
		// so those stages keep the reference's own 32768-point block and run on the generic kernel with its forward
		// array in global memory (r8b_kernels.hip k_conv_big; ratios 3/2 and 3/4 at a 0.5 % transition band)
		const bool exact_block = g.down_pow2 && g.down > 1 && g.n_in <= 32768 && g.n_out <= 16384;
		while (!exact_block && (g.n_in > 16384 || g.n_out > 16384) && g.bl2 / 2 - (f.kernel_len - 1) - g.down >= 64)
			shape(g.bl2 / 2);
	}
	else if (d.kind == kFrac)
	{
		s.ssr = d.a;
		s.dsr = d.b;
		s.whole = whole_stepping(d.a, d.b, &s.in_step, &s.out_step);
		// reference CDSPFracInterpolator.h:736-760
		s.bank = s.whole ? design_frac_bank(s.out_step, 1, 2, d.c, d.i0 != 0) :
			design_frac_bank(-1, 3, 8, d.c, d.i0 != 0);
		s.flen = s.bank->filter_len;
		s.fl2 = s.flen / 2;
		s.fll = s.fl2 - 1;
		// reference CDSPFracInterpolator.h:721-752.  (The integer part of prev_lat -- input samples the
		// interpolator swallows -- is added to the PREVIOUS stage's out_skip by ChainPlan::init.)
		const double ifp = prev_lat - (double) (long long) prev_lat;
		if (s.whole)
		{
			const double spos = ifp * s.out_step;
			s.pos0 = (int) spos;
			s.lat_frac = (spos - s.pos0) / s.in_step;
		}
		else
		{
			s.frac0 = ifp;
			s.lat_frac = 0.0;
		}
	}
	else
	{
		s.hb_n = select_hb_filter(d.a, d.i0, d.i1 != 0, &s.hb_taps, &s.hb_att);
		// reference CDSPHBUpsampler.h:610-612 / CDSPHBDownsampler.h:88-90
		double lf = d.kind == kHBUp ? prev_lat * 2.0 : prev_lat * 0.5;
		s.out_skip = (long long) lf;
		s.lat_frac = lf - (double) s.out_skip;
	}
	s.clear();
	return s;
}

void StagePlan::clear()
{
	m = 0;
	done = 0;
	poly = PolyState();
	// reference CDSPFracInterpolator.h:849-857
	poly.pos_frac = frac0;
	poly.pos_shift = whole || dsr == 0.0 ? 0.0 : frac0 * dsr / ssr;
}

long long StagePlan::total(long long mm) const
{
	switch (desc.kind)
	{
	case kConv:
	{
		// the stage swallows `latency` virtual samples, then keeps every down-th
		const long long v = (long long) cg.up * mm - cg.latency;
		return v <= 0 ? 0 : (v + cg.down - 1) / cg.down;
	}
	case kFrac:
	{
		// whole stepping: output j sits at input position floor(j*In/Out) and is emitted once
		// fl2 further input samples exist
		const long long lim = mm - fl2 - 1;
		if (lim < 0) return 0;
		const long long num = (lim + 1) * out_step - 1 - pos0; // position of output j: j*In + pos0
		return num < 0 ? 0 : num / in_step + 1;
	}
	case kHBUp:
		return mm > hb_n ? 2 * (mm - hb_n) : 0;
	case kHBDown:
	{
		const long long v = mm / 2 - hb_n + 1;
		return v > 0 ? v : 0;
	}
	}
	return 0;
}

void StagePlan::step(int l, long long* a, long long* b, PolyState* ps)
{
	m += l;
	*a = done;
	if (desc.kind == kFrac && !whole)
	{
		// literal restatement of the position counter (reference
		// CDSPFracInterpolator.h:1153-1168) and of its per-call re-base (:907-919)
		if (ps) *ps = poly;
		long long n = 0;
		PolyState st = poly;
		while (m - st.rpos - fl2 > 0)
		{
			n++;
			st.in_counter++;
			const double nxt = ((double) st.in_counter + st.pos_shift) * ssr / dsr;
			const int ni = (int) nxt;
			st.rpos += ni - st.in_pos_int;
			st.in_pos_int = ni;
			st.pos_frac = nxt - ni;
		}
		if (st.in_counter > 1000)
		{
			st.in_counter = 0;
			st.in_pos_int = 0;
			st.pos_shift = st.pos_frac * dsr / ssr;
		}
		poly = st;
		done += n;
		*b = done;
		return;
	}
	const long long t = total(m) - out_skip;
	if (t > done) done = t;
	*b = done;
}

int StagePlan::max_out_len(int maxin) const
{
	switch (desc.kind)
	{
	case kConv: // reference CDSPBlockConvolver.h:208-213
		return (int) (((long long) maxin * cg.up + cg.down - 1) / cg.down);
	case kFrac: // reference CDSPFracInterpolator.h:826-832
		return (int) std::ceil(maxin * dsr / ssr) + 1;
	case kHBUp: // reference CDSPHBUpsampler.h:647-652
		return maxin * 2;
	case kHBDown: // reference CDSPHBDownsampler.h:115-120
		return (maxin + 1) >> 1;
	}
	return 0;
}

int StagePlan::in_len_before_out_pos(int pos) const
{
	switch (desc.kind)
	{
	case kConv: // reference CDSPBlockConvolver.h:192-196
		return (int) ((cg.latency + (double) pos * cg.down) / cg.up + lat_frac * cg.down / cg.up);
	case kFrac: // reference CDSPFracInterpolator.h:802-815
		if (whole) return fl2 + (int) ((pos0 + (double) pos * in_step) / out_step +
			lat_frac * in_step / out_step);
		return fl2 + (int) (frac0 + pos * ssr / dsr);
	case kHBUp: // reference CDSPHBUpsampler.h:632-635
		return hb_n + (int) (((double) out_skip + lat_frac + pos) * 0.5);
	case kHBDown: // reference CDSPHBDownsampler.h:100-103
		return 2 * hb_n - 1 + (int) (((double) out_skip + lat_frac + pos) * 2.0);
	}
	return 0;
}

int StagePlan::history() const
{
	switch (desc.kind)
	{
	case kConv:
		// the earliest block the next call can touch starts less than 2*in_len virtual samples
		// before the stream end and reaches bl2-in_len further back
		// (+ up to one interpolator filter length when the next stage is fused in and starts a
		// little earlier than this stage's own next output)
		// (re-blocked geometry: the outputs due still lag the input by the REFERENCE's latency)
		// (polyphase 3x form, ConvGeom::p3: the next call's first block starts at most one block stride + the reference's
		// latency before the stream's end and reaches p3_b further back)
		return std::max((std::max(cg.in_len, cg.ref_in_len + cg.fl2) + cg.bl2) / cg.up,
			cg.p3 ? (cg.ref_in_len + cg.fl2) / cg.up + cg.p3_m + cg.p3_b + 8 : 0) + 64;
	case kFrac:
		return 2 * flen + 4;
	case kHBUp:
		// (+ what the later stages of a fused half-band run lag behind this stage's own output)
		return 2 * hb_n + 96;
	case kHBDown:
		return 4 * hb_n + 4;
	}
	return 0;
}

std::string StagePlan::describe() const
{
	char buf[256] = "(unknown stage kind)";
	switch (desc.kind)
	{
	case kConv:
		snprintf(buf, sizeof(buf), "BlockConvolver: flt_len=%d in_len=%d io=%d/%d fft=%d/%d "
			"latency=%d nfreq=%.6g tb=%.6g gain=%.6g\n", lp->kernel_len, cg.ref_in_len, cg.up,
			cg.down, cg.ref_n_in, cg.ref_n_out, cg.latency, desc.a, desc.b, desc.d);
		break;
	case kFrac:
		snprintf(buf, sizeof(buf), "FracInterpolator: %.10g->%.10g whole=%d step=%d/%d taps=%d "
			"fracs=%d order=%d\n", ssr, dsr, whole ? 1 : 0, in_step, out_step, flen, bank->fracs,
			whole ? 0 : 2);
		break;
	case kHBUp:
		snprintf(buf, sizeof(buf), "HBUpsampler: sti=%d third=%d taps=%d att=%.2f\n", desc.i0,
			desc.i1, hb_n, hb_att);
		break;
	case kHBDown:
		snprintf(buf, sizeof(buf), "HBDownsampler: sti=%d third=%d taps=%d att=%.2f\n", desc.i0,
			desc.i1, hb_n, hb_att);
		break;
	}
	return buf;
}

void ChainPlan::init(const std::vector<StageDesc>& descs, int maxin)
{
	stages.clear();
	stage_max_in.clear();
	max_in = maxin;
	int mo = maxin;
	double prev_lat = 0.0; // the fractional latency handed from stage to stage (reference CDSPResampler.h:688)
	for (const StageDesc& d : descs)
	{
		if (d.kind == kFrac && prev_lat >= 1.0 && !stages.empty())
		{
			// whole input samples an interpolator swallows: the previous stage simply never emits them
			stages.back().out_skip += (long long) prev_lat;
		}
		stages.push_back(make_stage_plan(d, prev_lat));
		prev_lat = stages.back().lat_frac;
		stage_max_in.push_back(mo);
		mo = stages.back().max_out_len(mo);
	}
	max_out_len = mo;
}

void ChainPlan::clear()
{
	for (StagePlan& s : stages) s.clear();
}

int ChainPlan::in_len_before_out_pos(int pos) const
{
	int r = pos;
	for (size_t i = stages.size(); i-- > 0;) r = stages[i].in_len_before_out_pos(r);
	return r;
}

int ChainPlan::input_required(int nout) const
{
	return nout < 1 ? 0 : in_len_before_out_pos(nout - 1) + 1;
}

std::string ChainPlan::describe() const
{
	std::string s;
	for (const StagePlan& st : stages) s += st.describe();
	return s;
}

} // namespace r8bhip
