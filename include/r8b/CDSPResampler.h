// include/r8b/CDSPResampler.h -- header-only C++ front-end with the reference's shape over the C ABI
// of libr8bsrc_hip.so.  Same class names, constructor arguments and member functions as
// r8b::CDSPResampler and its presets (reference CDSPResampler.h:117-120, 406-421, 476-519,
// 521-529, 559-575, 592-651, 729-810), so code written against the reference compiles against
// this header and runs on the GPU.  Both phase responses of the reference's constructor (fprLinearPhase, the
// default and the only mode of its DLL, DLL/r8bsrc.h:52; fprMinPhase through r8b_batch_create_ex).
//
// process() keeps the reference's host-pointer contract: `op0` receives a pointer to a buffer
// owned by the object (valid until the next call), or `ip0` itself when Src == Dst
// (reference CDSPResampler.h:534-552).  For throughput use the batch entry points of r8bsrc.h
// (many channels, device pointers); this class is the drop-in, not the fast path.
#ifndef R8B_HIP_CDSPRESAMPLER_INCLUDED
#define R8B_HIP_CDSPRESAMPLER_INCLUDED

#include <stdexcept>
#include <vector>

#include "../r8bsrc.h"

namespace r8b {

enum EDSPFilterPhaseResponse
{
	fprLinearPhase = 0,
	fprMinPhase
};

class CDSPResampler
{
public:
	CDSPResampler(const double SrcSampleRate, const double DstSampleRate, const int aMaxInLen,
		const double ReqTransBand = 2.0, const double ReqAtten = 206.91,
		const EDSPFilterPhaseResponse ReqPhase = fprLinearPhase)
		: SrcRate(SrcSampleRate), DstRate(DstSampleRate), MaxInLen(aMaxInLen)
	{
		h = r8b_batch_create_ex(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, ReqAtten,
			(int) ReqPhase, 1, -1);
		if (h == nullptr) throw std::runtime_error(r8b_last_error());
		const int mo = r8b_batch_max_out_len(h);
		OutBuf.resize((size_t) (mo > 0 ? mo : 1));
	}

	virtual ~CDSPResampler() { r8b_batch_delete(h); }
	CDSPResampler(const CDSPResampler&) = delete;
	CDSPResampler& operator=(const CDSPResampler&) = delete;

	int getInLenBeforeOutPos(const int ReqOutPos) const
	{
		return r8b_batch_inlen_before_outpos(h, ReqOutPos);
	}

	int getInputRequiredForOutput(const int ReqOutSamples) const
	{
		return r8b_batch_inlen(h, ReqOutSamples);
	}

	// legacy test helper of the reference (CDSPResampler.h:443-464)
	int getInLenBeforeOutStart(const int ReqOutPos = 0)
	{
		int inc = 0, outc = 0;
		while (true)
		{
			double ins = 0.0;
			double* op;
			outc += process(&ins, 1, op);
			if (outc > ReqOutPos)
			{
				clear();
				return inc;
			}
			inc++;
		}
	}

	int getLatency() const { return 0; }
	// reference CDSPResampler.h:491-494: the fractional latency the last stage reports (0 for linear phase)
	double getLatencyFrac() const { return r8b_batch_latency_frac(h); }
	int getMaxOutLen(const int /*MaxInLen*/) const { return r8b_batch_max_out_len(h); }
	void clear() { r8b_batch_clear(h); }

	int process(double* ip0, int l, double*& op0)
	{
		if (SrcRate == DstRate)
		{
			op0 = ip0;
			return l;
		}
		op0 = OutBuf.data();
		const int n = r8b_batch_process_host(h, ip0, l, l, OutBuf.data(), (long long) OutBuf.size());
		if (n < 0) throw std::runtime_error(r8b_last_error());
		return n;
	}

	// reference CDSPResampler.h:592-651
	template<typename Tin, typename Tout>
	void oneshot(const Tin* ip, int iplen, Tout* op, int oplen)
	{
		std::vector<double> Buf((size_t) MaxInLen);
		bool IsZero = false;
		while (oplen > 0)
		{
			int rc;
			if (iplen == 0)
			{
				rc = MaxInLen;
				if (!IsZero)
				{
					IsZero = true;
					for (double& v : Buf) v = 0.0;
				}
			}
			else
			{
				rc = iplen < MaxInLen ? iplen : MaxInLen;
				for (int i = 0; i < rc; i++) Buf[(size_t) i] = (double) ip[i];
				ip += rc;
				iplen -= rc;
			}
			double* p;
			int wc = process(Buf.data(), rc, p);
			if (wc > oplen) wc = oplen;
			for (int i = 0; i < wc; i++) op[i] = (Tout) p[i];
			op += wc;
			oplen -= wc;
		}
		clear();
	}

private:
	double SrcRate, DstRate;
	int MaxInLen;
	CR8BBatch h = nullptr;
	std::vector<double> OutBuf;
};

// presets, reference CDSPResampler.h:729-810
class CDSPResampler16 : public CDSPResampler
{
public:
	CDSPResampler16(const double SrcSampleRate, const double DstSampleRate, const int aMaxInLen,
		const double ReqTransBand = 2.0)
		: CDSPResampler(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, 136.45, fprLinearPhase)
	{
	}
};

class CDSPResampler16IR : public CDSPResampler
{
public:
	CDSPResampler16IR(const double SrcSampleRate, const double DstSampleRate, const int aMaxInLen,
		const double ReqTransBand = 2.0)
		: CDSPResampler(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, 109.56, fprLinearPhase)
	{
	}
};

class CDSPResampler24 : public CDSPResampler
{
public:
	CDSPResampler24(const double SrcSampleRate, const double DstSampleRate, const int aMaxInLen,
		const double ReqTransBand = 2.0)
		: CDSPResampler(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, 180.15, fprLinearPhase)
	{
	}
};

} // namespace r8b

#endif
