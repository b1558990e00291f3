// r8b_pcm_codec.h -- PCM sample codec shared by the stage kernels (src_load / dst_store on planar
// caller buffers) and the transposing ingest/egress kernels (r8b_pcm.h).  Conventions: see
// r8b_pcm.h and include/r8bsrc.h.  The includer defines R8B_HD.
#ifndef R8B_PCM_CODEC_H
#define R8B_PCM_CODEC_H

#include <math.h>

#include "r8b_launch.h"

namespace r8bhip {

R8B_HD int pcm_bytes(int fmt)
{
	switch (fmt)
	{
	case kPcmF64: return 8;
	case kPcmF32: return 4;
	case kPcmS16: return 2;
	case kPcmS24: return 3;
	case kPcmS32: return 4;
	}
	return 0;
}

R8B_HD double pcm_decode(const unsigned char* p, int fmt)
{
	switch (fmt)
	{
	case kPcmF64: return *reinterpret_cast<const double*>(p);
	case kPcmF32: return (double) *reinterpret_cast<const float*>(p);
	case kPcmS16: return (double) *reinterpret_cast<const short*>(p) * (1.0 / 32768.0);
	case kPcmS24:
	{
		// packed little-endian, no alignment: three byte loads
		int v = (int) p[0] | ((int) p[1] << 8) | ((int) (signed char) p[2] << 16);
		return (double) v * (1.0 / 8388608.0);
	}
	case kPcmS32: return (double) *reinterpret_cast<const int*>(p) * (1.0 / 2147483648.0);
	}
	return 0.0;
}

R8B_HD double pcm_quantize(double v, double scale)
{
	double q = rint(v * scale); // round half to even in both the HIP and the host build
	if (!(q >= -scale)) q = q != q ? 0.0 : -scale;
	if (q > scale - 1.0) q = scale - 1.0;
	return q;
}

R8B_HD void pcm_encode(unsigned char* p, int fmt, double v)
{
	switch (fmt)
	{
	case kPcmF64: *reinterpret_cast<double*>(p) = v; break;
	case kPcmF32: *reinterpret_cast<float*>(p) = (float) v; break;
	case kPcmS16: *reinterpret_cast<short*>(p) = (short) (int) pcm_quantize(v, 32768.0); break;
	case kPcmS24:
	{
		const int q = (int) pcm_quantize(v, 8388608.0);
		p[0] = (unsigned char) (q & 255);
		p[1] = (unsigned char) ((q >> 8) & 255);
		p[2] = (unsigned char) ((q >> 16) & 255);
		break;
	}
	case kPcmS32: *reinterpret_cast<int*>(p) = (int) (long long) pcm_quantize(v, 2147483648.0); break;
	}
}

} // namespace r8bhip

#endif
