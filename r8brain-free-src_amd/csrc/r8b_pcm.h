// r8b_pcm.h -- PCM ingest/egress next to the hot path: interleaved or planar
// int16 / packed int24 / int32 / float32 / float64 samples <-> the planar fp64 rows the resampler
// works on, so that 2-4 bytes per sample cross PCIe and the HBM edge instead of 8.
//
// The reference has no such code in-tree: its command-line tool converts through CWaveFile (the
// external "libvox", absent from the repository; call sites bench/r8bfreesrc.cpp:103,134), and
// oneshot<Tin,Tout> (CDSPResampler.h:592-651) only casts.  The conventions here are the usual PCM
// ones and are part of this library's interface (include/r8bsrc.h):
//   decode: integer / 2^(bits-1); float32 widened exactly
//   encode: v * 2^(bits-1), round to nearest even, saturate to [-2^(bits-1), 2^(bits-1)-1]
//           (no dither); float32 by round-to-nearest conversion; NaN encodes as 0 in the integer
//           formats
// The sample codec itself lives in r8b_pcm_codec.h (the stage kernels use it too, for planar PCM
// buffers read and written in place).  Phases are shared with the host emulation of tests/emul
// like every other kernel.
#ifndef R8B_PCM_H
#define R8B_PCM_H

#include "r8b_kernel_phases.h"

namespace r8bhip {

// byte offset of (frame f, channel c) in the PCM buffer
R8B_HD long long pcm_offset(const PcmLaunch& L, long long f, int c)
{
	const long long e = L.interleaved ? f * L.pcm_stride + c : (long long) c * L.pcm_stride + f;
	return e * pcm_bytes(L.fmt);
}

// A workgroup converts a tile of kPcmTile frames x kPcmTile channels.  Interleaved buffers are
// frame-major, the planar rows channel-major: the tile goes through LDS (pitch kPcmTile + 1) so
// that both sides are accessed with the fastest index on consecutive lanes.
static const int kPcmTile = 64;
static const int kPcmPitch = kPcmTile + 1;

// PCM -> tile (interleaved) : lanes walk channels
R8B_HD void pcm_in_gather(const PcmLaunch& L, double* tile, long long f0, int c0, int tid, int nthr)
{
	const unsigned char* src = static_cast<const unsigned char*>(L.pcm);
	for (int e = tid; e < kPcmTile * kPcmTile; e += nthr)
	{
		const int c = e & (kPcmTile - 1), f = e >> 6;
		if (f0 + f < L.n && c0 + c < L.nch)
			tile[c * kPcmPitch + f] = pcm_decode(src + pcm_offset(L, f0 + f, c0 + c), L.fmt);
	}
}

// tile -> planar rows : lanes walk frames
R8B_HD void pcm_in_scatter(const PcmLaunch& L, const double* tile, long long f0, int c0, int tid,
	int nthr)
{
	for (int e = tid; e < kPcmTile * kPcmTile; e += nthr)
	{
		const int f = e & (kPcmTile - 1), c = e >> 6;
		if (f0 + f < L.n && c0 + c < L.nch)
			L.planar[(long long) (c0 + c) * L.planar_stride + f0 + f] = tile[c * kPcmPitch + f];
	}
}

// planar PCM -> planar rows, no transposition
R8B_HD void pcm_in_direct(const PcmLaunch& L, long long f0, int c0, int tid, int nthr)
{
	const unsigned char* src = static_cast<const unsigned char*>(L.pcm);
	for (int e = tid; e < kPcmTile * kPcmTile; e += nthr)
	{
		const int f = e & (kPcmTile - 1), c = e >> 6;
		if (f0 + f < L.n && c0 + c < L.nch)
			L.planar[(long long) (c0 + c) * L.planar_stride + f0 + f] =
				pcm_decode(src + pcm_offset(L, f0 + f, c0 + c), L.fmt);
	}
}

R8B_HD void pcm_out_gather(const PcmLaunch& L, double* tile, long long f0, int c0, int tid, int nthr)
{
	for (int e = tid; e < kPcmTile * kPcmTile; e += nthr)
	{
		const int f = e & (kPcmTile - 1), c = e >> 6;
		if (f0 + f < L.n && c0 + c < L.nch)
			tile[c * kPcmPitch + f] = L.planar[(long long) (c0 + c) * L.planar_stride + f0 + f];
	}
}

R8B_HD void pcm_out_scatter(const PcmLaunch& L, const double* tile, long long f0, int c0, int tid,
	int nthr)
{
	unsigned char* dst = static_cast<unsigned char*>(L.pcm);
	for (int e = tid; e < kPcmTile * kPcmTile; e += nthr)
	{
		const int c = e & (kPcmTile - 1), f = e >> 6;
		if (f0 + f < L.n && c0 + c < L.nch)
			pcm_encode(dst + pcm_offset(L, f0 + f, c0 + c), L.fmt, tile[c * kPcmPitch + f]);
	}
}

R8B_HD void pcm_out_direct(const PcmLaunch& L, long long f0, int c0, int tid, int nthr)
{
	unsigned char* dst = static_cast<unsigned char*>(L.pcm);
	for (int e = tid; e < kPcmTile * kPcmTile; e += nthr)
	{
		const int f = e & (kPcmTile - 1), c = e >> 6;
		if (f0 + f < L.n && c0 + c < L.nch)
			pcm_encode(dst + pcm_offset(L, f0 + f, c0 + c), L.fmt,
				L.planar[(long long) (c0 + c) * L.planar_stride + f0 + f]);
	}
}

// Planar PCM <-> planar rows, row by row (no transposition): a workgroup converts kPcmRowChunk consecutive frames
// of ONE channel, thread tid the frames tid, tid + nthr, ... of the chunk -- consecutive lanes on consecutive
// samples, eight independent loads in flight per thread, the format switch outside the loop.  (The tile form
// above moves 64 frames of a row per instruction and decides the format per sample: 2.2 TB/s on the staging
// passes in front of / behind a fast-path convolver, where this streams.)
static const int kPcmRowChunk = 2048;

template<int FMT>
R8B_HD void pcm_row_in_t(const PcmLaunch& L, long long f0, int c, int tid, int nthr)
{
	const unsigned char* src = static_cast<const unsigned char*>(L.pcm) +
		(long long) c * L.pcm_stride * pcm_bytes(FMT);
	double* dst = L.planar + (long long) c * L.planar_stride;
	long long f1 = f0 + kPcmRowChunk;
	if (f1 > L.n) f1 = L.n;
	constexpr int B = FMT == kPcmF64 ? 8 : (FMT == kPcmS16 ? 2 : (FMT == kPcmS24 ? 3 : 4));
#pragma unroll 8
	for (long long f = f0 + tid; f < f1; f += nthr) dst[f] = pcm_decode(src + f * B, FMT);
}

template<int FMT>
R8B_HD void pcm_row_out_t(const PcmLaunch& L, long long f0, int c, int tid, int nthr)
{
	unsigned char* dst = static_cast<unsigned char*>(L.pcm) + (long long) c * L.pcm_stride * pcm_bytes(FMT);
	const double* src = L.planar + (long long) c * L.planar_stride;
	long long f1 = f0 + kPcmRowChunk;
	if (f1 > L.n) f1 = L.n;
	constexpr int B = FMT == kPcmF64 ? 8 : (FMT == kPcmS16 ? 2 : (FMT == kPcmS24 ? 3 : 4));
#pragma unroll 8
	for (long long f = f0 + tid; f < f1; f += nthr) pcm_encode(dst + f * B, FMT, src[f]);
}

R8B_HD void pcm_row_in(const PcmLaunch& L, long long f0, int c, int tid, int nthr)
{
	switch (L.fmt)
	{
	case kPcmF64: pcm_row_in_t<kPcmF64>(L, f0, c, tid, nthr); break;
	case kPcmF32: pcm_row_in_t<kPcmF32>(L, f0, c, tid, nthr); break;
	case kPcmS16: pcm_row_in_t<kPcmS16>(L, f0, c, tid, nthr); break;
	case kPcmS24: pcm_row_in_t<kPcmS24>(L, f0, c, tid, nthr); break;
	case kPcmS32: pcm_row_in_t<kPcmS32>(L, f0, c, tid, nthr); break;
	}
}

R8B_HD void pcm_row_out(const PcmLaunch& L, long long f0, int c, int tid, int nthr)
{
	switch (L.fmt)
	{
	case kPcmF64: pcm_row_out_t<kPcmF64>(L, f0, c, tid, nthr); break;
	case kPcmF32: pcm_row_out_t<kPcmF32>(L, f0, c, tid, nthr); break;
	case kPcmS16: pcm_row_out_t<kPcmS16>(L, f0, c, tid, nthr); break;
	case kPcmS24: pcm_row_out_t<kPcmS24>(L, f0, c, tid, nthr); break;
	case kPcmS32: pcm_row_out_t<kPcmS32>(L, f0, c, tid, nthr); break;
	}
}

} // namespace r8bhip

#endif
