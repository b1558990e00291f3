// r8b_launch.h -- the narrow interface between the host engine (r8b_engine.cpp, plain C++) and the
// device side (r8b_kernels.hip): launch descriptors, launchers and a handful of memory helpers.
// Pointers inside the descriptors are DEVICE pointers.  Positions are absolute stream positions
// (samples since construction/clear) of the respective stage's input or output stream.
#ifndef R8B_LAUNCH_H
#define R8B_LAUNCH_H

#include <cstddef>

namespace r8bhip {

// Where a stage reads its input stream x[pos] for channel ch:
//   pos <  0          -> 0.0 (the stream starts at 0)
//   pos >= cur_base   -> cur[ch*cur_stride + pos - cur_base]     (this call's fresh samples)
//   otherwise         -> ring[ch*ring_stride + (pos & ring_mask)] (history kept from earlier calls)
struct SrcView
{
	const double* ring;
	long long ring_stride;
	long long ring_mask;
	const double* cur;
	long long cur_stride;
	long long cur_base;
};

// Where a stage writes output sample q of channel ch: p[ch*stride + ((q + off) & mask)].
// Ring: off = 0, mask = size-1.  Linear (user buffer): mask = -1, off = -first_q.
struct DstView
{
	double* p;
	long long stride;
	long long mask;
	long long off;
};

struct cd;

static const int kMaxPasses = 16;

struct ConvLaunch
{
	// geometry (r8b_plan.h ConvGeom)
	int up, down, fl2, bl2, in_len, n_in, n_out;
	int up_pow2, down_pow2;
	// transform plan: radices of the forward passes in execution order (sub-length N, N/r0, ...)
	// and of the backward passes in execution order (sub-length grows to N2)
	int n_fwd, n_inv;
	int fwd_radix[kMaxPasses];
	int inv_radix[kMaxPasses];
	const double* H; // bl2/2+1 reals: zero-phase kernel spectrum / bl2
	const cd* tw;    // tw_len complex: exp(-2 pi i e / tw_len)
	int tw_len;
	// work: blocks [k0, k0+nblk) x channels [0, nch); outputs clipped to [a, b)
	long long k0;
	int nblk;
	long long a, b;
	int nch;
	int threads;
	SrcView src;
	DstView dst;
};

struct WholeLaunch
{
	int in_step, out_step, flen, fl2, fll;
	const double* table; // out_step rows x flen
	long long a, b;      // outputs to produce
	int tile;            // outputs per workgroup
	int span_max;        // LDS doubles per workgroup
	int nch;
	SrcView src;
	DstView dst;
};

struct PolyLaunch
{
	int flen, fl2, fll, fracs;
	const double* table; // (fracs+1) rows x flen x 3
	double ssr, dsr;
	// counter state at the start of the call (r8b_plan.h PolyState)
	long long rpos0;
	double fpos0;
	long long counter0, pos_int0;
	double shift;
	long long a, b;
	int nch;
	SrcView src;
	DstView dst;
};

struct HBLaunch
{
	int ntaps;
	double taps[16];
	long long a, b; // outputs to produce
	int tile;       // up: input indices per workgroup; down: outputs per workgroup
	int nch;
	SrcView src;
	DstView dst;
};

struct TailLaunch
{
	const double* cur;
	long long cur_stride;
	long long cur_base; // absolute position of cur[0]
	long long p0, p1;   // positions [p0, p1) to copy into the ring
	double* ring;
	long long ring_stride;
	long long ring_mask;
	int nch;
};

// launchers (asynchronous on `stream`, a hipStream_t)
void launch_conv(const ConvLaunch& L, void* stream);
void launch_whole(const WholeLaunch& L, void* stream);
void launch_poly(const PolyLaunch& L, void* stream);
void launch_hbup(const HBLaunch& L, void* stream);
void launch_hbdown(const HBLaunch& L, void* stream);
void launch_tail(const TailLaunch& L, void* stream);

// memory helpers; all throw std::runtime_error with the HIP error text on failure
void dev_select(int device);          // -1 keeps the current device
void* dev_alloc(size_t bytes);        // zero-initialised
void dev_free(void* p);
void dev_zero(void* p, size_t bytes, void* stream);
void dev_upload(void* dst, const void* src, size_t bytes);   // synchronous
void dev_download(void* dst, const void* src, size_t bytes, void* stream); // synchronous on stream
void dev_upload_async(void* dst, const void* src, size_t bytes, void* stream);
void dev_sync(void* stream);
void dev_check_last(const char* what);
// events for the optional per-stage timing (hipEvent_t on the launching stream)
void* dev_event_create();
void dev_event_destroy(void* ev);
void dev_event_record(void* ev, void* stream);
float dev_event_elapsed_ms(void* start, void* stop); // waits for `stop`

} // namespace r8bhip

#endif
