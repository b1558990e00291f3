// include/r8b/BatchSharded.h -- header-only helper for C++ hosts that spread ONE batch of channels over several GPUs
// (the Python twin is r8brain-free-src_amd/sharding.py ShardedBatchResampler).
//
// Channels never interact -- the reference keeps one resampler object per stream (README.md:53-55) --, so the only
// multi-GPU structure is a partition of the channel axis: device g owns the contiguous channels [lo(g), hi(g)), one
// r8b_batch object each (include/r8bsrc.h r8b_batch_create(..., device)), its own tables and stream history, NO
// collective on the data path.  Shards are whole channel PAIRS (channels 2c and 2c+1 of an object share one complex
// transform, so a cut between them would change the last bits of both): the sharded result is bit-identical to one
// object over all channels.  Data is sharded at rest: the caller hands every device ITS rows, in that device's memory.
//
//   r8b::BatchSharded rs(44100.0, 96000.0, 16384, 2.0, 180.15, /*channels*/ 8192, {0, 1, 2, 3, 4, 5, 6, 7});
//   for (;;) {                              // per call: enqueue on every device's stream, then wait where needed
//       for (int g = 0; g < rs.shards(); g++)
//           n = rs.process(g, d_in[g], in_stride, l, d_out[g], out_stride, stream[g]);   // same n on every shard
//       ...
//   }
//
// No HIP call is made here (the objects switch to their device themselves and leave the caller's current device
// alone); streams and buffers are the caller's.  The same device may be listed more than once (several shards on one
// GPU: what the test does on a one-GPU box).
#ifndef R8B_BATCHSHARDED_H
#define R8B_BATCHSHARDED_H

#include <stdexcept>
#include <string>
#include <vector>

#include "../r8bsrc.h"

namespace r8b {

class BatchSharded
{
public:
	// ReqPhase: 0 = fprLinearPhase, 1 = fprMinPhase (reference CDSPResampler.h:117-120)
	BatchSharded(double SrcSampleRate, double DstSampleRate, int MaxInLen, double ReqTransBand, double ReqAtten,
		int channels, const std::vector<int>& devices, int ReqPhase = 0)
		: nch_(channels)
	{
		if (channels < 1 || devices.empty()) throw std::runtime_error("BatchSharded: channels and devices");
		const int world = (int) devices.size();
		const int pairs = (channels + 1) / 2;
		for (int g = 0; g < world; g++)
		{
			// ceilings: when there are fewer pairs than devices the FIRST devices get them (as sharding.channel_shard)
			const int lo = clamp(2 * ceil_div((long long) pairs * g, world));
			const int hi = clamp(2 * ceil_div((long long) pairs * (g + 1), world));
			lo_.push_back(lo);
			hi_.push_back(hi);
			CR8BBatch h = nullptr;
			if (hi > lo)
			{
				h = r8b_batch_create_ex(SrcSampleRate, DstSampleRate, MaxInLen, ReqTransBand, ReqAtten, ReqPhase, hi - lo,
					devices[(size_t) g]);
				if (h == nullptr)
				{
					const std::string msg = r8b_last_error();
					release();
					throw std::runtime_error("BatchSharded: shard " + std::to_string(g) + ": " + msg);
				}
			}
			h_.push_back(h);
		}
	}
	~BatchSharded() { release(); }
	BatchSharded(const BatchSharded&) = delete;
	BatchSharded& operator=(const BatchSharded&) = delete;

	int shards() const { return (int) h_.size(); }
	int channels() const { return nch_; }
	// the channels shard g owns: [first_channel(g), first_channel(g) + shard_channels(g)); 0 channels: nothing to do
	int first_channel(int g) const { return lo_[(size_t) g]; }
	int shard_channels(int g) const { return hi_[(size_t) g] - lo_[(size_t) g]; }
	int device(int g) const { return h_[(size_t) g] ? r8b_batch_device(h_[(size_t) g]) : -1; }
	int getMaxOutLen() const { return first() ? r8b_batch_max_out_len(first()) : 0; }
	int getInLenBeforeOutPos(int ReqOutPos) const { return first() ? r8b_batch_inlen_before_outpos(first(), ReqOutPos) : 0; }
	double getLatencyFrac() const { return first() ? r8b_batch_latency_frac(first()) : 0.0; }

	// shard g's rows (device pointers in ITS device's memory, strides in doubles) through r8b_batch_process on
	// `stream` (a hipStream_t of that device); returns the output samples per channel of this call -- the same on every
	// shard, all follow one schedule -- or throws with the library's message
	int process(int g, const double* d_in, long long in_stride, int l, double* d_out, long long out_stride, void* stream)
	{
		if (h_[(size_t) g] == nullptr) return 0;
		const int n = r8b_batch_process(h_[(size_t) g], d_in, in_stride, l, d_out, out_stride, stream);
		if (n < 0) throw std::runtime_error(std::string("BatchSharded::process: ") + r8b_last_error());
		return n;
	}
	void clear()
	{
		for (CR8BBatch h : h_)
			if (h) r8b_batch_clear(h);
	}
	CR8BBatch handle(int g) const { return h_[(size_t) g]; }

private:
	static int ceil_div(long long a, int b) { return (int) ((a + b - 1) / b); }
	int clamp(int v) const { return v > nch_ ? nch_ : v; }
	CR8BBatch first() const
	{
		for (CR8BBatch h : h_)
			if (h) return h;
		return nullptr;
	}
	void release()
	{
		for (CR8BBatch h : h_)
			if (h) r8b_batch_delete(h);
		h_.clear();
	}
	int nch_;
	std::vector<int> lo_, hi_;
	std::vector<CR8BBatch> h_;
};

} // namespace r8b

#endif
