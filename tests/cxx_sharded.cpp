// tests/cxx_sharded.cpp -- compiled (hipcc) and run by tests/test_gpu_parity.py::test_cxx_batch_sharded: the C++ multi-device
// helper include/r8b/BatchSharded.h on the one GPU of the test box -- 11 channels over three shards on device 0 (4 + 4 + 3:
// whole pairs, the last shard ending in a channel without a partner), each on a stream of its own, against ONE object over all 11 channels:
// bitwise equal, call by call, ragged call lengths.  Exit code 0 and "OK" = equal.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/r8b/BatchSharded.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)

static double splitmix(uint64_t& s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return (double) (z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

int main()
{
	const int nch = 11, L = 3000;
	const int lens[] = { 3000, 3000, 1, 777, 3000, 2, 2999, 3000 };
	r8b::BatchSharded sh(44100.0, 96000.0, L, 2.0, 180.15, nch, { 0, 0, 0 });
	CR8BBatch all = r8b_batch_create(44100.0, 96000.0, L, 2.0, 180.15, nch, 0);
	if (all == nullptr) return 2;
	if (sh.shards() != 3 || sh.shard_channels(0) != 4 || sh.shard_channels(1) != 4 || sh.shard_channels(2) != 3 ||
		sh.first_channel(1) != 4 || sh.first_channel(2) != 8 || sh.device(0) != 0)
	{
		fprintf(stderr, "shards: %d %d %d\n", sh.shard_channels(0), sh.shard_channels(1), sh.shard_channels(2));
		return 3;
	}
	const int cap = sh.getMaxOutLen();
	if (cap != r8b_batch_max_out_len(all)) return 4;
	hipStream_t st[4];
	for (int i = 0; i < 4; i++) CHECK(hipStreamCreate(&st[i]));
	double *d_in, *d_out, *d_out2;
	CHECK(hipMalloc(&d_in, sizeof(double) * L * nch));
	CHECK(hipMalloc(&d_out, sizeof(double) * cap * nch));
	CHECK(hipMalloc(&d_out2, sizeof(double) * cap * nch));
	std::vector<double> in((size_t) L * nch), a((size_t) cap * nch), b((size_t) cap * nch);
	uint64_t seed = 11;
	long long total = 0;
	for (int l : lens)
	{
		for (int ch = 0; ch < nch; ch++)
			for (int i = 0; i < l; i++) in[(size_t) ch * L + i] = splitmix(seed);
		CHECK(hipMemcpy(d_in, in.data(), sizeof(double) * L * nch, hipMemcpyHostToDevice));
		CHECK(hipMemset(d_out, 0xff, sizeof(double) * cap * nch));
		CHECK(hipMemset(d_out2, 0xff, sizeof(double) * cap * nch));
		int n = -1;
		for (int g = 0; g < sh.shards(); g++)
		{
			const int c0 = sh.first_channel(g);
			const int m = sh.process(g, d_in + (size_t) c0 * L, L, l, d_out + (size_t) c0 * cap, cap, st[g]);
			if (n >= 0 && m != n) return 5;
			n = m;
		}
		const int n2 = r8b_batch_process(all, d_in, L, l, d_out2, cap, st[3]);
		if (n2 != n) return 6;
		for (int i = 0; i < 4; i++) CHECK(hipStreamSynchronize(st[i]));
		CHECK(hipMemcpy(a.data(), d_out, sizeof(double) * cap * nch, hipMemcpyDeviceToHost));
		CHECK(hipMemcpy(b.data(), d_out2, sizeof(double) * cap * nch, hipMemcpyDeviceToHost));
		for (int ch = 0; ch < nch; ch++)
			if (n > 0 && memcmp(&a[(size_t) ch * cap], &b[(size_t) ch * cap], sizeof(double) * (size_t) n) != 0)
			{
				fprintf(stderr, "channel %d differs (call of %d samples)\n", ch, l);
				return 7;
			}
		total += n;
	}
	if (total <= 0) return 8;
	r8b_batch_delete(all);
	printf("%lld outputs per channel, OK\n", total);
	return 0;
}
