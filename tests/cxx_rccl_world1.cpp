// tests/cxx_rccl_world1.cpp -- include/r8b/ShardTransfer.h on the one GPU a test box has: a world of ONE RCCL rank.
// scatter -> r8b_batch_process -> gather through the header equals the object driven on the root's rows directly, bit for
// bit, with pitched rows on both sides; once with the root's own shard as a device copy (what a real root does) and once
// with loopback = true, where the shard goes through ncclSend / ncclRecv to self inside the group (the only RCCL
// transfer a one-rank world can make).  Prints OK.  (More than one rank: unmeasured, no multi-GPU node -- DESIGN.md 7.)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/r8b/ShardTransfer.h"
#include "../include/r8bsrc.h"

#define CK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } } while (0)
#define NK(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { std::printf("RCCL error %d (%s) at %s:%d\n", (int) e_, ncclGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main()
{
	// shard boundaries: the table sharding.channel_shard / BatchSharded produce
	{
		const int want[4][2] = { { 0, 2 }, { 2, 4 }, { 4, 5 }, { 5, 5 } };
		for (int r = 0; r < 4; r++)
		{
			int lo, hi;
			r8b::channel_shard(5, r, 4, &lo, &hi);
			if (lo != want[r][0] || hi != want[r][1]) { std::printf("channel_shard(5, %d, 4) = [%d, %d)\n", r, lo, hi); return 1; }
		}
		int lo, hi;
		r8b::channel_shard(1024, 3, 8, &lo, &hi);
		if (lo != 384 || hi != 512) { std::printf("channel_shard(1024, 3, 8) = [%d, %d)\n", lo, hi); return 1; }
	}
	CK(hipSetDevice(0));
	ncclUniqueId id;
	NK(ncclGetUniqueId(&id));
	ncclComm_t comm;
	NK(ncclCommInitRank(&comm, 1, id, 0));
	hipStream_t stream;
	CK(hipStreamCreate(&stream));
	const int nch = 6, L = 4096, calls = 4, pitch_in = L + 24;
	CR8BBatch direct = r8b_batch_create(44100.0, 96000.0, L, 2.0, 180.15, nch, 0);
	CR8BBatch sharded = r8b_batch_create(44100.0, 96000.0, L, 2.0, 180.15, nch, 0);
	CR8BBatch looped = r8b_batch_create(44100.0, 96000.0, L, 2.0, 180.15, nch, 0);
	if (!direct || !sharded || !looped) { std::printf("create: %s\n", r8b_last_error()); return 1; }
	const int maxout = r8b_batch_max_out_len(direct), pitch_out = maxout + 8;
	double *root_in, *local_in, *local_out, *root_out, *ref_out, *scratch;
	CK(hipMalloc(&scratch, sizeof(double) * 2 * nch * (maxout > L ? maxout : L)));
	CK(hipMalloc(&root_in, sizeof(double) * nch * pitch_in));
	CK(hipMalloc(&local_in, sizeof(double) * nch * L));
	CK(hipMalloc(&local_out, sizeof(double) * nch * maxout));
	CK(hipMalloc(&root_out, sizeof(double) * nch * pitch_out));
	CK(hipMalloc(&ref_out, sizeof(double) * nch * pitch_out));
	std::vector<double> h((size_t) nch * pitch_in), a((size_t) nch * pitch_out), b((size_t) nch * pitch_out);
	unsigned long long s = 12345;
	long long total = 0;
	for (int c = 0; c < calls; c++)
	{
		for (double& v : h)
		{
			s = s * 6364136223846793005ull + 1442695040888963407ull;
			v = (double) (s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
		}
		CK(hipMemcpyAsync(root_in, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream));
		const int nref = r8b_batch_process(direct, root_in, pitch_in, L, ref_out, pitch_out, stream);
		if (nref < 0) { std::printf("process: %s\n", r8b_last_error()); return 1; }
		for (int pass = 0; pass < 2; pass++)
		{
			const bool loop = pass == 1;
			NK(r8b::scatter_channels(root_in, pitch_in, nch, L, local_in, L, 0, 0, 1, comm, stream, scratch, loop));
			const int n = r8b_batch_process(loop ? looped : sharded, local_in, L, L, local_out, maxout, stream);
			if (n != nref) { std::printf("counts differ: %d vs %d (%s)\n", n, nref, r8b_last_error()); return 1; }
			CK(hipMemsetAsync(root_out, 0xff, sizeof(double) * nch * pitch_out, stream));
			NK(r8b::gather_channels(local_out, maxout, nch, n, root_out, pitch_out, 0, 0, 1, comm, stream, scratch, loop));
			CK(hipMemcpyAsync(a.data(), ref_out, sizeof(double) * a.size(), hipMemcpyDeviceToHost, stream));
			CK(hipMemcpyAsync(b.data(), root_out, sizeof(double) * b.size(), hipMemcpyDeviceToHost, stream));
			CK(hipStreamSynchronize(stream));
			for (int ch = 0; ch < nch; ch++)
				for (int i = 0; i < n; i++)
					if (a[(size_t) ch * pitch_out + i] != b[(size_t) ch * pitch_out + i])
					{
						std::printf("call %d pass %d channel %d sample %d: %.17g vs %.17g\n", c, pass, ch, i,
							a[(size_t) ch * pitch_out + i], b[(size_t) ch * pitch_out + i]);
						return 1;
					}
		}
		total += nref;
	}
	if (total <= 0) { std::printf("no output\n"); return 1; }
	// argument errors come back as such
	if (r8b::scatter_channels(root_in, L - 1, nch, L, local_in, L, 0, 0, 1, comm, stream) != ncclInvalidArgument) { std::printf("pitch check\n"); return 1; }
	// ... and pitched rows that would have to travel without a scratch buffer
	if (r8b::gather_channels(local_out, maxout, nch, 16, root_out, pitch_out, 0, 0, 1, comm, stream, nullptr, true) != ncclInvalidArgument) { std::printf("scratch check\n"); return 1; }
	r8b_batch_delete(direct); r8b_batch_delete(sharded); r8b_batch_delete(looped);
	NK(ncclCommDestroy(comm));
	std::printf("%lld samples per channel through scatter / gather, both ways: OK\n", total);
	return 0;
}
