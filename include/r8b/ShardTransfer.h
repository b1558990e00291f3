// include/r8b/ShardTransfer.h -- header-only RCCL scatter / gather of channel shards for C++ hosts that run ONE PROCESS
// PER GPU (BASELINE.json north_star: "RCCL over xGMI used only to scatter/gather channel batches").
// The native twin of r8brain-free-src_amd/sharding.py scatter_channels / gather_channels.
//
// Channels never interact -- the reference keeps one resampler object per stream (README.md:53-55) and its callers loop
// over the channels (example.cpp:63-67) --, so a batch that lives on ONE rank is spread by cutting the channel axis:
// rank r owns the contiguous channels channel_shard(channels, r, world) -- whole channel PAIRS, as BatchSharded.h and
// sharding.channel_shard cut them, so that sharded == unsharded bit for bit --, resamples them with its own r8b_batch
// object (include/r8bsrc.h) and hands the result back.  There is NO collective on the data path; per call
//
//     scatter_channels(root rows -> every rank's rows)      one grouped point-to-point operation:
//     r8b_batch_process(...)  on every rank                 ncclGroupStart, one ncclSend per peer shard on the root /
//     gather_channels(every rank's rows -> root rows)       one ncclRecv on the peer, ncclGroupEnd -- xGMI is point to
//                                                           point, so every link carries exactly its shard and all
//                                                           links are busy together
//
// on the stream the caller passes (the root's own shard is a device-to-device copy on that stream).  Rows are fp64,
// channel-major; a shard travels as ONE dense message, pitched rows are packed / unpacked through a scratch buffer of
// the caller (below).  Nothing here touches the r8b_batch objects: keep data sharded at rest when the
// producer allows it (per call of BASELINE's 8192-channel configuration every link carries 128 MiB out and 279 MiB back
// -- 0.9 + 1.9 ms at 153 GB/s against 0.19 ms of compute: link bound) and use these two only where the batch really
// lives on one GPU.
//
// This header needs <hip/hip_runtime_api.h> and <rccl/rccl.h> (ROCm) and the host links librccl itself; the r8bsrc
// library does not depend on RCCL.  Functions return ncclSuccess or the first error; argument errors come back as
// ncclInvalidArgument.
#ifndef R8B_SHARDTRANSFER_H
#define R8B_SHARDTRANSFER_H

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace r8b {

// [lo, hi) of the channels rank `rank` of `world` owns: whole pairs, the LOW ranks first when there are fewer pairs than
// ranks (sharding.channel_shard, BatchSharded)
inline void channel_shard(int channels, int rank, int world, int* lo, int* hi)
{
	const long long pairs = ((long long) channels + 1) / 2;
	const long long a = 2 * ((pairs * rank + world - 1) / world), b = 2 * ((pairs * (rank + 1) + world - 1) / world);
	*lo = (int) (a < channels ? a : channels);
	*hi = (int) (b < channels ? b : channels);
}

namespace shard_detail {

inline ncclResult_t copy_rows(double* dst, long long dst_stride, const double* src, long long src_stride, int rows, int n,
	hipStream_t s)
{
	if (rows <= 0 || n <= 0) return ncclSuccess;
	const hipError_t e = hipMemcpy2DAsync(dst, (size_t) dst_stride * sizeof(double), src, (size_t) src_stride * sizeof(double),
		(size_t) n * sizeof(double), (size_t) rows, hipMemcpyDeviceToDevice, s);
	return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

inline bool args_ok(int channels, int n, long long local_stride, long long root_stride, int root, int rank, int world)
{
	return channels >= 1 && n >= 0 && world >= 1 && rank >= 0 && rank < world && root >= 0 && root < world &&
		local_stride >= n && (rank != root || root_stride >= n);
}

} // namespace shard_detail

// The wire format is ONE message per shard: rows x n doubles, dense -- both ends of a transfer must agree on the message
// sizes, whatever the pitch of their rows.  A side whose rows are pitched (stride != n: r8b_batch_process writes rows of
// a fixed capacity, n varies from call to call) goes through `scratch`, a device buffer of the caller with room for that
// side's share -- channels x n doubles on the root, (hi - lo) x n on the others --, packed / unpacked by one strided copy
// on the same stream; dense sides need none (scratch may be null then; a pitched side without scratch is
// ncclInvalidArgument).

// Rows [channels][n] on rank `root` (root_rows, pitch root_stride doubles; ignored elsewhere) -> this rank's shard
// (local_rows, pitch local_stride).  Every rank of the communicator calls it with the same channels / n / root.
// loopback: the root moves ITS OWN shard through ncclSend / ncclRecv to itself inside the group instead of copying it --
// for tests on a one-GPU box (a world of one rank has no other transfer to exercise); never needed in production.
inline ncclResult_t scatter_channels(const double* root_rows, long long root_stride, int channels, int n,
	double* local_rows, long long local_stride, int root, int rank, int world, ncclComm_t comm, hipStream_t stream,
	double* scratch = nullptr, bool loopback = false)
{
	if (!shard_detail::args_ok(channels, n, local_stride, root_stride, root, rank, world)) return ncclInvalidArgument;
	int lo, hi;
	channel_shard(channels, rank, world, &lo, &hi);
	if (n == 0) return ncclSuccess;
	const bool is_root = rank == root, remote = world > 1 || loopback;
	// root: a dense image of all rows to send from (its own shard never travels unless loopback);
	// others (and the looped-back root): where the dense message lands
	const double* send_base = root_rows;
	double* recv_base = local_rows;
	double* recv_scratch = nullptr;
	if (is_root && remote && root_stride != n)
	{
		if (scratch == nullptr) return ncclInvalidArgument;
		const ncclResult_t e = shard_detail::copy_rows(scratch, n, root_rows, root_stride, channels, n, stream);
		if (e != ncclSuccess) return e;
		send_base = scratch;
	}
	if ((!is_root || loopback) && local_stride != n)
	{
		// (a looped-back root with both sides pitched: its receive lands behind the send image)
		if (scratch == nullptr) return ncclInvalidArgument;
		recv_scratch = scratch + (is_root && root_stride != n ? (size_t) channels * (size_t) n : 0);
		recv_base = recv_scratch;
	}
	const long long send_stride = send_base == root_rows ? root_stride : n;
	ncclResult_t e = ncclGroupStart();
	if (e != ncclSuccess) return e;
	if (is_root)
		for (int r = 0; r < world && e == ncclSuccess; r++)
		{
			int a, b;
			channel_shard(channels, r, world, &a, &b);
			if ((r == root && !loopback) || b <= a) continue;
			e = ncclSend(send_base + (long long) a * send_stride, (size_t) (b - a) * (size_t) n, ncclDouble, r, comm, stream);
		}
	if ((!is_root || loopback) && hi > lo && e == ncclSuccess)
		e = ncclRecv(recv_base, (size_t) (hi - lo) * (size_t) n, ncclDouble, root, comm, stream);
	const ncclResult_t g = ncclGroupEnd();
	if (e != ncclSuccess) return e;
	if (g != ncclSuccess) return g;
	if (recv_scratch != nullptr) return shard_detail::copy_rows(local_rows, local_stride, recv_scratch, n, hi - lo, n, stream);
	if (is_root && !loopback)
		return shard_detail::copy_rows(local_rows, local_stride, root_rows + (long long) lo * root_stride, root_stride, hi - lo, n, stream);
	return ncclSuccess;
}

// The inverse: every rank's shard of n output samples per channel (the same n on every rank: all follow one schedule --
// a rank without channels passes the n the others report) -> rows [channels][n] on `root`.
inline ncclResult_t gather_channels(const double* local_rows, long long local_stride, int channels, int n,
	double* root_rows, long long root_stride, int root, int rank, int world, ncclComm_t comm, hipStream_t stream,
	double* scratch = nullptr, bool loopback = false)
{
	if (!shard_detail::args_ok(channels, n, local_stride, root_stride, root, rank, world)) return ncclInvalidArgument;
	int lo, hi;
	channel_shard(channels, rank, world, &lo, &hi);
	if (n == 0) return ncclSuccess;
	const bool is_root = rank == root, remote = world > 1 || loopback;
	const double* send_base = local_rows;
	double* recv_base = root_rows;
	double* recv_scratch = nullptr;
	if ((!is_root || loopback) && local_stride != n && hi > lo)
	{
		if (scratch == nullptr) return ncclInvalidArgument;
		const ncclResult_t e = shard_detail::copy_rows(scratch, n, local_rows, local_stride, hi - lo, n, stream);
		if (e != ncclSuccess) return e;
		send_base = scratch;
	}
	if (is_root && remote && root_stride != n)
	{
		if (scratch == nullptr) return ncclInvalidArgument;
		recv_scratch = scratch + (send_base == scratch ? (size_t) (hi - lo) * (size_t) n : 0);
		recv_base = recv_scratch;
	}
	const long long recv_stride = recv_base == root_rows ? root_stride : n;
	ncclResult_t e = ncclGroupStart();
	if (e != ncclSuccess) return e;
	if (is_root)
		for (int r = 0; r < world && e == ncclSuccess; r++)
		{
			int a, b;
			channel_shard(channels, r, world, &a, &b);
			if ((r == root && !loopback) || b <= a) continue;
			e = ncclRecv(recv_base + (long long) a * recv_stride, (size_t) (b - a) * (size_t) n, ncclDouble, r, comm, stream);
		}
	if ((!is_root || loopback) && hi > lo && e == ncclSuccess)
		e = ncclSend(send_base, (size_t) (hi - lo) * (size_t) n, ncclDouble, root, comm, stream);
	const ncclResult_t g = ncclGroupEnd();
	if (e != ncclSuccess) return e;
	if (g != ncclSuccess) return g;
	if (recv_scratch != nullptr)
	{
		// (the peers' shards out of the dense image; the root's own -- not looped back -- straight from its rows, below)
		for (int r = 0; r < world; r++)
		{
			int a, b;
			channel_shard(channels, r, world, &a, &b);
			if (r == root && !loopback) continue;
			const ncclResult_t c = shard_detail::copy_rows(root_rows + (long long) a * root_stride, root_stride,
				recv_scratch + (long long) a * n, n, b - a, n, stream);
			if (c != ncclSuccess) return c;
		}
	}
	if (is_root && !loopback)
		return shard_detail::copy_rows(root_rows + (long long) lo * root_stride, root_stride, local_rows, local_stride, hi - lo, n, stream);
	return ncclSuccess;
}

} // namespace r8b

#endif
