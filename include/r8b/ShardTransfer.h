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
// channel-major; a side whose row pitch equals the row length moves as ONE message per peer, other pitches as one
// message per row inside the same group.  Nothing here touches the r8b_batch objects: keep data sharded at rest when the
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

inline ncclResult_t send_rows(const double* p, long long stride, int rows, int n, int peer, ncclComm_t comm, hipStream_t s)
{
	if (rows <= 0 || n <= 0) return ncclSuccess;
	if (stride == n) return ncclSend(p, (size_t) rows * (size_t) n, ncclDouble, peer, comm, s);
	for (int r = 0; r < rows; r++)
	{
		const ncclResult_t e = ncclSend(p + (long long) r * stride, (size_t) n, ncclDouble, peer, comm, s);
		if (e != ncclSuccess) return e;
	}
	return ncclSuccess;
}

inline ncclResult_t recv_rows(double* p, long long stride, int rows, int n, int peer, ncclComm_t comm, hipStream_t s)
{
	if (rows <= 0 || n <= 0) return ncclSuccess;
	if (stride == n) return ncclRecv(p, (size_t) rows * (size_t) n, ncclDouble, peer, comm, s);
	for (int r = 0; r < rows; r++)
	{
		const ncclResult_t e = ncclRecv(p + (long long) r * stride, (size_t) n, ncclDouble, peer, comm, s);
		if (e != ncclSuccess) return e;
	}
	return ncclSuccess;
}

inline ncclResult_t copy_rows(double* dst, long long dst_stride, const double* src, long long src_stride, int rows, int n,
	hipStream_t s)
{
	if (rows <= 0 || n <= 0) return ncclSuccess;
	const hipError_t e = hipMemcpy2DAsync(dst, (size_t) dst_stride * sizeof(double), src, (size_t) src_stride * sizeof(double),
		(size_t) n * sizeof(double), (size_t) rows, hipMemcpyDeviceToDevice, s);
	return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

} // namespace shard_detail

// Rows [channels][n] on rank `root` (root_rows, pitch root_stride doubles; ignored elsewhere) -> this rank's shard
// (local_rows, pitch local_stride).  Every rank of the communicator calls it with the same channels / n / root.
// loopback: the root moves ITS OWN shard through ncclSend / ncclRecv to itself inside the group instead of copying it --
// for tests on a one-GPU box (a world of one rank has no other transfer to exercise); never needed in production.
inline ncclResult_t scatter_channels(const double* root_rows, long long root_stride, int channels, int n,
	double* local_rows, long long local_stride, int root, int rank, int world, ncclComm_t comm, hipStream_t stream,
	bool loopback = false)
{
	if (channels < 1 || n < 0 || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world ||
		local_stride < n || (rank == root && root_stride < n)) return ncclInvalidArgument;
	int lo, hi;
	channel_shard(channels, rank, world, &lo, &hi);
	if (n == 0) return ncclSuccess;
	ncclResult_t e = ncclGroupStart();
	if (e != ncclSuccess) return e;
	if (rank == root)
	{
		for (int r = 0; r < world && e == ncclSuccess; r++)
		{
			int a, b;
			channel_shard(channels, r, world, &a, &b);
			if (r == root && !loopback) continue;
			e = shard_detail::send_rows(root_rows + (long long) a * root_stride, root_stride, b - a, n, r, comm, stream);
		}
		if (loopback && e == ncclSuccess) e = shard_detail::recv_rows(local_rows, local_stride, hi - lo, n, root, comm, stream);
	}
	else e = shard_detail::recv_rows(local_rows, local_stride, hi - lo, n, root, comm, stream);
	const ncclResult_t g = ncclGroupEnd();
	if (e != ncclSuccess) return e;
	if (g != ncclSuccess) return g;
	if (rank == root && !loopback)
		return shard_detail::copy_rows(local_rows, local_stride, root_rows + (long long) lo * root_stride, root_stride, hi - lo, n, stream);
	return ncclSuccess;
}

// The inverse: every rank's shard of n output samples per channel (the same n on every rank: all follow one schedule --
// a rank without channels passes the n the others report, or 0 rows) -> rows [channels][n] on `root`.
inline ncclResult_t gather_channels(const double* local_rows, long long local_stride, int channels, int n,
	double* root_rows, long long root_stride, int root, int rank, int world, ncclComm_t comm, hipStream_t stream,
	bool loopback = false)
{
	if (channels < 1 || n < 0 || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world ||
		local_stride < n || (rank == root && root_stride < n)) return ncclInvalidArgument;
	int lo, hi;
	channel_shard(channels, rank, world, &lo, &hi);
	if (n == 0) return ncclSuccess;
	ncclResult_t e = ncclGroupStart();
	if (e != ncclSuccess) return e;
	if (rank == root)
	{
		for (int r = 0; r < world && e == ncclSuccess; r++)
		{
			int a, b;
			channel_shard(channels, r, world, &a, &b);
			if (r == root && !loopback) continue;
			e = shard_detail::recv_rows(root_rows + (long long) a * root_stride, root_stride, b - a, n, r, comm, stream);
		}
		if (loopback && e == ncclSuccess) e = shard_detail::send_rows(local_rows, local_stride, hi - lo, n, root, comm, stream);
	}
	else e = shard_detail::send_rows(local_rows, local_stride, hi - lo, n, root, comm, stream);
	const ncclResult_t g = ncclGroupEnd();
	if (e != ncclSuccess) return e;
	if (g != ncclSuccess) return g;
	if (rank == root && !loopback)
		return shard_detail::copy_rows(root_rows + (long long) lo * root_stride, root_stride, local_rows, local_stride, hi - lo, n, stream);
	return ncclSuccess;
}

} // namespace r8b

#endif
