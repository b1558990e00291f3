"""Channel sharding across the GPUs of one node.

Channels are independent streams (reference README.md:53-55: one resampler object per stream), so
the only multi-GPU structure is a partition of the channel axis: rank r owns the contiguous block
`channel_shard(total, r, world)`; every rank runs its own BatchResampler on its shard with its own
tables and history and no collective on the data path.

When a whole batch lives on one rank, `scatter_channels` / `gather_channels` move the shards with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Per
call of BASELINE config 4 that is 128 MiB out and 279 MiB back per peer link, several times the
compute time, so keep data sharded at rest whenever the producer/consumer allow it (bench.py does).
"""
import torch
import torch.distributed as dist


def channel_shard(total_channels, rank, world):
    """[lo, hi) of the channels owned by `rank`."""
    lo = total_channels * rank // world
    hi = total_channels * (rank + 1) // world
    return lo, hi


def scatter_channels(x_full, total_channels, length, src=0, device=None, dtype=torch.float64):
    """x_full: [total_channels, length] on rank `src` (ignored elsewhere).  Returns this rank's
    shard [hi-lo, length].  One send per peer, so on xGMI each point-to-point link carries exactly
    one shard."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = channel_shard(total_channels, rank, world)
    device = device if device is not None else (x_full.device if x_full is not None else "cpu")
    local = torch.empty((hi - lo, length), dtype=dtype, device=device)
    if rank == src:
        reqs = []
        for r in range(world):
            a, b = channel_shard(total_channels, r, world)
            if r == src:
                local.copy_(x_full[a:b, :length])
            elif b > a:
                reqs.append(dist.isend(x_full[a:b, :length].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif hi > lo:
        dist.recv(local, src=src)
    return local


def gather_channels(y_local, total_channels, dst=0):
    """Inverse of scatter_channels for the per-rank outputs [hi-lo, n] (same n on every rank: all
    ranks follow the same schedule).  Returns [total_channels, n] on `dst`, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n = y_local.shape[1]
    if rank == dst:
        out = torch.empty((total_channels, n), dtype=y_local.dtype, device=y_local.device)
        reqs = []
        for r in range(world):
            a, b = channel_shard(total_channels, r, world)
            if r == dst:
                out[a:b].copy_(y_local)
            elif b > a:
                reqs.append((dist.irecv(out[a:b], src=r), None))
        for q, _ in reqs:
            q.wait()
        return out
    if y_local.shape[0] > 0:
        dist.send(y_local.contiguous(), dst=dst)
    return None


class ShardedBatchResampler:
    """`total_channels` streams partitioned over the ranks of the default process group; each rank
    holds a BatchResampler for its shard.  process() works on the local shard (data sharded at
    rest); process_from_root() scatters a batch that lives on one rank, resamples, and gathers."""

    def __init__(self, factory, total_channels):
        """factory(nch) -> object with process(x_local) -> y_local (e.g. a BatchResampler)"""
        self.total = int(total_channels)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.lo, self.hi = channel_shard(self.total, self.rank, self.world)
        self.local = factory(self.hi - self.lo) if self.hi > self.lo else None

    def process(self, x_local):
        return self.local.process(x_local)

    def process_from_root(self, x_full, length, root=0, device=None):
        x = scatter_channels(x_full, self.total, length, src=root, device=device)
        y = self.local.process(x) if self.local is not None else x[:, :0]
        return gather_channels(y, self.total, dst=root)
