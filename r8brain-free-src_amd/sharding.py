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


def _rank_world():
    """(rank, world) of the default process group; (0, 1) when none is initialised (single GPU)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def channel_shard(total_channels, rank, world):
    """[lo, hi) of the channels owned by `rank`."""
    # whole channel PAIRS per rank: the pair kernel packs channels 2c and 2c+1 into one complex
    # transform, so a shard boundary between them would change which channels share a transform (and
    # with it the last bits of their samples); with pairs kept together sharded == unsharded bit for bit
    # (ceilings: when there are fewer pairs than ranks the LOW ranks get them, so that rank 0 -- the usual
    # root of scatter / gather -- owns channels whenever anybody does)
    pairs = (total_channels + 1) // 2
    lo = min(2 * -(-pairs * rank // world), total_channels)
    hi = min(2 * -(-pairs * (rank + 1) // world), total_channels)
    return lo, hi


def _some_shard_empty(total_channels, world):
    return any(channel_shard(total_channels, r, world)[1] <= channel_shard(total_channels, r, world)[0]
               for r in range(world))


def _run_p2p(ops):
    """All point-to-point transfers of one scatter / gather as ONE group (dist.batch_isend_irecv: a
    single ncclGroupStart/End on RCCL), so that every xGMI link carries its shard at the same time
    instead of one transfer after the other."""
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def scatter_channels(x_full, total_channels, length, src=0, device=None, dtype=torch.float64, out=None):
    """x_full: [total_channels, length] on rank `src` (ignored elsewhere).  Returns this rank's
    shard [hi-lo, length] (in `out` when given: a preallocated [hi-lo, >= length] buffer).  One grouped send per
    peer: on xGMI each point-to-point link carries exactly one shard and all links are busy together."""
    rank, world = _rank_world()
    lo, hi = channel_shard(total_channels, rank, world)
    device = device if device is not None else (x_full.device if x_full is not None else "cpu")
    local = out[:, :length] if out is not None else torch.empty((hi - lo, length), dtype=dtype, device=device)
    if world > 1 and not local.is_contiguous():
        local = torch.empty((hi - lo, length), dtype=dtype, device=device)  # (receives need a dense buffer)
    ops = []
    if rank == src:
        keep = []
        for r in range(world):
            a, b = channel_shard(total_channels, r, world)
            if r == src:
                local.copy_(x_full[a:b, :length])
            elif b > a:
                keep.append(x_full[a:b, :length].contiguous())
                ops.append(dist.P2POp(dist.isend, keep[-1], r))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, local, src))
    _run_p2p(ops)
    return local


def gather_channels(y_local, total_channels, dst=0, n=None, out=None):
    """Inverse of scatter_channels for the per-rank outputs [hi-lo, n] (same n on every rank that owns
    channels: all ranks follow the same schedule).  Returns [total_channels, n] on `dst`, None elsewhere.
    A rank WITHOUT channels (fewer channel pairs than ranks) has no resampler and so no n of its own:
    whenever some shard is empty -- every rank can tell from (total_channels, world) alone -- the ranks
    agree on n with one all_reduce(MAX) first (or the caller passes n), so that a channel-less `dst`
    still allocates the full result and posts a receive for every sender.

    `out` (optional, `dst` only): STORAGE for the result, not a tensor to read the result through.  A contiguous
    tensor with room for total_channels * n elements of y_local's dtype; its first total_channels * n elements are
    used as a DENSE [total_channels, n] tensor, which is what is returned -- row c starts at element c * n of the
    storage, NOT at c * out.shape[1].  A caller that passes rows wider than n must read the RETURNED tensor; reading
    its own `out` through its original shape yields scrambled rows.  An `out` that cannot hold the
    result (too small, another dtype, not contiguous) is a ValueError rather than a silent fresh allocation."""
    rank, world = _rank_world()
    if n is None:
        n = y_local.shape[1]
        if world > 1 and _some_shard_empty(total_channels, world):
            t = torch.tensor([n], dtype=torch.int64, device=y_local.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n = int(t.item())
    if rank == dst:
        # (`out`: a preallocated contiguous buffer of the caller with room for [total_channels, n], e.g. RootPipeline's
        # rotation of [total_channels, max_out_len] tensors: its storage is used as a DENSE [total_channels, n] result --
        # a [:, :n] view of wider rows is not contiguous, and a receive wants contiguous rows)
        if out is not None:
            if not (out.is_contiguous() and out.numel() >= total_channels * n and out.dtype == y_local.dtype):
                raise ValueError("gather_channels: `out` must be contiguous storage for %d x %d elements of %s"
                                 % (total_channels, n, y_local.dtype))
            out = out.reshape(-1)[:total_channels * n].view(total_channels, n)
        else:
            out = torch.empty((total_channels, n), dtype=y_local.dtype, device=y_local.device)
        ops = []
        for r in range(world):
            a, b = channel_shard(total_channels, r, world)
            if r == dst:
                if b > a:
                    out[a:b].copy_(y_local)
            elif b > a and n > 0:
                ops.append(dist.P2POp(dist.irecv, out[a:b], r))
        _run_p2p(ops)
        return out
    if y_local.shape[0] > 0 and n > 0:
        _run_p2p([dist.P2POp(dist.isend, y_local.contiguous(), dst)])
    return None


class ShardedBatchResampler:
    """`total_channels` streams partitioned over the ranks of the default process group; each rank
    holds a BatchResampler for its shard.  process() works on the local shard (data sharded at
    rest); process_from_root() scatters a batch that lives on one rank, resamples, and gathers."""

    def __init__(self, factory, total_channels):
        """factory(nch) -> object with process(x_local) -> y_local (e.g. a BatchResampler)"""
        self.total = int(total_channels)
        self.rank, self.world = _rank_world()
        self.lo, self.hi = channel_shard(self.total, self.rank, self.world)
        self.local = factory(self.hi - self.lo) if self.hi > self.lo else None

    def process(self, x_local):
        return self.local.process(x_local)

    def process_from_root(self, x_full, length, root=0, device=None):
        x = scatter_channels(x_full, self.total, length, src=root, device=device)
        y = self.local.process(x) if self.local is not None else x[:, :0]
        return gather_channels(y, self.total, dst=root)


class RootPipeline:
    """A batch that lives on ONE rank (SURVEY.md 8e "end-to-end"): per call scatter the channel shards,
    resample them everywhere, gather the outputs back -- double buffered, so that the transfers of
    call i+1 (out-bound) and call i-1 (in-bound) run on a side stream while call i is resampled.
    On xGMI every peer link carries exactly one shard per direction and call (grouped sends).

    `sharded`: a ShardedBatchResampler; `length`: samples per channel and call.  With CPU tensors
    (gloo, the tests) there are no streams and the steps simply run in order."""

    def __init__(self, sharded, length, root=0, device=None, keep_last_only=False):
        """keep_last_only: run() reuses three result buffers in rotation (a streaming consumer that is done with a
        result before the third call after it); False: every result is a tensor of its own."""
        self.sh, self.length, self.root = sharded, int(length), root
        self.keep_last_only = bool(keep_last_only)
        self.device = torch.device(device if device is not None else "cpu")
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        self._ybuf = [None, None]  # two output buffers in rotation (process(out=)): no copy per call
        # shard and result buffers in rotation as well: a buffer that has been used on two streams goes back to the
        # caching allocator only when both have passed it, so a fresh allocation per call ends in a device malloc per
        # call (measured on one GPU: 8.5 ms per step instead of 0.6)
        self._xbuf = [None, None, None]
        self._obuf = [None, None, None]
        import inspect
        loc = sharded.local
        try:
            self._has_out = loc is not None and "out" in inspect.signature(loc.process).parameters
        except (TypeError, ValueError):
            self._has_out = False

    def _on_side(self):
        import contextlib
        return torch.cuda.stream(self.side) if self.cuda else contextlib.nullcontext()

    def run(self, batches):
        """batches: list of [total_channels, length] tensors on the root (anything elsewhere; only
        the length of the list matters there).  Returns the list of gathered outputs on the root
        (None elsewhere).  Call i's scatter is issued before call i-1's gather, both on the side
        stream; the resampling of call i-1 runs on the current stream in between."""
        n = len(batches)
        outs = [None] * n
        shards = [None] * n
        ready = [None] * n   # scatter of call i complete (event on the side stream)
        done = [None] * n    # resampling of call i complete (event on the current stream)
        rank = self.sh.rank

        def scatter(i):
            with self._on_side():
                x = batches[i] if rank == self.root else None
                if self.cuda and self._xbuf[i % 3] is None:
                    self._xbuf[i % 3] = torch.empty((self.sh.hi - self.sh.lo, self.length), dtype=torch.float64,
                                                    device=self.device)
                # (shard i - 3 was consumed by the resampling of call i - 3, which the gather of that call -- queued on
                # this side stream before this scatter -- waited for)
                shards[i] = scatter_channels(x, self.sh.total, self.length, src=self.root,
                                             device=self.device, out=self._xbuf[i % 3] if self.cuda else None)
                if self.cuda:
                    ready[i] = torch.cuda.Event()
                    ready[i].record(self.side)

        def gather(i, y):
            with self._on_side():
                if self.cuda:
                    self.side.wait_event(done[i])
                ob = None
                if self.keep_last_only and rank == self.root and hasattr(self.sh.local, "max_out_len"):
                    # (three result buffers in rotation, only when the caller asked for them -- keep_last_only --: the
                    # caller gets the tensors the gather RETURNS; a result stays valid until the third call after it has
                    # been gathered.  Each buffer is sized for max_out_len per channel and handed to the gather as
                    # STORAGE: the result is the dense [total, n] tensor over its first total * n elements (not a
                    # [:, :n] view of max_out_len-wide rows), so the rotation also holds when the per-call count
                    # changes -- 35666 / 35667 for 44100 -> 96000 -- and nobody reads the buffer through its own shape)
                    if self._obuf[i % 3] is None:
                        self._obuf[i % 3] = torch.empty((self.sh.total, self.sh.local.max_out_len), dtype=y.dtype,
                                                        device=self.device)
                    ob = self._obuf[i % 3]
                outs[i] = gather_channels(y, self.sh.total, dst=self.root, out=ob)

        if n:
            scatter(0)
        for i in range(n):
            if i + 1 < n:
                scatter(i + 1)
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_event(ready[i])
            x = shards[i]
            if self.cuda:
                # allocated on the side stream, read on this one: tell the caching allocator
                x.record_stream(torch.cuda.current_stream(self.device))
            if self.sh.local is None:
                y = x[:, :0]
            elif self._has_out and hasattr(self.sh.local, "max_out_len"):
                # the gather of call i-2 out of this buffer was queued on the side stream before scatter(i),
                # which `ready[i]` (waited for above) follows: the buffer is free again
                if self._ybuf[i & 1] is None:
                    self._ybuf[i & 1] = torch.empty((x.shape[0], self.sh.local.max_out_len), dtype=x.dtype,
                                                    device=x.device)
                y = self.sh.local.process(x, out=self._ybuf[i & 1])
            else:
                y = self.sh.local.process(x)
                if self.cuda:
                    y = y.clone()  # the resampler reuses its output buffer on the next call
            if self.cuda:
                y.record_stream(self.side)
                done[i] = torch.cuda.Event()
                done[i].record(torch.cuda.current_stream(self.device))
            gather(i, y)
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return outs
