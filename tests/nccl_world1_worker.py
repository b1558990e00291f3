"""tests/nccl_world1_worker.py -- run by tests/test_gpu_parity.py::test_root_pipeline_on_one_gpu_nccl in a process of its
own: a world-size-1 process group on the NCCL (= RCCL) backend on cuda:0, the scatter -> resample -> gather pipeline of
r8brain-free-src_amd/sharding.py with its side stream, events and record_stream calls LIVE (under gloo / CPU tensors they
are no-ops), >= 50 pipelined steps, against the unsharded object bit for bit.  With one rank the shard moves are local
copies (no peer exists), so this exercises the stream logic and RCCL's initialisation, not xGMI.  Prints OK <n> <sum>."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port = sys.argv[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    r8b = importlib.import_module("r8brain-free-src_amd")
    # one real collective through RCCL (world 1: the identity), so that the communicator exists
    t = torch.ones(4, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    assert float(t.sum().item()) == 4.0
    total, L, steps = 10, 3000, 56
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    batches = [torch.rand((total, L), generator=g, dtype=torch.float64, device=dev) * 2.0 - 1.0 for _ in range(steps)]
    sh = r8b.ShardedBatchResampler(
        lambda nch: r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=nch, device=0), total)
    pipe = r8b.RootPipeline(sh, L, root=0, device=dev)
    assert pipe.cuda and pipe.side is not None
    outs = pipe.run(batches)
    torch.cuda.synchronize()
    plain = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=total, device=0)
    n = 0
    acc = 0.0
    for i in range(steps):
        y = plain.process(batches[i]).clone()
        torch.cuda.synchronize()
        assert outs[i].shape == y.shape, (i, outs[i].shape, y.shape)
        assert torch.equal(outs[i], y), "step %d differs from the unsharded object" % i
        n += y.shape[1]
        acc += float(y.sum().item())
    # ... and the one-call form
    sh2 = r8b.ShardedBatchResampler(
        lambda nch: r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=nch, device=0), total)
    plain.clear()
    for i in range(3):
        a = sh2.process_from_root(batches[i], L, root=0, device=dev)
        b = plain.process(batches[i])
        assert torch.equal(a, b)
    dist.barrier()
    dist.destroy_process_group()
    print("OK %d %r" % (n, acc))


if __name__ == "__main__":
    main()
