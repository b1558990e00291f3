"""CPU tier, 2 processes over gloo: the multi-GPU path of bench.py shards CHANNELS across ranks
with no data-path collective (channels are independent streams, reference README.md:53-55).
Checked here without GPUs: every rank derives its shard from (rank, world) alone, shards are
disjoint and cover the batch, each rank's per-call output counts (host plan through the C ABI) are
identical -- so the only cross-rank traffic bench.py needs is the barrier and the MAX of the timed
region, which are exercised with the same torch.distributed calls.  The scatter -> resample ->
gather path (a batch that lives on one rank) runs the real engine on every rank through the host
emulation of the kernels and must reproduce the unsharded result bit for bit."""
import importlib
import os
import socket
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def emul_built():
    import subprocess
    subprocess.run(["make"], cwd=os.path.join(ROOT, "tests", "emul"), check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    r8b = importlib.import_module("r8brain-free-src_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 2048
    lo, hi = bench.channel_shard(total, rank, world)
    lib = r8b.load()
    p = lib.r8b_plan_create(44100.0, 96000.0, 16384, 2.0, 180.15)
    counts = [lib.r8b_plan_step(p, 16384) for _ in range(4)]
    t = torch.tensor([float(hi - lo), float(sum(counts)), 0.001 * (rank + 1)], dtype=torch.float64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    dist.barrier()
    # scatter a batch that lives on rank 0, resample every shard with the ENGINE (host emulation of the
    # kernels, tests/emul: the same schedule and arithmetic as the HIP path), gather it back: the
    # sharded result must equal the unsharded one bit for bit, call by call
    import numpy as np
    emul = r8b.bind(os.path.join(ROOT, "tests", "emul", "_build", "libr8bsrc_emul.so"))

    class Local:
        def __init__(self, nch):
            self.rs = r8b.BatchResampler(44100.0, 96000.0, 2048, 2.0, 180.15, nch=nch, lib=emul)

        def process(self, x):
            return torch.from_numpy(self.rs.process_host(x.numpy()))

    total_ch, L, calls = 10, 2048, 3
    sh = r8b.ShardedBatchResampler(Local, total_ch)
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from cases import make_input
        xall = make_input(total_ch, L * calls, 3)
        whole = Local(total_ch)
    for i in range(calls):
        full = torch.from_numpy(np.ascontiguousarray(xall[:, i * L:(i + 1) * L])) if rank == 0 else None
        back = sh.process_from_root(full, L, root=0, device="cpu")
        if rank == 0:
            ref = whole.process(full)
            assert back.shape == ref.shape and back.shape[1] > 0 or i == 0
            assert torch.equal(back, ref)
        else:
            assert back is None
    # the same through the double-buffered pipeline bench.py --e2e uses (fresh streams)
    sh2 = r8b.ShardedBatchResampler(Local, total_ch)
    pipe = r8b.RootPipeline(sh2, L, root=0, device="cpu")
    batches = [torch.from_numpy(np.ascontiguousarray(xall[:, i * L:(i + 1) * L])) if rank == 0 else None
               for i in range(calls)]
    got = pipe.run(batches)
    if rank == 0:
        whole2 = Local(total_ch)
        for i in range(calls):
            assert torch.equal(got[i], whole2.process(batches[i]))
    else:
        assert all(g is None for g in got)
    # ... and with the three result buffers in rotation (keep_last_only, what bench.py --e2e asks for): per-call counts
    # that change from call to call (44100 -> 96000: 4457 / 4458 / ...) still land in the SAME three buffers, as dense
    # [channels, n] results (a fresh tensor per call defeated the rotation whenever n != max_out_len)
    class LocalMax(Local):
        def __init__(self, nch):
            super().__init__(nch)
            self.max_out_len = self.rs.max_out_len
    calls3 = 7
    sh3 = r8b.ShardedBatchResampler(LocalMax, total_ch)
    pipe3 = r8b.RootPipeline(sh3, L, root=0, device="cpu", keep_last_only=True)
    if rank == 0:
        xall3 = make_input(total_ch, L * calls3, 5)
    whole3 = Local(total_ch) if rank == 0 else None
    ptrs, lens3 = set(), set()
    for i in range(calls3):
        b3 = [torch.from_numpy(np.ascontiguousarray(xall3[:, i * L:(i + 1) * L])) if rank == 0 else None]
        g3 = pipe3.run(b3)[0]
        if rank == 0:
            assert g3.is_contiguous() and torch.equal(g3, whole3.process(b3[0]))
            ptrs.add(g3.data_ptr())
            lens3.add(g3.shape[1])
        else:
            assert g3 is None
    if rank == 0:
        # (run() is called once per batch here, so every call uses buffer 0 of the rotation: one address)
        assert len(lens3) > 1 and len(ptrs) == 1, (lens3, ptrs)
    tmax = t[2:3].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put(([(float(g[0]), float(g[1])) for g in gathered], float(tmax), (lo, hi)))
    dist.destroy_process_group()


def test_channel_sharding_two_ranks(emul_built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per_rank, tmax, shard0 = res
    assert sum(n for n, _ in per_rank) == 2048 and all(n == 1024 for n, _ in per_rank)
    assert per_rank[0][1] == per_rank[1][1] > 0     # same schedule on every rank
    assert abs(tmax - 0.002) < 1e-12                 # MAX over ranks
    assert shard0 == (0, 1024)


def _bench_cmd(gpus, extra=()):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "2",
            "--settle", "0", "--channels", "6", "--block", "2048", "--backend", "gloo",
            "--lib", os.path.join(ROOT, "tests", "emul", "_build", "libr8bsrc_emul.so")] + list(extra)


def _bench_env():
    return {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_starts_its_own_ranks(emul_built, gpus):
    """`python3 bench.py --gpus N` with WORLD_SIZE unset -- the driver's command shape -- starts its N
    ranks itself (torch.distributed.run, 127.0.0.1, a free port) and prints ONE line from rank 0 with n_gpus = N.  Run
    here on gloo over the host emulation of the kernels (the --backend gloo --lib side door): the control flow of the
    N-rank bench, not a measurement."""
    import json
    import subprocess
    r = subprocess.run(_bench_cmd(gpus), env=_bench_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["channels_per_gpu"] == 6
    # whole-job value: N ranks x 6 channels x 2048 samples x 3 steps over the MAX-over-ranks time
    assert abs(d["value"] - gpus * 6 * 2048 / d["ms_per_step"] / 1e3) / d["value"] < 0.01
    assert "HOST EMULATION" in d["data"] and "cpu_baseline" not in d


def test_bench_under_a_launcher_is_a_rank(emul_built):
    """... and launched by torch.distributed.run (the contract's other shape) it must not spawn again"""
    import json
    import subprocess
    cmd = _bench_cmd(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + cmd[1:]
    r = subprocess.run(cmd, env=_bench_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_exits_nonzero_when_a_rank_fails(emul_built):
    """a rank that cannot start (here: the product path without a GPU) fails the whole command, no JSON line"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_bench_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.split("\n") if l.startswith("{")]
    # mismatched launcher: --gpus 2 inside a one-rank world is refused, not silently run as one rank
    env = dict(_bench_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(_bench_cmd(2), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE 1" in r.stderr


def test_channel_shard_function():
    sys.path.insert(0, ROOT)
    import bench
    for total, world in [(8192, 8), (1000, 3), (7, 8), (1, 1)]:
        cover = []
        for r in range(world):
            lo, hi = bench.channel_shard(total, r, world)
            assert 0 <= lo <= hi <= total
            cover += list(range(lo, hi))
        assert cover == list(range(total))


def _few_worker(rank, world, port, q):
    """fewer channel pairs than ranks: some shards are empty, possibly the root's own"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    r8b = importlib.import_module("r8brain-free-src_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Local:  # a stand-in stage with a fixed schedule: n outputs = 2 * inputs - 3
        def __init__(self, nch):
            self.nch = nch

        def process(self, x):
            assert x.shape[0] == self.nch
            return torch.cat([x, -x], dim=1)[:, :2 * x.shape[1] - 3].contiguous()

    ok = True
    for total in (2, 5):
        for root in (0, world - 1):  # (the last rank owns nothing when total = 2)
            sh = r8b.ShardedBatchResampler(Local, total)
            assert (sh.local is None) == (sh.hi <= sh.lo)
            full = (torch.arange(total * 8, dtype=torch.float64).reshape(total, 8) + 1.0) if rank == root else None
            back = sh.process_from_root(full, 8, root=root, device="cpu")
            if rank == root:
                ok = ok and back is not None and tuple(back.shape) == (total, 13)
                ok = ok and torch.equal(back, Local(total).process(full))
            else:
                ok = ok and back is None
            pipe = r8b.RootPipeline(r8b.ShardedBatchResampler(Local, total), 8, root=root, device="cpu")
            got = pipe.run([full, full])
            if rank == root:
                ok = ok and all(torch.equal(g, Local(total).process(full)) for g in got)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(float(t.item()))
    dist.destroy_process_group()


def test_fewer_channel_pairs_than_ranks_four_ranks():
    """stereo on 4 ranks -- rank 0 owns the pair, ranks 1-3 nothing; gather must neither hang
    nor return an empty result, also when the destination itself owns no channels"""
    import torch.multiprocessing as mp
    r8b = importlib.import_module("r8brain-free-src_amd")
    assert r8b.channel_shard(2, 0, 4) == (0, 2) and r8b.channel_shard(2, 3, 4) == (2, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_few_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == 1.0


def test_native_channel_shards_equal_the_python_ones(tmp_path):
    """include/r8b/ShardTransfer.h (the C++ / RCCL twin of sharding.scatter_channels / gather_channels): its shard
    boundaries are sharding.channel_shard's for every (channels, world) of a sweep; the header and its GPU test compile
    against the ROCm headers (the transfers themselves run in the GPU tier on a one-rank world)."""
    import importlib
    import os
    import shutil
    import subprocess
    import pytest
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h") or shutil.which("g++") is None:
        pytest.skip("ROCm headers / g++ not here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sharding = importlib.import_module("r8brain-free-src_amd.sharding")
    src = tmp_path / "shards.cpp"
    src.write_text('#include <cstdio>\n#include "%s/include/r8b/ShardTransfer.h"\n'
                   'int main() { for (int c = 1; c <= 40; c++) for (int w = 1; w <= 9; w++) for (int r = 0; r < w; r++) {'
                   ' int lo, hi; r8b::channel_shard(c, r, w, &lo, &hi); std::printf("%%d %%d %%d %%d %%d\\n", c, w, r, lo, hi); }'
                   ' int lo, hi; r8b::channel_shard(8192, 5, 8, &lo, &hi); std::printf("8192 8 5 %%d %%d\\n", lo, hi); return 0; }\n' % root)
    exe = str(tmp_path / "shards")
    subprocess.run(["g++", "-std=c++17", "-O0", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
    n = 0
    for line in out:
        if line.strip():
            c, w, r, lo, hi = (int(v) for v in line.split())
            assert (lo, hi) == sharding.channel_shard(c, r, w), line
            n += 1
    assert n > 1000
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(root, "tests", "cxx_rccl_world1.cpp")], check=True)
