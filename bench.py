#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched 44100->96000 fp64 resampling on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a launcher (no WORLD_SIZE in the environment) and with --gpus N > 1 the script starts its own N ranks --
it re-executes itself through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free port>` --, forwards rank 0's ONE JSON line and exits non-zero if any rank failed; started by
torch.distributed.run (the driver's multi-GPU shape) it is a rank.

A "step" is one process() call of the whole hot path (r8b::CDSPResampler::process semantics) over
one batch of synthetic input: 1024 channels x 16384 samples per GPU (BASELINE.json configs[1];
configs[3] = 8192 channels over 8 GPUs is the same per-GPU shard, i.e. weak scaling).  Channels
are independent streams, so ranks own disjoint channel shards and the data path has no collective;
torch.distributed (RCCL) is used for the barrier and the MAX over ranks of the timed region only.
Inputs are resident in HBM before the timed region starts; outputs stay in HBM.

`value` / `ms_per_step` are ALWAYS the K steps timed straight after the W warm-up calls (barrier + synchronize on both
sides), with every call's outputs written from column 0 of the output rows (`config.placement`).  Side fields (with
--settle N > 0, the default): `settled` = the same K steps timed again after N further untimed calls (the board's
power controller answers the load step after an idle gap with a clock dip that a 5 + 20-call window falls into), and
`other_placement` = that window once more with the stream-aligned output placement (--align-out).

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel, timed live with HIP events on the
                  launching stream (engine option "timing"), algorithmic bytes 8*(N_in+N_out);
  "cpu_baseline": the real reference (oracle/_ref, kind "reference") or, if that library did not
                  travel, the numpy restatement (kind "port"), timed on this box's host cores on a
                  bounded sample of the same workload.  The oracle is used here as the timed
                  baseline and as the checker of a sample channel only -- never as the product.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def channel_shard(total_channels, rank, world):
    """Contiguous channel block [lo, hi) owned by `rank` (r8brain-free-src_amd/sharding.py)."""
    # whole channel PAIRS per rank: the pair kernel packs channels 2c and 2c+1 into one complex
    # transform, so a shard boundary between them would change which channels share a transform (and
    # with it the last bits of their samples); with pairs kept together sharded == unsharded bit for bit
    pairs = (total_channels + 1) // 2
    lo = min(2 * -(-pairs * rank // world), total_channels)
    hi = min(2 * -(-pairs * (rank + 1) // world), total_channels)
    return lo, hi


def measured_traffic(kernel, cfg="cfg2"):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/pmc.sh
    (FETCH_SIZE x2 per MI355X_MICROARCH.md section HBM, + WRITE_SIZE), as recorded in
    profiles/traffic.json for this build's default workload; None if not recorded.  Counters
    cannot be read from inside the timed process, so this is the committed measurement."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return rec.get(cfg + ":" + kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def kernel_ms(timings, steps):
    """ms per step per kernel name (stages that run the same kernel are summed)."""
    out = {}
    for name, ms, launches, _, _ in timings:
        out[name] = round(out.get(name, 0.0) + ms / max(steps, 1), 4)
    return out


def splitmix_uniform(seed, n):
    """splitmix64 -> uniform [-1, 1) doubles (vectorised numpy)."""
    import numpy as np
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def host_cpu_info():
    """Usable host cores: the affinity mask capped by the cgroup CPU quota (os.cpu_count() reports
    the machine, not what this container may use)."""
    info = {"cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = info["cpu_count"]
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    info["cgroup_quota"] = quota
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    info["cpu_model"] = model
    usable = info["affinity"]
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    info["usable"] = usable
    return info


def cpu_baseline(src, dst, L, gpu_sample=None, budget_s=24.0):
    """Reference CPU path on a bounded sample of the same workload: one
    CDSPResampler24 per channel, channels statically partitioned over T std::threads, timing only
    the process() loops (reference bench/r8bfreesrc.cpp:118-126); T = 1 and T = all usable cores,
    >= 2 s timed each, median of three repeats."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    cpu = host_cpu_info()
    try:
        import refwrap as R
        have_ref = R.available()
    except Exception:
        have_ref = False
    if have_ref:
        def leg(threads, seconds):
            nch = threads * 2
            t = R.bench(src, dst, L, nch, 1, 2, threads)          # calibration
            per_call = max(t["seconds"] / 2, 1e-6)                # seconds per call round
            calls = int(min(max(seconds / 3.0 / per_call, 4), 100000))
            runs = [R.bench(src, dst, L, nch, 2, calls, threads) for _ in range(3)]
            rates = sorted(r["in_samples"] / r["seconds"] / 1e6 for r in runs)
            return {"threads": runs[0]["threads"], "value": round(rates[1], 3),
                    "per_core": round(rates[1] / runs[0]["threads"], 3),
                    "spread": round((rates[2] - rates[0]) / rates[1], 4),
                    "timed_s": round(sum(r["seconds"] for r in runs), 2),
                    "sample": "%d channels x %d calls x %d samples, 3 repeats" % (nch, calls, L),
                    "flags": runs[0]["flags"]}
        t1 = leg(1, budget_s * 0.4)
        tall = leg(cpu["usable"], budget_s * 0.6) if cpu["usable"] > 1 else t1
        res = {"value": tall["value"], "unit": "Msamples/s", "cores": tall["threads"],
               "kind": "reference",
               "sample": "CDSPResampler24 %g->%g, one resampler per channel, process() loops only; "
                         "T=all: %s; T=1: %s; reference built %s" %
                         (src, dst, tall["sample"], t1["sample"], tall["flags"]),
               "per_core": tall["per_core"], "t1": t1, "tall": tall, "host": cpu,
               "note": "the reference publishes ~38 Msamples/s per core for this conversion on a "
                       "3.5 GHz desktop core (reference README.md:111-114); server cores with all "
                       "threads busy run lower clocks and share memory bandwidth"}
        if gpu_sample is not None:
            import numpy as np
            rows, xs, ys = gpu_sample  # channel indices, their input streams, what the timed batch object produced
            sq, pk, cnt = 0.0, 0.0, 0
            for x, y in zip(xs, ys):
                r = R.RefResampler(src, dst, L, 2.0, 180.15)
                yr = np.concatenate([r.process(x[i:i + L]) for i in range(0, len(x), L)])
                assert len(yr) == len(y), (len(yr), len(y))
                d = y - yr
                sq += float(np.sum(d * d))
                pk = max(pk, float(np.abs(d).max()))
                cnt += len(y)
            res["gpu_vs_reference"] = {"rms_err": float(np.sqrt(sq / max(cnt, 1))), "peak_err": pk,
                                       "samples": int(cnt), "channels": [int(c) for c in rows],
                                       "what": "rows of the timed batch's own output (the object's first calls, "
                                               "captured during warm-up) against the compiled reference"}
        return res
    import numpy as np
    import r8b_oracle as O
    o = O.OracleResampler(src, dst, L, 2.0, 180.15)
    x = O.splitmix_uniform(1, L * 4)
    t0 = time.perf_counter()
    for i in range(4):
        o.process(x[i * L:(i + 1) * L])
    dt = time.perf_counter() - t0
    return {"value": round(4 * L / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "host": cpu,
            "sample": "1 channel x 4 calls x %d samples through the numpy restatement "
                      "(oracle/_ref not present)" % L}


def spawn_ranks(n):
    """--gpus N > 1 without a launcher: run this very command line as N ranks of one node under
    torch.distributed.run (RCCL rendezvous on 127.0.0.1, a free port), pass rank 0's JSON line through, return the
    launcher's exit status (non-zero when any rank failed)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL across processes needs it on these hosts)
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    other = [ln for ln in r.stdout.splitlines() if ln not in lines]
    if other:
        sys.stderr.write("\n".join(other) + "\n")
    if r.returncode == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: %d JSON lines from %d ranks (expected one, from rank 0)\n" % (len(lines), n))
        return 1
    for ln in lines:
        print(ln, flush=True)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--block", type=int, default=16384, help="input samples per channel per step")
    ap.add_argument("--src", type=float, default=44100.0)
    ap.add_argument("--dst", type=float, default=96000.0)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg5"], default=None,
                    help="BASELINE.json config preset: cfg2 = the default (1024 ch x 16384, 44100->96000), "
                         "cfg3 = 1024 ch x 16384, 96000->44100, cfg5 = 64 ch x 1024, 44100->2822400; the "
                         "same JSON line (with its own roofline block) for each")
    ap.add_argument("--tb", type=float, default=2.0, help="transition band, percent (side runs)")
    ap.add_argument("--atten", type=float, default=180.15, help="stop-band attenuation (side runs)")
    ap.add_argument("--phase", type=int, default=0, choices=[0, 1],
                    help="1 = minimum-phase filters (side runs; the CPU leg and the error report are linear phase: use "
                         "--no-cpu)")
    ap.add_argument("--settle", type=int, default=150,
                    help="`value` is ALWAYS the K steps timed straight after the W warm-up calls (the driver's contract).  "
                         "With --settle N > 0 the line also carries, as side fields, the same K steps timed again after N further "
                         "untimed calls (`settled`: after an idle gap the board's power controller answers the load step with a "
                         "clock dip that lasts about 40 calls of this batch and has fully recovered after about 150, DESIGN.md "
                         "section 5 -- a 5 + 20 call window lies inside it) and once more with the other output placement "
                         "(`other_placement`, see --align-out); 0: neither")
    ap.add_argument("--align-out", type=int, default=0,
                    help="where the caller (this script) puts a call's outputs.  0 (default, `placement` \"column 0\"): every "
                         "call writes its rows from column 0 of the output buffer, as a caller of the reference's process() "
                         "does; 1 (\"stream-aligned\"): at column (outputs so far) mod 8 of 64-byte-aligned rows, so that every "
                         "64-byte piece the kernel stores is a whole aligned segment in every call (INTEGRATION.md section 5).  "
                         "The line reports the other placement's rate beside the headline when --settle > 0")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--e2e", action="store_true",
                    help="side measurement (SURVEY.md 8e): the whole batch lives on rank 0; every step "
                         "scatters the channel shards over xGMI (one grouped send per peer), resamples "
                         "them on all ranks and gathers the outputs back, double buffered; reports the "
                         "end-to-end rate next to the kernel-only one")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value")
    ap.add_argument("--pcm", choices=["s16", "s24", "s32", "f32"], default=None,
                    help="side measurement: interleaved PCM in/out through the ingest/egress "
                         "kernels (r8b_batch_process_pcm) instead of planar fp64; not the headline")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the barrier / MAX-over-ranks.  nccl (= RCCL; default): the "
                         "product on GPUs.  gloo: the CPU tier's side door -- needs --lib with a library that exports "
                         "the same C ABI over HOST memory (tests/emul: the engine's schedule over a host emulation of "
                         "the kernels); tensors stay on the CPU, there is no HIP-event pass and no CPU-baseline leg.  "
                         "It exists so that the N-rank control flow of this script runs where there is no GPU "
                         "(tests/test_dist.py); its numbers mean nothing")
    ap.add_argument("--lib", default=None, help="with --backend gloo: the library to bind instead of the HIP one")
    ap.add_argument("--planar", action="store_true",
                    help="with --pcm: planar [channel][frame] buffers, decoded/encoded inside the "
                         "first/last stage kernels")
    args = ap.parse_args()
    if args.config == "cfg3":
        args.src, args.dst = 96000.0, 44100.0
    elif args.config == "cfg5":
        args.src, args.dst, args.block, args.channels = 44100.0, 2822400.0, 1024, 64
    cfg_name = args.config or ("cfg2" if (args.src, args.dst, args.block, args.channels, args.tb, args.atten) ==
                               (44100.0, 96000.0, 16384, 1024, 2.0, 180.15) else "custom")

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not started by a launcher: start the N ranks ourselves (the driver's `python3 bench.py --gpus N` shape)
        sys.exit(spawn_ranks(args.gpus))
    host_side = args.backend == "gloo"
    if host_side and not args.lib:
        raise SystemExit("--backend gloo is the CPU tier's side door and needs --lib (an emulation library); the "
                         "product has no CPU path")
    if args.lib and not host_side:
        raise SystemExit("--lib goes with --backend gloo only (kernel variants of the HIP library: R8B_HIP_LIB)")
    if host_side and (args.pcm or args.e2e):
        raise SystemExit("--backend gloo: fp64 rows only")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d (the launcher's --nproc-per-node must equal --gpus)" %
                         (args.gpus, world))
    r8b = importlib.import_module("r8brain-free-src_amd")
    if host_side:
        dev = torch.device("cpu")
        lib = r8b.bind(args.lib)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d: --gpus %d but this node shows %d GPU(s)" %
                             (rank, args.gpus, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        lib = None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=dev)

    def dev_sync():
        if not host_side:
            torch.cuda.synchronize()

    # weak scaling: --channels per GPU; the global batch is channels*world, rank r owns
    # channel_shard(channels*world, r, world)
    lo, hi = channel_shard(args.channels * world, rank, world)
    C, L = hi - lo, args.block
    rs = r8b.BatchResampler(args.src, args.dst, L, args.tb, args.atten, nch=C, device=-1 if host_side else local_rank,
                            phase=args.phase, lib=lib)
    for o in args.opt:
        k, v = o.split("=")
        rs.set_option(k, int(v))

    # synthetic input: fp64 uniform noise in [-1, 1) from splitmix64, seed = 1 + global
    # channel index, one continuous stream per channel; a rotation of three resident buffers (= three
    # consecutive blocks of every stream) so that a step's input was not just produced in cache
    nbuf = 3
    host_x = np.stack([splitmix_uniform(1 + lo + c, L * nbuf) for c in range(C)])
    xin = [torch.from_numpy(np.ascontiguousarray(host_x[:, i * L:(i + 1) * L])).to(dev)
           for i in range(nbuf)]
    if args.pcm:
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
    # (output rows on a 64-byte pitch: the interpolator stores pairs of outputs as 16 bytes when both rows of a
    # channel pair are 16-byte aligned; max_out_len itself is odd for this conversion)
    # ... and the caller -- this script -- places a call's outputs at column (outputs produced so far) mod 8 of those
    # rows (--align-out, default): output j of the STREAM then always sits at a column congruent to j mod 8, so the
    # 64-byte pieces the kernel stores (four adjacent phase pairs) are whole aligned 64-byte segments in every call,
    # not only in the calls whose first output index happens to be a multiple of 8
    pitch = (rs.max_out_len + 7) // 8 * 8 + 8
    outs_full = [torch.empty((C, pitch), dtype=torch.float64, device=dev) for _ in range(2)]
    produced = [0]

    placement = [bool(args.align_out)]  # (switched for the `other_placement` window below)

    def out_view(i):
        off = produced[0] % 8 if placement[0] else 0
        return outs_full[i % 2][:, off:off + rs.max_out_len]

    def barrier():
        dev_sync()
        if world > 1:
            dist.barrier()
        dev_sync()

    def process(x, out):
        if not host_side:
            return rs.process(x, out=out)
        # (the side door: the same entry point, r8b_batch_process, over host memory)
        n = rs.process_ptr(x.data_ptr(), x.stride(0), x.shape[1], out.data_ptr(), out.stride(0), 0)
        return out[:, :n]

    if args.pcm:
        fmt = {"s16": r8b.PCM_S16, "s24": r8b.PCM_S24, "s32": r8b.PCM_S32, "f32": r8b.PCM_F32}[args.pcm]
        dt_ = {"s16": torch.int16, "s24": torch.uint8, "s32": torch.int32, "f32": torch.float32}[args.pcm]
        tail = (3,) if args.pcm == "s24" else ()
        if args.pcm == "f32":
            pin = [(x if args.planar else x.t().contiguous()).to(torch.float32) for x in xin]
        else:
            pin = [torch.randint(0, 255, ((C, L) if args.planar else (L, C)) + tail, generator=g,
                                 device=dev, dtype=torch.int32).to(dt_) for _ in range(nbuf)]
        oshape = (C, rs.max_out_len) if args.planar else (rs.max_out_len, C)
        pouts = [torch.empty(oshape + tail, dtype=dt_, device=dev) for _ in range(2)]

    # rows {0, C/2, C-1} of the batch's own output for the first nbuf calls of the object's life (a continuous
    # stream: buffers 0, 1, 2), kept for the error report against the reference (copied during warm-up, not timed)
    chk_rows = sorted(set([0, C // 2, C - 1]))
    captured = []

    def run(k0, k, capture=False):
        n_out = 0
        for i in range(k0, k0 + k):
            if args.pcm:
                n_out += rs.process_pcm(pin[i % nbuf], out_format=fmt, out=pouts[i % 2],
                                        planar=args.planar).shape[1 if args.planar else 0]
            else:
                y = process(xin[i % nbuf], out_view(i))
                produced[0] += y.shape[1]
                n_out += y.shape[1]
                if capture and i < nbuf:
                    captured.append(y[chk_rows].clone())
        return n_out

    e2e = None
    if args.e2e:
        # the batch at rest on rank 0; shards travel every step (RootPipeline: scatter of step i+1 and
        # gather of step i-1 on a side stream while step i is resampled)
        total = args.channels * world
        sh = r8b.ShardedBatchResampler(
            lambda nch: r8b.BatchResampler(args.src, args.dst, L, args.tb, args.atten, nch=nch,
                                           device=local_rank), total)
        pipe = r8b.RootPipeline(sh, L, root=0, device=dev, keep_last_only=True)
        if rank == 0:
            full = [torch.from_numpy(np.stack([splitmix_uniform(1 + c, L * nbuf)[i * L:(i + 1) * L]
                                               for c in range(total)])).to(dev) for i in range(nbuf)]
        else:
            full = [None] * nbuf
        pipe.run([full[i % nbuf] for i in range(args.warmup)])
        barrier()
        t0 = time.perf_counter()
        got = pipe.run([full[i % nbuf] for i in range(args.steps)])
        barrier()
        dte = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dte], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dte = float(t.item())
        e2e = {"ms_per_step": round(dte / args.steps * 1e3, 4),
               "value": round(total * L * args.steps / dte / 1e6, 3), "unit": "Msamples/s",
               "what": "scatter (root -> shards) + resample + gather (-> root) per step, double "
                       "buffered, batch resident on rank 0",
               "out_samples_per_step": int(got[-1].shape[1]) if rank == 0 and got else None}
        del got

    def timed(k0):
        barrier()
        t0 = time.perf_counter()
        n = run(k0, args.steps)
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return n, d

    def window(n, d, what):
        return {"ms_per_step": round(d / args.steps * 1e3, 4),
                "value": round(C * L * args.steps * world / d / 1e6, 3),
                "out_msamples_per_s": round(n * C * world / d / 1e6, 3), "what": what}

    # THE measurement: W untimed warm-up calls, then exactly K timed steps, barrier + synchronize on both sides
    run(0, args.warmup, capture=not args.pcm and not args.no_cpu and world == 1 and args.phase == 0 and not host_side)
    n_out, dt = timed(args.warmup)
    calls = args.warmup + args.steps
    settled = other = None
    if args.settle > 0:
        # side fields: the same K steps once more after `settle` further untimed calls (the stream simply
        # continues), and once more with the other output placement
        run(calls, args.settle)
        calls += args.settle
        n2, dt2 = timed(calls)
        calls += args.steps
        settled = window(n2, dt2, "the same %d steps timed again after %d further untimed calls (clock settled; same "
                                  "placement as `value`)" % (args.steps, args.settle))
        if not args.pcm:
            placement[0] = not placement[0]
            run(calls, 8)
            calls += 8
            n3, dt3 = timed(calls)
            calls += args.steps
            other = window(n3, dt3, "the same %d steps, settled, with the caller's other output placement" % args.steps)
            other["placement"] = "stream-aligned" if placement[0] else "column 0"
            placement[0] = not placement[0]
            run(calls, 8)
            calls += 8

    # second pass, same steps, with per-kernel HIP events (kept out of the headline timing); its own wall time is taken
    # too: wall time per step minus the kernels' time per step = what a step spends BETWEEN kernels (launch gaps)
    rs.set_option("timing", 1)
    dev_sync()
    t_ev0 = time.perf_counter()
    run(calls, args.steps)
    dev_sync()
    ev_wall_ms = (time.perf_counter() - t_ev0) / args.steps * 1e3
    timings = rs.stage_timings()
    symbols = rs.stage_symbols()
    rs.set_option("timing", 0)

    if rank == 0:
        in_samples = C * L * args.steps * world
        value = in_samples / dt / 1e6
        # dominant kernel and its algorithmic bytes: 8*(samples read + samples written) by that
        # kernel per launch (tables excluded)
        # (kernels within 10 % of the longest: the one that moves more bytes, so that the choice does not
        # flip from run to run when two launches take the same time -- cfg5)
        tmax = max(t[1] for t in timings)
        dom = max((i for i in range(len(timings)) if timings[i][1] >= 0.9 * tmax),
                  key=lambda i: timings[i][3] + timings[i][4])
        label, ms_sum, launches, s_in, s_out = timings[dom]
        # `kernel`: the device symbol as rocprofv3 prints it (profiles/*_kernel_stats.csv, profiles/traffic.json);
        # `label`: the engine's name for the stage's form (k_convp_whole = convolver + interpolator in one launch)
        name = symbols[dom] or label
        avg_ms = ms_sum / max(launches, 1)
        plan_out = n_out / args.steps  # average final outputs per channel per step
        alg_bytes = 8.0 * C * (s_in + s_out) / max(launches, 1)
        # (the emulation library's events read 0: no kernel time, no fraction)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        path_bytes = 8.0 * (C * L + C * plan_out)
        res = {
            "metric": "Msamples/sec 44.1k\u219296k, N-channel batch, 1/2/4/8 GPU; RMS err vs ref",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: CDSPResampler24 %g->%g, %d channels/GPU x %d-sample blocks, "
                                   "fp64 splitmix64 noise (seed 1 + channel), inputs and outputs "
                                   "resident in HBM%s" % (cfg_name, args.src, args.dst, C, L,
                                                         ", output rows 64-byte aligned with a call's outputs at column "
                                                         "(outputs so far) mod 8" if args.align_out and not args.pcm else
                                                         ", a call's outputs from column 0 of 64-byte-aligned rows"),
                       "placement": "stream-aligned" if args.align_out else "column 0",
                       "channels_per_gpu": C, "block": L, "io": ((args.pcm + (" planar" if args.planar else " interleaved"))
                                                   if args.pcm else "f64 planar"),
                       "out_msamples_per_s":
                           round(n_out * C * world / dt / 1e6, 3),
                       "chain": rs.describe().strip().split("\n")},
            "roofline": {"bound": "hbm", "kernel": name, "label": label, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(name, cfg_name) or measured_traffic(label, cfg_name),
                         "alg_bytes_per_launch": alg_bytes, "avg_kernel_ms": round(avg_ms, 4),
                         "launches": launches,
                         "kernels_ms_per_step": kernel_ms(timings, args.steps),
                         "kernel_symbols": [sym for sym in symbols if sym],
                         "path_frac": round(path_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                                            4)},
        }
        # `frac` is the dominant kernel as the event pass saw it -- AFTER every timed window, i.e. at the settled clock;
        # `path_frac` is the whole call in the window `value` is quoted on (algorithmic bytes of the call / ms_per_step).
        # (Round 5's derived `frac_value_window` is gone: path_frac is the measured figure of that window.)
        ksum = sum(t[1] for t in timings) / args.steps
        # (what a step spends between its kernels, where the events' own overhead does not drown it: informational)
        ref_ms = settled["ms_per_step"] if settled is not None else ev_wall_ms
        res["roofline"]["launch_gap_ms_per_step"] = round(ref_ms - ksum, 4) if ref_ms >= ksum else None
        res["roofline"]["frac_clock_state"] = ("`frac`: per-kernel HIP events of a pass AFTER every timed window (settled "
                                               "clock); `path_frac`: the whole call, measured in the window `value` is "
                                               "quoted on")
        res["value_window"] = "the %d steps timed straight after the %d warm-up calls" % (args.steps, args.warmup)
        if settled is not None:
            res["settle_calls"] = args.settle
            res["settled"] = settled
        if other is not None:
            res["other_placement"] = other
        if e2e is not None:
            # (kernel-only = the line's own `value`: shards at rest; end-to-end beside it)
            res["e2e"] = e2e
        if host_side:
            res["data"] = "synthetic; HOST EMULATION of the kernels (--backend gloo --lib): control flow only, not a measurement"
        if not args.no_cpu and world == 1 and args.phase == 0 and not host_side:
            # (N = 1 only: the CPU leg is a per-box baseline, not part of the scaling runs)
            # error report: rows {0, C/2, C-1} of the timed batch's own output (what the object produced for its
            # first calls, captured during warm-up) against the reference on the same samples
            sample = None
            if captured:
                ycap = torch.cat(captured, dim=1).cpu().numpy()
                nb = len(captured)
                sample = (chk_rows, [host_x[c, :nb * L] for c in chk_rows], [ycap[j] for j in range(len(chk_rows))])
            res["cpu_baseline"] = cpu_baseline(args.src, args.dst, L, sample)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
