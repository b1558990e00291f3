#!/usr/bin/env python3
"""tools/graph_probe.py [--config cfg5] [--calls N] -- what a hipGraph could buy a small call: the launches
of ONE process() call captured into a graph and replayed against the same call issued eagerly.  The replay repeats the
captured call's stream positions (its outputs are those of that one call again: a TIMING experiment, results are not
used), which is exactly what a per-call `hipGraphExecKernelNodeSetParams` update + `hipGraphLaunch` would put on the
GPU.  Prints eager / graph ms per call for 1 and for 8 calls per graph, and the one-channel form of the convolver
(engine option pair_conv = 0: twice the workgroups for the same blocks) beside the pair form."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg5")
    ap.add_argument("--calls", type=int, default=2000)
    args = ap.parse_args()
    import torch
    r8b = importlib.import_module("r8brain-free-src_amd")
    src, dst, L, C = {"cfg5": (44100.0, 2822400.0, 1024, 64), "cfg2": (44100.0, 96000.0, 16384, 1024),
                      "cfg3": (96000.0, 44100.0, 16384, 1024)}[args.config]
    dev = torch.device("cuda", 0)
    x = torch.rand((C, L), dtype=torch.float64, device=dev) * 2.0 - 1.0

    def eager(opts):
        rs = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=C, device=0)
        for k, v in opts.items():
            rs.set_option(k, v)
        out = torch.empty((C, (rs.max_out_len + 15) // 8 * 8), dtype=torch.float64, device=dev)
        for _ in range(200):
            rs.process(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.calls):
            rs.process(x, out=out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.calls * 1e3, rs, out

    def graph(rs, out, per):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            for _ in range(20):
                rs.process(x, out=out)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(per):
                    rs.process(x, out=out)
        torch.cuda.synchronize()
        reps = max(args.calls // per, 1)
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * per) * 1e3

    for name, opts in (("pair form", {}), ("one-channel convolver (pair_conv=0)", {"pair_conv": 0})):
        e, rs, out = eager(opts)
        g1 = graph(rs, out, 1)
        g8 = graph(rs, out, 8)
        print("%s %-36s eager %.4f ms/call   graph(1 call) %.4f   graph(8 calls) %.4f" % (args.config, name, e, g1, g8), flush=True)


if __name__ == "__main__":
    main()
