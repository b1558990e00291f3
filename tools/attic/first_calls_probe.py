#!/usr/bin/env python3
"""tools/first_calls_probe.py [n]: per-call GPU time of the first n calls of a fresh cfg2 object (events on the launching
stream around every call), the way bench.py's warm-up + value window sees them; run twice in one process to tell what the
process's first use costs from what every object's first calls cost."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nch, L = 1024, 16384
x = [torch.rand((nch, L), dtype=torch.float64, device="cuda:0") * 2 - 1 for _ in range(3)]
for rep in range(2):
    rs = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=nch, device=0)
    out = torch.empty((nch, rs.max_out_len), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    if rep == 1:
        time.sleep(2.0)  # (an idle GPU in front of the second object)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        rs.process(x[i % 3], out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print("object %d: wall %.2f ms for %d calls; per call (ms):" % (rep, wall * 1e3, n))
    print("  " + " ".join("%.3f" % v for v in ms))
    print("  calls 5..24 mean %.4f   calls 40..59 mean %.4f" % (np.mean(ms[5:25]), np.mean(ms[40:60]) if n >= 60 else float("nan")))
