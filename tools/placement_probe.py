#!/usr/bin/env python3
"""tools/placement_probe.py [calls] : what the column of a call's first output costs.  cfg2 object, every call's outputs
from column c of 64-byte-pitched rows, GPU time of each call (events), for each delta = (column - outputs so far) mod 8
-- the misalignment of the kernel's 64-byte store pieces (four adjacent phase pairs of a lane quad) against the rows'
64-byte lines.  Column 0 for every call (the reference's caller) walks delta through all eight values."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 480
src, dst, nch, L = 44100.0, 96000.0, 1024, 16384
rs = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=nch, device=0)
for kv in os.environ.get("R8B_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        rs.set_option(k, int(v))
x = torch.rand((nch, L), dtype=torch.float64, device="cuda:0") * 2 - 1
pitch = (rs.max_out_len + 7) // 8 * 8 + 16
out = torch.empty((nch, pitch), dtype=torch.float64, device="cuda:0")
for i in range(200):
    rs.process(x, out=out[:, :rs.max_out_len])
torch.cuda.synchronize()
produced = 0
res = []
for delta in range(8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        if rep == 1:
            e0.record()
        for i in range(ncalls):
            c = (produced + delta) % 8
            y = rs.process(x, out=out[:, c:c + rs.max_out_len])
            produced += y.shape[1]
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / ncalls)
print("ms per call, every call's first output at column (outputs so far + delta) mod 8 of 64-byte-pitched rows:")
print("  " + "  ".join("delta %d: %.4f" % (k, res[k]) for k in range(8)))
print("  column 0 for every call walks delta through all eight: mean %.4f; aligned (delta 0) %.4f" % (sum(res) / 8, res[0]))
