#!/usr/bin/env python3
"""tools/stamps_probe.py [bench-like args] : per-phase cycle stamps of eight workgroups of the pair kernel under the real load
(development build with -DR8B_CP_STAMPS=<first workgroup>, tools/variant.sh; R8B_HIP_LIB points at it)."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
src, dst, nch, L = 44100.0, 96000.0, 1024, 16384
if len(sys.argv) > 2:
    src, dst = float(sys.argv[1]), float(sys.argv[2])
# (BLOCK / TB / ATTEN in the environment: input samples per call, transition band, stop-band attenuation)
L = int(os.environ.get("BLOCK", L))
rs = r8b.BatchResampler(src, dst, L, float(os.environ.get("TB", "2.0")), float(os.environ.get("ATTEN", "180.15")), nch=nch, device=0)
x = torch.rand((nch, L), dtype=torch.float64, device="cuda:0") * 2 - 1
out = torch.empty((nch, rs.max_out_len), dtype=torch.float64, device="cuda:0")
for i in range(int(os.environ.get("NCALLS", "30"))):
    rs.process(x, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["R8B_HIP_LIB"])
buf = (ctypes.c_longlong * (8 * 8 * 32))()
lib.r8b_dev_stamps(buf)
a = np.array(buf[:], dtype=np.int64).reshape(8, 8, 32)
names = None
for wg in range(8):
    for w in range(8):
        n = int(a[wg, w, 0])
        if n <= 1:
            continue
        d = np.diff(a[wg, w, 1:1 + n])
        print("wg+%d wave %d:" % (wg, w), " ".join("%6d" % v for v in d), "| total", int(d.sum()))
