#!/usr/bin/env python3
"""tools/timeline_probe.py [src dst] : where and when every workgroup of ONE launch of the pair kernel ran (development
build with -DR8B_TIMELINE, tools/variant.sh; R8B_HIP_LIB points at it): lifetimes, the gap between a workgroup's end and
the start of the next one in the same slot of its CU, how many workgroups a CU holds over time."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
src, dst, nch, L = 44100.0, 96000.0, 1024, 16384
if len(sys.argv) > 2:
    src, dst = float(sys.argv[1]), float(sys.argv[2])
rs = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=nch, device=0)
for o in os.environ.get("R8B_OPTS", "").split():  # (engine options, "name=value ...")
    rs.set_option(o.split("=")[0], int(o.split("=")[1]))
x = torch.rand((nch, L), dtype=torch.float64, device="cuda:0") * 2 - 1
out = torch.empty((nch, rs.max_out_len + 8), dtype=torch.float64, device="cuda:0")
for i in range(200):
    rs.process(x, out=out[:, :rs.max_out_len])
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["R8B_HIP_LIB"])
N = 16384
buf = (ctypes.c_longlong * (4 * N))()
lib.r8b_dev_timeline(buf, N)
a = np.array(buf[:], dtype=np.int64).reshape(N, 4)
a = a[a[:, 1] > 0]
a = a[a[:, 3] == a[:, 3].max()]  # (the last launch: the tag is its first block index)
# (the cycle counters of the XCDs are not synchronised: every XCD is put on its own time base, first start = 0)
xcc0 = a[:, 2] >> 32
for x in np.unique(xcc0):
    m = xcc0 == x
    base = a[m, 0].min()
    a[m, 0] -= base
    a[m, 1] -= base
t0, t1 = a[:, 0], a[:, 1]
hw = a[:, 2] & 0xffffffff
xcc = a[:, 2] >> 32
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("workgroups", len(a), "distinct CUs", len(np.unique(key)), "launch span (cycles)", int(t1.max() - t0.min()))
life = t1 - t0
print("lifetime: mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % (life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
gaps, busy = [], []
kstart, kend = 0, t1.max()
for k in np.unique(key):
    m = key == k
    s, e = np.sort(t0[m]), np.sort(t1[m])
    # two slots per CU: the i-th start (i >= 2) follows the (i-2)-th end
    for i in range(2, len(s)):
        gaps.append(s[i] - e[i - 2])
    busy.append(life[m].sum() / (2.0 * (kend - kstart)))
gaps = np.array(gaps)
print("gap from a workgroup's end to the next start in that CU slot: mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f  (n=%d)"
      % (gaps.mean(), *np.percentile(gaps, [10, 50, 90]), gaps.max(), len(gaps)))
print("slot occupancy per CU (lifetimes / 2 slots / span): mean %.3f  min %.3f  max %.3f" % (np.mean(busy), np.min(busy), np.max(busy)))
for x in np.unique(xcc):
    m = xcc == x
    print("XCD %d: %d workgroups, span %d cycles, last start %d, mean lifetime %.0f, slots busy %.3f" %
          (x, m.sum(), t1[m].max(), t0[m].max(), life[m].mean(), life[m].sum() / (64.0 * t1[m].max())))
slow = life > 1.5 * np.median(life)
print("workgroups slower than 1.5 x the median: %d (%.1f %%), of which first blocks of their pair (blockIdx %% nbg == 0 is not known here): lifetimes" % (slow.sum(), 100.0 * slow.mean()), np.sort(life[slow])[-8:])
cnt = np.array([np.sum(key == k) for k in np.unique(key)])
print("workgroups per CU: min %d max %d" % (cnt.min(), cnt.max()))
first = np.sort(t0)[:512] - kstart
print("start of the first 512 workgroups after the launch's first: p50 %.0f p90 %.0f max %.0f" % (*np.percentile(first, [50, 90]), first.max()))
last_start = np.sort(t0)[-1] - kstart
print("last workgroup starts at %.0f, launch ends at %.0f" % (last_start, kend - kstart))
# lifetime by the block's place in its channel pair's run (k_convp's mapping: workgroup w -> block group (w >> 3) % nbg
# when the pair count is a multiple of 8); the tail of the history goes with the last ones
idx = np.nonzero(np.array(buf[:], dtype=np.int64).reshape(N, 4)[:, 1] > 0)[0]
idx = idx[np.array(buf[:], dtype=np.int64).reshape(N, 4)[idx, 3] == a[:, 3].max()]
npair = (nch + 1) // 2
if len(idx) == len(a) and len(a) % npair == 0 and npair % 8 == 0:
    nbg = len(a) // npair
    bg = (idx >> 3) % nbg
    print("blocks per pair", nbg, "- lifetime by block (mean / p50 / p90):")
    for b in range(nbg):
        m = bg == b
        print("  block %2d: %6.0f %6.0f %6.0f" % (b, life[m].mean(), *np.percentile(life[m], [50, 90])))
