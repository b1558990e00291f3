"""timing study: per-phase wave clocks of the pair kernel (variants/trace.so built with -DR8B_P_TRACE)"""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["R8B_HIP_LIB"] = os.path.join(ROOT, "variants", "trace.so")
r8b = importlib.import_module("r8brain-free-src_amd")
src, dst = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (44100.0, 96000.0)
C, L = 1024, 16384
rs = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=C, device=0)
for o in sys.argv[3:]:
    k, v = o.split("="); rs.set_option(k, int(v))
x = torch.rand((C, L), dtype=torch.float64, device="cuda") * 2 - 1
out = torch.empty((C, rs.max_out_len), dtype=torch.float64, device="cuda")
for i in range(6):
    rs.process(x, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["R8B_HIP_LIB"])
n = 4096 * 4 * 32
buf = (ctypes.c_ulonglong * n)()
assert lib.r8b_ptrace_dump(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4, 32).astype(np.int64)
nst = int((a[10, 0] > 0).sum())
print("stamps", nst)
a = a[:, :, :nst]
t0 = a[:, :, 0].min(axis=1, keepdims=True)[:, :, None]
rel = a - t0
print("stamps per wave:", nst)
# stamp 0 = kernel entry; then (arrive, leave) per barrier; last = exit
d = np.diff(a, axis=2)  # [wg, wave, nst-1]
names = []
for i in range(nst - 1):
    names.append(("work%d" % (i // 2)) if i % 2 == 0 else ("wait%d" % (i // 2)))
nwg = int((a[:, 0, 1] > 0).sum())
sel = slice(nwg // 8, nwg - nwg // 8)  # steady state workgroups
print('workgroups with stamps:', nwg)
print("%-8s %10s %10s %10s" % ("segment", "mean", "p50", "p90"))
for i, nm in enumerate(names):
    v = d[sel, :, i].reshape(-1)
    print("%-8s %10.0f %10.0f %10.0f" % (nm, v.mean(), np.percentile(v, 50), np.percentile(v, 90)))
tot = (a[sel, :, -1] - a[sel, :, 0]).reshape(-1)
print("total    %10.0f %10.0f %10.0f" % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90)))
