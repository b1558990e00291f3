"""Where the fixed cost of a short timed region goes (20 calls of the cfg2 batch): wall clock vs stream
events, CPU time per process() call, sync latency."""
import importlib, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
dev = torch.device("cuda", 0)
C, L = 1024, 16384
rs = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=C, device=0)
x = [torch.rand((C, L), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3)]
outs = [torch.empty((C, rs.max_out_len), dtype=torch.float64, device=dev) for _ in range(2)]
res = []
for rep in range(6):
    time.sleep(0.5)
    for i in range(5):
        rs.process(x[i % 3], out=outs[i % 2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    tl = []
    for i in range(20):
        a = time.perf_counter()
        rs.process(x[i % 3], out=outs[i % 2])
        tl.append(time.perf_counter() - a)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.append({"wall_ms": round((t2 - t0) * 1e3, 3), "events_ms": round(e0.elapsed_time(e1), 3),
                "issue_ms": round((t1 - t0) * 1e3, 3), "first_call_us": round(tl[0] * 1e6, 1),
                "call_us_median": round(float(np.median(tl)) * 1e6, 1), "call_us_max": round(max(tl) * 1e6, 1)})
# sync latency on an idle device
t = []
for i in range(20):
    a = time.perf_counter(); torch.cuda.synchronize(); t.append(time.perf_counter() - a)
print(json.dumps({"runs": res, "idle_sync_us": round(float(np.median(t)) * 1e6, 1)}, indent=1))
