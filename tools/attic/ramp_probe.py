"""Per-call kernel time of the cfg2 batch right after an idle period (how the driver's 5 + 20 calls see the
GPU) and after a busy period: shows what the clock ramp costs the 20-step bench figure."""
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
torch.cuda.synchronize()
res = {}
for label, idle in (("idle_2s", 2.0), ("idle_0.2s", 0.2), ("idle_0.02s", 0.02), ("busy", 0.0)):
    time.sleep(idle)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(ev):
        a.record()
        rs.process(x[i % 3], out=outs[i % 2])
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in ev]
    res[label] = {"first10": [round(m, 3) for m in ms[:10]], "mean_5_25": round(float(np.mean(ms[5:25])), 4),
                  "mean_40_60": round(float(np.mean(ms[40:60])), 4), "wall_ms_60": round(wall * 1e3, 2)}
print(json.dumps(res, indent=1))
