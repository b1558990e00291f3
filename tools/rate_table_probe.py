"""tools/rate_table_probe.py: every ordered pair of the common audio rates at the 24-bit preset, cfg2-sized
batch (1024 channels x 16384 samples): ms per call, kernels, fraction of the HBM roofline of the whole path.
Finds topologies left on slow paths."""
import importlib, os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
rates = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]
dev = torch.device("cuda", 0)
C, L = 1024, 16384
x = [torch.rand((C, L), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3)]
rows = []
for s in rates:
    for d in rates:
        if s == d: continue
        try:
            rs = r8b.BatchResampler(float(s), float(d), L, 2.0, 180.15, nch=C, device=0)
        except Exception as e:
            print(s, d, "ERR", e); continue
        outs = [torch.empty((C, rs.max_out_len), dtype=torch.float64, device=dev) for _ in range(2)]
        for i in range(6): rs.process(x[i % 3], out=outs[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        for i in range(30): n += rs.process(x[i % 3], out=outs[i % 2]).shape[1]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        rs.set_option("timing", 1)
        for i in range(6): rs.process(x[i % 3], out=outs[i % 2])
        torch.cuda.synchronize()
        tm = {}
        for name, ms, launches, _, _ in rs.stage_timings():
            tm[name] = round(tm.get(name, 0) + ms / 6, 3)
        bytes_ = 8.0 * C * (L + n / 30)
        rows.append((s, d, round(dt * 1e3, 3), round(bytes_ / dt / 8e12, 3), tm))
        print(s, d, round(dt * 1e3, 3), "ms", "path_frac", round(bytes_ / dt / 8e12, 3), tm, flush=True)
        del rs, outs
