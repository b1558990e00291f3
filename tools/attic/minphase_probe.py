import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
dev = torch.device("cuda", 0)
C, L = 1024, 16384
x = [torch.rand((C, L), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3)]
for s, d in [(44100., 96000.), (96000., 44100.), (44100., 88200.), (88200., 44100.)]:
    for ph in (0, 1):
        rs = r8b.BatchResampler(s, d, L, 2.0, 180.15, nch=C, device=0, phase=ph)
        outs = [torch.empty((C, rs.max_out_len), dtype=torch.float64, device=dev) for _ in range(2)]
        for i in range(6): rs.process(x[i % 3], out=outs[i % 2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(30): rs.process(x[i % 3], out=outs[i % 2])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        rs.set_option("timing", 1)
        for i in range(6): rs.process(x[i % 3], out=outs[i % 2])
        torch.cuda.synchronize()
        tm = {}
        for name, ms, launches, _, _ in rs.stage_timings(): tm[name] = round(tm.get(name, 0) + ms / 6, 3)
        print(s, d, "minphase" if ph else "linear", round(dt * 1e3, 3), tm, flush=True)
