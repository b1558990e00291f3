"""Is the headline kernel held back by the board's power limit?  The same launches (identical instruction streams,
cfg2 batch) on inputs that toggle the multipliers differently: uniform noise, noise of tiny amplitude (the same
mantissa activity), a constant (the transforms' outputs are almost all exact zeros), all zeros (the silence path: every
product is 0 * x).  If the time per call follows the data, the clock / issue rate follows the power drawn."""
import importlib, os, sys, time, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
dev = torch.device("cuda", 0)
C, L = 1024, 16384

def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(o)
        c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if "ower" in k or "sclk" in k}
    except Exception as e:
        return {"err": str(e)[:80]}

def run(name, make):
    rs = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=C, device=0)
    x = [make() for _ in range(3)]
    out = [torch.empty((C, rs.max_out_len), dtype=torch.float64, device=dev) for _ in range(2)]
    for i in range(60):
        rs.process(x[i % 3], out=out[i % 2])
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(600):
            rs.process(x[i % 3], out=out[i % 2])
            if i == 400 and rep == 2:
                s = smi()
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / 600, 4))
    print(json.dumps({"input": name, "ms_per_call": res, "smi_mid_run": s}), flush=True)

run("uniform noise +-1", lambda: torch.rand((C, L), dtype=torch.float64, device=dev) * 2 - 1)
run("uniform noise +-1e-30", lambda: (torch.rand((C, L), dtype=torch.float64, device=dev) * 2 - 1) * 1e-30)
run("constant 1.0", lambda: torch.ones((C, L), dtype=torch.float64, device=dev))
run("zeros", lambda: torch.zeros((C, L), dtype=torch.float64, device=dev))
run("uniform noise +-1 (again)", lambda: torch.rand((C, L), dtype=torch.float64, device=dev) * 2 - 1)
