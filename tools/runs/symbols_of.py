"""tools/runs/symbols_of.py: device symbols and chain of a few conversions at BASELINE's batch size (which kernel forms they run)"""
import sys, importlib
sys.path.insert(0, '.')
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
rates = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]
pairs = [(s, d) for s in rates for d in rates if s != d] if len(sys.argv) > 1 and sys.argv[1] == "all" else [
    (8000, 11025), (16000, 11025), (32000, 22050), (88200, 48000), (48000, 88200), (96000, 176400), (22050, 32000),
    (11025, 48000), (44100, 48000), (48000, 32000), (44100, 22050), (96000, 32000), (44100, 16000)]
x = torch.rand((1024, 16384), dtype=torch.float64, device="cuda:0") * 2 - 1
for s, d in pairs:
    b = r8b.BatchResampler(float(s), float(d), 16384, 2.0, 180.15, nch=1024, device=0)
    b.set_option("timing", 1)
    for _ in range(4):
        b.process(x)
    torch.cuda.synchronize()
    print(s, d, b.stage_symbols(), [(t[0], round(t[1] / max(t[2], 1), 4)) for t in b.stage_timings()], "|", b.describe().replace("\n", " | ")[:260])
