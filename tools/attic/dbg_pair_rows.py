"""identical data in every channel: which outputs of the even rows of a 5-channel batch differ from a 1-channel object?"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
r8b = importlib.import_module("r8brain-free-src_amd")
L = 2000
b5 = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=5, device=0)
b1 = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=1, device=0)
rng = np.random.default_rng(3)
stride = L + 24
for c in range(4):
    x = rng.uniform(-1.0, 1.0, L)
    X = torch.zeros((5, stride), dtype=torch.float64, device="cuda:0")
    X[:, :L] = torch.from_numpy(np.tile(x, (5, 1))).cuda()
    y5 = b5.process(X[:, :L]).cpu().numpy()
    y1 = b1.process(torch.from_numpy(x[None, :].copy()).cuda()).cpu().numpy()
    for ch in range(5):
        idx = np.nonzero(y5[ch] != y1[0])[0]
        print(c, ch, y5.shape, len(idx), idx[:8], idx[-3:] if len(idx) else None,
              float(np.abs(y5[ch] - y1[0]).max()))
