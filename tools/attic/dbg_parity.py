#!/usr/bin/env python3
"""tools/dbg_parity.py <src> <dst> <nch> <L> <calls> [opt=v ...] : HIP path (R8B_DBG_TB = transition band, default 2; (R8B_HIP_LIB honoured) vs the oracle on a few channels"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
r8b = importlib.import_module("r8brain-free-src_amd")
import r8b_oracle as O
src, dst, nch, L, calls = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
TB = float(os.environ.get("R8B_DBG_TB", "2.0"))
rs = r8b.BatchResampler(src, dst, L, TB, 180.15, nch=nch, device=0)
for o in sys.argv[6:]:
    k, v = o.split("="); rs.set_option(k, int(v))
chk = sorted(set([0, 1, nch // 2, nch - 1]))
orc = {c: O.OracleResampler(src, dst, L, TB, 180.15) for c in chk}
x = np.stack([O.splitmix_uniform(1 + c, L * calls) for c in range(nch)])
worst = 0.0
for i in range(calls):
    y = rs.process(torch.from_numpy(np.ascontiguousarray(x[:, i * L:(i + 1) * L])).cuda()).cpu().numpy()
    for c in chk:
        yo = orc[c].process(x[c, i * L:(i + 1) * L])
        assert len(yo) == y.shape[1]
        if len(yo):
            d = np.abs(y[c] - yo)
            worst = max(worst, d.max())
            if d.max() > 1e-12:
                bad = np.nonzero(d > 1e-12)[0]
                print("call %d ch %d: %d bad of %d, first %d last %d, max %.3g" % (i, c, len(bad), len(yo), bad[0], bad[-1], d.max()))
print("worst peak", worst)
