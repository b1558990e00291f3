import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import importlib
r8b = importlib.import_module("r8brain-free-src_amd")
from cases import make_input
import os
KW = {"lib": r8b.bind(os.environ["R8B_LIB"], test_hooks=True)} if "R8B_LIB" in os.environ else {"device": 0}
for (src, dst, maxin) in ((44100.0, 88200.0, 6000), (44100.0, 2822400.0, 1024), (44100.0, 44101.0, 3000)):
    nch = 37
    x = make_input(nch, 5 * maxin, 43)
    outs = []
    for h in (0, 1, 1):
        b = r8b.BatchResampler(src, dst, maxin, 2.0, 180.15, nch=nch, **KW)
        b.set_option("half", h)
        ys, pos = [], 0
        for l in [maxin, maxin, maxin // 5, 17, 1, maxin - 1, maxin // 2]:
            ys.append(b.process_host(x[:, pos:pos + l])); pos += l
        outs.append(np.concatenate(ys, axis=1))
    d = outs[0] - outs[1]
    print(src, dst, "rms", np.sqrt((d*d).mean()), "peak", np.abs(d).max(), "nonzero share", (d != 0).mean(), "rerun equal", np.array_equal(outs[1], outs[2]), "level", np.sqrt((outs[0]**2).mean()))
