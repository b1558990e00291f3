"""tools/runs/fuzz_one.py: one fuzz case on the GPU under several option sets -- RMS / peak against the compiled reference"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import test_fuzz as T
import refwrap as R
O = T.O
r8b = T.r8b
case = tuple(float(v) if i < 2 or i in (3, 4) else int(v) for i, v in enumerate(sys.argv[1:7])) if len(sys.argv) > 6 else (67337.82745715551, 46197.03787634388, 2823, 0.88, 60.88, 242469838)
src, dst, maxin, tb, att, seed = case
for opts in ({"half": 0, "half_fused": 0}, {"half": 2, "half_fused": 2}, {"half": 0, "half_fused": 0, "pair_conv": 0}):
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=3)
    for k, v in opts.items():
        b.set_option(k, v)
    print(b.describe())
    refs = [R.RefResampler(src, dst, maxin, tb, att) for _ in range(3)]
    rng = np.random.default_rng(seed)
    total = int(min(400000, max(6000, b.getInputRequiredForOutput(300) + 4 * maxin)))
    x = np.stack([O.splitmix_uniform(seed % 1000 + c, total) for c in range(3)])
    pos, sq, cnt, pk = 0, 0.0, 0, 0.0
    while pos < total:
        l = int(min(total - pos, rng.integers(maxin // 2, maxin + 1)))
        y = b.process_host(x[:, pos:pos + l])
        for c in range(3):
            yr = refs[c].process(x[c, pos:pos + l])
            if len(yr):
                d = y[c] - yr
                sq += float(np.sum(d * d)); pk = max(pk, float(np.abs(d).max())); cnt += len(yr)
        pos += l
    print(opts, "rms", (sq / cnt) ** 0.5, "peak", pk, "n", cnt)
