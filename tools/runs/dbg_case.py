import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import pytest, numpy as np
import test_fuzz as T
import refwrap as R
case = (159208.0, 131316.0, 2951, 3.71, 66.79, 217628576)
print(T.r8b.BatchResampler(case[0], case[1], case[2], case[3], case[4], nch=1).describe())
for opts in ([], [("quad", 1)], [("fuse_hbconv", 1)], [("pair_conv", 0)]):
    _orig = T.r8b.BatchResampler
    def _with(*a, **kw):
        b = _orig(*a, **kw)
        for k, v in opts: b.set_option(k, v)
        return b
    T.r8b.BatchResampler = _with
    try:
        T.test_fuzz_gpu_vs_reference(R, case); print(opts, "ok")
    except AssertionError as e:
        print(opts, "FAIL", str(e)[:200])
    T.r8b.BatchResampler = _orig
