import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import pytest, numpy as np
import test_fuzz as T
import refwrap as R
bad = 0; n = 0; skipped = 0
for seed in (101, 102, 103):
    for case in [c for c in T._cases(150, seed) if c[2] >= 300][:60]:
        n += 1
        try:
            T.test_fuzz_gpu_vs_reference(R, case)
        except pytest.skip.Exception:
            skipped += 1
        except AssertionError as e:
            bad += 1; print("FAIL", case, str(e)[:300])
        except Exception as e:
            bad += 1; print("ERR", case, repr(e)[:300])
print("gpu fuzz done", n, "bad", bad, "skipped", skipped)
