"""tools/gpu_fuzz.py [n] [seed] [wide]: one-off GPU fuzz (the test suite's fuzz function on fresh random
cases); `wide` draws the filter from the reference's whole range (transition band 0.5 ... 45 %, attenuation
49 ... 218 dB) so that every block geometry of the pair kernel is hit.  Test infrastructure."""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import pytest, numpy as np
import test_fuzz as T
import refwrap as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 180
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 101
wide = len(sys.argv) > 3
rng = np.random.default_rng(seed)
# (R8B_FUZZ_TB=lo,hi narrows the transition band range, as in tools/wide_fuzz.py)
TB = [float(v) for v in os.environ.get("R8B_FUZZ_TB", "0.5,45").split(",")]
# (R8B_FUZZ_OPTS="walk=2 solo_fuse=0": engine options set on every object the fuzz creates -- e.g. the walk form forced
# on the fuzz's three-channel batches, which the engine would not walk by itself)
OPTS = [o.split("=") for o in os.environ.get("R8B_FUZZ_OPTS", "").split()]
if OPTS:
    _orig = T.r8b.BatchResampler
    def _with_opts(*a, **kw):
        b = _orig(*a, **kw)
        for k, v in OPTS:
            b.set_option(k, int(v))
        return b
    T.r8b.BatchResampler = _with_opts
bad = 0; done = 0; skipped = 0; known = 0
for case in [c for c in T._cases(3 * n, seed) if c[2] >= 300][:n]:
    if wide:
        tb = float(np.round(np.exp(rng.uniform(np.log(TB[0]), np.log(TB[1]))), 2))
        att = float(np.round(rng.uniform(49.0, 218.0), 2))
        case = (case[0], case[1], case[2], tb, att, case[5])
    done += 1
    try:
        T.test_fuzz_gpu_vs_reference(R, case)
    except pytest.skip.Exception:
        skipped += 1
    except AssertionError as e:
        # (the one known difference, as in tools/wide_fuzz.py: a one-tap half-band up-sampler's first odd output, where
        # the reference's own value is indeterminate.  Until round 6 also: truncated 32768-point reference blocks within
        # 1e-10 -- those chains now run the reference's own block and have to meet the bound)
        desc = T.r8b.BatchResampler(case[0], case[1], case[2], case[3], case[4], nch=1).describe()
        a = e.args[0] if e.args and isinstance(e.args[0], tuple) and len(e.args[0]) in (3, 4) else None
        if "taps=1 " in desc:
            known += 1
        else:
            bad += 1; print("FAIL", case, str(e)[:300])
    except Exception as e:
        bad += 1; print("ERR", case, repr(e)[:300])
print("gpu fuzz done", done, "bad", bad, "skipped", skipped, "known differences", known)
