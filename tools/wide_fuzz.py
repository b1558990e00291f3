"""tools/wide_fuzz.py [n] [seed]: one-off CPU fuzz of the emulated engine against the compiled reference over
the reference's WHOLE filter range (transition band 0.5 ... 45 %, attenuation 49 ... 218 dB): every block
geometry of the pair kernel (64 ... 8192 points, decimating, radix-3 edges) gets hit.  Test infrastructure."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np, pytest
import test_fuzz as T
import refwrap as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
# (R8B_FUZZ_TB=lo,hi narrows the transition band range: 0.5,0.62 = the 8192 -> 16384-point blocks of the split form)
TB = [float(v) for v in os.environ.get("R8B_FUZZ_TB", "0.5,45").split(",")]
emul = T.r8b.bind(os.path.join(ROOT, "tests", "emul", "_build", "libr8bsrc_emul.so"))
bad = skipped = known = known32 = 0
geoms = {}
for i, c in enumerate(T._cases(n, seed)):
    src, dst, maxin, _, _, s = c
    tb = float(np.round(np.exp(rng.uniform(np.log(TB[0]), np.log(TB[1]))), 2))
    att = float(np.round(rng.uniform(49.0, 218.0), 2))
    case = (src, dst, maxin, tb, att, s)
    try:
        T.test_fuzz_emulated_engine_vs_reference(emul, R, case)
    except pytest.skip.Exception:
        skipped += 1
    except AssertionError as e:
        # (the one known difference: the reference's indeterminate start-of-stream sample behind a one-tap half-band
        # up-sampler -- DESIGN.md section 6, tests/test_emul.py test_one_tap_halfband_start_of_stream)
        desc = T.r8b.BatchResampler(src, dst, maxin, tb, att, nch=1, lib=emul).describe()
        a = e.args[0] if e.args and isinstance(e.args[0], tuple) and len(e.args[0]) == 3 else None
        if "taps=1 " in desc:
            known += 1
        else:
            bad += 1; print("FAIL", case, str(e)[:300], flush=True)
    except Exception as e:
        bad += 1; print("ERR", case, repr(e)[:300], flush=True)
print("wide fuzz done", n, "bad", bad, "skipped", skipped, "one-tap half-band chains (reference indeterminate)", known,
      "(32768-point reference blocks are no exception any more: round 6)", known32)
