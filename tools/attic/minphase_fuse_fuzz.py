"""tools/minphase_fuse_fuzz.py [n] [seed]: one-off CPU fuzz of the emulated engine -- minimum-phase chains with the convolver +
interpolator pair as ONE launch (Engine::fused_shift) against the two-launch form (option fuse_latency = 0) over random
ratios, filters and ragged calls: same counts per call, same samples to rounding.  Test infrastructure."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np
import test_fuzz as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
rng = np.random.default_rng(seed)
# (a third argument "gpu": the HIP library on device 0 instead of the emulation)
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
emul = None if GPU else T.r8b.bind(os.path.join(ROOT, "tests", "emul", "_build", "libr8bsrc_emul.so"))
KW = {"device": 0} if GPU else {"lib": emul}
bad = fused = skipped = 0
for case in T._cases(n, seed):
    src, dst, maxin, _, _, s = case
    tb = float(np.round(np.exp(rng.uniform(np.log(0.8), np.log(30.0))), 2))
    att = float(np.round(rng.uniform(60.0, 218.0), 2))
    try:
        objs = [T.r8b.BatchResampler(src, dst, maxin, tb, att, nch=3, phase=1, **KW) for _ in range(2)]
    except RuntimeError:
        skipped += 1
        continue
    objs[1].set_option("fuse_latency", 0)
    objs[0].set_option("timing", 1)
    objs[1].set_option("timing", 1)
    # (chains whose kernels differ between the two: a fused convolver + interpolator pair, a half-band cascade)
    if [t[0] for t in objs[0].stage_timings()] == [t[0] for t in objs[1].stage_timings()]:
        continue
    objs[0].set_option("timing", 0)
    objs[1].set_option("timing", 0)
    fused += 1
    total = int(min(120000, max(6000, objs[0].getInputRequiredForOutput(300) + 4 * maxin)))
    x = rng.uniform(-1.0, 1.0, (3, total))
    pos, worst, cnt = 0, 0.0, 0
    ok = True
    while pos < total:
        l = int(min(total - pos, rng.integers(1, maxin + 1)))
        try:
            ya, yb = objs[0].process_host(x[:, pos:pos + l]), objs[1].process_host(x[:, pos:pos + l])
        except RuntimeError as e:
            ok = False
            print("ERR", (src, dst, maxin, tb, att), pos, l, str(e)[-60:], objs[0].describe().replace("\n", " | "), flush=True)
            break
        if ya.shape != yb.shape:
            ok = False
            print("COUNT", (src, dst, maxin, tb, att), pos, l, ya.shape, yb.shape, flush=True)
            break
        if ya.shape[1]:
            worst = max(worst, float(np.abs(ya - yb).max()))
            cnt += ya.shape[1]
        pos += l
    if ok and cnt == 0:
        continue  # (the chain's latency is longer than the fuzz's stream: nothing to compare)
    if not ok or worst > 1e-13:
        bad += 1
        print("FAIL", (src, dst, maxin, tb, att), cnt, worst, objs[0].describe().replace("\n", " | "), flush=True)
print("minimum-phase fuse fuzz done", n, "fused chains", fused, "bad", bad, "skipped", skipped)
