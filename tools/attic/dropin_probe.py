"""Single-channel drop-in path (r8b_create / r8b_process with host pointers, BASELINE configs[0]): samples per
second through the GPU per call of 16384 samples, against the compiled reference on one core."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
r8b = importlib.import_module("r8brain-free-src_amd")
x = np.random.default_rng(1).uniform(-1, 1, 16384 * 64)
for L in (16384, 2048, 256):
    rs = r8b.CDSPResampler24(44100.0, 96000.0, L)
    for i in range(0, 8 * L, L): rs.process(x[i:i + L])
    t0 = time.perf_counter(); n = 0
    reps = len(x) // L
    for i in range(reps): n += len(rs.process(x[i * L:(i + 1) * L]))
    dt = time.perf_counter() - t0
    print("GPU drop-in  MaxInLen %5d: %.1f M input samples/s, %.1f us per call" % (L, reps * L / dt / 1e6, dt / reps * 1e6))
try:
    import refwrap as R
    for L in (16384, 2048, 256):
        rr = R.RefResampler(44100.0, 96000.0, L)
        for i in range(0, 8 * L, L): rr.process(x[i:i + L])
        t0 = time.perf_counter(); reps = len(x) // L
        for i in range(reps): rr.process(x[i * L:(i + 1) * L])
        dt = time.perf_counter() - t0
        print("CPU reference MaxInLen %5d: %.1f M input samples/s, %.1f us per call" % (L, reps * L / dt / 1e6, dt / reps * 1e6))
except Exception as e:
    print("reference not available:", e)
