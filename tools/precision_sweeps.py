#!/usr/bin/env python3
"""tools/precision_sweeps.py -- the reference's round-trip precision sweeps at their REAL extent, on the HIP path and on
the compiled reference side by side (VERDICT r5 missing #4 / next #8).

  zerotest  reference bench/zerotest.cpp:98-141   20 -> k -> 20 for k = 21 ... 640 (620 ratios), ReqAtten 180.15,
            tb = 0.5 + 4.5 U, MaxInLen = 50 + 1500 U, 50000 samples, edges of 5000 skipped
  snrtest   reference bench/snrtest.cpp:69-99     ReqAtten = 49, 55, ... 217 (29 values) x k = 21, 28, ... 595 (83
            ratios), tb = 0.5 + 4 U, MaxInLen = 50 + 1500 U, 100000 samples, edges of 20000 skipped
  masstest  reference bench/masstest.cpp:111-172  1000 random ratios 1 -> 1 + 44 U -> 1, ReqAtten 180.15,
            tb = 0.5 + 4.5 U, MaxInLen = 50 + 2000 U, 50000 samples, edges of 5000 skipped

What cannot be the reference's: its input WAV file and its random generator (CRnd) come from the author's libvox,
which is not part of the reference sources -- the signal is splitmix64 noise, band-limited
exactly as the reference does it (10 -> bw -> 10 with CDSPResampler24, bw = 9.3 / 9.0), and U comes from splitmix64 too
(seeds in the CSV header).  Every round trip runs through oneshot() (CDSPResampler.h:592-651) twice: on the library
under test (--backend hip: libr8bsrc_hip.so, samples resident on the GPU; --backend emul: the host emulation of the CPU
tier) and on oracle/_ref (the reference compiled from /root/reference), the latter on a thread pool.

Output: one CSV row per round trip -- parameters, RMS and peak of (reference signal - round trip) in dB for both -- and a
summary block like the one the reference prints (average rms, peak diff).  The committed files are
profiles/r06_zerotest.csv, r06_snrtest.csv, r06_masstest.csv; tests/test_roundtrip.py asserts their worst difference.
"""
import argparse
import concurrent.futures as cf
import importlib
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import r8b_oracle as O  # noqa: E402  (splitmix64 only)
import refwrap as R     # noqa: E402  (the compiled reference: the thing compared against)


def db(v):
    return -400.0 if v <= 0.0 else 20.0 * math.log10(v)


def uniform01(seed, n):
    return (O.splitmix_uniform(seed, n) + 1.0) * 0.5


class RefOneshot:
    """oneshot() (CDSPResampler.h:592-651) over the compiled reference's process()"""

    def __init__(self, src, dst, maxin, tb, att):
        self.r = R.RefResampler(src, dst, maxin, tb, att)
        self.maxin = maxin

    def oneshot(self, ip, oplen):
        out, got, pos = [], 0, 0
        zeros = np.zeros(self.maxin)
        while got < oplen:
            if pos < len(ip):
                blk = ip[pos:pos + self.maxin]
                pos += len(blk)
            else:
                blk = zeros
            y = self.r.process(blk)[:oplen - got]
            out.append(y)
            got += len(y)
        return np.concatenate(out)


class HipOneshot:
    """the same over r8b_batch_process with the samples resident on the GPU: a call only enqueues, the per-call
    count comes from the host plan, so a round trip is two streams of launches and one copy back"""

    def __init__(self, r8b, torch, src, dst, maxin, tb, att):
        self.rs = r8b.BatchResampler(src, dst, maxin, tb, att, nch=1)
        self.maxin = maxin
        self.torch = torch

    def oneshot(self, ip, oplen):
        t = self.torch
        x = ip if t.is_tensor(ip) else t.from_numpy(np.ascontiguousarray(ip)).to("cuda").reshape(1, -1)
        n = x.shape[1]
        cap = self.rs.max_out_len
        out = t.empty((1, oplen + cap + 8), dtype=t.float64, device="cuda")
        zeros = t.zeros((1, self.maxin), dtype=t.float64, device="cuda")
        got = pos = 0
        while got < oplen:
            if pos < n:
                blk = x[:, pos:pos + self.maxin]
                pos += blk.shape[1]
            else:
                blk = zeros
            got += self.rs.process(blk, out=out[:, got:got + cap]).shape[1]
        return out[:, :oplen]

    @staticmethod
    def to_host(y):
        return y[0].cpu().numpy()


class EmulOneshot:
    def __init__(self, r8b, lib, src, dst, maxin, tb, att):
        self.rs = r8b.CDSPResampler(src, dst, maxin, tb, att, lib=lib)

    def oneshot(self, ip, oplen):
        return self.rs.oneshot(ip, oplen)

    @staticmethod
    def to_host(y):
        return y


def err_db(ref, back, skip):
    d = ref[skip:len(ref) - skip] - back[skip:len(ref) - skip]
    return db(float(np.sqrt(np.mean(d * d)))), db(float(np.abs(d).max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("test", choices=["zerotest", "snrtest", "masstest"])
    ap.add_argument("--backend", choices=["hip", "emul"], default="hip")
    ap.add_argument("--out", required=True)
    ap.add_argument("--limit", type=int, default=0, help="first N round trips only (smoke runs)")
    ap.add_argument("--threads", type=int, default=max(1, min(16, (os.cpu_count() or 2) - 1)))
    args = ap.parse_args()
    assert R.available(), "oracle/_ref not built"
    r8b = importlib.import_module("r8brain-free-src_amd")
    if args.backend == "hip":
        import torch
        assert torch.cuda.is_available()
        make = lambda s, d, m, tb, att: HipOneshot(r8b, torch, s, d, m, tb, att)  # noqa: E731
        to_host = HipOneshot.to_host
    else:
        lib = r8b.bind(os.path.join(ROOT, "tests", "emul", "_build", "libr8bsrc_emul.so"))
        make = lambda s, d, m, tb, att: EmulOneshot(r8b, lib, s, d, m, tb, att)  # noqa: E731
        to_host = EmulOneshot.to_host

    if args.test == "snrtest":
        n_in, skip, bw, sig_seed, rnd_seed = 100000, 20000, 9.0, 1000000, 1000001
    else:
        n_in, skip, bw, sig_seed, rnd_seed = 50000, 5000, 9.3, 4242, (7 if args.test == "zerotest" else 11)
    x = O.splitmix_uniform(sig_seed, n_in)
    # reference signal (zerotest.cpp:63-82, snrtest.cpp:46-64): made ONCE, with the compiled reference, for both sides
    n0 = int(n_in * bw / 10.0)
    ref = RefOneshot(bw, 10.0, 521, 2.0, 180.15).oneshot(RefOneshot(10.0, bw, 521, 2.0, 180.15).oneshot(x, n0), n_in)

    # the parameter draws, in the order the reference draws them
    cases = []
    if args.test == "zerotest":
        u = uniform01(rnd_seed, 4 * 620)
        for i, k in enumerate(range(21, 641)):
            tb = 0.5 + u[4 * i] * 4.5
            maxin = int(50 + u[4 * i + 1] * 1500)
            cases.append((20.0, float(k), maxin, tb, 180.15))
    elif args.test == "snrtest":
        ks = list(range(21, 600, 7))
        atts = list(range(49, 219, 6))
        u = uniform01(rnd_seed, 2 * len(ks) * len(atts))
        i = 0
        for att in atts:
            for k in ks:
                maxin = int(50 + u[i] * 1500)
                tb = 0.5 + u[i + 1] * 4.0
                i += 2
                cases.append((20.0, float(k), maxin, tb, float(att)))
    else:
        u = uniform01(rnd_seed, 5 * 1000)
        for i in range(1000):
            dst = 1.0 + 44.0 * u[5 * i]
            tb = 0.5 + 4.5 * u[5 * i + 1]
            maxin = 50 + int(2000 * u[5 * i + 2])
            cases.append((1.0, dst, maxin, tb, 180.15))
    if args.limit:
        cases = cases[:args.limit]

    def ref_trip(c):
        src, dst, maxin, tb, att = c
        ol = int(n_in * dst / src)
        up = RefOneshot(src, dst, maxin, tb, att).oneshot(ref, ol)
        back = RefOneshot(dst, src, maxin, tb, att).oneshot(up, n_in)
        return err_db(ref, back, skip)

    def lib_trip(c):
        src, dst, maxin, tb, att = c
        ol = int(n_in * dst / src)
        up = make(src, dst, maxin, tb, att).oneshot(ref, ol)
        back = make(dst, src, maxin, tb, att).oneshot(up, n_in)
        return err_db(ref, to_host(back), skip)

    t0 = time.time()
    rows = []
    with cf.ThreadPoolExecutor(args.threads) as pool:
        futs = [pool.submit(ref_trip, c) for c in cases]
        for i, c in enumerate(cases):
            lr, lp = lib_trip(c)
            rows.append([c, lr, lp])
            if (i + 1) % 50 == 0:
                sys.stderr.write("%s: %d / %d round trips, %.0f s\n" % (args.test, i + 1, len(cases), time.time() - t0))
        for i, f in enumerate(futs):
            rows[i] += list(f.result())
    with open(args.out, "w") as f:
        f.write("# %s: %s vs the compiled reference (oracle/_ref), %d round trips, %d-sample splitmix64 noise (seed %d) "
                "band-limited to %.1f/10, parameter draws from splitmix64 seed %d; tools/precision_sweeps.py\n"
                % (args.test, "libr8bsrc_hip.so on MI355X" if args.backend == "hip" else "host emulation", len(rows),
                   n_in, sig_seed, bw, rnd_seed))
        f.write("src,dst,maxin,tb,atten,lib_rms_db,ref_rms_db,lib_peak_db,ref_peak_db\n")
        for c, lr, lp, rr, rp in rows:
            f.write("%.10g,%.10g,%d,%.6f,%.2f,%.3f,%.3f,%.3f,%.3f\n" % (c[0], c[1], c[2], c[3], c[4], lr, rr, lp, rp))
        # the reference's own summary lines (zerotest.cpp:158-160, snrtest.cpp:95-99, masstest.cpp:174-176)
        def avg_db(vals):
            return 10.0 * math.log10(sum(10.0 ** (v / 10.0) for v in vals) / len(vals))
        lib_r, ref_r = [r[1] for r in rows], [r[3] for r in rows]
        f.write("# summary: average rms lib %.2f dB, reference %.2f dB; max rms lib %.2f, reference %.2f; peak diff lib "
                "%.2f, reference %.2f; worst |lib - reference| rms %.3f dB; %.0f s\n"
                % (avg_db(lib_r), avg_db(ref_r), max(lib_r), max(ref_r), max(r[2] for r in rows), max(r[4] for r in rows),
                   max(abs(a - b) for a, b in zip(lib_r, ref_r)), time.time() - t0))
        if args.test == "snrtest":
            for att in sorted(set(r[0][4] for r in rows)):
                sel = [r for r in rows if r[0][4] == att]
                f.write("# ReqAtten=%.2f avg lib %.2f reference %.2f max lib %.2f reference %.2f\n"
                        % (att, avg_db([r[1] for r in sel]), avg_db([r[3] for r in sel]), max(r[1] for r in sel),
                           max(r[3] for r in sel)))
    print(open(args.out).read().split("# summary")[1].split("\n")[0])


if __name__ == "__main__":
    main()
