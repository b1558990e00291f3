"""Round-trip precision tests: restatements of the reference's own bench/zerotest.cpp and
bench/snrtest.cpp (the precision criteria BASELINE.json's north_star
names).  A band-limited noise signal is resampled 20 -> k and back k -> 20 with `oneshot`, and the
difference to the original (edges skipped) is the figure of merit: about -180 dB RMS at
ReqAtten = 180.15, tracking ReqAtten in snrtest.

The same procedure runs through the real reference (oracle/_ref) and the results must agree to a
fraction of a dB; absolute bounds apply when the reference library is not present.  CPU tier: the
engine under the host emulation of tests/emul (test infrastructure); GPU tier: libr8bsrc_hip.so.
"""
import importlib
import math
import os
import subprocess

import numpy as np
import pytest

import r8b_oracle as O
from conftest import ROOT

r8b = importlib.import_module("r8brain-free-src_amd")

N_IN = 30000
SKIP = 5000


def _db(v):
    return -400.0 if v <= 0.0 else 20.0 * math.log10(v)


def _uniform01(seed, n):
    return (O.splitmix_uniform(seed, n) + 1.0) * 0.5


class _Ref:
    """oneshot() of the reference (CDSPResampler.h:592-651) on top of the compiled reference."""

    def __init__(self, R, src, dst, maxin, tb, att):
        self.r = R.RefResampler(src, dst, maxin, tb, att)
        self.maxin = maxin

    def oneshot(self, ip, oplen):
        out, got, pos = [], 0, 0
        while got < oplen:
            if pos < len(ip):
                blk = ip[pos:pos + self.maxin]
                pos += len(blk)
            else:
                blk = np.zeros(self.maxin)
            y = self.r.process(blk)[:oplen - got]
            out.append(y)
            got += len(y)
        self.r.clear()
        return np.concatenate(out)


def _band_limited(make, bw):
    """zerotest.cpp:63-82 / snrtest.cpp: reference signal = noise through 10 -> bw -> 10."""
    x = O.splitmix_uniform(4242, N_IN)
    n0 = int(N_IN * bw / 10.0)
    y0 = make(10.0, bw, 521, 2.0, 180.15).oneshot(x, n0)
    return make(bw, 10.0, 521, 2.0, 180.15).oneshot(y0, N_IN)


def _roundtrip(make, ref, k, maxin, tb, att):
    ol1 = int(N_IN * k / 20.0)
    up = make(20.0, float(k), maxin, tb, att).oneshot(ref, ol1)
    assert len(up) == ol1
    back = make(float(k), 20.0, maxin, tb, att).oneshot(up, N_IN)
    assert len(back) == N_IN
    d = ref[SKIP:N_IN - SKIP] - back[SKIP:N_IN - SKIP]
    return float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())


# (k, tb, MaxInLen): 20 -> k -> 20.  Ratios cover the 2x convolver + whole-step interpolator,
# 3/2, pure 2^k chains with half-band stages, intermediate-interpolation chains and strong
# decimation on the way back; transition bands span the reference's range down to 0.5 %.
ZERO_CASES = [
    (21, 0.5, 700), (27, 3.1, 233), (30, 1.7, 1021), (40, 0.6, 512), (63, 4.4, 350),
    (147, 2.0, 1500), (320, 1.2, 640), (441, 0.8, 997), (640, 2.6, 421),
]
SNR_CASES = [(49.0, 21, 1.0, 300), (109.0, 77, 2.2, 1100), (163.0, 168, 3.9, 520),
             (218.0, 35, 1.5, 777), (218.0, 420, 0.7, 1234)]


def _check(make, refmake):
    ref = _band_limited(make, 9.3)
    if refmake is not None:
        ref_r = _band_limited(refmake, 9.3)
        d = ref - ref_r
        assert np.sqrt(np.mean(d * d)) <= 1e-15 and np.abs(d).max() <= 1e-13
    worst = -400.0
    for k, tb, maxin in ZERO_CASES:
        rms, pk = _roundtrip(make, ref, k, maxin, tb, 180.15)
        worst = max(worst, _db(rms))
        # reference's own run of this test: average -179.8 dB RMS, peak -166.8 dB
        assert _db(rms) <= -172.0 and _db(pk) <= -158.0, (k, _db(rms), _db(pk))
        if refmake is not None:
            rr, rp = _roundtrip(refmake, ref, k, maxin, tb, 180.15)
            assert abs(_db(rms) - _db(rr)) <= 0.5, (k, _db(rms), _db(rr))
    return worst


def _check_snr(make, refmake):
    ref = _band_limited(make, 9.0)
    for att, k, tb, maxin in SNR_CASES:
        rms, _ = _roundtrip(make, ref, k, maxin, tb, att)
        # the reference lands within a few dB of -ReqAtten over its whole sweep, until double
        # precision itself limits the round trip near -200 dB
        assert _db(rms) <= max(-att + 8.0, -198.0), (att, k, _db(rms))
        if refmake is not None:
            rr, _ = _roundtrip(refmake, ref, k, maxin, tb, att)
            assert abs(_db(rms) - _db(rr)) <= 0.5, (att, k, _db(rms), _db(rr))


def _inlen_consistency(make):
    """zerotest.cpp:105-118: getInLenBeforeOutStart(n-1)+1 vs getInputRequiredForOutput(n); the
    reference prints their difference (0 for every ratio)."""
    u = _uniform01(99, 8)
    for i, k in enumerate((21, 40, 147, 640)):
        for src, dst in ((20.0, float(k)), (float(k), 20.0)):
            r = make(src, dst, 256, 2.0, 180.15)
            n = 1 + int(300 * u[2 * i + (src > dst)])
            assert r.getInLenBeforeOutStart(n - 1) + 1 == r.getInputRequiredForOutput(n), (k, n)


@pytest.fixture(scope="module")
def refmake():
    """Factory over the compiled reference, or None when oracle/_ref is not present (the
    absolute bounds still apply then)."""
    import refwrap as R
    if not R.available():
        return None
    return lambda src, dst, maxin, tb, att: _Ref(R, src, dst, maxin, tb, att)


@pytest.fixture(scope="module")
def emul():
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.run(["make"], cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return r8b.bind(os.path.join(d, "_build", "libr8bsrc_emul.so"))


def test_zerotest_roundtrip_emulated(emul, refmake):
    make = lambda s, d, m, tb, att: r8b.CDSPResampler(s, d, m, tb, att, lib=emul)
    _check(make, refmake)


def test_snrtest_roundtrip_emulated(emul, refmake):
    make = lambda s, d, m, tb, att: r8b.CDSPResampler(s, d, m, tb, att, lib=emul)
    _check_snr(make, refmake)


def test_inlen_consistency_emulated(emul):
    _inlen_consistency(lambda s, d, m, tb, att: r8b.CDSPResampler(s, d, m, tb, att, lib=emul))


@pytest.mark.gpu
def test_zerotest_roundtrip_gpu(refmake):
    make = lambda s, d, m, tb, att: r8b.CDSPResampler(s, d, m, tb, att)
    _check(make, refmake)


@pytest.mark.gpu
def test_snrtest_roundtrip_gpu(refmake):
    make = lambda s, d, m, tb, att: r8b.CDSPResampler(s, d, m, tb, att)
    _check_snr(make, refmake)


@pytest.mark.gpu
def test_inlen_consistency_gpu():
    _inlen_consistency(lambda s, d, m, tb, att: r8b.CDSPResampler(s, d, m, tb, att))


# ---- the reference's sweeps at their real extent, as committed artefacts ----------------------------
# tools/precision_sweeps.py ran bench/zerotest.cpp:98-141 (620 ratios), bench/snrtest.cpp:69-99 (29 attenuations x 83 ratios)
# and bench/masstest.cpp:111-172 (1000 random ratios) through libr8bsrc_hip.so on an MI355X and through the compiled
# reference on the same box; one CSV row per round trip.  This test reads the committed files: every round trip of the
# sweep is there, and the library's figure equals the reference's to a small fraction of a dB.
SWEEPS = {"zerotest": 620, "snrtest": 29 * 83, "masstest": 1000}


def _sweep_rows(name):
    import csv
    path = os.path.join(ROOT, "profiles", "r06_%s.csv" % name)
    with open(path) as f:
        lines = [l for l in f if not l.startswith("#")]
    rows = list(csv.DictReader(lines))
    return [{k: float(v) for k, v in r.items()} for r in rows]


@pytest.mark.parametrize("name", sorted(SWEEPS))
def test_committed_precision_sweep(name):
    rows = _sweep_rows(name)
    assert len(rows) == SWEEPS[name], (name, len(rows))
    worst_rms = max(abs(r["lib_rms_db"] - r["ref_rms_db"]) for r in rows)
    worst_pk = max(abs(r["lib_peak_db"] - r["ref_peak_db"]) for r in rows)
    # (VERDICT's bar is 0.5 dB; measured: 0.001 dB, the CSV's resolution)
    assert worst_rms <= 0.05 and worst_pk <= 0.05, (name, worst_rms, worst_pk)
    if name == "zerotest":
        assert [r["dst"] for r in rows] == [float(k) for k in range(21, 641)] and all(r["src"] == 20.0 for r in rows)
        assert all(0.5 <= r["tb"] <= 5.0 and 50 <= r["maxin"] <= 1550 and r["atten"] == 180.15 for r in rows)
    elif name == "snrtest":
        assert sorted(set(r["atten"] for r in rows)) == [float(a) for a in range(49, 219, 6)]
        assert sorted(set(r["dst"] for r in rows)) == [float(k) for k in range(21, 600, 7)]
        # the round trip tracks ReqAtten until double precision limits it near -200 dB (snrtest's own finding)
        for r in rows:
            assert r["lib_rms_db"] <= max(-r["atten"] + 8.0, -196.0), r
    else:
        assert all(1.0 < r["dst"] <= 45.0 and r["src"] == 1.0 for r in rows)
