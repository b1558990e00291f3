"""CPU tier: the product library (libr8bsrc_hip.so) loads without a GPU, exports every symbol
include/r8bsrc.h declares, and its host-side designer + schedule reproduce the reference's
coefficient sets and per-call output counts.  No compute call is made here (that needs a GPU).

Reference anchors: CDSPFIRFilter.h:220-537 (low-pass), CDSPFracInterpolator.h:61-189 (bank),
CDSPHBUpsampler.h:47-552 (half-band taps), CDSPResampler.h:135-394 (topology), :476-519 (inlen,
max out len).  Golden fixtures were generated from the compiled reference
(tests/golden/make_golden.py); `refwrap` tests run against it directly when oracle/_ref exists.
"""
import ctypes as C
import importlib
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, peak

r8b = importlib.import_module("r8brain-free-src_amd")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(r8b.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return r8b.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "r8bsrc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    # what only TEST builds declare and export (-DR8B_TEST_HOOKS) must be absent from the shipped library
    hooks = re.findall(r"#ifdef R8B_TEST_HOOKS(.*?)#endif", hdr, flags=re.S)
    hook_names = set(re.findall(r"R8BSRC_DECL[^;(]*?\b(r8b_\w+)\s*\(", "".join(hooks)))
    assert hook_names == {p[0] for p in r8b._capi.TEST_HOOK_PROTOTYPES} and hook_names
    for n in hook_names:
        assert not hasattr(lib, n), n + " is exported by the product library"
    hdr = re.sub(r"#ifdef R8B_TEST_HOOKS.*?#endif", "", hdr, flags=re.S)
    names = set(re.findall(r"R8BSRC_DECL[^;(]*?\b(r8b_\w+)\s*\(", hdr))
    assert {"r8b_create", "r8b_delete", "r8b_inlen", "r8b_clear", "r8b_process"} <= names
    assert len(names) >= 30
    bound = {p[0] for p in r8b.PROTOTYPES}
    assert names == bound, names ^ bound
    for n in names:
        assert getattr(lib, n) is not None


def test_no_cpu_fallback_without_gpu(lib):
    """Creating a resampler without a HIP device must fail loudly, not fall back."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    h = lib.r8b_batch_create(44100.0, 96000.0, 1024, 2.0, 180.15, 2, -1)
    assert not h
    assert b"HIP" in lib.r8b_last_error() or b"hip" in lib.r8b_last_error()


def _lp(lib, nf, tb, att, g):
    bits, lat = C.c_int(), C.c_int()
    n = lib.r8b_design_lpfilter(nf, tb, att, g, bits, lat, None, 0)
    taps = np.empty(n)
    lib.r8b_design_lpfilter(nf, tb, att, g, bits, lat, taps.ctypes.data_as(r8b._capi.dp), n)
    return taps, bits.value, lat.value


def _spectrum(taps, fl2, n):
    hz = np.zeros(n)
    hz[:fl2 + 1] = taps[fl2:]
    hz[n - fl2:] = taps[:fl2]
    return np.fft.rfft(hz).real


def test_lp_filter_vs_golden(lib, golden_tables):
    i = 0
    while "lp%d/params" % i in golden_tables:
        nf, tb, att, g, klen, bits, lat = golden_tables["lp%d/params" % i]
        taps, b, l = _lp(lib, nf, tb, att, g)
        assert len(taps) == int(klen) and b == int(bits) and l == int(lat)
        n = 2 << b
        H = _spectrum(taps, l, n) * (2.0 / n)
        Hg = golden_tables["lp%d/H" % i]
        assert peak(H - Hg) <= 4e-16 * peak(Hg) + 1e-18
        i += 1
    assert i >= 5


def test_frac_bank_vs_golden(lib, golden_tables):
    dp = r8b._capi.dp
    i = 0
    while "ws%d/params" % i in golden_tables:
        fr, att, third, flen = golden_tables["ws%d/params" % i]
        fl, nf = C.c_int(), C.c_int()
        n = lib.r8b_design_fracbank(int(fr), 1, 2, att, int(third), fl, nf, None, 0)
        t = np.empty(n)
        lib.r8b_design_fracbank(int(fr), 1, 2, att, int(third), fl, nf, t.ctypes.data_as(dp), n)
        assert fl.value == int(flen) and nf.value == int(fr)
        assert peak(t.reshape(nf.value + 1, -1) - golden_tables["ws%d/table" % i]) <= 2e-15
        i += 1
    assert i >= 3
    i = 0
    while "poly%d/params" % i in golden_tables:
        att, third, flen, fracs = golden_tables["poly%d/params" % i]
        fl, nf = C.c_int(), C.c_int()
        n = lib.r8b_design_fracbank(-1, 3, 8, att, int(third), fl, nf, None, 0)
        t = np.empty(n)
        lib.r8b_design_fracbank(-1, 3, 8, att, int(third), fl, nf, t.ctypes.data_as(dp), n)
        assert fl.value == int(flen) and nf.value == int(fracs)
        rows = golden_tables["poly%d/rows" % i]
        assert peak(t.reshape(nf.value + 1, -1)[rows] - golden_tables["poly%d/table" % i]) <= 1e-14
        i += 1
    assert i >= 2


def test_tables_vs_real_reference(lib, refwrap):
    """Tap-level comparison with the compiled reference, tighter than the fixtures allow."""
    dp = r8b._capi.dp
    for nf, tb, att, g in [(0.5, 2.0, 180.15, 2.0), (0.459375, 2.0, 180.15, 1.0),
                           (0.25, 0.7, 206.91, 1.0), (0.5, 12.0, 90.0, 2.0), (0.4, 30.0, 60.0, 1.0),
                           (1.0 / 3.0, 4.0, 136.45, 3.0), (0.5, 45.0, 49.0, 2.0)]:
        taps, bits, lat = _lp(lib, nf, tb, att, g)
        f = refwrap.lpfilter_real_spectrum(nf, tb, att, g)
        assert len(taps) == f["kernel_len"] and bits == f["block_len_bits"] and lat == f["latency"]
        n = 2 << bits
        H = _spectrum(taps, lat, n) * (2.0 / n)
        assert peak(H - f["H"]) <= 4e-16 * peak(f["H"]) + 1e-18
    for att in (60.0, 109.56, 136.45, 180.15, 206.91):
        for steep in range(0, 8):
            for third in (False, True):
                t = np.zeros(16)
                a = C.c_double()
                n = lib.r8b_design_hbfilter(att, steep, int(third), t.ctypes.data_as(dp), a)
                tr, ar = refwrap.hbfilter(att, steep, third)
                assert n == len(tr) and a.value == ar and np.array_equal(t[:n], tr)
    for fr, att, third in [(160, 180.15, False), (147, 180.15, True), (80, 136.45, False),
                           (1500, 109.56, False), (3, 206.91, False)]:
        fl, nfr = C.c_int(), C.c_int()
        n = lib.r8b_design_fracbank(fr, 1, 2, att, int(third), fl, nfr, None, 0)
        t = np.empty(n)
        lib.r8b_design_fracbank(fr, 1, 2, att, int(third), fl, nfr, t.ctypes.data_as(dp), n)
        b = refwrap.fracbank(fr, 1, 2, att, third)
        assert fl.value == b["filter_len"]
        assert peak(t.reshape(fr + 1, -1) - b["table"]) <= 2e-15
    for s, d in [(88200.0, 96000.0), (96000.0, 44100.0), (44100.0, 44101.0), (1.0, 1.5),
                 (48000.0, 44111.0), (44100.0 * 2, 48000.0), (3.0, 4500.0), (3.0, 4503.0)]:
        i, o = C.c_int(), C.c_int()
        ok = lib.r8b_design_whole_stepping(s, d, i, o)
        rok, ri, ro = refwrap.whole_stepping(s, d)
        assert bool(ok) == rok
        if rok:
            assert (i.value, o.value) == (ri, ro)


def _plan(lib, src, dst, maxin, tb, att):
    p = lib.r8b_plan_create(src, dst, int(maxin), tb, att)
    assert p
    return p


def test_plan_counts_and_inlen_vs_golden(lib, golden_streams):
    with open(os.path.join(GOLDEN, "inlen.json")) as f:
        g = json.load(f)
    for name, rec in g.items():
        src, dst, maxin, chunk, n, tb, att, seed = golden_streams[name + "/params"]
        p = _plan(lib, src, dst, maxin, tb, att)
        assert lib.r8b_plan_max_out_len(p) == rec["max_out_len"], name
        assert [lib.r8b_plan_inlen(p, k) for k in rec["n"]] == rec["inlen"], name
        n, chunk = int(n), int(chunk)
        counts = [lib.r8b_plan_step(p, min(chunk, n - i)) for i in range(0, n, chunk)]
        assert counts == list(golden_streams[name + "/counts"]), name
        lib.r8b_plan_clear(p)
        counts2 = [lib.r8b_plan_step(p, min(chunk, n - i)) for i in range(0, n, chunk)]
        assert counts2 == counts
        lib.r8b_plan_delete(p)


PLAN_CASES = [
    (44100.0, 96000.0, 16384, [16384] * 6),
    (96000.0, 44100.0, 16384, [16384] * 6),
    (44100.0, 2822400.0, 1024, [1024] * 6),
    (44100.0, 96000.0, 1024, [1, 2, 3, 500, 1024, 7, 1024, 1024, 999]),
    (176400.0, 44100.0, 4096, [1000] * 12),
    (2822400.0, 176400.0, 4096, [4096] * 8),
    (44100.0, 44101.0, 1024, [100] * 40 + [1024] * 5),
    (48000.0, 44111.0, 512, [512] * 10),
    (44100.0, 192000.0, 512, [512] * 8),
    (44100.0, 132300.0, 512, [512] * 8),
    (48000.0, 32000.0, 512, [511] * 12),
    (96000.0, 11025.0, 512, [512] * 30),
    (11025.0, 96000.0, 512, [512] * 6),
    (1.0, 1.0, 100, [100, 3]),
]


@pytest.mark.parametrize("case", PLAN_CASES)
def test_plan_counts_vs_real_reference(lib, refwrap, case):
    src, dst, maxin, chunks = case
    for att in (180.15, 136.45):
        p = _plan(lib, src, dst, maxin, 2.0, att)
        r = refwrap.RefResampler(src, dst, maxin, 2.0, att)
        for l in chunks:
            assert lib.r8b_plan_step(p, l) == len(r.process(np.zeros(l))), (case, l)
        if src != dst:
            assert lib.r8b_plan_max_out_len(p) == r.maxout
            for k in (0, 1, 2, 17, 1000, 54321):
                assert lib.r8b_plan_inlen(p, k) == r.input_required(k)
                assert lib.r8b_plan_inlen_before_outpos(p, k) == r.inlen_before_outpos(k)
        lib.r8b_plan_delete(p)


def test_plan_topology_matches_reference_console(lib, refwrap):
    """Stage kinds/sizes against what the reference constructors print (R8BCONSOLE)."""
    for src, dst in [(44100.0, 96000.0), (96000.0, 44100.0), (44100.0, 2822400.0),
                     (176400.0, 44100.0), (44100.0, 192000.0), (44100.0, 44101.0),
                     (48000.0, 32000.0), (44100.0, 132300.0), (192000.0, 44100.0)]:
        p = _plan(lib, src, dst, 1024, 2.0, 180.15)
        n = lib.r8b_plan_describe(p, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.r8b_plan_describe(p, buf, n + 1)
        mine = buf.value.decode().strip().split("\n")
        ref = [l for l in refwrap.topology(src, dst, 1024).split("\n")
               if l.startswith(("CDSPBlockConvolver", "CDSPFracInterpolator", "CDSPHBUpsampler",
                                "CDSPHBDownsampler"))]
        assert len(mine) == len(ref), (mine, ref)
        for a, b in zip(mine, ref):
            assert b[4:].split(":")[0] == a.split(":")[0], (a, b)
            if a.startswith("BlockConvolver"):
                ka = dict(re.findall(r"(\w+)=([\d/]+)", a))
                kb = dict(re.findall(r"(\w+)=([\d/]+)", b))
                for key in ("flt_len", "in_len", "io", "fft"):
                    assert ka[key] == kb[key], (a, b)
            if a.startswith("HB"):
                assert re.search(r"taps=(\d+)", a).group(1) == re.search(r"taps=(\d+)", b).group(1)
        lib.r8b_plan_delete(p)


def test_designer_caches_are_bounded_like_the_reference():
    """VERDICT r4 missing #6 / reference r8bconf.h:90,103 (R8B_FILTER_CACHE_MAX 96, R8B_FRACBANK_CACHE_MAX 12),
    CDSPFIRFilter.h:598-694: a host that sweeps ratios (bench/masstest.cpp: 1 000 of them) must not grow without limit,
    and a live object must never lose its tables.  300 distinct ratios through the host plan, each deleted again: the
    entry counts stay at the bounds; objects that are alive keep theirs beyond the bound."""
    import ctypes as C
    lib = r8b.load()

    def counts():
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib.r8b_design_cache_counts(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    keep = lib.r8b_plan_create(44100.0, 96000.0, 1024, 2.0, 180.15)       # alive throughout
    n0 = [lib.r8b_plan_step(keep, 1024) for _ in range(3)]
    lib.r8b_plan_clear(keep)
    for i in range(300):
        p = lib.r8b_plan_create(44100.0, 44100.0 + 147.0 * (i + 1), 1024, 2.0 + 0.01 * i, 140.0)
        assert p
        lib.r8b_plan_delete(p)
        f, b, _ = counts()
        assert f <= 96 + 2 and b <= 12 + 1, (i, f, b)   # (+ what `keep` holds beside a full cache)
    f, b, _ = counts()
    assert 90 <= f <= 98 and 2 <= b <= 13, (f, b)
    # objects in use at the same time may exceed the bound (reference: "the actual number can be higher") ...
    live = [lib.r8b_plan_create(48000.0, 48000.0 + 100.0 * (i + 1), 512, 3.0 + 0.01 * i, 120.0) for i in range(120)]
    f2, _, _ = counts()
    assert f2 >= 120
    for p in live:
        lib.r8b_plan_delete(p)
    # ... and the next creations bring it back under it
    for i in range(3):
        lib.r8b_plan_delete(lib.r8b_plan_create(32000.0, 32000.0 + 10.0 * (i + 1), 512, 2.5, 100.0))
    f3, _, _ = counts()
    assert f3 <= 96 + 2, f3
    # the object that lived through all of it still follows its schedule
    assert [lib.r8b_plan_step(keep, 1024) for _ in range(3)] == n0
    lib.r8b_plan_delete(keep)
