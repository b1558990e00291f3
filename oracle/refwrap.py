"""oracle/refwrap.py -- ctypes access to the REAL reference built by oracle/Makefile.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  The product (r8brain-free-src_amd) never does.

Loads oracle/_ref/libr8bref.so (oracle/ref_shim.cpp over the unmodified reference
headers) and oracle/_ref/libr8bsrc_ref.so (the reference's own DLL/r8bsrc.cpp).
`available()` is False when the libraries were not built (e.g. no /root/reference and
no prebuilt oracle/_ref/): callers then fall back to the numpy restatement
(oracle/r8b_oracle.py) and say so.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REFDIR = os.path.join(_HERE, "_ref")

_lib = None
_fast = None
_dll = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _bind(lib):
    lib.refx_create.restype = C.c_void_p
    lib.refx_create.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]
    lib.refx_delete.argtypes = [C.c_void_p]
    lib.refx_clear.argtypes = [C.c_void_p]
    lib.refx_process.restype = C.c_int
    lib.refx_process.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int]
    for f in (lib.refx_input_required, lib.refx_inlen_before_outpos,
              lib.refx_inlen_before_outstart):
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int]
    lib.refx_maxoutlen.restype = C.c_int
    lib.refx_maxoutlen.argtypes = [C.c_void_p]
    lib.refx_topology.restype = C.c_int
    lib.refx_topology.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                  C.c_char_p, C.c_int]
    lib.refx_lpfilter.restype = C.c_int
    lib.refx_lpfilter.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double,
                                  _ip, _ip, _ip, _dp, C.c_int]
    lib.refx_fracbank.restype = C.c_int
    lib.refx_fracbank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                  _ip, _ip, _dp, C.c_int]
    lib.refx_fracbank_round_atten.restype = C.c_double
    lib.refx_fracbank_round_atten.argtypes = [C.c_double, C.c_int]
    lib.refx_hbfilter.restype = C.c_int
    lib.refx_hbfilter.argtypes = [C.c_double, C.c_int, C.c_int, _dp, _dp]
    lib.refx_whole_stepping.restype = C.c_int
    lib.refx_whole_stepping.argtypes = [C.c_double, C.c_double, _ip, _ip]
    lib.refx_stage_create.restype = C.c_void_p
    lib.refx_stage_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_int, C.c_int]
    lib.refx_stage_delete.argtypes = [C.c_void_p]
    lib.refx_stage_clear.argtypes = [C.c_void_p]
    lib.refx_stage_maxoutlen.restype = C.c_int
    lib.refx_stage_maxoutlen.argtypes = [C.c_void_p, C.c_int]
    lib.refx_stage_inlen_before_outpos.restype = C.c_int
    lib.refx_stage_inlen_before_outpos.argtypes = [C.c_void_p, C.c_int]
    lib.refx_stage_process.restype = C.c_int
    lib.refx_stage_process.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int]
    lib.refx_bench.restype = C.c_double
    lib.refx_bench.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    lib.refx_batch_check.restype = C.c_int
    lib.refx_batch_check.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.c_int, _ip, _dp, C.c_longlong, _dp, C.c_longlong, _ip, C.c_int,
                                     _dp, _dp, C.POINTER(C.c_longlong)]
    lib.refx_lpfilter_ex.restype = C.c_int
    lib.refx_lpfilter_ex.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                     _ip, _ip, _ip, _dp, _dp, C.c_int]
    lib.refx_create_ex.restype = C.c_void_p
    lib.refx_create_ex.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int]
    lib.refx_batch_check_ex.restype = C.c_int
    lib.refx_batch_check_ex.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                        C.c_int, C.c_int, _ip, _dp, C.c_longlong, _dp, C.c_longlong, _ip,
                                        C.c_int, _dp, _dp, C.POINTER(C.c_longlong)]
    lib.refx_latency_frac.restype = C.c_double
    lib.refx_latency_frac.argtypes = [C.c_void_p]
    lib.refx_version.restype = C.c_char_p
    return lib


def available():
    return os.path.exists(os.path.join(_REFDIR, "libr8bref.so"))


def lib():
    global _lib
    if _lib is None:
        _lib = _bind(C.CDLL(os.path.join(_REFDIR, "libr8bref.so")))
    return _lib


_pffft = None


def pffft_lib():
    """The reference built over its OTHER FFT back-end (oracle/Makefile libr8bref_pffft.so: -DR8B_PFFFT_DOUBLE=1, AVX);
    None when it was not built or the host has no AVX.  Used only to measure the reference against itself."""
    global _pffft
    if _pffft is None:
        p = os.path.join(_REFDIR, "libr8bref_pffft.so")
        ok = False
        try:
            with open("/proc/cpuinfo") as f:
                ok = any(line.startswith("flags") and " avx " in line + " " for line in f)
        except OSError:
            pass
        _pffft = _bind(C.CDLL(p)) if ok and os.path.exists(p) else False
    return _pffft or None


def _cpu_has_avx2_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = set(line.split(":", 1)[1].split())
                    return {"avx2", "fma", "bmi2"} <= fl
    except OSError:
        pass
    return False


def fast_lib():
    """-O3 -march=x86-64-v3 build for the timed CPU baseline; plain build otherwise."""
    global _fast
    if _fast is None:
        p = os.path.join(_REFDIR, "libr8bref_fast.so")
        if os.path.exists(p) and _cpu_has_avx2_fma():
            _fast = (_bind(C.CDLL(p)), "-O3 -march=x86-64-v3")
        else:
            _fast = (lib(), "-O2 x86-64 baseline (SSE2)")
    return _fast


def dll():
    """The reference's own C ABI (DLL/r8bsrc.cpp): r8b_create ... r8b_process."""
    global _dll
    if _dll is None:
        d = C.CDLL(os.path.join(_REFDIR, "libr8bsrc_ref.so"))
        d.r8b_create.restype = C.c_void_p
        d.r8b_create.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_int]
        d.r8b_delete.argtypes = [C.c_void_p]
        d.r8b_clear.argtypes = [C.c_void_p]
        d.r8b_inlen.restype = C.c_int
        d.r8b_inlen.argtypes = [C.c_void_p, C.c_int]
        d.r8b_process.restype = C.c_int
        d.r8b_process.argtypes = [C.c_void_p, _dp, C.c_int, C.POINTER(_dp)]
        _dll = d
    return _dll


ATTEN = {"16": 136.45, "16IR": 109.56, "24": 180.15}


class RefResampler:
    """r8b::CDSPResampler(src, dst, maxin, tb, atten, fprLinearPhase or fprMinPhase)."""

    def __init__(self, src, dst, maxin, tb=2.0, atten=180.15, phase=0, backend=None):
        # (backend: another build of the same shim -- pffft_lib() --, default the Ooura build)
        self._l = backend if backend is not None else lib()
        self.h = (self._l.refx_create_ex(src, dst, maxin, tb, atten, int(phase)) if phase else
                  self._l.refx_create(src, dst, maxin, tb, atten))
        self.maxin = maxin
        self.maxout = self._l.refx_maxoutlen(self.h) if src != dst else maxin
        self._buf = np.empty(self.maxout + 16, dtype=np.float64)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert len(x) <= self.maxin
        n = self._l.refx_process(self.h, _ptr(x), len(x), _ptr(self._buf), len(self._buf))
        assert n <= len(self._buf)
        return self._buf[:n].copy()

    def stream(self, x, chunk=None):
        chunk = chunk or self.maxin
        outs = [self.process(x[i:i + chunk]) for i in range(0, len(x), chunk)]
        return np.concatenate(outs) if outs else np.zeros(0)

    def clear(self):
        self._l.refx_clear(self.h)

    def input_required(self, n):
        return self._l.refx_input_required(self.h, n)

    def inlen_before_outpos(self, p):
        return self._l.refx_inlen_before_outpos(self.h, p)

    def inlen_before_outstart(self, p=0):
        return self._l.refx_inlen_before_outstart(self.h, p)

    def latency_frac(self):
        """CDSPResampler::getLatencyFrac (CDSPResampler.h:491-494)"""
        return self._l.refx_latency_frac(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self._l.refx_delete(self.h)
            self.h = None


class RefStage:
    KINDS = {"conv": 0, "frac": 1, "hbup": 2, "hbdown": 3}

    def __init__(self, kind, a=0.0, b=0.0, c=0.0, d=0.0, i0=0, i1=0):
        self.h = lib().refx_stage_create(self.KINDS[kind], a, b, c, d, i0, i1)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        cap = lib().refx_stage_maxoutlen(self.h, len(x)) + 64
        out = np.empty(cap, dtype=np.float64)
        n = lib().refx_stage_process(self.h, _ptr(x), len(x), _ptr(out), cap)
        return out[:n].copy()

    def stream(self, x, chunk):
        outs = [self.process(x[i:i + chunk]) for i in range(0, len(x), chunk)]
        return np.concatenate(outs) if outs else np.zeros(0)

    def clear(self):
        lib().refx_stage_clear(self.h)

    def inlen_before_outpos(self, p):
        return lib().refx_stage_inlen_before_outpos(self.h, p)

    def __del__(self):
        if getattr(self, "h", None):
            lib().refx_stage_delete(self.h)
            self.h = None


def topology(src, dst, maxin, tb=2.0, atten=180.15):
    buf = C.create_string_buffer(8192)
    lib().refx_topology(src, dst, maxin, tb, atten, buf, len(buf))
    return buf.value.decode()


def lpfilter(normfreq, tb, atten, gain):
    """Returns dict(kernel_len, block_len_bits, latency, kernel_block) where kernel_block is
    the reference's packed zero-phase spectrum block (Ooura layout, CDSPRealFFT.h:395-414)."""
    kl, bb, lat = C.c_int(), C.c_int(), C.c_int()
    n = lib().refx_lpfilter(normfreq, tb, atten, gain, kl, bb, lat, None, 0)
    kb = np.empty(n, dtype=np.float64)
    lib().refx_lpfilter(normfreq, tb, atten, gain, kl, bb, lat, _ptr(kb), n)
    return dict(kernel_len=kl.value, block_len_bits=bb.value, latency=lat.value, kernel_block=kb)


def lpfilter_real_spectrum(normfreq, tb, atten, gain):
    """Decodes the packed block to the Len/2+1 real zero-phase bins H[0..Len/2] (still scaled
    by the reference's InvMulConst=2/Len and the requested gain)."""
    f = lpfilter(normfreq, tb, atten, gain)
    kb = f["kernel_block"]
    n = len(kb)
    H = np.empty(n // 2 + 1)
    H[0] = kb[0]
    H[n // 2] = kb[1]
    H[1:n // 2] = kb[2::2]
    f["H"] = H
    return f


def fracbank(fracs, elsize, interppoints, atten, third=False):
    fl, nf = C.c_int(), C.c_int()
    n = lib().refx_fracbank(fracs, elsize, interppoints, atten, int(third), fl, nf, None, 0)
    t = np.empty(n, dtype=np.float64)
    lib().refx_fracbank(fracs, elsize, interppoints, atten, int(third), fl, nf, _ptr(t), n)
    return dict(filter_len=fl.value, fracs=nf.value,
                table=t.reshape(nf.value + 1, fl.value * elsize))


def fracbank_round_atten(atten, third=False):
    return lib().refx_fracbank_round_atten(atten, int(third))


def hbfilter(atten, steep, third=False):
    taps = np.zeros(16)
    att = C.c_double()
    n = lib().refx_hbfilter(atten, steep, int(third), _ptr(taps), att)
    return taps[:n].copy(), att.value


def whole_stepping(ssr, dsr):
    a, b = C.c_int(), C.c_int()
    ok = lib().refx_whole_stepping(ssr, dsr, a, b)
    return (bool(ok), a.value, b.value)


def bench(src, dst, L, nch, warm, calls, nthreads, tb=2.0, atten=180.15):
    l, flags = fast_lib()
    outs = C.c_longlong()
    secs = l.refx_bench(src, dst, L, tb, atten, nch, warm, calls, nthreads, outs)
    return dict(seconds=secs, in_samples=nch * L * calls, out_samples=outs.value,
                threads=min(nthreads, nch), flags=flags)


def lpfilter_taps(normfreq, tb, atten, gain, phase=0):
    """(taps, latency, latency fraction) of the reference's low-pass, recovered from its spectrum block
    (phase 1: the minimum-phase filter, causal taps)"""
    kl, bb, lat, lf = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    n = lib().refx_lpfilter_ex(normfreq, tb, atten, gain, phase, kl, bb, lat, lf, None, 0)
    blk = np.zeros(n)
    lib().refx_lpfilter_ex(normfreq, tb, atten, gain, phase, kl, bb, lat, lf, blk.ctypes.data_as(_dp), n)
    # Ooura rdft: blk[2k] + i blk[2k+1] = sum_j a_j e^{+2 pi i jk/n}; blk[1] = Nyquist; includes 2/n
    spec = np.zeros(n // 2 + 1, dtype=complex)
    spec[0] = blk[0]
    spec[-1] = blk[1]
    spec[1:-1] = blk[2::2] - 1j * blk[3::2]
    taps = np.fft.irfft(spec, n) * (n / 2.0)
    return taps[:kl.value].copy(), lat.value, lf.value


def batch_check(src, dst, maxin, lens, x, y, counts, tb=2.0, atten=180.15, nthreads=None, phase=0):
    """One reference resampler per row of x (nch x sum(lens), C order) walks the calls of lens[k]
    samples on nthreads threads; y (nch x >= sum(counts)) holds the outputs of the path under test,
    calls back to back.  Returns (rms per channel, peak per channel); raises if a call's output
    count differs from counts[k]."""
    import os
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    nch = x.shape[0]
    assert x.shape[1] >= int(lens.sum()) and y.shape[0] == nch and y.shape[1] >= int(counts.sum())
    if nthreads is None:
        nthreads = max(1, min(64, len(os.sched_getaffinity(0))))
    sq = np.zeros(nch)
    pk = np.zeros(nch)
    tot = C.c_longlong()
    rc = lib().refx_batch_check_ex(src, dst, maxin, tb, atten, phase, nch, len(lens), lens.ctypes.data_as(_ip), x.ctypes.data_as(_dp), x.shape[1],
                                y.ctypes.data_as(_dp), y.shape[1], counts.ctypes.data_as(_ip),
                                nthreads, sq.ctypes.data_as(_dp), pk.ctypes.data_as(_dp), tot)
    if rc != 0:
        raise AssertionError("output count of call %d differs from the reference's" % (rc - 1))
    n = max(int(tot.value), 1)
    return np.sqrt(sq / n), pk


def splitmix_uniform(seed, n):
    """SURVEY Appendix B PRNG: splitmix64 -> uniform [-1, 1) doubles (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0
