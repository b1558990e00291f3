"""oracle/r8b_oracle.py -- CPU restatement of the reference's resampling path (numpy).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; the product path (r8brain-free-src_amd + libr8bsrc_hip.so) never does and
fails loudly without its HIP library.

Parity status: PINNED.  tests/test_oracle.py checks this restatement against (a) the real
reference compiled from /root/reference into oracle/_ref (oracle/Makefile) on identical inputs,
(b) the known-answer vectors of SURVEY.md Appendix B, (c) the committed fixtures under
tests/golden/ (generated from the real reference by tests/golden/make_golden.py).

It restates, as *stream functions* (SURVEY.md Appendix C), what each reference stage computes:
every output sample is written as a pure function of the stage's input stream, with no FFT
blocks and no ring buffers, plus the integer bookkeeping that decides HOW MANY samples each
process() call returns.  Because it shares no structure with either the reference's block/ring
implementation or the HIP kernels, agreement between the three is meaningful.

Reference anchors (file:line under /root/reference):
  topology selection            CDSPResampler.h:135-394
  low-pass design               CDSPFIRFilter.h:220-537, CDSPSincFilterGen.h:114-123,230-241,312-338,586-605
  besselI0 / pow_a / asinh      r8bbase.h:1192-1212, 1154-1157, 1176-1179
  fractional-delay bank         CDSPFracInterpolator.h:61-189, 279-341; CDSPSincFilterGen.h:168-177,452-552
  block convolver semantics     CDSPBlockConvolver.h:62-185 (geometry), 252-354, 512-593
  whole-step interpolator       CDSPFracInterpolator.h:736-747, 802-815, 834-859, 991-1060
  polynomial interpolator       CDSPFracInterpolator.h:1069-1179, 907-919
  half-band up / down           CDSPHBUpsampler.h:605-635, 674-732 (.inc:564-709);
                                CDSPHBDownsampler.h:80-103, 137-239 (.inc:571-716)
"""
import json
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(_HERE, "data", "tables.json")) as f:
            _TABLES = json.load(f)
    return _TABLES


# --------------------------------------------------------------------------- helpers

def bit_occupancy(v):
    """r8bbase.h:766-803: number of significant low bits (1 for v == 0)."""
    return max(1, int(v).bit_length())


def bessel_i0(x):
    """r8bbase.h:1192-1212 -- the Abramowitz-Stegun POLYNOMIAL, not the true I0."""
    x = np.asarray(x, dtype=np.float64)
    ax = np.abs(x)
    small = ax < 3.75
    y = np.where(small, (x / 3.75) ** 2, 0.0)
    r1 = 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 + y * (
        0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))))
    axs = np.where(small, 3.75, ax)
    y2 = 3.75 / axs
    r2 = np.exp(axs) / np.sqrt(axs) * (0.39894228 + y2 * (0.1328592e-1 + y2 * (
        0.225319e-2 + y2 * (-0.157565e-2 + y2 * (0.916281e-2 + y2 * (
            -0.2057706e-1 + y2 * (0.2635537e-1 + y2 * (-0.1647633e-1 + y2 * 0.392377e-2))))))))
    return np.where(small, r1, r2)


def pow_a(v, p):
    """r8bbase.h:1154-1157."""
    return np.exp(p * np.log(np.abs(v) + 1e-300))


def _asinh_ref(v):
    """r8bbase.h:1176-1179 (the reference's own asinh, cancellation and all)."""
    return math.log(v + math.sqrt(v * v + 1.0))


def kaiser_pow_window(pos, len2, beta, power):
    """CDSPSincFilterGen.h:230-241 + 586-605: window value at (possibly fractional)
    positions `pos` measured from the window centre, raised to `power`."""
    beta = min(max(beta, 1.0), 350.0)
    n = 1.0 - (np.asarray(pos, dtype=np.float64) * (1.0 / len2)) ** 2
    w = np.where(n > 0.0, bessel_i0(beta * np.sqrt(np.maximum(n, 0.0))) * (
        1.0 / float(bessel_i0(beta))), 0.0)
    return pow_a(w, abs(power))


# --------------------------------------------------------------------------- low-pass design

def lp_design_params(norm_freq, trans_band, atten_req):
    """CDSPFIRFilter.h:222-448 -> (pwr, hl, fo1)."""
    tb = trans_band * 0.01
    atten = -atten_req
    band = 0 if tb >= 0.25 else (1 if tb >= 0.10 else 2)
    lvl = 0 if atten_req >= 117.0 else (1 if atten_req >= 60.0 else 2)
    atten -= [[1.60, 1.91, 2.25], [0.69, 0.73, 1.13], [0.21, 0.25, 0.36]][band][lvl]
    idx = int(math.floor((-atten - 49.0) * 264 / 176.25 + 0.5))
    idx = min(264, max(0, idx))
    t = tables()["atten_corrs"][band]
    atten -= t["values"][idx] / t["scale"]
    a = atten
    cos, sin, tan, atan, atan2 = math.cos, math.sin, math.tan, math.atan, math.atan2
    exp, sqrt, tanh, cosh, log = math.exp, math.sqrt, math.tanh, math.cosh, math.log
    pwr = (7.43932822146293e-8 * a * a + 0.000102747434588003 *
           cos(0.00785021930010397 * a) * cos(0.633854318781239 + 0.103208573657699 * a) -
           0.00798132247867036 - 0.000903555213543865 * a - 0.0969365532127236 * exp(
               0.0779275237937911 * a) - 1.37304948662012e-5 * a * cos(0.00785021930010397 * a))
    if pwr <= 0.067665322581:
        if band == 0:
            hl = 2.6778150875894 / tb + 300.547590563091 * atan(atan(
                2.68959772209918 * pwr)) / (5.5099277187035 * tb - tb * tanh(cos(_asinh_ref(a))))
            fo1 = 0.987205355829873 * tb + 1.00011788929851 * atan2(
                -0.321432067051302 - 6.19131357321578 * sqrt(pwr),
                hl + -1.14861472207245 / (hl - 14.1821147585957) + math.pow(
                    0.9521145021664, math.pow(atan2(1.12018764830637, tb),
                                              2.10988901686912 * hl - 20.9691278378345)))
        elif band == 1:
            hl = (1.56688617018066 + 142.064321294568 * pwr + 0.00419441117131136 * cos(
                243.633511747297 * pwr) - 0.022953443903576 * a - 0.026629568860284 * cos(
                    127.715550622571 * pwr)) / tb
            fo1 = 0.982299356642411 * tb + 0.999441744774215 * _asinh_ref(
                (-0.361783054039583 - 5.80540593623676 * sqrt(pwr)) / hl)
        else:
            hl = (2.45739657014937 + 269.183679500541 * pwr * cos(5.73225668178813 + atan2(
                cosh(0.988861169868941 - 17.2201556280744 * pwr), 1.08340138240431 * pwr))) / tb
            fo1 = (2.291956939 * tb + 0.01942450693 * tb * tb * hl - 4.67538973161837 * pwr * tb -
                   1.668433124 * tb * math.pow(pwr, pwr))
    else:
        if band == 0:
            hl = (1.50258368698213 + 158.556968859477 * _asinh_ref(pwr) * tanh(
                57.9466246871383 * tanh(pwr)) - 0.0105440479814834 * a) / tb
            fo1 = 0.994024401639321 * tb + (-0.236282717577215 - 6.8724924545387 * sqrt(
                sin(pwr))) / hl
        elif band == 1:
            hl = (1.50277377248945 + 158.222625721046 * _asinh_ref(pwr) * tanh(
                1.02875299001715 + 42.072277322604 * pwr) - 0.0108380943845632 * a) / tb
            fo1 = 0.992539376734551 * tb + (-0.251747813037178 - 6.74159892452584 * sqrt(
                tanh(tanh(tan(pwr))))) / hl
        else:
            hl = (1.15990238966306 * pwr - 5.02124037125213 * pwr * pwr - 0.158676856669827 * a *
                  cos(1.1609073390614 * pwr - 6.33932586197475 * pwr * pwr * pwr)) / tb
            fo1 = (0.867344453126885 * tb + 0.052693817907757 * tb * log(pwr) +
                   0.0895511178735932 * tb * atan(59.7538527741309 * pwr) -
                   0.0745653568081453 * pwr * tb)
    return pwr, hl, fo1


_LP_CACHE = {}


def lp_filter(norm_freq, trans_band, atten_req, gain):
    """Zero-phase low-pass taps h[-fl2..fl2] with DC gain `gain` (C.1 of SURVEY).

    Returns dict(taps, fl2, kernel_len, block_len_bits)."""
    key = (norm_freq, trans_band, atten_req, gain)
    if key in _LP_CACHE:
        return _LP_CACHE[key]
    pwr, hl, fo1 = lp_design_params(norm_freq, trans_band, atten_req)
    len2 = 0.25 * hl / norm_freq
    freq2 = math.pi * (1.0 - fo1) * norm_freq
    fl2 = int(math.floor(len2))
    t = np.arange(1, fl2 + 1, dtype=np.float64)
    win = kaiser_pow_window(np.arange(0, fl2 + 1, dtype=np.float64), len2, 125.0, pwr)
    half = np.empty(fl2 + 1)
    half[0] = freq2 * win[0] / math.pi
    # sine via the reference's 2-term recurrence (r8bbase.h:741-749, CDSPSincFilterGen.h:315-316)
    s1 = 0.0
    s2 = math.sin(-freq2) * (1.0 / math.pi)
    incr = 2.0 * math.cos(freq2)
    sines = np.empty(fl2 + 1)
    for i in range(fl2 + 1):
        sines[i] = s1
        s1, s2 = incr * s1 - s2, s1
    half[1:] = sines[1:] * win[1:] / t
    taps = np.concatenate([half[:0:-1], half])
    s = 0.0
    for v in taps:  # same left-to-right accumulation as CDSPFIRFilter.h:492-498
        s += v
    taps = taps * (gain / s)
    res = dict(taps=taps, fl2=fl2, kernel_len=2 * fl2 + 1,
               block_len_bits=bit_occupancy(2 * fl2))
    _LP_CACHE[key] = res
    return res


# --------------------------------------------------------------------------- fractional bank

def frac_win_params(atten, third):
    """CDSPFracInterpolator.h:279-341 -> (beta, power, rounded atten, filter_len)."""
    fr = tables()["frac"]
    rows = fr["Coeffs3" if third else "Coeffs2"]
    base = fr["Coeffs3Base" if third else "Coeffs2Base"]
    i = 0
    while i != len(rows) - 1 and rows[i][2] < atten:
        i += 1
    return rows[i][0], rows[i][1], rows[i][2], base + 2 * i


def frac_filter(frac_delay, filter_len, beta, power):
    """One windowed-sinc fractional-delay filter (CDSPSincFilterGen.h:168-177, 452-552),
    normalised to unit DC gain (r8bbase.h:931-961)."""
    len2 = float(filter_len // 2)
    fl2 = int(math.ceil(len2))
    fd = frac_delay
    t = np.arange(-fl2, fl2, dtype=np.float64)  # tap i <-> time t = -fl2 + i
    ut = t + fd
    win = kaiser_pow_window(ut, len2, beta, power)
    f0 = math.sin(fd * math.pi) / math.pi
    sign = np.where((np.arange(-fl2, fl2) & 1) != 0, -1.0, 1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = np.where(np.abs(ut) > 0.0, sign * f0 * win / ut, 0.0)
    # the sample where t + fd == 0 (fd ~ 0 at t = 0, fd ~ 1 at t = -1) is the window itself
    if abs(fd) < 2.3e-13:
        v[fl2] = win[fl2]
    elif abs(fd - 1.0) < 2.3e-13:
        v[fl2 - 1] = win[fl2 - 1]
    # edges: first tap is zero when t+fd < -Len2; last tap is zero when ut > Len2
    if (-fl2 + fd) < -len2:
        v[0] = 0.0
    if ut[-1] > len2:
        v[-1] = 0.0
    s = 0.0
    for x in v:
        s += x
    return v * (1.0 / s)


_BANK_CACHE = {}


def frac_bank(fracs, element_size, interp_points, atten, third):
    """CDSPFracDelayFilterBank ctor (CDSPFracInterpolator.h:61-189).

    Returns dict(filter_len, fracs, table[(fracs+1), filter_len*element_size]) where for
    element_size == 3 each tap holds (c0, c1, c2) of the 8-point 2nd-order spline in the
    NATURAL (unshuffled) order (the reference shuffles pairs for SIMD, :369-384)."""
    key = (fracs, element_size, interp_points, atten, bool(third))
    if key in _BANK_CACHE:
        return _BANK_CACHE[key]
    beta, power, att_r, flen = frac_win_params(atten, third)
    if fracs == -1:
        fracs = int(math.ceil(math.pow(6.4, att_r / 50.0)))
    pc2 = interp_points // 2
    idx = list(range(-pc2 + 1, fracs + pc2 + 1))
    filt = np.stack([frac_filter((fracs - i) / float(fracs), flen, beta, power) for i in idx])
    if element_size == 1:
        table = filt[:fracs + 1].copy()
    elif element_size == 3 and interp_points == 8:
        # row r of `filt` is bank index i = r - 3; spline for index i uses rows i-3..i+4 of the
        # bank, i.e. filt[r .. r+7] with x0 = filt[r+3] (r8bbase.h:1014-1024)
        xm3, xm2, xm1, x0, x1, x2, x3, x4 = [filt[k:k + fracs + 1] for k in range(8)]
        k = 1.31578947368421052e-2
        c0 = x0
        c1 = (61.0 * (x1 - xm1) + 16.0 * (xm2 - x2) + 3.0 * (x3 - xm3)) * k
        c2 = (106.0 * (xm1 + x1) + 10.0 * x3 + 6.0 * xm3 - 3.0 * x4 - 29.0 * (xm2 + x2) -
              167.0 * x0) * k
        table = np.stack([c0, c1, c2], axis=2).reshape(fracs + 1, flen * 3)
    else:
        raise NotImplementedError("element_size %d" % element_size)
    res = dict(filter_len=flen, fracs=fracs, table=table, atten=att_r)
    _BANK_CACHE[key] = res
    return res


def find_gcd(l, s):
    """CDSPFracInterpolator.h:609-628 (double subtraction Euclid, <150 iterations)."""
    it = 0
    while True:
        it += 1
        if it >= 150:
            return None
        r = l - s
        if r == 0.0:
            return s if s > 0.0 else None
        l, s = s, abs(r)


def whole_stepping(ssr, dsr):
    """CDSPFracInterpolator.h:644-673 -> (ok, in_step, out_step)."""
    g = find_gcd(ssr, dsr)
    if g is None:
        return False, 0, 0
    i0 = ssr / g
    o0 = dsr / g
    if i0 != int(i0) or o0 != int(o0):
        return False, int(i0), int(o0)
    if int(o0) > 1500:
        return False, int(i0), int(o0)
    return True, int(i0), int(o0)


def hb_filter(atten, steep, third):
    """CDSPHBUpsampler.h:47-316 / 331-552 selection rule over the generated tap data."""
    rows = tables()["hb"]["third" if third else "half"][min(max(steep, 0), 6)]
    k = 0
    while k != len(rows) - 1 and rows[k]["att"] < atten:
        k += 1
    return np.array(rows[k]["taps"]), rows[k]["att"]


# --------------------------------------------------------------------------- stages

def _take(x, idx):
    """x[idx] with zeros outside [0, len(x))."""
    idx = np.asarray(idx)
    ok = (idx >= 0) & (idx < len(x))
    return np.where(ok, x[np.clip(idx, 0, max(len(x) - 1, 0))] if len(x) else 0.0, 0.0)


class _Stage:
    """Common driver: keeps the whole input stream, emits outputs [done, total(M))."""

    def __init__(self):
        self.x = np.zeros(0)
        self.done = 0

    def clear(self):
        self.x = np.zeros(0)
        self.done = 0

    def process(self, chunk):
        self.x = np.concatenate([self.x, np.asarray(chunk, dtype=np.float64)])
        tot = self.total(len(self.x))
        out = self.eval(self.done, tot) if tot > self.done else np.zeros(0)
        self.done = max(self.done, tot)
        return out


class ConvStage(_Stage):
    """CDSPBlockConvolver (linear phase, DoConsumeLatency): C.2 of SURVEY.

    y_full[t] = sum_k h[k] * xu[t - k], xu[Up*n] = x[n]; output q is y_full[Down*q].
    For power-of-2 Down the reference decimates by truncating each block's spectrum
    (CDSPBlockConvolver.h:108-111, 329-344, 543-562); `exact_down` reproduces that block
    procedure (with numpy's FFT) so that parity holds to 1e-15 there as well."""

    def __init__(self, norm_freq, tb, atten, gain, up, down):
        super().__init__()
        f = lp_filter(norm_freq, tb, atten, gain)
        self.h = f["taps"]
        self.fl2 = f["fl2"]
        self.up, self.down = up, down
        klen = f["kernel_len"]
        self.bl2 = 2 << f["block_len_bits"]
        ups = bit_occupancy(up) - 1
        if (1 << ups) == up:
            self.prev_len = (klen - 1 + up - 1) // up
            self.in_len = self.bl2 - self.prev_len * up
        else:
            self.prev_len = klen - 1
            self.in_len = self.bl2 - self.prev_len
        self.latency = self.in_len + self.fl2
        dsh = bit_occupancy(down) - 1
        self.down_pow2 = (1 << dsh) == down and down > 1
        if self.down_pow2:
            assert (1 << ups) != up or up == 1  # CDSPBlockConvolver.h:114-121
            ilc = self.in_len & (down - 1)
            self.prev_len += ilc
            self.in_len -= ilc
            self.latency -= ilc
        self.fft_in = self.bl2 // up if (1 << ups) == up else self.bl2
        self.fft_out = self.bl2 // down if self.down_pow2 else self.bl2

    def total(self, n):
        v = self.up * n - self.latency
        return 0 if v <= 0 else (v + self.down - 1) // self.down

    def in_len_before_out_pos(self, pos):
        return int((self.latency + float(pos) * self.down) / self.up)

    def max_out_len(self, maxin):
        return (maxin * self.up + self.down - 1) // self.down

    def eval(self, a, b):
        if self.down_pow2:
            return self._eval_down_blocks(a, b)
        # direct (time-domain) form over the zero-stuffed stream; np.convolve is a plain
        # O(N*K) correlation, not an FFT, so this shares nothing with the block procedure
        t0, t1 = self.down * a, self.down * (b - 1) + 1  # y_full times needed: [t0, t1)
        lo, hi = t0 - self.fl2, t1 + self.fl2  # xu indices needed: [lo, hi)
        idx = np.arange(lo, hi)
        xu = np.where(idx % self.up == 0, _take(self.x, idx // self.up), 0.0)
        yf = np.convolve(xu, self.h, mode="valid")  # yf[i] = y_full[t0 + i]
        return yf[::self.down].copy()

    def _xu(self, idx):
        """zero-stuffed input stream: xu[Up*n] = x[n]."""
        return np.where(idx % self.up == 0, _take(self.x, idx // self.up), 0.0)

    def _eval_down_blocks(self, a, b):
        # block blk covers absolute (full-rate) times [blk*IL - fl2, blk*IL + IL - fl2)
        il, bl2, d = self.in_len, self.bl2, self.down
        hz = np.zeros(bl2)
        hz[:self.fl2 + 1] = self.h[self.fl2:]
        hz[bl2 - self.fl2:] = self.h[:self.fl2]
        H = np.fft.rfft(hz).real
        out = np.empty(b - a)
        cache = {}
        for i, q in enumerate(range(a, b)):
            T = q * d
            blk = (T + self.fl2) // il
            if blk not in cache:
                cur = np.zeros(bl2)
                cur[:il] = self._xu(np.arange(blk * il, blk * il + il))
                cur[il:] = self._xu(np.arange(blk * il - self.prev_len, blk * il))
                X = np.fft.rfft(cur) * H
                z = bl2 // d // 2
                Y = X[:z + 1].copy()
                # reference Nyquist fix (CDSPBlockConvolver.h:329-342): multiplyBlocksZP ran over
                # the SHORT (output) length only, so bin z is still unmultiplied there and
                # p[1] = kb[z]*p[z] - kb[z+1]*p[z+1] is H[z]*(Re - Im_ooura) = H[z]*(Re + Im) in the
                # e^{-i...} convention used here (Ooura stores +sin imaginary parts).
                Y[z] = X[z].real + X[z].imag
                cache = {blk: np.fft.irfft(Y, bl2 // d) * (1.0 / d)}
            c = (T - blk * il) % bl2
            out[i] = cache[blk][c // d]
        return out


class WholeStepStage(_Stage):
    """CDSPFracInterpolator in whole-stepping mode (C.3 of SURVEY)."""

    def __init__(self, ssr, dsr, atten, third):
        super().__init__()
        ok, self.in_step, self.out_step = whole_stepping(ssr, dsr)
        assert ok
        self.ssr, self.dsr = ssr, dsr
        bank = frac_bank(self.out_step, 1, 2, atten, third)
        self.table = bank["table"]
        self.flen = bank["filter_len"]
        self.fl2 = self.flen // 2
        self.fll = self.fl2 - 1

    def total(self, m):
        lim = m - self.fl2 - 1  # last allowed integer input position
        if lim < 0:
            return 0
        # largest j with floor(j*In/Out) <= lim  <=>  j*In < (lim+1)*Out
        return ((lim + 1) * self.out_step - 1) // self.in_step + 1

    def in_len_before_out_pos(self, pos):
        return self.fl2 + int((0 + float(pos) * self.in_step) / self.out_step)

    def max_out_len(self, maxin):
        return int(math.ceil(maxin * self.dsr / self.ssr)) + 1

    def eval(self, a, b):
        j = np.arange(a, b, dtype=np.int64)
        p = j * self.in_step
        ph = p % self.out_step
        r = p // self.out_step
        idx = r[:, None] - self.fll + np.arange(self.flen)[None, :]
        return np.sum(self.table[ph] * _take(self.x, idx), axis=1)


class PolyStage:
    """CDSPFracInterpolator, non-whole stepping (convolve2, :1069-1179) -- stateful
    double-precision position counter restated literally, including the per-call counter
    re-base (:907-919)."""

    def __init__(self, ssr, dsr, atten, third):
        self.ssr, self.dsr = ssr, dsr
        bank = frac_bank(-1, 3, 8, atten, third)
        self.fracs = bank["fracs"]
        self.flen = bank["filter_len"]
        self.tab = bank["table"].reshape(self.fracs + 1, self.flen, 3)
        self.fl2 = self.flen // 2
        self.fll = self.fl2 - 1
        self.clear()

    def clear(self):
        self.x = np.zeros(0)
        self.rpos = 0  # absolute integer input position of the next output
        self.in_pos_frac = 0.0
        self.in_counter = 0
        self.in_pos_int = 0
        self.in_pos_shift = 0.0

    def in_len_before_out_pos(self, pos):
        return self.fl2 + int(0.0 + pos * self.ssr / self.dsr)

    def max_out_len(self, maxin):
        return int(math.ceil(maxin * self.dsr / self.ssr)) + 1

    def process(self, chunk):
        self.x = np.concatenate([self.x, np.asarray(chunk, dtype=np.float64)])
        m = len(self.x)
        out = []
        fpos = self.in_pos_frac
        taps = np.arange(self.flen)
        while m - self.rpos - self.fl2 > 0:
            xx = fpos * self.fracs
            fti = int(xx)
            xx -= fti
            c = self.tab[fti]
            coef = c[:, 0] + c[:, 1] * xx + c[:, 2] * (xx * xx)
            out.append(float(np.sum(coef * _take(self.x, self.rpos - self.fll + taps))))
            self.in_counter += 1
            nxt = (self.in_counter + self.in_pos_shift) * self.ssr / self.dsr
            nxt_i = int(nxt)
            self.rpos += nxt_i - self.in_pos_int
            self.in_pos_int = nxt_i
            fpos = nxt - nxt_i
        self.in_pos_frac = fpos
        if self.in_counter > 1000:
            self.in_counter = 0
            self.in_pos_int = 0
            self.in_pos_shift = self.in_pos_frac * self.dsr / self.ssr
        return np.array(out)


class HBUpStage(_Stage):
    """CDSPHBUpsampler (C.4): y[2n] = x[n]; y[2n+1] = sum_k f[k] (x[n+1+k] + x[n-k])."""

    def __init__(self, atten, steep, third):
        super().__init__()
        self.f, self.att = hb_filter(atten, steep, third)
        self.T = len(self.f)

    def total(self, m):
        return 2 * max(0, m - self.T)

    def in_len_before_out_pos(self, pos):
        return self.T + int((0 + 0.0 + pos) * 0.5)

    def max_out_len(self, maxin):
        return maxin * 2

    def eval(self, a, b):
        n0, n1 = a // 2, (b + 1) // 2
        n = np.arange(n0, n1)
        k = np.arange(self.T)
        odd = np.sum(self.f[None, :] * (_take(self.x, n[:, None] + 1 + k[None, :]) +
                                        _take(self.x, n[:, None] - k[None, :])), axis=1)
        y = np.empty(2 * len(n))
        y[0::2] = _take(self.x, n)
        y[1::2] = odd
        return y[a - 2 * n0:b - 2 * n0]


class HBDownStage(_Stage):
    """CDSPHBDownsampler (C.5): y[n] = x[2n] + sum_k f[k] (x[2n+1+2k] + x[2n-1-2k])."""

    def __init__(self, atten, steep, third):
        super().__init__()
        self.f, self.att = hb_filter(atten, steep, third)
        self.T = len(self.f)

    def total(self, m):
        return max(0, m // 2 - self.T + 1)

    def in_len_before_out_pos(self, pos):
        return 2 * self.T - 1 + int((0 + 0.0 + pos) * 2.0)

    def max_out_len(self, maxin):
        return (maxin + 1) >> 1

    def eval(self, a, b):
        n = np.arange(a, b)
        k = np.arange(self.T)
        return _take(self.x, 2 * n) + np.sum(self.f[None, :] * (
            _take(self.x, 2 * n[:, None] + 1 + 2 * k[None, :]) +
            _take(self.x, 2 * n[:, None] - 1 - 2 * k[None, :])), axis=1)


# --------------------------------------------------------------------------- front-end

def build_topology(src, dst, tb=2.0, atten=206.91):
    """CDSPResampler ctor (CDSPResampler.h:135-394) -> list of stage descriptors."""
    steps = []
    if src == dst:
        return steps
    for num, den in ((1, 2), (1, 3), (2, 3), (3, 2), (3, 4)):
        if src * num == dst * den:
            steps.append(("conv", 1.0 / max(num, den), tb, atten, float(num), num, den))
            return steps
    for i in (2, 3):
        c = 0
        found = False
        while True:
            nsr = src * (i << c)
            if nsr == dst:
                found = True
                break
            if nsr > dst:
                break
            c += 1
        if found:
            steps.append(("conv", 1.0 / i, tb, atten, float(i), i, 1))
            for s in range(c):
                steps.append(("hbup", atten, s, i == 3))
            return steps
    if dst * 2.0 > src:
        nf = 0.5 if dst > src else 0.5 * dst / src
        steps.append(("conv", nf, tb, atten, 2.0, 2, 1))
        tbw = 0.0175
        thresh = src / (1.0 - tbw * tb)
        c, div = 0, 1
        while True:
            nd = div * 2
            if dst < thresh * nd:
                break
            div = nd
            c += 1
        c2, div2 = 0, 1
        while True:
            nd = div * (3 if c2 == 0 else 2)
            if dst < thresh * nd:
                break
            div2 = nd
            c2 += 1
        src2 = src * 2.0
        if c == 1 and whole_stepping(src2, dst)[0]:
            c = 0
        if c > 0:
            if c2 > 0 and div2 > div:
                div, c, num = div2, c2, 3
            else:
                num = 2
            steps.append(("frac", src2 * div, dst, atten, False))
            tb2 = (1.0 - src * div / dst) / tbw
            tb2 = min(tb2, 45.0)
            steps.append(("conv", 1.0 / num, tb2, atten, float(num), num, 1))
            for s in range(1, c):
                steps.append(("hbup", atten, s - 1, num == 3))
        else:
            steps.append(("frac", src2, dst, atten, False))
        return steps
    check = dst * 4.0
    c = 0
    fin_gain = 1.0
    while check <= src:
        c += 1
        check *= 2.0
        fin_gain *= 0.5
    srdiv = 1 << c
    nf = 0.5
    use_interp = True
    third = False
    downf = 1
    for df in (2, 3):
        if dst * srdiv * df == src:
            nf = 1.0 / df
            use_interp = False
            third = df == 3
            downf = df
            break
    if use_interp:
        downf = 1
        nf = dst * srdiv / src
        third = nf * 3.0 <= 1.0
    for i in range(c):
        steps.append(("hbdown", atten, c - 1 - i, third))
    steps.append(("conv", nf, tb, atten, fin_gain, 1, downf))
    if use_interp:
        steps.append(("frac", src, dst * srdiv, atten, third))
    return steps


def make_stage(d):
    kind = d[0]
    if kind == "conv":
        return ConvStage(*d[1:])
    if kind == "frac":
        ssr, dsr, atten, third = d[1:]
        if whole_stepping(ssr, dsr)[0]:
            return WholeStepStage(ssr, dsr, atten, third)
        return PolyStage(ssr, dsr, atten, third)
    if kind == "hbup":
        return HBUpStage(*d[1:])
    if kind == "hbdown":
        return HBDownStage(*d[1:])
    raise ValueError(kind)


class OracleResampler:
    """Restatement of r8b::CDSPResampler (linear phase).  process() returns exactly the
    samples (count and values, to fp64 rounding) the reference's process() returns."""

    def __init__(self, src, dst, maxin, tb=2.0, atten=180.15):
        self.src, self.dst, self.maxin = src, dst, maxin
        self.desc = build_topology(src, dst, tb, atten)
        self.stages = [make_stage(d) for d in self.desc]
        mo = maxin
        for s in self.stages:
            mo = s.max_out_len(mo)
        self.max_out_len = mo

    def process(self, x):
        y = np.asarray(x, dtype=np.float64)
        for s in self.stages:
            y = s.process(y)
        return y

    def stream(self, x, chunk=None):
        chunk = chunk or self.maxin
        outs = [self.process(x[i:i + chunk]) for i in range(0, len(x), chunk)]
        return np.concatenate(outs) if outs else np.zeros(0)

    def clear(self):
        for s in self.stages:
            s.clear()

    def in_len_before_out_pos(self, pos):
        r = pos
        for s in reversed(self.stages):
            r = s.in_len_before_out_pos(r)
        return r

    def input_required(self, n):
        return 0 if n < 1 else self.in_len_before_out_pos(n - 1) + 1


def splitmix_uniform(seed, n):
    """SURVEY Appendix B PRNG (splitmix64 -> uniform [-1,1))."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0
