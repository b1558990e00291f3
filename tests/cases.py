"""Shared parity cases and the stream comparison helper used by the emulation tier (CPU) and the
GPU tier.  Tolerance: direct parity vs the oracle on identical input, RMS <= 1e-15
and peak <= 1e-13 on +-1.0 full-scale noise (the reference's own cross-build noise floor is
3e-16 / 2.3e-15)."""
import numpy as np

import r8b_oracle as O

RMS_TOL = 1e-15
PEAK_TOL = 1e-13


def tol_scale(ref_sq_sum, ref_count):
    """The stated bound is relative to FULL-SCALE noise: +-1.0 uniform input has RMS 0.577, and so do its resampled
    outputs through the reference's usual filters.  Outputs are not bounded by full scale, though (short filters of
    50-60 dB overshoot to 1.9x): rounding errors grow with the signal they ride on, in the reference's own builds as
    in ours.  Streams louder than full-scale noise get the bound scaled by their RMS; quieter ones
    keep the absolute bound."""
    if ref_count <= 0:
        return 1.0
    return max(1.0, (ref_sq_sum / ref_count) ** 0.5 / 0.577)

# (src, dst, maxin, chunk, n_in, tb, atten)
STREAM_CASES = [
    (44100.0, 96000.0, 4096, 4096, 4096 * 4, 2.0, 180.15),      # cfg2 topology
    (96000.0, 44100.0, 4096, 1000, 4096 * 4, 2.0, 180.15),      # cfg3 topology, ragged chunks
    (44100.0, 2822400.0, 1024, 1024, 4096, 2.0, 180.15),        # cfg5: convolver + 5 half-band up
    (176400.0, 44100.0, 4096, 1000, 30000, 2.0, 180.15),        # half-band down + 2^k-down convolver
    (2822400.0, 176400.0, 4096, 4096, 65536, 2.0, 180.15),      # sacd.cpp second pass
    (44100.0, 44101.0, 1024, 100, 5000, 2.0, 180.15),           # polynomial-interpolated bank
    (44100.0, 96000.0, 1024, 1, 1800, 2.0, 180.15),             # one sample per call
    (48000.0, 44111.0, 512, 512, 5000, 2.0, 136.45),            # 16-bit preset, non-whole step
    (64000.0, 48000.0, 1024, 333, 15000, 2.0, 180.15),          # 3/4: up 3, down 4
    (44100.0, 132300.0, 1024, 1024, 15000, 2.0, 180.15),        # 3x up (zero stuffing)
    (48000.0, 32000.0, 1024, 1024, 15000, 2.0, 180.15),         # 2/3: up 2, strided down 3
    (11025.0, 96000.0, 512, 512, 3000, 2.0, 136.45),            # intermediate interpolation chain
    (96000.0, 11025.0, 512, 512, 30000, 2.0, 136.45),           # deep decimation
    (44100.0, 192000.0, 512, 512, 6000, 2.0, 180.15),           # interp -> convolver -> half-band
    (96000.0, 48000.0, 512, 512, 9000, 5.0, 109.56),            # 1/2 via one convolver, 16IR preset
    (44100.0, 48000.0, 512, 512, 9000, 0.5, 109.56),            # narrow transition band (long filter)
    (44100.0, 88200.0, 512, 512, 6000, 45.0, 49.0),             # widest band, lowest attenuation
    (44100.0, 44100.0, 512, 512, 1024, 2.0, 180.15),            # Src == Dst passthrough
    (44100.0, 96000.0, 70000, 70000, 140000, 2.0, 180.15),      # 53 FFT blocks per call
    (44100.0, 529200.0, 512, 300, 3000, 2.0, 180.15),           # 3x convolver + two third-band half-bands
    (44100.0, 705600.0, 256, 256, 1024, 2.0, 136.45),           # 2x + three half-bands, 16-bit preset
    # narrowest transition band at 24-bit attenuation: 16384-point blocks, in-place kernels only
    (44100.0, 96000.0, 2048, 2048, 36000, 0.5, 180.15),         # 8192 -> 16384-point transforms, fused
    (44100.0, 88200.0, 2048, 1500, 36000, 0.5, 180.15),         # same, convolver only
    (192000.0, 44100.0, 4096, 4096, 72000, 0.5, 180.15),        # 16384 -> 16384, fused
    (176400.0, 44100.0, 4096, 3000, 72000, 0.5, 180.15),        # 16384 -> 8192 (2x decimating)
    (192000.0, 44100.0, 4096, 4096, 40000, 1.0, 180.15),        # 8192 -> 8192
    (88200.0, 44100.0, 1024, 700, 12000, 2.0, 180.15),          # 2x decimating convolver alone
    (32000.0, 48000.0, 1024, 777, 20000, 2.0, 180.15),          # 3/2: zero stuffing + 2x decimating spectrum
    (96000.0, 32000.0, 2048, 2048, 30000, 2.0, 136.45),         # 1/3: strided decimation alone
    (705600.0, 44100.0, 2048, 1333, 70000, 2.0, 136.45),        # three half-band decimators as one kernel
    (20.0, 21.0, 300, 300, 6000, 1.0, 49.0),                    # 8-tap interpolator rows (zero padding)
    (32000.0, 96000.0, 2048, 2048, 30000, 1.0, 180.15),         # 3x zero stuffing, 16384-point, in place
    (96000.0, 32000.0, 4096, 4096, 60000, 1.0, 180.15),         # strided 3x decimation, 16384-point
    # polynomial interpolator at other steps: 3.2 inputs per output with 18 taps (tile spans beyond the one-phase
    # front of the tiled kernel, tap loop over zero-padded rows), ~1 and 1.84 (row pitch chosen per step)
    (96000.0, 30001.0, 1024, 777, 12000, 2.0, 180.15),
    (22050.0, 44101.0, 1024, 500, 6000, 2.0, 180.15),
    (44100.0, 48001.0, 1024, 1024, 6000, 2.0, 180.15),
    # a 2x half-band up-sampler as the LAST stage behind a fused interpolator: its 16-byte output pairs start at even
    # and at odd elements of the caller's rows from call to call (ragged chunks)
    (8000.0, 44100.0, 1024, 777, 6000, 2.0, 180.15),
]


# Shorter filters (wider transition band / lower attenuation than the 24-bit preset): backward
# transforms of 256 ... 2048 points, where a workgroup of the pair kernel (r8b_convp.h) carries 2 ... 16
# consecutive blocks of its channel pair.  Ragged chunks leave block groups partly filled.
# (src, dst, maxin, chunk, n_in, tb, atten, expected describe() fragment)
SHORT_CASES = [
    (44100.0, 96000.0, 4096, 1000, 14000, 2.0, 109.56, "fft=1024/2048"),   # 16IR preset, 2x up, fused
    (96000.0, 44100.0, 4096, 4096, 16384, 2.0, 109.56, "fft=2048/2048"),   # 16IR preset, 1:1, fused
    (44100.0, 88200.0, 2048, 777, 9000, 5.0, 109.56, "fft=512/1024"),      # convolver alone
    (96000.0, 44100.0, 4096, 1500, 16000, 5.0, 136.45, "fft=1024/1024"),
    (44100.0, 96000.0, 4096, 4096, 12288, 10.0, 109.56, "fft=256/512"),
    (96000.0, 44100.0, 2048, 900, 9000, 10.0, 109.56, "fft=512/512"),
    (44100.0, 96000.0, 4096, 333, 6000, 20.0, 109.56, "fft=128/256"),
    (96000.0, 44100.0, 2048, 2048, 8192, 20.0, 109.56, "fft=256/256"),
    (44100.0, 44101.0, 1024, 1024, 5000, 5.0, 109.56, "fft=512/1024"),     # polynomial bank behind it
    (44100.0, 705600.0, 512, 300, 2000, 5.0, 109.56, "fft=512/1024"),      # half-band cascade behind it
    (44100.0, 96000.0, 50000, 50000, 100000, 5.0, 109.56, "fft=512/1024"), # 148 blocks per call: split launches
    (44100.0, 96000.0, 2048, 700, 6000, 30.0, 109.56, "fft=64/128"),       # 8 threads per block, fused
    (96000.0, 44100.0, 2048, 2048, 8192, 45.0, 109.56, "fft=128/128"),
    (44100.0, 96000.0, 2048, 1111, 6000, 45.0, 49.0, "fft=32/64"),         # 4 threads per block; run does not
    (96000.0, 44100.0, 1024, 1024, 6000, 45.0, 49.0, "fft=64/64"),         # fit the block's array: unfused
    (44100.0, 132300.0, 2048, 1500, 9000, 10.0, 109.56, "io=3/1"),         # 3x zero stuffing in the pair load
    (96000.0, 32000.0, 4096, 4096, 20000, 10.0, 109.56, "io=1/3"),         # 3x strided store from registers
    (48000.0, 32000.0, 2048, 900, 12000, 5.0, 109.56, "io=2/3"),           # 2x up transform + strided store
    (44100.0, 132300.0, 1024, 1024, 6000, 45.0, 49.0, "io=3/1"),
    (48000.0, 16000.0, 2048, 2048, 12000, 2.0, 109.56, "io=1/3"),          # 4096-point blocks
    # 2x / 4x decimation in the spectrum (incl. the reference's Nyquist fix-up, visible at low attenuation)
    (88200.0, 44100.0, 4096, 1000, 24000, 2.0, 180.15, "fft=4096/2048"),
    (88200.0, 44100.0, 4096, 4096, 24000, 2.0, 109.56, "fft=2048/1024"),
    (96000.0, 48000.0, 2048, 700, 16000, 10.0, 109.56, "io=1/2"),
    (88200.0, 44100.0, 2048, 2048, 12000, 30.0, 60.0, "io=1/2"),
    (88200.0, 44100.0, 2048, 333, 8000, 45.0, 49.0, "fft=64/32"),
    (176400.0, 44100.0, 4096, 3000, 30000, 5.0, 109.56, "io=1/2"),         # half-band down, then 2x decimating
    (32000.0, 48000.0, 2048, 2048, 16000, 2.0, 109.56, "io=3/2"),          # 3x zero stuffing + 2x decimation
    (32000.0, 48000.0, 1024, 500, 8000, 45.0, 49.0, "io=3/2"),
    (64000.0, 48000.0, 2048, 1100, 16000, 2.0, 109.56, "io=3/4"),          # 4x decimation
    (64000.0, 48000.0, 2048, 2048, 12000, 10.0, 109.56, "io=3/4"),
    (64000.0, 48000.0, 1024, 1024, 8000, 45.0, 49.0, "io=3/4"),
    # more interpolator phases than threads (fused: a thread walks phases tid, tid + 256, ...)
    (32000.0, 44100.0, 4096, 3000, 20000, 2.0, 180.15, "step=640/441"),
    (32000.0, 44100.0, 4096, 1000, 12000, 5.0, 109.56, "step=640/441"),    # ... with 4 blocks per workgroup
    (16000.0, 44100.0, 2048, 2048, 10000, 2.0, 180.15, "step=320/441"),    # two phases per thread, one group set
    (8000.0, 44100.0, 1024, 700, 6000, 2.0, 136.45, "step=640/441"),       # ... inside a longer chain
    # fewer phases than threads: several lanes share a phase and take its output groups in turn
    (88200.0, 48000.0, 4096, 3000, 30000, 2.0, 180.15, "step=147/40"),     # 40 phases, 512 threads (12 sets)
    (88200.0, 96000.0, 4096, 1500, 20000, 10.0, 109.56, "step=147/80"),    # 80 phases, 256 threads (3 sets), 8 blocks
    # 8192-point blocks: 512-thread workgroups
    (44100.0, 96000.0, 8192, 5000, 40000, 1.0, 180.15, "fft=4096/8192"),   # fused, two phases per thread
    (96000.0, 44100.0, 8192, 8192, 50000, 1.0, 180.15, "fft=8192/8192"),   # fused, one phase per thread
    (88200.0, 44100.0, 8192, 3000, 50000, 1.0, 180.15, "fft=8192/4096"),   # decimating
    (64000.0, 48000.0, 4096, 4096, 40000, 2.0, 180.15, "fft=8192/2048"),   # 4x decimating: one barrier more
]


# Filters too long for the reference's own block to fit LDS (32768-point blocks: transition band 0.5 %
# at 180 dB with a radix-3 factor): the engine runs the same filter on 16384-point blocks.  Exact for
# plain and strided decimation; where the reference decimates by TRUNCATING the block spectrum (2^k
# down factors, reference CDSPBlockConvolver.h:329-344) the truncation residue depends on the block
# length, so those two ratios agree to the filter's own stop-band level (-219 dB)
# instead of 1e-15: tolerance stated per case.
# (src, dst, maxin, chunk, n_in, tb, atten, rms_tol, peak_tol)
REBLOCK_CASES = [
    (48000.0, 32000.0, 4096, 3000, 90000, 0.5, 180.15, RMS_TOL, PEAK_TOL),    # 2/3
    (44100.0, 132300.0, 2048, 1000, 50000, 0.5, 180.15, RMS_TOL, PEAK_TOL),   # 3/1
    (96000.0, 32000.0, 4096, 4096, 120000, 0.5, 180.15, RMS_TOL, PEAK_TOL),   # 1/3
    (32000.0, 48000.0, 2048, 2048, 60000, 0.5, 180.15, RMS_TOL, PEAK_TOL),    # 3/2 (truncated spectrum: the reference's own 32768-point block)
    (64000.0, 48000.0, 2048, 777, 70000, 0.5, 180.15, RMS_TOL, PEAK_TOL),     # 3/4 (truncated spectrum: ...)
]


# 8192 -> 16384-point blocks (2x up-sampling filters with a transition band of 0.5 ... 0.6 %, among them the re-blocked
# 2/3 filters above): the split 2x up-sampling form of the pair kernel (r8b_convp.h cp_sp_*, modes 8 / 9), and the
# one-channel kernel behind option pair_split = 0.  (src, dst, maxin, chunk, n_in, tb, atten)
SPLIT_CASES = [
    (44100.0, 88200.0, 4096, 3000, 70000, 0.5, 180.15),       # convolver alone, ragged calls
    (44100.0, 96000.0, 4096, 4096, 60000, 0.5, 180.15),       # ... in front of the whole-step interpolator
    (44100.0, 176400.0, 2048, 2048, 50000, 0.55, 206.91),     # ... in front of a half-band up-sampler
    (96000.0, 64000.0, 8192, 5000, 90000, 0.5, 180.15),       # strided store (mode 9)
    (44100.0, 44101.0, 4096, 1500, 60000, 0.5, 180.15),       # ... in front of the polynomial interpolator
]


# 16384-point blocks (1:1 and 2x decimating filters with a transition band of 0.5 ... 0.6 %, the re-blocked 1/3, 3/1, 3/2 and
# 3/4 filters): the one-channel form of the pair kernel (r8b_convp.h cp_solo_*, modes 10 / 11), and the one-channel kernel
# k_convx behind option pair_solo = 0.  (src, dst, maxin, chunk, n_in, tb, atten[, rms_tol, peak_tol])
SOLO_CASES = [
    (96000.0, 44100.0, 8192, 5000, 100000, 0.5, 180.15),      # 1:1 in front of the whole-step interpolator, ragged calls
    (48000.0, 16000.0, 8192, 8192, 110000, 0.5, 180.15),      # strided store (mode 11)
    (44100.0, 132300.0, 2048, 2048, 40000, 0.5, 180.15),      # 3x zero stuffing load (mode 11)
    (88200.0, 44100.0, 8192, 3000, 90000, 0.5, 180.15),       # decimating by 2 in the spectrum
    (176400.0, 44100.0, 16384, 16384, 200000, 0.55, 206.91),  # ... behind a half-band decimator (input from a ring)
    (32000.0, 48000.0, 2048, 2048, 60000, 0.6, 180.15),       # 3x zero stuffing + decimating (at 0.5 % the reference's block is 32768 points: REBLOCK_CASES)
    (64000.0, 48000.0, 2048, 777, 70000, 0.7, 180.15),        # ... decimating by 4 (32768 points up to 0.6 %)
    (192000.0, 44100.0, 8192, 8192, 150000, 0.5, 180.15),     # half-band decimator + 16384 points 1:1 + interpolator
]


# Minimum-phase chains (reference fprMinPhase).  Per-call counts must equal the reference's exactly.
# Samples: the reference derives the filter by a cepstral transform whose result in the deep stop band
# is set by the rounding noise of ITS fp64 FFT (CDSPRealFFT.h:681-785): two correct evaluations of the
# same transform agree on the taps to ~1e-8 at 136 dB, ~1e-5 at 180 dB and ~2e-3 for the 1/3-band
# filter at 180 dB (the reference's comes out 0.011 samples later), and the streams inherit that.
# The tests bound the deviation by the reference's OWN noise, measured in the same run (test_emul.run_minphase_case: the
# reference over its other FFT back-end against the default build, times 3); the tolerances below -- 3x what was
# measured against oracle/_ref -- are only the fallback for hosts where that build cannot run (no AVX).
# (src, dst, maxin, chunk, n_in, tb, atten, rms_tol, peak_tol)
MINPHASE_CASES = [
    (44100.0, 88200.0, 2048, 2048, 20000, 2.0, 180.15, 6e-5, 3e-4),     # convolver alone
    (44100.0, 96000.0, 2048, 1000, 20000, 2.0, 180.15, 6e-5, 3e-4),     # + whole-step interpolator (InitFracPosW)
    (96000.0, 44100.0, 2048, 2048, 30000, 2.0, 180.15, 6e-6, 3e-5),     # 1:1 convolver + interpolator
    (44100.0, 44101.0, 1024, 1000, 9000, 2.0, 180.15, 2e-5, 8e-5),      # polynomial bank (InitFracPos)
    (44100.0, 176400.0, 1024, 1024, 8000, 2.0, 180.15, 6e-5, 3e-4),     # half-band up with inherited latency
    (44100.0, 2822400.0, 512, 512, 3000, 2.0, 180.15, 6e-5, 3e-4),      # five of them
    (176400.0, 44100.0, 4096, 3000, 40000, 2.0, 180.15, 3e-5, 1.2e-4),  # decimators + 2x-decimating convolver
    (88200.0, 44100.0, 2048, 2048, 20000, 2.0, 180.15, 4e-5, 2e-4),     # 2x-decimating convolver
    (44100.0, 192000.0, 1024, 1024, 8000, 2.0, 180.15, 6e-5, 3e-4),     # interpolator -> convolver -> half-band
    (64000.0, 48000.0, 2048, 2048, 30000, 2.0, 180.15, 5e-5, 2.5e-4),   # 3/4
    (48000.0, 32000.0, 2048, 2048, 30000, 2.0, 180.15, 1e-2, 4e-2),     # 2/3: the 1/3-band filter (see above)
    (44100.0, 96000.0, 1024, 777, 12000, 2.0, 136.45, 3e-7, 2e-6),      # 16-bit preset: taps agree to 2e-8
]


def make_input(nch, n, seed0=1):
    return np.stack([O.splitmix_uniform(seed0 + c, n) for c in range(nch)])


def compare_stream(batch, src, dst, maxin, chunk, n, tb, att, nch, x=None):
    """Feeds x (nch x n) through `batch` (anything with process_host) and through one oracle per
    channel, call by call; asserts equal per-call counts and returns the (rms, peak) difference
    over the whole stream (worst channel)."""
    x = make_input(nch, n) if x is None else x
    oracles = [O.OracleResampler(src, dst, maxin, tb, att) for _ in range(nch)]
    sq = np.zeros(nch)
    cnt = 0
    worst_peak = 0.0
    for i in range(0, n, chunk):
        y = batch.process_host(x[:, i:i + chunk])
        for c in range(nch):
            yo = oracles[c].process(x[c, i:i + chunk]) if src != dst else x[c, i:i + chunk]
            assert len(yo) == y.shape[1], (i, len(yo), y.shape)
            if len(yo):
                d = y[c] - yo
                sq[c] += float(np.sum(d * d))
                worst_peak = max(worst_peak, float(np.abs(d).max()))
        cnt += y.shape[1]
    return (float(np.sqrt(sq.max() / cnt)) if cnt else 0.0), worst_peak


# Partner scales in the pair kernel (r8b_convp.h: channels 2c and 2c+1 share one complex transform).  Each channel
# picks up rounding residue of the order of 1e-16 of its PARTNER's amplitude, so the error bound of a channel is
# relative to the louder of the two; a channel of exact zeros must come out as exact zeros whatever its partner
# carries (the reference keeps one object per channel, README.md:53-55).
# (src, dst, maxin, chunk, n_in, tb, atten): fused 2x-up + interpolator, fused 1:1 + interpolator (In > Out),
# convolver alone, 2x-decimating form, 3x zero-stuffing load, several blocks per workgroup, half-band cascade behind
PAIR_SCALE_CASES = [
    (44100.0, 96000.0, 4096, 4096, 4096 * 3, 2.0, 180.15),
    (96000.0, 44100.0, 4096, 1500, 4096 * 3, 2.0, 180.15),
    (44100.0, 88200.0, 2048, 2048, 2048 * 4, 2.0, 180.15),
    (88200.0, 44100.0, 4096, 3000, 4096 * 3, 2.0, 180.15),
    (44100.0, 132300.0, 2048, 2048, 2048 * 4, 10.0, 109.56),
    (44100.0, 96000.0, 4096, 1000, 4096 * 3, 10.0, 109.56),
    (44100.0, 705600.0, 512, 512, 2048, 5.0, 109.56),
    (96000.0, 48000.0, 4096, 4096, 4096 * 10, 0.5, 180.15),     # one-channel form (16384 points), decimating
    (48000.0, 16000.0, 8192, 5000, 8192 * 6, 0.5, 180.15),      # ... 1:1 with the strided store
    (44100.0, 88200.0, 4096, 4096, 4096 * 8, 0.5, 180.15),      # split 2x up-sampling form (8192 -> 16384 points)
    (48000.0, 32000.0, 4096, 3000, 4096 * 8, 0.5, 180.15),      # ... with the strided store
]


# full scale of the channels of pair_scale_input() (0: silent)
PAIR_SCALES = [1.0, 1e-6, 0.0, 1.0, 0.0, 0.0, 1e-12, 1.0, 1.0]


def pair_scale_input(n, seed0=11):
    """eight channels = four pairs: (full scale, 1e-6 of full scale), (silence, full scale), (silence, silence),
    (1e-12 of full scale, full scale); a ninth, unpaired channel that is silent for the first third of the stream and
    full scale after it"""
    x = np.zeros((9, n))
    x[0] = O.splitmix_uniform(seed0, n)
    x[1] = 1e-6 * O.splitmix_uniform(seed0 + 1, n)
    x[3] = O.splitmix_uniform(seed0 + 3, n)
    x[6] = 1e-12 * O.splitmix_uniform(seed0 + 7, n)
    x[7] = O.splitmix_uniform(seed0 + 8, n)
    x[8, n // 3:] = O.splitmix_uniform(seed0 + 6, n)[n // 3:]
    return x


def check_pair_scales(batch, case):
    """runs pair_scale_input() through `batch` (9 channels) and one oracle per channel; asserts exact zeros for the
    silent channels and, for every other channel, the tolerance relative to THAT CHANNEL'S OWN level -- RMS <= 1e-15 and
    peak <= 1e-13 of its full scale, whatever its partner in the pair kernel's complex transform carries (round 6: the
    partners are brought to one binary order of magnitude per block, r8b_convp.h cp_level_bits; until round 5 the bound
    was relative to the LOUDER partner, i.e. 1e-9 of its own level for a channel at 1e-6)"""
    src, dst, maxin, chunk, n, tb, att = case
    x = pair_scale_input(n)
    nch = x.shape[0]
    oracles = [O.OracleResampler(src, dst, maxin, tb, att) for _ in range(nch)]
    ys, yos = [], [[] for _ in range(nch)]
    for i in range(0, n, chunk):
        ys.append(batch.process_host(x[:, i:i + chunk]))
        for c in range(nch):
            yos[c].append(oracles[c].process(x[c, i:i + chunk]))
    y = np.concatenate(ys, axis=1)
    yo = np.stack([np.concatenate(v) for v in yos])
    assert y.shape == yo.shape and y.shape[1] > 0
    for c in (2, 4, 5):
        assert not y[c].any(), (c, float(np.abs(y[c]).max()))         # silence in, exact zeros out
        assert not yo[c].any()
    d = y - yo
    rms = np.sqrt((d * d).mean(axis=1))
    pk = np.abs(d).max(axis=1)
    rel_rms, rel_pk = [], []
    for c, sc in enumerate(PAIR_SCALES):
        if sc > 0.0:
            rel_rms.append(rms[c] / sc)
            rel_pk.append(pk[c] / sc)
            assert rms[c] <= RMS_TOL * sc and pk[c] <= PEAK_TOL * sc, (c, sc, rms[c] / sc, pk[c] / sc)
    # the unpaired channel must leave silence exactly when its samples arrive
    # (a block that holds the first non-zero sample is transformed as a whole, here as in the reference: only blocks
    # that end before it are exact -- the first eighth of the stream is well clear of it)
    assert not y[8, :int(n // 8 * dst / src)].any() and not yo[8, :int(n // 8 * dst / src)].any()
    return float(max(rel_rms)), float(max(rel_pk))


# Every block once (Engine::launch_fused / launch_stage, ConvxLaunch::park_*): the block that holds a call's last output
# is computed once; what it holds of the next call(s) waits in a park buffer (end of the chain) or goes ahead into the
# next stage's ring.  (src, dst, maxin, tb, atten, kind[, engine options]): "park" / "ahead" / "none" = what the chain's
# convolver kernels do
PARK_CASES = [
    (44100.0, 96000.0, 8192, 2.0, 180.15, "park"),        # cfg2: fused, two phases per thread
    (96000.0, 44100.0, 16384, 2.0, 180.15, "park"),       # cfg3
    (44100.0, 96000.0, 4096, 2.0, 109.56, "park"),        # 16IR preset: two blocks per workgroup
    (48000.0, 44100.0, 6000, 2.0, 109.56, "park"),
    (44100.0, 48000.0, 5000, 2.0, 180.15, "park"),
    (44100.0, 88200.0, 6000, 2.0, 180.15, "park"),        # convolver alone (mode 0), 2x up
    (88200.0, 44100.0, 12000, 2.0, 180.15, "park"),       # 2x decimating form
    (48000.0, 32000.0, 16384, 2.0, 180.15, "park"),       # 8192-point blocks, strided store (mode 3)
    (44100.0, 132300.0, 3000, 10.0, 109.56, "park"),      # 3x zero stuffing, several blocks per workgroup
    (176400.0, 44100.0, 9000, 2.0, 180.15, "park"),       # half-band decimator in front (input from a ring)
    (44100.0, 2822400.0, 1024, 2.0, 180.15, "ahead"),     # cfg5: convolver -> half-band cascade (ahead into its ring)
    (44100.0, 44101.0, 4096, 2.0, 180.15, "ahead"),       # convolver -> polynomial interpolator
    (44100.0, 192000.0, 2048, 2.0, 180.15, "ahead"),      # fused pair -> ring -> convolver -> half-band
    (44100.0, 96000.0, 4096, 10.0, 109.56, "none"),       # 512-point blocks, one phase per thread: as before
    # 8192 -> 16384-point blocks: the split 2x up-sampling form of the pair kernel (modes 8 / 9)
    (48000.0, 32000.0, 4096, 0.5, 180.15, "park"),        # re-blocked 8 507-tap filter, strided store
    (44100.0, 88200.0, 2048, 0.5, 180.15, "park"),
    (44100.0, 96000.0, 2048, 0.5, 180.15, "ahead"),       # ... in front of the (unfused) interpolator
    # 16384-point blocks: the one-channel form of the pair kernel (modes 10 / 11)
    (48000.0, 16000.0, 8192, 0.5, 180.15, "park"),        # re-blocked 8 507-tap filter, 16384 points 1:1, strided store
    (176400.0, 44100.0, 16384, 0.5, 180.15, "park"),      # half-band decimator + 16384 -> 8192 points (decimating form)
    (96000.0, 44100.0, 8192, 0.5, 180.15, "park"),        # 16384 points 1:1 with the interpolator fused in (mode 18, round 5)
    (96000.0, 44100.0, 8192, 0.5, 180.15, "ahead", {"solo_fuse": 0}),   # ... in front of the unfused interpolator
    (44100.0, 132300.0, 3000, 0.5, 180.15, "park"),       # 3x zero stuffing into 16384 points
    (48000.0, 36000.0, 6000, 0.7, 180.15, "park"),        # 3x zero stuffing into 16384 points, decimated by 4
    # the one-channel KERNEL (what is left for it: option pair_solo = 0): at the end of a chain through an output ring of
    # its own and a copy
    (48000.0, 36000.0, 6000, 0.7, 180.15, "ahead", {"pair_solo": 0}),
    (48000.0, 16000.0, 8192, 0.5, 180.15, "ahead", {"pair_solo": 0}),
    (96000.0, 44100.0, 8192, 0.5, 180.15, "ahead", {"pair_solo": 0}),   # fused with the interpolator (output ring)
    (192000.0, 44100.0, 8192, 0.5, 180.15, "park"),       # half-band decimator + fused 16384 -> 16384 points (mode 18)
]
# Minimum-phase filters on the long blocks (transition band 0.5 ... 1 %): 8192 -> 16384 points on the split form, 16384
# points 1:1 and decimating by 2 on the one-channel form, with a complex kernel spectrum (modes 12 ... 15) -- until round 4
# these chains were refused (2x up-sampling, decimating) or ran on the generic kernel in place (1:1).  (Decimating
# chains whose latency is odd carry the reference's InputDelay exception -- test_emul.run_minphase_reference_taps, the
# 64000 -> 48000 case -- on every kernel path; the cases here have even latencies.)  Run on the REFERENCE's own taps
# (test_emul.run_minphase_reference_taps: RMS <= 1e-15 / peak <= 1e-13).  (src, dst, maxin, chunk, n_in, tb, atten)
MINPHASE_LONG_CASES = [
    (44100.0, 88200.0, 4096, 3000, 60000, 0.5, 180.15),      # split form
    (48000.0, 32000.0, 4096, 4096, 90000, 0.5, 180.15),      # ... with the strided store (the 1/3-band filter)
    (44100.0, 96000.0, 4096, 4096, 60000, 0.5, 180.15),      # ... in front of the interpolator (fractional start)
    (96000.0, 44100.0, 8192, 5000, 90000, 0.5, 180.15),      # one-channel form 1:1 + interpolator
    (48000.0, 16000.0, 8192, 8192, 100000, 1.0, 180.15),     # ... strided store
    (44100.0, 132300.0, 2048, 2048, 40000, 1.0, 180.15),     # ... 3x zero stuffing load
    (88200.0, 44100.0, 8192, 3000, 90000, 0.5, 180.15),      # ... decimating by 2 in the spectrum (Nyquist fix-up Re(H X))
    (176400.0, 44100.0, 16384, 16384, 200000, 0.5, 180.15),  # ... behind a half-band decimator with inherited latency
]


PARK_CASES_MINPHASE = [
    (44100.0, 88200.0, 4096, 2.0, 180.15, "park"),        # complex kernel spectrum (mode 6)
    (48000.0, 32000.0, 8192, 2.0, 180.15, "park"),        # mode 7
    (44100.0, 96000.0, 4096, 2.0, 180.15, "park"),        # fused with the interpolator, fractional start (mode 16)
    (96000.0, 44100.0, 4096, 2.0, 180.15, "park"),        # ... In > Out (mode 17)
    (44100.0, 96000.0, 4096, 2.0, 180.15, "ahead", {"fuse_latency": 0}),   # convolver -> k_whole with a fractional start
    (44100.0, 88200.0, 2048, 0.5, 180.15, "park"),        # split form, complex spectrum (mode 12)
    (48000.0, 16000.0, 8192, 1.0, 180.15, "park"),        # one-channel form, complex spectrum, strided store (mode 15)
]


def check_parked_outputs(make, case):
    """`make(park)` builds a 3-channel batch object (odd: a block pair without a partner) with option park = 1 / 0.
    Calls of every length -- MaxInLen, a third, a few samples, one sample (served from the park buffer alone) -- give
    the same outputs BIT FOR BIT with every block computed once and with the call's last block computed again by the
    next call, across clear(); the counters say which of the two happened."""
    src, dst, maxin, tb, att, kind = case[:6]
    lens = [maxin, maxin, maxin // 3, 300 % maxin + 1, 1, 1, 2, maxin, 17, 1, maxin - 5, 2500 % maxin + 1, maxin, 40,
            maxin]
    a, b = make(1), make(0)
    rng = np.random.default_rng(7)
    for rep in range(2):
        for i, l in enumerate(lens):
            x = rng.uniform(-1.0, 1.0, (3, l))
            ya, yb = a.process_host(x), b.process_host(x)
            assert ya.shape == yb.shape and np.array_equal(ya, yb), (rep, i, l)
        if rep == 0:
            na, nb = a.stat("conv_blocks"), b.stat("conv_blocks")
            a.clear()
            b.clear()
    parked = a.stat("park_calls")
    assert b.stat("park_calls") == 0
    if kind == "none":
        assert na == nb and parked == 0, (na, nb, parked)
    elif kind == "ahead":
        assert na < nb and parked == 0, (na, nb, parked)
    else:
        # without parking nearly every call with work computes one block twice
        assert na < nb and parked > 0 and a.stat("park_only_calls") > 0, (na, nb, parked)
    return parked, na, nb
