"""GPU tier (-m gpu): the HIP path, called through the C ABI of libr8bsrc_hip.so, against
  * the oracle (oracle/r8b_oracle.py, pinned to the real reference) on identical seeded input,
  * the committed golden streams generated from the real reference (tests/golden/),
  * the real reference itself (oracle/_ref) when the prebuilt libraries travelled with the repo,
and, at BASELINE.json's full batch sizes, through size-independent properties.
Tolerance: RMS <= 1e-15, peak <= 1e-13 vs the oracle; bitwise where stated.
"""
import importlib
import os

import numpy as np
import pytest

import r8b_oracle as O
import test_emul
from cases import (STREAM_CASES, SHORT_CASES, REBLOCK_CASES, SPLIT_CASES, SOLO_CASES, MINPHASE_CASES, MINPHASE_LONG_CASES, PAIR_SCALE_CASES, PARK_CASES,
                   PARK_CASES_MINPHASE, RMS_TOL, PEAK_TOL, compare_stream, make_input, check_pair_scales,
                   check_parked_outputs)
from conftest import rms, peak

pytestmark = pytest.mark.gpu

r8b = importlib.import_module("r8brain-free-src_amd")


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "GPU tier needs a GPU"
    assert os.path.exists(r8b.lib_path()), "libr8bsrc_hip.so missing: no CPU fallback exists"
    return t


@pytest.mark.parametrize("case", STREAM_CASES)
def test_hip_matches_oracle(torch, case):
    src, dst, maxin, chunk, n, tb, att = case
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=3, device=0)
    r, p = compare_stream(b, src, dst, maxin, chunk, n, tb, att, 3)
    assert r <= RMS_TOL and p <= PEAK_TOL, (r, p)


GOLDEN_NAMES = ["cfg2_44k_96k", "cfg3_96k_44k", "cfg5_44k_2822k", "hbdown_176k_44k",
                "sacd_down_2822k_176k", "poly_44100_44101", "up3_44k_132k", "down3_48k_32k",
                "ratio32_32k_48k", "interm_44k_192k", "res16_44k_48k", "res16ir_48k_44k",
                "impulse_44k_96k"]


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_hip_matches_golden_stream(torch, golden_streams, name):
    """fixtures are outputs of the compiled reference (tests/golden/make_golden.py)"""
    src, dst, maxin, chunk, n, tb, att, seed = golden_streams[name + "/params"]
    n, chunk, seed = int(n), int(chunk), int(seed)
    if seed == 0:
        x = np.zeros(n)
        x[0] = 1.0
    else:
        x = O.splitmix_uniform(seed, n)
    b = r8b.BatchResampler(src, dst, int(maxin), tb, att, nch=1, device=0)
    outs = [b.process_host(x[None, i:i + chunk])[0] for i in range(0, n, chunk)]
    assert [len(o) for o in outs] == list(golden_streams[name + "/counts"])
    g = golden_streams[name + "/y"]
    y = np.concatenate(outs)[:len(g)]
    assert rms(y - g) <= RMS_TOL and peak(y - g) <= PEAK_TOL, (rms(y - g), peak(y - g))


STAGES = [
    ("conv", (0, 0.5, 2.0, 180.15, 2.0, 2, 1), 1024, 7000),
    ("conv", (0, 0.459375, 2.0, 180.15, 1.0, 1, 1), 1000, 9000),
    ("conv", (0, 0.5, 2.0, 180.15, 0.5, 1, 2), 999, 12000),
    ("conv", (0, 1.0 / 3.0, 2.0, 136.45, 3.0, 3, 1), 512, 6000),
    ("conv", (0, 1.0 / 3.0, 2.0, 136.45, 2.0, 2, 3), 512, 9000),
    ("frac", (1, 88200.0, 96000.0, 180.15, 0.0, 0, 0), 1024, 6000),
    ("frac", (1, 96000.0, 44100.0, 180.15, 0.0, 0, 0), 777, 6000),
    ("frac", (1, 44100.0, 44101.0, 180.15, 0.0, 0, 0), 300, 4000),
    ("hbup", (2, 180.15, 0.0, 0.0, 0.0, 0, 0), 512, 3000),
    ("hbup", (2, 180.15, 0.0, 0.0, 0.0, 4, 0), 512, 3000),
    ("hbdown", (3, 180.15, 0.0, 0.0, 0.0, 0, 0), 1000, 6000),
    ("hbdown", (3, 136.45, 0.0, 0.0, 0.0, 1, 1), 1001, 6000),
]


@pytest.mark.parametrize("stage", STAGES)
def test_hip_single_stage_matches_reference_stage(torch, refwrap, stage):
    """the reference's CDSPProcessor boundary: one stage object vs one stage kernel"""
    kind, desc, chunk, n = stage
    _, a, b_, c, d, i0, i1 = desc
    x = O.splitmix_uniform(21, n)
    rs = refwrap.RefStage(kind, a, b_, c, d, i0, i1)
    hb = r8b.BatchResampler(0, 0, 1024, nch=2, device=0, stage=desc)
    for i in range(0, n, chunk):
        yr = rs.process(x[i:i + chunk])
        y = hb.process_host(np.stack([x[i:i + chunk]] * 2))
        assert y.shape[1] == len(yr)
        if len(yr):
            # (the pair kernel carries channel 0 in the real and channel 1 in the imaginary part of one
            # complex transform: equal inputs give equal outputs to rounding, not bit for bit)
            assert peak(y[0] - y[1]) <= 4e-15
            for c in range(2):
                assert rms(y[c] - yr) <= RMS_TOL and peak(y[c] - yr) <= PEAK_TOL


def test_hip_chunk_invariance_is_bitwise(torch):
    x = make_input(1, 9000, 11)
    ref = None
    for chunk in (1024, 1000, 777, 64):
        b = r8b.BatchResampler(44100.0, 96000.0, 1024, 2.0, 180.15, nch=1, device=0)
        y = np.concatenate([b.process_host(x[:, i:i + chunk])[0] for i in range(0, 9000, chunk)])
        if ref is None:
            ref = y
        else:
            assert np.array_equal(y, ref)


def test_hip_clear_restores_state(torch):
    x = make_input(2, 5000, 5)
    b = r8b.BatchResampler(96000.0, 44100.0, 1024, 2.0, 180.15, nch=2, device=0)
    y1 = np.concatenate([b.process_host(x[:, i:i + 1000]) for i in range(0, 5000, 1000)], axis=1)
    b.clear()
    y2 = np.concatenate([b.process_host(x[:, i:i + 1000]) for i in range(0, 5000, 1000)], axis=1)
    assert np.array_equal(y1, y2)


def test_hip_dll_abi_matches_reference_dll(torch, refwrap):
    """the five drop-in symbols, called exactly like the reference's DLL, on both libraries"""
    import ctypes as C
    ref = refwrap.dll()
    x = O.splitmix_uniform(9, 8192)
    for res, (src, dst) in [(2, (44100.0, 96000.0)), (0, (48000.0, 44100.0)), (1, (44100.0, 48000.0))]:
        mine = r8b.DLLResampler(src, dst, 1024, 2.0, res)
        h = ref.r8b_create(src, dst, 1024, 2.0, res)
        for k in (1, 2, 100, 5000):
            assert mine.inlen(k) == ref.r8b_inlen(h, k)
        for i in range(0, 8192, 1024):
            xin = np.ascontiguousarray(x[i:i + 1024])
            op = C.POINTER(C.c_double)()
            n = ref.r8b_process(h, xin.ctypes.data_as(C.POINTER(C.c_double)), 1024, C.byref(op))
            a = np.ctypeslib.as_array(op, shape=(n,)).copy() if n else np.zeros(0)
            b = mine.process(xin)
            assert len(a) == len(b)
            if n:
                assert rms(a - b) <= RMS_TOL and peak(a - b) <= PEAK_TOL
        ref.r8b_delete(h)


def test_hip_frontend_mirror(torch):
    """CDSPResampler24 mirror: oneshot(), getInLenBeforeOutStart() vs getInputRequiredForOutput()
    (the consistency check of the reference's bench/zerotest.cpp:115-116,166)"""
    r = r8b.CDSPResampler24(44100.0, 48000.0, 512)
    o = O.OracleResampler(44100.0, 48000.0, 512, 2.0, 180.15)
    assert r.getInputRequiredForOutput(1) == o.input_required(1)
    assert r.getInLenBeforeOutStart(0) == r.getInLenBeforeOutPos(0)
    x = O.splitmix_uniform(3, 3000)
    y = r.oneshot(x, 3000)
    yo = np.concatenate([o.process(x[i:i + 512]) for i in range(0, 3000, 512)] +
                        [o.process(np.zeros(512)) for _ in range(8)])[:3000]
    assert len(y) == 3000 and rms(y - yo) <= RMS_TOL and peak(y - yo) <= PEAK_TOL


def test_hip_device_tensor_entry_and_strides(torch):
    """device-pointer entry: padded row strides, offset views, guard regions stay untouched"""
    nch, L = 5, 2048
    b = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=nch, device=0)
    x = make_input(nch, L * 3, 31)
    oracles = [O.OracleResampler(44100.0, 96000.0, L, 2.0, 180.15) for _ in range(nch)]
    xin = torch.full((nch, L + 37), float("nan"), dtype=torch.float64, device="cuda:0")
    cap = b.max_out_len + 64
    out = torch.full((nch, cap), -7.0, dtype=torch.float64, device="cuda:0")
    for i in range(3):
        xin[:, 5:5 + L] = torch.from_numpy(x[:, i * L:(i + 1) * L]).cuda()
        y = b.process(xin[:, 5:5 + L], out=out)
        torch.cuda.synchronize()
        yh = y.cpu().numpy()
        n = yh.shape[1]
        for c in range(nch):
            yo = oracles[c].process(x[c, i * L:(i + 1) * L])
            assert len(yo) == n
            if n:
                assert peak(yh[c] - yo) <= PEAK_TOL
        assert bool((out[:, n:] == -7.0).all())  # nothing written past the returned count


FULL_CFGS = [(44100.0, 96000.0, 1024, 16384), (96000.0, 44100.0, 1024, 16384),
             (44100.0, 2822400.0, 64, 1024)]


@pytest.mark.parametrize("cfg", FULL_CFGS)
def test_hip_full_size_all_channels_vs_reference(torch, refwrap, cfg):
    """BASELINE.json configs 2, 3 and 5 at their full batch sizes: every channel its own
    splitmix64 stream (seed 1 + channel), three calls, EVERY channel compared with the compiled
    reference (one CDSPResampler24 per channel, run multi-threaded by oracle/ref_shim.cpp) --
    like the reference's bench/r8bfreesrc.cpp:118-126 loop, but checking the samples."""
    src, dst, nch, L = cfg
    calls = 3
    x = make_input(nch, L * calls, 1)
    b = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=nch, device=0)
    xd = torch.from_numpy(x).cuda()
    outs, counts = [], []
    for i in range(calls):
        y = b.process(xd[:, i * L:(i + 1) * L].contiguous())
        counts.append(y.shape[1])
        outs.append(y.clone())
    y = torch.cat(outs, dim=1).cpu().numpy()
    assert sum(counts) > 0
    r, p = refwrap.batch_check(src, dst, L, [L] * calls, x, y, counts)
    assert r.max() <= RMS_TOL and p.max() <= PEAK_TOL, (r.max(), p.max(), int(r.argmax()))


@pytest.mark.parametrize("cfg", FULL_CFGS)
def test_hip_full_size_properties(torch, cfg):
    """size-independent properties at the BASELINE.json batch sizes:
      * linearity: R(a*x + b*z) == a*R(x) + b*R(z) to rounding;
      * a constant stream resamples to the same constant after the transient (DC gain 1 up to the
        filters' own imaging residue: <= 3e-13 through the convolver+interpolator, ~7e-10 through
        the five half-band stages -- the oracle shows the same numbers);
      * a channel does not depend on what its neighbours carry beyond rounding: the pair kernel packs
        channels 2c and 2c+1 into one complex transform, so a loud neighbour leaves a residue of the
        order of 1e-16 of ITS amplitude in the quiet channel (measured and bounded here)."""
    src, dst, nch, L = cfg
    calls = 3
    b = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=nch, device=0)
    x = make_input(nch, L * calls, 101)
    xd = torch.from_numpy(x).cuda()
    y = torch.cat([b.process(xd[:, i * L:(i + 1) * L].contiguous()).clone() for i in range(calls)],
                  dim=1).cpu().numpy()
    assert y.shape[1] > 0
    # linearity on two channels
    b2 = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=1, device=0)
    mix = 0.75 * x[0] - 0.5 * x[1]
    ym = np.concatenate([b2.process_host(mix[None, i * L:(i + 1) * L])[0] for i in range(calls)])
    yl = 0.75 * y[0] - 0.5 * y[1]
    assert peak(ym - yl) <= 1e-13
    # DC gain
    b2.clear()
    yc = np.concatenate([b2.process_host(np.full((1, L), 0.5))[0] for _ in range(calls + 2)])
    tail = yc[len(yc) // 2:]
    assert len(tail) > 100 and peak(tail - 0.5) <= (5e-9 if dst > 1e6 else 2e-12)
    # neighbour independence: channel 0 next to silence vs next to full-scale noise
    b3 = r8b.BatchResampler(src, dst, L, 2.0, 180.15, nch=2, device=0)
    quiet = np.stack([x[0], np.zeros_like(x[0])])
    loud = np.stack([x[0], x[1]])
    yq = np.concatenate([b3.process_host(quiet[:, i * L:(i + 1) * L]) for i in range(calls)], axis=1)
    b3.clear()
    yn = np.concatenate([b3.process_host(loud[:, i * L:(i + 1) * L]) for i in range(calls)], axis=1)
    assert peak(yq[0] - yn[0]) <= 4e-15
    assert not yq[1].any()  # a silent channel beside a loud one: exact zeros (silence detection per block pair)


@pytest.mark.parametrize("case", PAIR_SCALE_CASES)
def test_hip_pair_partner_scales_and_silence(torch, case):
    """channel scales 1 : 1e-6, 1e-12 : 1 and 1 : 0 across a pair of the pair kernel -- every channel within RMS 1e-15 /
    peak 1e-13 of ITS OWN level (round 6: partners equalised per block by an exact power of two), exact zeros for silent
    channels whatever the partner carries (cases.check_pair_scales)"""
    src, dst, maxin, chunk, n, tb, att = case
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=9, device=0)
    rel_rms, rel_pk = check_pair_scales(b, case)
    assert rel_rms <= RMS_TOL and rel_pk <= PEAK_TOL


@pytest.mark.parametrize("nch", [5, 16])
@pytest.mark.parametrize("case", SHORT_CASES)
def test_hip_short_filters_in_block_groups(torch, case, nch):
    """shorter filters (16IR preset, 5 ... 20 % transition bands): the pair kernel with 2 ... 16 blocks per
    workgroup; every channel against its oracle, ragged calls, odd channel count"""
    src, dst, maxin, chunk, n, tb, att, frag = case
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=nch, device=0)
    assert frag in b.describe(), b.describe()
    b.set_option("timing", 1)
    assert any(t[0].startswith("k_convp") for t in b.stage_timings()), b.stage_timings()
    b.set_option("timing", 0)
    r, p = compare_stream(b, src, dst, maxin, chunk, n, tb, att, nch)
    assert r <= RMS_TOL and p <= PEAK_TOL, (r, p)


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("case", SPLIT_CASES)
def test_hip_split_upsampling_form(torch, case, split):
    """8192 -> 16384-point blocks: the pair kernel's split 2x up-sampling form (modes 8 / 9) and the one-channel
    kernel behind option pair_split = 0, each against the oracle"""
    from test_emul import run_split_form_case
    run_split_form_case({"device": 0}, case, split, nch=5)


@pytest.mark.parametrize("solo", [1, 0])
@pytest.mark.parametrize("case", SOLO_CASES)
def test_hip_one_channel_form(torch, case, solo):
    """16384-point blocks: the pair kernel's one-channel form (modes 10 / 11, 1:1 and decimating by 2) and the one-channel
    kernel behind option pair_solo = 0, each against the oracle"""
    from test_emul import run_solo_form_case
    run_solo_form_case({"device": 0}, case, solo, nch=5)


@pytest.mark.parametrize("case", REBLOCK_CASES)
def test_hip_long_filters_on_shorter_blocks(torch, refwrap, case):
    """transition band 0.5 % at 180 dB with a radix-3 factor (the reference's block is 32768 points):
    r8b_batch_create succeeds, counts equal the reference's, samples to the stated tolerance"""
    src, dst, maxin, chunk, n, tb, att, rtol, ptol = case
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=2, device=0)
    b.set_option("timing", 1)
    x = make_input(2, n, 5)
    lens, ys, counts, pos = [], [], [], 0
    while pos < n:
        l = min(chunk, n - pos)
        y = b.process_host(x[:, pos:pos + l])
        lens.append(l)
        counts.append(y.shape[1])
        ys.append(y)
        pos += l
    r, p = refwrap.batch_check(src, dst, maxin, lens, x, np.concatenate(ys, axis=1), counts, tb, att)
    assert r.max() <= rtol and p.max() <= ptol, (r.max(), p.max())
    from test_emul import exact_block_ratio
    assert ([t[0] for t in b.stage_timings()] == ["k_conv"]) == exact_block_ratio(src, dst)
    if exact_block_ratio(src, dst):
        assert b.stage_symbols() == ["k_conv_big"], b.stage_symbols()


@pytest.mark.parametrize("src,dst,maxin,tb", [(32000.0, 48000.0, 5000, 0.5), (64000.0, 48000.0, 7000, 0.6)])
def test_hip_exact_block_chunk_invariance(torch, src, dst, maxin, tb):
    """k_conv_big: ragged calls and checkpoints == MaxInLen calls bit for bit"""
    from test_emul import run_exact_block_chunk_invariance
    run_exact_block_chunk_invariance({"device": 0}, src, dst, maxin, tb)


def test_hip_exact_block_many_items(torch, refwrap):
    """k_conv_big with more (block, channel) items than workgroup slots (512): every channel against the reference"""
    src, dst, maxin, tb, att, nch = 32000.0, 48000.0, 16384, 0.5, 180.15, 300
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=nch, device=0)
    x = make_input(nch, 3 * maxin, 11)
    ys, counts = [], []
    for i in range(3):
        y = b.process_host(x[:, i * maxin:(i + 1) * maxin])
        ys.append(y)
        counts.append(y.shape[1])
    r, p = refwrap.batch_check(src, dst, maxin, [maxin] * 3, x, np.concatenate(ys, axis=1), counts, tb, att)
    assert r.max() <= RMS_TOL and p.max() <= PEAK_TOL, (r.max(), p.max())


@pytest.mark.parametrize("case", MINPHASE_CASES)
def test_hip_minimum_phase_chains(torch, refwrap, case):
    """fprMinPhase on the HIP path (generic kernels): counts equal the reference's, samples to the
    stated tolerance (tests/cases.py)"""
    from test_emul import run_minphase_case
    run_minphase_case({"device": 0}, refwrap, case)


@pytest.mark.parametrize("case", MINPHASE_CASES)
def test_hip_minimum_phase_kernels_on_reference_taps(torch, refwrap, hip_hooks, case):
    """the minimum-phase chains on the real kernels with the REFERENCE's own minimum-phase taps
    (parity-test hook r8b_design_set_lp_provider of the test build, conftest.hip_hooks): pair-kernel modes 6 / 7, k_whole / k_poly / half-band kernels with
    fractional start positions -- RMS <= 1e-15 / peak <= 1e-13 against the reference stream"""
    from test_emul import run_minphase_reference_taps
    run_minphase_reference_taps(hip_hooks, {"device": 0, "lib": hip_hooks}, refwrap, case)


@pytest.mark.parametrize("case", MINPHASE_LONG_CASES)
def test_hip_minimum_phase_long_blocks_on_reference_taps(torch, refwrap, hip_hooks, case):
    """cases.MINPHASE_LONG_CASES on the real kernels: the split and one-channel forms of the pair kernel with a complex
    kernel spectrum (modes 12 ... 15), on the reference's own minimum-phase taps"""
    from test_emul import run_minphase_reference_taps
    run_minphase_reference_taps(hip_hooks, {"device": 0, "lib": hip_hooks}, refwrap, case)


def test_hip_poly_channel_groups(torch):
    from test_emul import run_poly_channel_groups
    run_poly_channel_groups({"device": 0})


def test_hip_minphase_pair_kernel_vs_generic(torch):
    """complex kernel spectrum on the pair kernel against the generic kernel, real GPU"""
    from test_emul import run_minphase_pair_vs_generic, MINPHASE_PAIR_TOPOLOGIES
    for topo in MINPHASE_PAIR_TOPOLOGIES:
        run_minphase_pair_vs_generic({"device": 0}, topo)


@pytest.mark.parametrize("topo", [(44100.0, 96000.0, 2048, 2.0, 180.15, 2400),   # cfg2 topology
                                  (44100.0, 96000.0, 2048, 10.0, 109.56, 1200),  # 8 blocks per workgroup, fused
                                  (88200.0, 44100.0, 2048, 5.0, 109.56, 1200),   # decimating, 4 blocks per workgroup
                                  (48000.0, 32000.0, 2048, 2.0, 180.15, 1200),   # 8192-point blocks, 3x strided store
                                  # the polynomial interpolator's counter is re-based after every call that brings it
                                  # past 1000 outputs (reference CDSPFracInterpolator.h:909-917): thousands of times here
                                  (44100.0, 44101.0, 1024, 2.0, 180.15, 3000),
                                  (44100.0, 2822400.0, 128, 2.0, 180.15, 1200),  # cfg5: convolver + k_hbcascade (5 stages)
                                  (2822400.0, 176400.0, 4096, 2.0, 180.15, 1200),   # k_hbdcascade + decimating convolver
                                  # the long-block forms of the pair kernel: split 2x up-sampling, one-channel 1:1 (+
                                  # k_whole) and decimating, one-channel with the strided store
                                  (44100.0, 88200.0, 4096, 0.5, 180.15, 800), (96000.0, 44100.0, 8192, 0.5, 180.15, 600),
                                  (88200.0, 44100.0, 8192, 0.5, 180.15, 600), (48000.0, 16000.0, 8192, 1.0, 180.15, 600)])
def test_hip_soak_ragged_calls_vs_reference(torch, refwrap, topo):
    """thousands of ragged process() calls of one stream per channel on the real kernels (position
    wrap, ring masks, block schedule and partly filled block groups over a long run), every call's
    count and samples against the compiled reference"""
    src, dst, maxin, tb, att, ncalls = topo
    nch = 3
    rng = np.random.default_rng(2024)
    lens = rng.integers(1, maxin + 1, size=ncalls).astype(np.int32)
    lens[::7] = maxin
    lens[3::11] = 1
    n = int(lens.sum())
    x = make_input(nch, n, 31)
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=nch, device=0)
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((nch, b.max_out_len), dtype=torch.float64, device="cuda")
    ys, counts, pos = [], [], 0
    for l in lens:
        yv = b.process(xd[:, pos:pos + int(l)].contiguous(), out=out)
        counts.append(yv.shape[1])
        if yv.shape[1]:
            ys.append(yv.clone())
        pos += int(l)
    y = torch.cat(ys, dim=1).cpu().numpy()
    r, p = refwrap.batch_check(src, dst, maxin, lens, x, y, counts, tb, att)
    assert r.max() <= RMS_TOL and p.max() <= PEAK_TOL, (r.max(), p.max())


def test_cxx_frontend(torch, tmp_path):
    """include/r8b/CDSPResampler.h + the five DLL symbols from a C++ host, the way the reference's
    example.cpp uses its classes; compared with the oracle"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "cxx_frontend")
    libdir = os.path.dirname(r8b.lib_path())
    subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cxx_frontend.cpp"),
                    "-L" + libdir, "-lr8bsrc_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout.split("\n")
    assert out[0].split()[1] == out[0].split()[2]
    vals = np.array([float.fromhex(l) for l in out if l.startswith(("0x", "-0x"))])
    o = O.OracleResampler(44100.0, 96000.0, 1024, 2.0, 180.15)
    x = O.splitmix_uniform(7, 1024 * 6)
    yo = np.concatenate([o.process(x[i:i + 1024]) for i in range(0, len(x), 1024)])
    assert len(vals) == len(yo) and rms(vals - yo) <= RMS_TOL and peak(vals - yo) <= PEAK_TOL
    # getLatencyFrac through the mirror header: 0.0 for linear phase as in the reference, the reference's value for
    # fprMinPhase to what the designer's group-delay fraction allows (equality on the reference's own taps:
    # test_hip_latency_frac_matches_reference)
    lat = [l.split() for l in out if l.startswith("latfrac")]
    assert len(lat) == 8
    import refwrap as R
    for _, a, b, lin, mp in lat:
        a, b = float(a), float(b)
        assert float.fromhex(lin) == 0.0
        if R.available():
            want = R.RefResampler(a, b, 1024, 2.0, 180.15, phase=1).latency_frac()
            assert R.RefResampler(a, b, 1024, 2.0, 180.15).latency_frac() == 0.0
            assert abs(float.fromhex(mp) - want) < 0.02 * max(1.0, b / a), (a, b, mp, want)


def test_hip_latency_frac_matches_reference(torch, refwrap, hip_hooks):
    """r8b_batch_latency_frac == CDSPResampler::getLatencyFrac of the compiled reference: 0.0 on linear phase, equal to
    1e-9 on ten minimum-phase chains run on the reference's own taps (test build of the HIP library)"""
    from test_emul import check_latency_frac, reference_minphase_taps
    check_latency_frac(hip_hooks, {"lib": hip_hooks, "device": 0}, refwrap, reference_minphase_taps(hip_hooks, refwrap))


@pytest.mark.parametrize("opts", [{"pair_two": 0}, {"fuse": 0}, {"fuse": 0, "fast_conv": 0},
                                  {"fuse_hb": 0}, {"pair_conv": 0}, {"pair_conv": 0, "fuse": 0}])
@pytest.mark.parametrize("case", [STREAM_CASES[0], STREAM_CASES[1], STREAM_CASES[2], STREAM_CASES[4],
                                  STREAM_CASES[6]])
def test_hip_alternative_kernel_paths(torch, case, opts):
    """every kernel path behind the engine options (one-phase pair interpolator, unfused fast
    convolver, generic kernels, unfused half-bands, one-channel fast path) produces the same stream"""
    src, dst, maxin, chunk, n, tb, att = case
    b = r8b.BatchResampler(src, dst, maxin, tb, att, nch=2, device=0)
    for k, v in opts.items():
        b.set_option(k, v)
    r, p = compare_stream(b, src, dst, maxin, chunk, n, tb, att, 2)
    assert r <= RMS_TOL and p <= PEAK_TOL, (r, p)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [(44100.0, 96000.0), (44100.0, 44101.0), (44100.0, 2822400.0),
                                     (176400.0, 44100.0)])
def test_gpu_checkpoint_resume(src, dst):
    """r8b_batch_state_save / _load: a stream resumed from a checkpoint in a fresh object continues
    bit-identically (same procedure as the emulation tier)."""
    from test_emul import checkpoint_roundtrip
    blob = checkpoint_roundtrip(lambda: r8b.BatchResampler(src, dst, 700, 2.0, 136.45, nch=3),
                                src, dst)
    assert blob.size > 64


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 3])
def test_hip_eight_elements_per_thread_form(torch, case):
    """r8b_convq.h (engine option "quad": the 2048 -> 4096-point convolver-only block pair on 512 threads) against the
    256-thread form and the oracle -- the measured E = 8 experiment of round 6 (profiles/r06_experiments.txt) stays correct"""
    from test_emul import QUAD_CASES, run_quad_case
    run_quad_case({"device": 0}, QUAD_CASES[case], nch=37)
    b = r8b.BatchResampler(44100.0, 88200.0, 4096, 2.0, 180.15, nch=4, device=0)
    b.set_option("quad", 1)
    b.set_option("timing", 1)
    b.process_host(make_input(4, 4096, 3))
    assert "k_convq" in b.stage_symbols(), b.stage_symbols()


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(test_emul.HALF_CASES)))
def test_hip_half_array_form_against_the_full_one(torch, case):
    """r8b_convp.h cp_ha_* (kernel mode 21, engine option "half": the 2048 -> 4096-point convolver-only block pair with the
    backward side's exchanges by parts through 32 KB of LDS, four workgroups per CU) against the 64 KB form: the same
    arithmetic, equal to rounding on the device (bit for bit under emulation: tests/test_emul.py), ragged calls"""
    from test_emul import HALF_CASES, run_half_case
    y = run_half_case({"device": 0}, HALF_CASES[case], nch=37, bitwise=False)
    assert y.shape[0] == 37


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(test_emul.HALF_FUSED_CASES)))
def test_hip_half_array_fused_form_against_mode_4(torch, case):
    """kernel modes 23 / 25 (the fused two-phase block pair in the half-array form, option half_fused) against a workgroup
    per block of modes 4 / 5: equal to rounding on the device (bit for bit under emulation), ragged calls"""
    test_emul.run_half_fused_case({"device": 0}, test_emul.HALF_FUSED_CASES[case], nch=37, bitwise=False,
                                  must_run=case not in (4,))


@pytest.mark.gpu
def test_hip_half_array_form_is_chunk_invariant_and_the_default_of_large_objects(torch):
    """an object whose largest call holds 512 workgroups and more (channel pairs x blocks: here 65 x 13) runs the half-array
    forms by default (options half = 1, half_fused = 1), a small one (4 x 13) the full-array kernels; each stays bitwise
    independent of how the stream is cut into calls"""
    nch, n = 130, 16000
    x = make_input(nch, n, 5)
    for dst, sym in ((88200.0, "k_convp<11, 1, 21, 24>"), (96000.0, "k_convp<11, 1, 23, 24>"), (48000.0, "k_convp<11, 1, 25, 24>")):
        outs = []
        for lens in ([n], [7000, 1, 4000, 99, 4900]):
            b = r8b.BatchResampler(44100.0, dst, 16384, 2.0, 180.15, nch=nch, device=0)
            b.set_option("timing", 1)
            ys, pos = [], 0
            for l in lens:
                ys.append(b.process_host(x[:, pos:pos + l]))
                pos += l
            assert b.stage_symbols()[0] == sym, b.stage_symbols()
            outs.append(np.concatenate(ys, axis=1))
        assert np.array_equal(outs[0], outs[1]), dst
    s = r8b.BatchResampler(44100.0, 88200.0, 16384, 2.0, 180.15, nch=8, device=0)
    s.set_option("timing", 1)
    s.process_host(x[:8, :16384])
    assert s.stage_symbols() == ["k_convp<11, 1, 0, 24>"], s.stage_symbols()


@pytest.mark.gpu
def test_bench_contract(tmp_path):
    """bench.py prints ONE JSON line carrying the driver's contract fields plus `roofline` (live
    HIP-event timing of the dominant kernel) -- a short run of the real script."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup",
                          "2", "--no-cpu"], check=True, stdout=subprocess.PIPE, text=True).stdout
    lines = [l for l in out.split("\n") if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    assert d["value"] > 0 and abs(d["value"] - 1024 * 16384 / d["ms_per_step"] / 1e3) / d["value"] < 0.01
    # `value` is the window straight after the warm-up calls, from column 0 of the caller's rows; the settled window and
    # the other output placement are side fields (bench.py --settle / --align-out)
    assert d["config"]["placement"] == "column 0" and "straight after the 2 warm-up calls" in d["value_window"]
    assert d["settle_calls"] == 150
    for side in (d["settled"], d["other_placement"]):
        assert side["value"] > 0 and abs(side["value"] - 1024 * 16384 / side["ms_per_step"] / 1e3) / side["value"] < 0.01
    assert d["other_placement"]["placement"] == "stream-aligned"
    # `kernel` is the device symbol rocprofv3 prints for the dominant stage (profiles/*_kernel_stats.csv are keyed by it),
    # `label` the engine's name for the stage's form; `path_frac` is the whole call in the value window
    assert r["label"] == "k_convp_whole" and r["kernel"] in ("k_convp_walk<11, 1, 4, 24>", "k_convp<11, 1, 4, 24>",
                                                            "k_convp<11, 1, 23, 24>"), r
    assert r["kernel"] in r["kernel_symbols"] and "frac_value_window" not in r
    path_bytes = 8.0 * 1024 * (16384 + d["config"]["out_msamples_per_s"] / d["value"] * 16384)
    assert abs(r["path_frac"] - path_bytes / (d["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 2e-3
    assert r["launch_gap_ms_per_step"] is None or 0.0 <= r["launch_gap_ms_per_step"] < d["settled"]["ms_per_step"]


@pytest.mark.gpu
def test_bench_gpus_1_without_a_launcher_is_a_plain_run():
    """`python bench.py --gpus 1` must not go through the self-spawn path: WORLD_SIZE unset,
    one rank, one line"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--no-cpu", "--settle", "0", "--channels", "64", "--block", "4096"], check=True, env=env,
                         stdout=subprocess.PIPE, text=True).stdout
    d = json.loads([l for l in out.split("\n") if l.strip()][0])
    assert d["n_gpus"] == 1 and d["value"] > 0


@pytest.mark.gpu
def test_gpu_deep_decimation_full_size(torch):
    """sacd.cpp's second pass at a large block (256 channels x 65536 samples, 2822400 -> 176400: a run
    of half-band decimators as one kernel, then the convolver): exact-size device buffers (reads past
    the end of the caller's buffer would fault), channel 0 against the oracle, all channels
    against the unfused kernels."""
    nch, L = 256, 65536
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    a = r8b.BatchResampler(2822400.0, 176400.0, L, 2.0, 180.15, nch=nch)
    a.set_option("fuse_hbd", 1)  # (by default a batch this large keeps the first stages unfused)
    b = r8b.BatchResampler(2822400.0, 176400.0, L, 2.0, 180.15, nch=nch)
    b.set_option("fuse_hb", 0)
    o = O.OracleResampler(2822400.0, 176400.0, L, 2.0, 180.15)
    for _ in range(3):
        x = torch.rand((nch, L), generator=g, dtype=torch.float64, device="cuda") * 2.0 - 1.0
        ya = a.process(x).clone()
        yb = b.process(x).clone()
        torch.cuda.synchronize()
        assert ya.shape == yb.shape
        if ya.shape[1]:
            dd = (ya - yb).abs().max().item()  # same filters, different summation order
            assert dd <= 1e-14, dd
        yo = o.process(x[0].cpu().numpy())
        assert len(yo) == ya.shape[1]
        if len(yo):
            d = ya[0].cpu().numpy() - yo
            assert rms(d) <= RMS_TOL and peak(d) <= PEAK_TOL


@pytest.mark.gpu
def test_cxx_batch_device(tmp_path):
    """the batch C ABI from a plain C++/HIP host program with its own device buffers and stream
    (tests/cxx_batch.cpp): even rows equal the single-stream entry bitwise, odd rows (the imaginary
    halves of the pair kernel's transforms) to rounding, across a checkpoint"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "cxx_batch")
    libdir = os.path.dirname(r8b.lib_path())
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cxx_batch.cpp"),
                    "-L" + libdir, "-lr8bsrc_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.returncode, out.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("walk_len", [0, 2])
@pytest.mark.parametrize("case", test_emul.WALK_CASES)
def test_hip_walk_form_equals_one_block_form(torch, case, walk_len):
    """the GPU twin of test_emulated_walk_form_equals_one_block_form"""
    walked, blocks = test_emul.run_walk_form_case({"device": 0}, case, walk_len)
    if case[6]:
        assert walked >= blocks // 3, (walked, blocks)


@pytest.mark.gpu
@pytest.mark.parametrize("case", test_emul.UP3_CASES)
def test_hip_polyphase_up3_equals_zero_stuffing(torch, refwrap, case):
    """the GPU twin of test_emulated_polyphase_up3_equals_zero_stuffing"""
    test_emul.run_polyphase_up3_case({"device": 0}, refwrap, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", test_emul.SOLO_FUSE_CASES)
def test_hip_one_channel_form_fused_with_the_interpolator(torch, case):
    """the GPU twin of test_emulated_one_channel_form_fused_with_the_interpolator"""
    test_emul.run_solo_fused_case({"device": 0}, case)


@pytest.mark.gpu
def test_hip_walk_form_full_batch(torch):
    """BASELINE's cfg2 batch (1024 channels x 16384) with the engine's own choice (option walk = 1: by batch size) against
    a workgroup per block: every channel bit for bit, calls whose outputs start at even and at odd columns"""
    a = r8b.BatchResampler(44100.0, 96000.0, 16384, 2.0, 180.15, nch=1024, device=0)
    b = r8b.BatchResampler(44100.0, 96000.0, 16384, 2.0, 180.15, nch=1024, device=0)
    a.set_option("walk", 0)
    b.set_option("walk", 1)
    # (objects of this size run the half-array form by default: the walk form is what this test is about)
    a.set_option("half_fused", 0)
    b.set_option("half_fused", 0)
    w0 = b.stat("walk_blocks")
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    for _ in range(4):
        x = torch.rand((1024, 16384), generator=g, dtype=torch.float64, device="cuda") * 2.0 - 1.0
        ya = a.process(x).clone()
        yb = b.process(x).clone()
        assert ya.shape == yb.shape and torch.equal(ya, yb)
    assert b.stat("walk_blocks") - w0 >= 4 * 8


@pytest.mark.gpu
def test_cxx_threads_device(tmp_path):
    """VERDICT r4 weak #2 / reference README.md:52-55 ("thread-safe ... a separate object per stream"): ten host threads
    r8b_create + 50 r8b_process calls each at once (eight ratios, one of them on three threads), then eight threads with a
    batch object, device buffers and a stream each; every stream bit for bit what one thread gives (tests/cxx_threads.cpp)"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "cxx_threads")
    libdir = os.path.dirname(r8b.lib_path())
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-pthread", "-DWITH_HIP",
                    os.path.join(ROOT, "tests", "cxx_threads.cpp"),
                    "-L" + libdir, "-lr8bsrc_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    for _ in range(2):
        out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.returncode, out.stdout[-3000:])


@pytest.mark.parametrize("src,dst,maxin,tb", [(44100.0, 96000.0, 16384, 2.0), (96000.0, 44100.0, 16384, 2.0),
                                              (88200.0, 44100.0, 12000, 2.0), (44100.0, 88200.0, 6000, 2.0),
                                              (48000.0, 32000.0, 16384, 2.0), (44100.0, 88200.0, 9000, 0.5),
                                              (96000.0, 44100.0, 16384, 0.5), (88200.0, 44100.0, 12000, 0.5),
                                              (48000.0, 16000.0, 30000, 0.5)])
def test_hip_history_from_registers_equals_the_copy_kernel(torch, src, dst, maxin, tb):
    """the GPU twin of tests/test_emul.py test_emulated_history_from_registers_equals_the_copy_kernel: the next call's
    history stored from the blocks' registers / copied in slices / copied by a kernel of its own -- same stream bit for
    bit, odd channel count, ragged calls"""
    lens = [maxin, maxin, maxin // 3, 300, maxin, 17, 1, maxin - 5, 2500, maxin]
    a = r8b.BatchResampler(src, dst, maxin, tb, 180.15, nch=5, device=0)
    b = r8b.BatchResampler(src, dst, maxin, tb, 180.15, nch=5, device=0)
    b.set_option("fold_tail", 0)
    rng = np.random.default_rng(5)
    for i, l in enumerate(lens):
        x = rng.uniform(-1.0, 1.0, (5, l))
        ya, yb = a.process_host(x), b.process_host(x)
        assert ya.shape == yb.shape and np.array_equal(ya, yb), (i, l)


def _hbconv_topologies():
    from test_emul import HBCONV_TOPOLOGIES
    return HBCONV_TOPOLOGIES


@pytest.mark.parametrize("src,dst,maxin", _hbconv_topologies())
def test_hip_half_band_front_of_the_convolver(torch, refwrap, src, dst, maxin):
    """kernel mode 20 (k_convp<12, -1, 20, 24>) == k_hbdown + k_convp bit for bit, and the reference's stream"""
    from test_emul import run_hbconv_fused
    run_hbconv_fused({"device": 0}, refwrap, src, dst, maxin, nch=5)


def _hb_carry_topologies():
    from test_emul import HB_CARRY_TOPOLOGIES
    return HB_CARRY_TOPOLOGIES


@pytest.mark.parametrize("src,dst,maxin,tb", _hb_carry_topologies())
def test_hip_history_copy_carried_by_a_half_band_launch(torch, src, dst, maxin, tb):
    """the history copy as extra workgroups of a half-band launch (k_hbdown, k_hbcascade, k_hbdcascade) == k_tail"""
    from test_emul import run_history_copy_carried, run_history_three_ways
    run_history_copy_carried({"device": 0}, src, dst, maxin, tb)
    run_history_three_ways({"device": 0}, src, dst, maxin, tb)


@pytest.mark.parametrize("case", PARK_CASES)
def test_hip_parked_outputs_equal_recomputation(torch, case):
    """every block once -- the call's last block parks what it holds of the next call (r8b_convp.h cp_park_*, the second
    cp_whole2_compute of the last block, cp_store_conv's park view) or writes it ahead into the next stage's ring -- ==
    the block computed again by the next call, bit for bit, on the real kernels (cases.check_parked_outputs)"""
    from test_emul import run_parked_outputs
    run_parked_outputs({"device": 0}, case)


@pytest.mark.parametrize("case", PARK_CASES_MINPHASE)
def test_hip_parked_outputs_minimum_phase(torch, case):
    from test_emul import run_parked_outputs
    run_parked_outputs({"device": 0}, case, phase=1)


@pytest.mark.parametrize("src,dst,maxin,tb", [(44100.0, 96000.0, 8192, 2.0), (96000.0, 44100.0, 16384, 2.0),
                                             (88200.0, 44100.0, 12000, 2.0), (44100.0, 88200.0, 9000, 0.5),
                                             (96000.0, 44100.0, 16384, 0.5), (88200.0, 44100.0, 12000, 0.5),
                                             (48000.0, 16000.0, 30000, 0.5)])
def test_hip_chunk_invariance_with_no_work_calls_and_checkpoints(torch, src, dst, maxin, tb):
    from test_emul import run_chunk_invariance_with_no_work_calls_and_checkpoints
    run_chunk_invariance_with_no_work_calls_and_checkpoints({"device": 0}, src, dst, maxin, tb)


@pytest.mark.parametrize("src,dst,maxin,tb", [(44100.0, 96000.0, 8192, 2.0), (96000.0, 44100.0, 16384, 2.0),
                                             (44100.0, 192000.0, 4096, 2.0), (48000.0, 44100.0, 6000, 2.0),
                                             (44100.0, 88200.0, 9000, 0.5), (96000.0, 44100.0, 16384, 0.5)])
def test_hip_chunk_invariance_minimum_phase(torch, src, dst, maxin, tb):
    """minimum-phase chains -- convolver + interpolator as one launch on the renumbered canonical stream
    (Engine::fused_shift), long-block forms with a complex spectrum --: ragged calls, single samples and checkpoints
    leave the stream bit for bit the same"""
    from test_emul import run_chunk_invariance_with_no_work_calls_and_checkpoints
    run_chunk_invariance_with_no_work_calls_and_checkpoints({"device": 0}, src, dst, maxin, tb, phase=1)


@pytest.mark.parametrize("src,dst,maxin", [(44100.0, 96000.0, 8192), (96000.0, 44100.0, 16384), (44100.0, 192000.0, 4096),
                                          (48000.0, 44100.0, 6000), (192000.0, 44100.0, 8192)])
def test_hip_fused_minimum_phase_equals_unfused(torch, src, dst, maxin):
    """option fuse_latency = 0 (two launches) against the fused launch of a minimum-phase convolver + interpolator
    pair on the real kernels: same counts per call, same samples to rounding"""
    x = make_input(5, maxin * 4 + 777, 8)
    lens = [maxin, 1, maxin // 3, 17, maxin, maxin - 5, 300, maxin]
    ys = []
    for fl in (1, 0):
        b = r8b.BatchResampler(src, dst, maxin, 2.0, 180.15, nch=5, phase=1, device=0)
        b.set_option("fuse_latency", fl)
        b.set_option("timing", 1)
        assert any(t[0] == "k_convp_whole" for t in b.stage_timings()) == bool(fl), b.stage_timings()
        b.set_option("timing", 0)
        out, pos = [], 0
        for l in lens:
            l = min(l, x.shape[1] - pos)
            out.append(b.process_host(x[:, pos:pos + l]))
            pos += l
        ys.append(out)
    assert [o.shape for o in ys[0]] == [o.shape for o in ys[1]]
    d = np.abs(np.concatenate(ys[0], axis=1) - np.concatenate(ys[1], axis=1))
    assert d.shape[1] > 1000 and d.max() <= 2e-14, d.max()


# Error budget: the tolerance of the path is RMS 1e-15 / peak 1e-13, the reference's own cross-build
# noise 3e-16 / 2.3e-15.  "Derive instead of fetch" optimisations stack roundings, so the TREND is pinned too: the
# levels measured on the final build of round 4 against the compiled reference (4 channels, 6-24 calls) plus 12 %.
# (src, dst, maxin, n, rms measured, peak measured)
ERROR_BUDGET = [
    (44100.0, 96000.0, 16384, 16384 * 6, 3.202e-16, 1.887e-15),
    (96000.0, 44100.0, 16384, 16384 * 6, 2.159e-16, 1.221e-15),
    (44100.0, 2822400.0, 1024, 1024 * 24, 3.394e-16, 2.442e-15),
    (44100.0, 88200.0, 8192, 8192 * 8, 3.152e-16, 1.665e-15),
    (88200.0, 44100.0, 8192, 8192 * 8, 2.066e-16, 9.437e-16),
    (48000.0, 32000.0, 8192, 8192 * 8, 2.684e-16, 1.221e-15),
]


@pytest.mark.parametrize("case", ERROR_BUDGET)
def test_hip_error_budget_does_not_creep(torch, refwrap, case):
    src, dst, maxin, n, r0, p0 = case
    x = make_input(4, n, 1)
    b = r8b.BatchResampler(src, dst, maxin, 2.0, 180.15, nch=4, device=0)
    ys = [b.process_host(x[:, i:i + maxin]) for i in range(0, n, maxin)]
    r, p = refwrap.batch_check(src, dst, maxin, [maxin] * (n // maxin), x, np.concatenate(ys, axis=1),
                               [y.shape[1] for y in ys], 2.0, 180.15)
    assert r.max() <= 1.12 * r0 and p.max() <= 1.12 * p0 + 2.3e-16, (r.max(), r0, p.max(), p0)


def test_cxx_batch_sharded(torch, tmp_path):
    """include/r8b/BatchSharded.h, the C++ host's helper for several devices -- three shards of whole
    channel pairs (here all on device 0, each on its own stream) equal ONE object over all channels bit for bit
    (tests/cxx_sharded.cpp)"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "cxx_sharded")
    libdir = os.path.dirname(r8b.lib_path())
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cxx_sharded.cpp"),
                    "-L" + libdir, "-lr8bsrc_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.returncode, out.stdout)


def test_cxx_rccl_scatter_gather_world1(torch, tmp_path):
    """VERDICT r5 missing #5: include/r8b/ShardTransfer.h, the native (C++ / RCCL) scatter and gather of channel shards
    for one-process-per-GPU hosts -- on a world of ONE rank (all a one-GPU box can form): scatter -> process -> gather
    equals the object driven directly bit for bit, with the root's shard as a device copy and, loopback, through
    ncclSend / ncclRecv to self inside the group (tests/cxx_rccl_world1.cpp).  Several ranks: unmeasured."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "cxx_rccl_world1")
    libdir = os.path.dirname(r8b.lib_path())
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cxx_rccl_world1.cpp"),
                    "-L" + libdir, "-lr8bsrc_hip", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.returncode, out.stdout[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("case", test_emul.COLUMN_CASES)
def test_hip_output_columns_are_bitwise_alike(torch, case):
    """the GPU twin of test_emulated_output_columns_are_bitwise_alike: here the odd columns take the 16-byte store at
    element alignment (one global_store_dwordx4 at an address that is 8 mod 16)"""
    def device_buffers(xa, rows, cols):
        xin = torch.from_numpy(xa).to("cuda:0")
        obuf = torch.full((rows, cols), float("nan"), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        keep.append(xin)

        def back(b):
            torch.cuda.synchronize()
            return b.cpu().numpy()
        return xin.data_ptr(), xin.stride(0), obuf, obuf.data_ptr(), back
    keep = []
    test_emul.run_output_columns_case({"device": 0}, case, device_buffers)
