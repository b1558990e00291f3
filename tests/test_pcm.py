"""PCM ingest/egress kernels next to the hot path (r8b_pcm.h): interleaved or
planar int16 / packed int24 / int32 / float32 / float64 buffers in and out of r8b_batch_process_pcm.

Checked against numpy restatements of the documented conventions (decode = value / 2^(bits-1);
encode = round-half-even of v * 2^(bits-1), saturated) applied around the fp64 path itself:
bit-exact for the integer formats.  CPU tier: host emulation (its "device memory" is host memory,
so numpy buffers are passed as device pointers); GPU tier: torch tensors."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import r8b_oracle as O
from conftest import ROOT

r8b = importlib.import_module("r8brain-free-src_amd")

BITS = {r8b.PCM_S16: 16, r8b.PCM_S24: 24, r8b.PCM_S32: 32}


def np_decode(raw, fmt):
    """raw: integer/float array of sample values (int24 already as int32 values)."""
    if fmt in BITS:
        return raw.astype(np.float64) / float(1 << (BITS[fmt] - 1))
    return raw.astype(np.float64)


def np_encode(v, fmt):
    if fmt in BITS:
        s = float(1 << (BITS[fmt] - 1))
        q = np.rint(v * s)  # half to even
        q = np.where(np.isnan(q), 0.0, q)
        return np.clip(q, -s, s - 1.0).astype(np.int64)
    if fmt == r8b.PCM_F32:
        return v.astype(np.float32)
    return v


def pack24(vals):
    """int values [..] -> uint8 [.., 3], little-endian two's complement."""
    u = (vals.astype(np.int64) & 0xFFFFFF).astype(np.uint32)
    return np.stack([(u & 255), (u >> 8) & 255, (u >> 16) & 255], axis=-1).astype(np.uint8)


def unpack24(b):
    u = b[..., 0].astype(np.int64) | (b[..., 1].astype(np.int64) << 8) | (b[..., 2].astype(np.int64) << 16)
    return np.where(u >= 1 << 23, u - (1 << 24), u)


NP_DTYPE = {r8b.PCM_F64: np.float64, r8b.PCM_F32: np.float32, r8b.PCM_S16: np.int16,
            r8b.PCM_S32: np.int32}


def make_pcm(fmt, frames, nch, seed):
    """Random full-range samples as (storage array [frames, nch(,3)], sample values)."""
    u = np.stack([O.splitmix_uniform(seed + c, frames) for c in range(nch)], axis=1)
    if fmt in BITS:
        vals = np_encode(u, fmt)
        store = pack24(vals) if fmt == r8b.PCM_S24 else vals.astype(NP_DTYPE[fmt])
        return np.ascontiguousarray(store), vals
    store = u.astype(NP_DTYPE[fmt])
    return np.ascontiguousarray(store), store


def values_of(store, fmt):
    return unpack24(store) if fmt == r8b.PCM_S24 else store


@pytest.fixture(scope="module")
def emul():
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.run(["make"], cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return r8b.bind(os.path.join(d, "_build", "libr8bsrc_emul.so"))


CASES = [(r8b.PCM_S16, r8b.PCM_S16, 2), (r8b.PCM_S24, r8b.PCM_S24, 3), (r8b.PCM_S32, r8b.PCM_F32, 70),
         (r8b.PCM_F32, r8b.PCM_S24, 5), (r8b.PCM_S16, r8b.PCM_F64, 65), (r8b.PCM_F64, r8b.PCM_S32, 1)]


def run_emul(lib, fin, fout, nch, interleaved, src=44100.0, dst=48000.0, att=136.45, staged_sides=None, tb=2.0,
             frames=3000, chunk=1000):
    store, vals = make_pcm(fin, frames, nch, 11)
    x = np_decode(vals, fin)  # [frames, nch]
    a = r8b.BatchResampler(src, dst, chunk, tb, att, nch=nch, lib=lib)
    b = r8b.BatchResampler(src, dst, chunk, tb, att, nch=nch, lib=lib)
    cap = a.max_out_len
    tail = (3,) if fout == r8b.PCM_S24 else ()
    odt = np.uint8 if fout == r8b.PCM_S24 else NP_DTYPE[fout]
    for i in range(0, frames, chunk):
        want = b.process_host(np.ascontiguousarray(x[i:i + chunk].T))  # [nch, n]
        if interleaved:
            inp = np.ascontiguousarray(store[i:i + chunk])
            out = np.zeros((cap, nch) + tail, dtype=odt)
            n = a.process_pcm_ptr(inp.ctypes.data, fin, True, nch, chunk, out.ctypes.data, fout,
                                  True, nch)
            got = values_of(out[:n], fout)
            ref = np_encode(want.T, fout)
        else:
            inp = np.ascontiguousarray(np.moveaxis(store[i:i + chunk], 1, 0))  # [nch, l(,3)]
            out = np.zeros((nch, cap) + tail, dtype=odt)
            n = a.process_pcm_ptr(inp.ctypes.data, fin, False, chunk, chunk, out.ctypes.data, fout,
                                  False, cap)
            got = values_of(out[:, :n], fout)
            ref = np_encode(want, fout)
        assert n == want.shape[1]
        assert np.array_equal(got, ref), (fin, fout, nch, interleaved, i)
    if staged_sides is not None:
        assert a.stat("pcm_staged_sides") == staged_sides * (frames // chunk), a.stat("pcm_staged_sides")


@pytest.mark.parametrize("fin,fout,nch", CASES)
@pytest.mark.parametrize("interleaved", [True, False])
def test_pcm_emulated(emul, fin, fout, nch, interleaved):
    run_emul(emul, fin, fout, nch, interleaved)


# planar buffers are decoded / encoded inside the first / last stage kernel: one case per kernel
# that can sit at either end of a chain
EDGE_TOPOLOGIES = [(44100.0, 44101.0),     # fast convolver first, polynomial interpolator last
                   (176400.0, 44100.0),    # half-band decimator first, 2x-decimating convolver last
                   (44100.0, 2822400.0),   # half-band cascade last
                   (48000.0, 32000.0),     # generic convolver first and last
                   (44100.0, 705600.0),    # half-band up-samplers last
                   (96000.0, 11025.0),     # half-band decimators, convolver, whole-step interpolator
                   (48000.0, 48000.0)]     # pass-through (staged)


@pytest.mark.parametrize("src,dst", EDGE_TOPOLOGIES)
def test_pcm_planar_fused_edges_emulated(emul, src, dst):
    run_emul(emul, r8b.PCM_S16, r8b.PCM_S24, 3, False, src, dst)
    run_emul(emul, r8b.PCM_F32, r8b.PCM_S32, 2, False, src, dst)


# Planar PCM at a compile-time-sized convolver (the 24-bit preset's chains start and / or end in one) goes through the
# staging rows (counter pcm_staged_sides); the fp64 values behind the codec are the fp64 path's own bit for bit.  A PCM
# twin of the two headline pair kernels (decode in the block's loads, encode in its stores) was built and measured in
# round 4: 0.352 ms per cfg2 call with planar s16 both ways against 0.33 through the staging rows -- two-byte stores
# from the interpolator's lanes and the general load path cost more than the staging pass --, so it is not shipped.
# (src, dst, planar sides that are staged)
PCM_PAIR_TOPOLOGIES = [(44100.0, 96000.0, 2),      # fused pair, two phases per thread (cfg2)
                       (96000.0, 44100.0, 2),      # fused 1:1 pair (cfg3)
                       (44100.0, 88200.0, 2),      # convolver alone: both edges in one kernel
                       (44100.0, 2822400.0, 1),    # cfg5: pair convolver first (staged), cascade last (encodes itself)
                       (88200.0, 44100.0, 2)]      # decimating form


@pytest.mark.parametrize("src,dst,staged", PCM_PAIR_TOPOLOGIES)
def test_pcm_planar_at_the_pair_kernels_emulated(emul, src, dst, staged):
    run_emul(emul, r8b.PCM_S16, r8b.PCM_S24, 3, False, src, dst, att=180.15, staged_sides=staged)
    run_emul(emul, r8b.PCM_F32, r8b.PCM_S32, 2, False, src, dst, att=180.15, staged_sides=staged)


@pytest.mark.parametrize("src,dst", [(44100.0, 88200.0), (88200.0, 44100.0), (48000.0, 16000.0)])
def test_pcm_planar_at_the_long_block_forms_emulated(emul, src, dst):
    """transition band 0.5 %: the split 2x up-sampling form and the one-channel form of the pair kernel (fp64 rows only)
    between planar PCM edges -- both sides through the staging rows, enough frames to get past the filter's latency"""
    run_emul(emul, r8b.PCM_S24, r8b.PCM_S16, 3, False, src, dst, att=180.15, staged_sides=2, tb=0.5, frames=45000, chunk=9000)


def test_pcm_rounding_and_saturation(emul):
    """Src == Dst passes samples through: the codec alone.  Half-way cases round to even, values
    beyond full scale saturate, NaN encodes as 0."""
    v = np.array([0.5 / 32768, 1.5 / 32768, 2.5 / 32768, -0.5 / 32768, -1.5 / 32768, 1.0, -1.0,
                  2.0, -3.0, np.nan, 32766.5 / 32768, 0.25 / 32768])
    a = r8b.BatchResampler(48000.0, 48000.0, len(v), 2.0, 136.45, nch=1, lib=emul)
    out = np.zeros((len(v), 1), dtype=np.int16)
    x = np.ascontiguousarray(v[:, None])
    n = a.process_pcm_ptr(x.ctypes.data, r8b.PCM_F64, True, 1, len(v), out.ctypes.data,
                          r8b.PCM_S16, True, 1)
    assert n == len(v)
    assert out[:, 0].tolist() == [0, 2, 2, 0, -2, 32767, -32768, 32767, -32768, 0, 32766, 0]
    assert np.array_equal(out[:, 0], np_encode(v, r8b.PCM_S16))
    with pytest.raises(RuntimeError, match="format"):
        a.process_pcm_ptr(x.ctypes.data, 9, True, 1, 4, out.ctypes.data, r8b.PCM_S16, True, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("fin,fout,nch", CASES)
def test_pcm_gpu(fin, fout, nch):
    import torch
    frames, chunk = 3000, 1000
    store, vals = make_pcm(fin, frames, nch, 11)
    x = np_decode(vals, fin)
    a = r8b.BatchResampler(44100.0, 48000.0, chunk, 2.0, 136.45, nch=nch)
    b = r8b.BatchResampler(44100.0, 48000.0, chunk, 2.0, 136.45, nch=nch)
    for i in range(0, frames, chunk):
        want = b.process_host(np.ascontiguousarray(x[i:i + chunk].T))
        t = torch.from_numpy(np.ascontiguousarray(store[i:i + chunk])).cuda()
        y = a.process_pcm(t, out_format=fout)
        torch.cuda.synchronize()
        got = values_of(y.cpu().numpy(), fout)
        assert got.shape[0] == want.shape[1]
        assert np.array_equal(got, np_encode(want.T, fout)), (fin, fout, nch, i)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", EDGE_TOPOLOGIES)
def test_pcm_planar_fused_edges_gpu(src, dst):
    import torch
    nch, frames, chunk = 3, 3000, 1000
    store, vals = make_pcm(r8b.PCM_S16, frames, nch, 23)
    x = np_decode(vals, r8b.PCM_S16)
    a = r8b.BatchResampler(src, dst, chunk, 2.0, 136.45, nch=nch)
    b = r8b.BatchResampler(src, dst, chunk, 2.0, 136.45, nch=nch)
    for i in range(0, frames, chunk):
        want = b.process_host(np.ascontiguousarray(x[i:i + chunk].T))
        t = torch.from_numpy(np.ascontiguousarray(store[i:i + chunk].T)).cuda()  # [nch, l]
        y = a.process_pcm(t, out_format=r8b.PCM_S24, planar=True)
        torch.cuda.synchronize()
        got = unpack24(y.cpu().numpy())
        assert got.shape == want.shape
        assert np.array_equal(got, np_encode(want, r8b.PCM_S24)), (src, dst, i)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst,staged", PCM_PAIR_TOPOLOGIES)
def test_pcm_planar_at_the_pair_kernels_gpu(src, dst, staged):
    import torch
    nch, frames, chunk = 3, 6000, 2000
    store, vals = make_pcm(r8b.PCM_S16, frames, nch, 29)
    x = np_decode(vals, r8b.PCM_S16)
    a = r8b.BatchResampler(src, dst, chunk, 2.0, 180.15, nch=nch)
    b = r8b.BatchResampler(src, dst, chunk, 2.0, 180.15, nch=nch)
    for i in range(0, frames, chunk):
        want = b.process_host(np.ascontiguousarray(x[i:i + chunk].T))
        t = torch.from_numpy(np.ascontiguousarray(store[i:i + chunk].T)).cuda()  # [nch, l]
        y = a.process_pcm(t, out_format=r8b.PCM_S24, planar=True)
        torch.cuda.synchronize()
        got = unpack24(y.cpu().numpy())
        assert got.shape == want.shape
        assert np.array_equal(got, np_encode(want, r8b.PCM_S24)), (src, dst, i)
    assert a.stat("pcm_staged_sides") == staged * (frames // chunk)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [(44100.0, 88200.0), (88200.0, 44100.0), (48000.0, 16000.0)])
def test_pcm_planar_at_the_long_block_forms_gpu(src, dst):
    """the GPU twin of test_pcm_planar_at_the_long_block_forms_emulated (transition band 0.5 %)"""
    import torch
    nch, frames, chunk = 3, 45000, 9000
    store, vals = make_pcm(r8b.PCM_S24, frames, nch, 31)
    x = np_decode(vals, r8b.PCM_S24)
    a = r8b.BatchResampler(src, dst, chunk, 0.5, 180.15, nch=nch)
    b = r8b.BatchResampler(src, dst, chunk, 0.5, 180.15, nch=nch)
    seen = 0
    for i in range(0, frames, chunk):
        want = b.process_host(np.ascontiguousarray(x[i:i + chunk].T))
        t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(store[i:i + chunk], 1, 0))).cuda()  # [nch, l, 3]
        y = a.process_pcm(t, out_format=r8b.PCM_S16, planar=True)
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got, np_encode(want, r8b.PCM_S16)), (src, dst, i)
        seen += want.shape[1]
    assert seen > 0 and a.stat("pcm_staged_sides") == 2 * (frames // chunk)


@pytest.mark.gpu
def test_pcm_gpu_full_size_roundtrip():
    """1024 channels x 16384 frames of int16 through 44100 -> 96000: the egress of the fp64 result
    equals the numpy encoding of the fp64 path's own output (bit-exact), at BASELINE's size."""
    import torch
    nch, L = 1024, 16384
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x16 = torch.randint(-32768, 32767, (L, nch), generator=g, dtype=torch.int16, device="cuda")
    a = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=nch)
    b = r8b.BatchResampler(44100.0, 96000.0, L, 2.0, 180.15, nch=nch)
    xf = (x16.to(torch.float64) / 32768.0).t().contiguous()
    for _ in range(2):
        y16 = a.process_pcm(x16)
        yf = b.process(xf)
        assert y16.shape[0] == yf.shape[1]
        ref = torch.clamp(torch.round(yf * 32768.0), -32768, 32767).to(torch.int16).t()
        assert torch.equal(y16, ref)
