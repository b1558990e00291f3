"""Host-side mirror of the reference's front-end over the C ABI (include/r8bsrc.h).

Names, arguments and behaviour follow r8b::CDSPResampler (reference CDSPResampler.h:117-120,
406-421, 476-519, 521-529, 559-575, 592-651) and its presets CDSPResampler16 / 16IR / 24
(:729-810), so the parity tests read like calls into the reference.  `BatchResampler` is the
N-channel object the throughput numbers are quoted on: process() takes torch CUDA tensors (device
memory and streams are the only things torch is used for).

`lib=` lets the tests bind another library that exports the same ABI; the default is the HIP
library and nothing else.
"""
import ctypes as C

import numpy as np

from . import _capi

fprLinearPhase = 0
fprMinPhase = 1


# r8b_pcm_format (include/r8bsrc.h)
PCM_F64, PCM_F32, PCM_S16, PCM_S24, PCM_S32 = 0, 1, 2, 3, 4


def _dptr(a):
    return a.ctypes.data_as(_capi.dp)


class _Base:
    def __init__(self, lib):
        self._lib = lib if lib is not None else _capi.load()
        self._h = None

    def _err(self):
        return self._lib.r8b_last_error().decode()


class BatchResampler(_Base):
    """`nch` independent CDSPResampler streams sharing one schedule (C ABI part 2)."""

    def __init__(self, SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand=2.0,
                 ReqAtten=206.91, nch=1, device=-1, lib=None, stage=None, phase=0):
        """phase: 0 = fprLinearPhase (the reference's default), 1 = fprMinPhase"""
        super().__init__(lib)
        self.nch = int(nch)
        self.MaxInLen = int(aMaxInLen)
        if stage is not None:
            kind, a, b, c, d, i0, i1 = stage
            self._h = self._lib.r8b_batch_create_stage(int(kind), a, b, c, d, int(i0), int(i1),
                                                       self.MaxInLen, self.nch, int(device))
        elif phase:
            self._h = self._lib.r8b_batch_create_ex(SrcSampleRate, DstSampleRate, self.MaxInLen,
                                                    ReqTransBand, ReqAtten, int(phase), self.nch,
                                                    int(device))
        else:
            self._h = self._lib.r8b_batch_create(SrcSampleRate, DstSampleRate, self.MaxInLen,
                                                 ReqTransBand, ReqAtten, self.nch, int(device))
        if not self._h:
            raise RuntimeError(self._err())
        self.max_out_len = self._lib.r8b_batch_max_out_len(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.r8b_batch_delete(self._h)
            self._h = None

    def getMaxOutLen(self, MaxInLen=0):
        return self.max_out_len

    def getInLenBeforeOutPos(self, ReqOutPos):
        return self._lib.r8b_batch_inlen_before_outpos(self._h, int(ReqOutPos))

    def getInputRequiredForOutput(self, ReqOutSamples):
        return self._lib.r8b_batch_inlen(self._h, int(ReqOutSamples))

    def clear(self):
        self._lib.r8b_batch_clear(self._h)
        self._produced = 0

    def describe(self):
        n = self._lib.r8b_batch_describe(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self._lib.r8b_batch_describe(self._h, buf, n + 1)
        return buf.value.decode()

    def set_option(self, name, value):
        if self._lib.r8b_batch_set_option(self._h, name.encode(), int(value)) != 0:
            raise KeyError(name)

    def getLatencyFrac(self):
        """reference CDSPResampler.h:491-494: the chain's residual fractional latency (0.0 for linear phase)"""
        return self._lib.r8b_batch_latency_frac(self._h)

    def stat(self, name):
        """a counter of the engine since creation (include/r8bsrc.h r8b_batch_stat)"""
        v = self._lib.r8b_batch_stat(self._h, name.encode())
        if v < 0:
            raise KeyError(name)
        return v

    def stage_timings(self):
        """[(kernel name, total ms, launches, samples in, samples out)] per stage since the last
        query (needs set_option("timing", 1)); sample counts are per channel; waits for the
        recorded events."""
        res = []
        for s in range(self._lib.r8b_batch_stage_count(self._h)):
            ms, n = C.c_double(), C.c_int()
            si, so = C.c_longlong(), C.c_longlong()
            name = C.create_string_buffer(64)
            if self._lib.r8b_batch_stage_timing(self._h, s, ms, n, si, so, name, 64) != 0:
                raise RuntimeError(self._err())
            res.append((name.value.decode(), ms.value, n.value, si.value, so.value))
        return res

    def stage_symbols(self):
        """device symbol of every stage's most recent timed launch, as rocprofv3 names it (r8b_batch_stage_symbol)"""
        res = []
        for s in range(self._lib.r8b_batch_stage_count(self._h)):
            buf = C.create_string_buffer(96)
            if self._lib.r8b_batch_stage_symbol(self._h, s, buf, 96) != 0:
                raise RuntimeError(self._err())
            res.append(buf.value.decode())
        return res

    def process_ptr(self, d_in, in_stride, l, d_out, out_stride, stream=0):
        """Raw device-pointer entry (r8b_batch_process)."""
        n = self._lib.r8b_batch_process(self._h, C.c_void_p(d_in), in_stride, int(l),
                                        C.c_void_p(d_out), out_stride, C.c_void_p(stream))
        if n < 0:
            raise RuntimeError(self._err())
        return n

    def process(self, x, out=None):
        """x: float64 CUDA tensor [nch, l] (row stride >= l); returns a view [nch, n] of `out`
        (allocated [nch, max_out_len] if not given).  Enqueues on torch's current stream."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.shape[0] == self.nch
        assert x.stride(1) == 1
        dev = self._lib.r8b_batch_device(self._h)
        if x.device.index != dev or (out is not None and out.device.index != dev):
            raise ValueError("tensors on cuda:%s, the resampler lives on cuda:%d" % (x.device.index, dev))
        l = x.shape[1]
        if out is None:
            # rows on a 64-byte pitch, this call's outputs at column (outputs produced so far) mod 8: output j of
            # the stream always sits at a column congruent to j mod 8, so the 64-byte pieces the fused kernels store
            # (four adjacent phase pairs of a group) are whole aligned segments in every call
            cap = max(self.max_out_len, 1)
            off = getattr(self, "_produced", 0) % 8
            out = torch.empty((self.nch, (cap + 7) // 8 * 8 + 8), dtype=torch.float64, device=x.device)[:, off:off + cap]
        assert out.is_cuda and out.dtype == torch.float64 and out.stride(1) == 1
        assert out.shape[0] == self.nch and out.shape[1] >= self.max_out_len
        stream = torch.cuda.current_stream(x.device).cuda_stream
        n = self.process_ptr(x.data_ptr(), x.stride(0), l, out.data_ptr(), out.stride(0), stream)
        self._produced = getattr(self, "_produced", 0) + n
        return out[:, :n]

    def state_dict(self, stream=0):
        """Checkpoint of the streaming state of all channels as one numpy uint8 blob
        (r8b_batch_state_save); waits for `stream`."""
        size = self._lib.r8b_batch_state_size(self._h)
        buf = np.empty(size, dtype=np.uint8)
        n = self._lib.r8b_batch_state_save(self._h, C.c_void_p(buf.ctypes.data), size,
                                           C.c_void_p(stream))
        if n < 0:
            raise RuntimeError(self._err())
        return buf[:n]

    def load_state_dict(self, blob, stream=0):
        """Resume from a blob of state_dict() taken from an equally configured object."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        if self._lib.r8b_batch_state_load(self._h, C.c_void_p(blob.ctypes.data), blob.size,
                                          C.c_void_p(stream)) < 0:
            raise RuntimeError(self._err())

    def process_pcm_ptr(self, d_in, in_format, in_interleaved, in_stride, l, d_out, out_format,
                        out_interleaved, out_stride, stream=0):
        """Raw device-pointer PCM entry (r8b_batch_process_pcm); strides in samples."""
        n = self._lib.r8b_batch_process_pcm(self._h, C.c_void_p(d_in), int(in_format),
                                            int(bool(in_interleaved)), in_stride, int(l),
                                            C.c_void_p(d_out), int(out_format),
                                            int(bool(out_interleaved)), out_stride,
                                            C.c_void_p(stream))
        if n < 0:
            raise RuntimeError(self._err())
        return n

    def process_pcm(self, x, out_format=None, out=None, planar=False):
        """PCM in, PCM out, both CUDA tensors.  Interleaved (default): x is [frames, nch] of
        int16 / int32 / float32 / float64, or uint8 [frames, nch, 3] for packed 24-bit; returns a
        view [n, nch(, 3)] of `out` (allocated [max_out_len, nch(, 3)] if not given) in
        `out_format` (default: the input's).  planar=True: x is [nch, frames(, 3)] and the result
        [nch, n(, 3)]; the conversion then happens inside the first and last stage kernels, no
        staging copy.  Enqueues on torch's current stream."""
        import torch
        fmt_of = {torch.float64: PCM_F64, torch.float32: PCM_F32, torch.int16: PCM_S16,
                  torch.int32: PCM_S32, torch.uint8: PCM_S24}
        dtype_of = {v: k for k, v in fmt_of.items()}
        if planar:
            return self._process_pcm_planar(x, fmt_of, dtype_of, out_format, out)
        assert x.is_cuda and x.is_contiguous() and x.shape[1] == self.nch
        in_format = fmt_of[x.dtype]
        assert (x.dim() == 3 and x.shape[2] == 3) if in_format == PCM_S24 else x.dim() == 2
        if out_format is None:
            out_format = in_format
        tail = (3,) if out_format == PCM_S24 else ()
        if out is None:
            out = torch.empty((max(self.max_out_len, 1), self.nch) + tail,
                              dtype=dtype_of[out_format], device=x.device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == dtype_of[out_format]
        assert out.shape[0] >= self.max_out_len and tuple(out.shape[1:]) == (self.nch,) + tail
        stream = torch.cuda.current_stream(x.device).cuda_stream
        n = self.process_pcm_ptr(x.data_ptr(), in_format, True, self.nch, x.shape[0],
                                 out.data_ptr(), out_format, True, self.nch, stream)
        return out[:n]

    def _process_pcm_planar(self, x, fmt_of, dtype_of, out_format, out):
        import torch
        assert x.is_cuda and x.is_contiguous() and x.shape[0] == self.nch
        in_format = fmt_of[x.dtype]
        assert (x.dim() == 3 and x.shape[2] == 3) if in_format == PCM_S24 else x.dim() == 2
        if out_format is None:
            out_format = in_format
        tail = (3,) if out_format == PCM_S24 else ()
        cap = max(self.max_out_len, 1)
        if out is None:
            out = torch.empty((self.nch, cap) + tail, dtype=dtype_of[out_format], device=x.device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == dtype_of[out_format]
        assert out.shape[0] == self.nch and out.shape[1] >= self.max_out_len
        stream = torch.cuda.current_stream(x.device).cuda_stream
        n = self.process_pcm_ptr(x.data_ptr(), in_format, False, x.shape[1], x.shape[1],
                                 out.data_ptr(), out_format, False, out.shape[1], stream)
        return out[:, :n]

    def process_host(self, x):
        """x: float64 numpy [nch, l]; synchronous; returns numpy [nch, n]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.ndim == 2 and x.shape[0] == self.nch
        l = x.shape[1]
        cap = max(self.max_out_len, 1)
        out = np.empty((self.nch, cap), dtype=np.float64)
        n = self._lib.r8b_batch_process_host(self._h, _dptr(x), l, l, _dptr(out), cap)
        if n < 0:
            raise RuntimeError(self._err())
        return out[:, :n].copy()


class CDSPResampler(_Base):
    """Single stream through the reference's own five C-ABI symbols is `DLLResampler`; this class
    mirrors the C++ front-end, i.e. arbitrary ReqAtten (the DLL only offers three presets)."""

    def __init__(self, SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand=2.0,
                 ReqAtten=206.91, ReqPhase=fprLinearPhase, lib=None, device=-1):
        super().__init__(lib)
        self._b = BatchResampler(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, ReqAtten,
                                 nch=1, device=device, lib=self._lib, phase=int(ReqPhase))
        self.SrcSampleRate, self.DstSampleRate = SrcSampleRate, DstSampleRate
        self.MaxInLen = int(aMaxInLen)

    def getInLenBeforeOutPos(self, ReqOutPos):
        return self._b.getInLenBeforeOutPos(ReqOutPos)

    def getInputRequiredForOutput(self, ReqOutSamples):
        return self._b.getInputRequiredForOutput(ReqOutSamples)

    def getInLenBeforeOutStart(self, ReqOutPos=0):
        """reference CDSPResampler.h:443-464: feed one zero sample at a time until the output
        length passes ReqOutPos, then clear().  (Legacy/test helper; slow by design.)"""
        inc = 0
        outc = 0
        one = np.zeros(1)
        while True:
            outc += len(self.process(one))
            if outc > ReqOutPos:
                self.clear()
                return inc
            inc += 1

    def getMaxOutLen(self, MaxInLen=0):
        return self._b.max_out_len

    def getLatencyFrac(self):
        """reference CDSPResampler.h:491-494"""
        return self._b.getLatencyFrac()

    def getLatency(self):
        return 0

    def clear(self):
        self._b.clear()

    def process(self, ip0):
        x = np.ascontiguousarray(ip0, dtype=np.float64).reshape(1, -1)
        if self.SrcSampleRate == self.DstSampleRate:
            return x[0].copy()
        return self._b.process_host(x)[0]

    def oneshot(self, ip, oplen):
        """reference CDSPResampler.h:592-651."""
        ip = np.asarray(ip, dtype=np.float64)
        out = []
        got = 0
        pos = 0
        while got < oplen:
            if pos < len(ip):
                blk = ip[pos:pos + self.MaxInLen]
                pos += len(blk)
            else:
                blk = np.zeros(self.MaxInLen)
            y = self.process(blk)
            y = y[:oplen - got]
            out.append(y)
            got += len(y)
        self.clear()
        return np.concatenate(out) if out else np.zeros(0)


class CDSPResampler16(CDSPResampler):
    def __init__(self, SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand=2.0, **kw):
        super().__init__(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, 136.45, **kw)


class CDSPResampler16IR(CDSPResampler):
    def __init__(self, SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand=2.0, **kw):
        super().__init__(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, 109.56, **kw)


class CDSPResampler24(CDSPResampler):
    def __init__(self, SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand=2.0, **kw):
        super().__init__(SrcSampleRate, DstSampleRate, aMaxInLen, ReqTransBand, 180.15, **kw)


class DLLResampler(_Base):
    """The five drop-in symbols exactly as a C host would call them (reference DLL/r8bsrc.h)."""

    r8brr16, r8brr16IR, r8brr24 = 0, 1, 2

    def __init__(self, SrcSampleRate, DstSampleRate, MaxInLen, ReqTransBand=2.0, Res=2, lib=None):
        super().__init__(lib)
        self._h = self._lib.r8b_create(SrcSampleRate, DstSampleRate, int(MaxInLen), ReqTransBand,
                                       int(Res))
        if not self._h:
            raise RuntimeError(self._err())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.r8b_delete(self._h)
            self._h = None

    def inlen(self, n):
        return self._lib.r8b_inlen(self._h, int(n))

    def clear(self):
        self._lib.r8b_clear(self._h)

    def process(self, ip0):
        x = np.ascontiguousarray(ip0, dtype=np.float64)
        op = _capi.dp()
        n = self._lib.r8b_process(self._h, _dptr(x), len(x), C.byref(op))
        return np.ctypeslib.as_array(op, shape=(n,)).copy() if n > 0 else np.zeros(0)
