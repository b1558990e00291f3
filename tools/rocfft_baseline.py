#!/usr/bin/env python3
"""tools/rocfft_baseline.py -- the "obvious alternative" the hand-written FFT has to beat (north_star:
"rocFFT only as a baseline"): the block convolver of BASELINE config 2 (1024 channels x
16384 samples, 44100 -> 88200: 2048-point blocks, 1417-tap kernel, 2x up by spectrum replication) as
batched rocFFT R2C -> pointwise multiply -> rocFFT C2R through torch.fft (hipFFT/rocFFT back end),
timed with HIP events on data resident in HBM.  It computes the same overlap-save stream as
reference CDSPBlockConvolver.h:307-344 (spectra not bit-compatible: timing study only).
Compare with `python bench.py --src 44100 --dst 88200 --no-cpu` (the hand-written pair kernel)."""
import json
import torch

C, L = 1024, 16384
NIN, NOUT, FL2 = 2048, 4096, 708        # block geometry of the reference for this filter
INLEN = (NOUT - 2 * FL2) // 2            # 1340 new input samples per block
dev = torch.device("cuda")
x = torch.rand((C, L + NIN), dtype=torch.float64, device=dev) * 2 - 1   # history in front
nblk = L // INLEN                                                        # 12 whole blocks per call
H = torch.rand((NOUT // 2 + 1,), dtype=torch.float64, device=dev)        # zero-phase kernel spectrum (real)


def step(fused_view=True):
    # overlapping blocks as a strided view (no gather copy): [C, nblk, NIN]
    blocks = x.as_strided((C, nblk, NIN), (x.stride(0), INLEN, 1), storage_offset=NIN - (NIN - INLEN))
    X = torch.fft.rfft(blocks, dim=2)                                    # [C, nblk, 1025]
    # 2x zero stuffing = the spectrum repeated: bins 0..2048 of the 4096-point transform
    Xr = torch.cat([X, torch.conj(torch.flip(X[:, :, :-1], dims=[2]))], dim=2)[:, :, :NOUT // 2 + 1]
    Y = Xr * H
    y = torch.fft.irfft(Y, n=NOUT, dim=2)                                # [C, nblk, 4096]
    return y[:, :, FL2:FL2 + 2 * INLEN].reshape(C, -1)                   # valid outputs


for _ in range(5):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
e0.record()
for _ in range(K):
    out = step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
# rfft + irfft alone (no replication / multiply / slicing passes)
blocks = x.as_strided((C, nblk, NIN), (x.stride(0), INLEN, 1), storage_offset=INLEN)
Y = torch.fft.rfft(blocks, dim=2)
Y2 = torch.zeros((C, nblk, NOUT // 2 + 1), dtype=torch.complex128, device=dev)
torch.cuda.synchronize()
e0.record()
for _ in range(K):
    Yf = torch.fft.rfft(blocks, dim=2)
    yb = torch.fft.irfft(Y2, n=NOUT, dim=2)
e1.record()
torch.cuda.synchronize()
ms_fft = e0.elapsed_time(e1) / K
print(json.dumps({"workload": "%d ch x %d blocks of %d -> %d points (44100->88200 convolver, cfg2 batch)" % (C, nblk, NIN, NOUT),
                  "rocfft_r2c_mul_c2r_ms": round(ms, 4), "rocfft_transforms_only_ms": round(ms_fft, 4),
                  "in_msamples_per_s": round(C * nblk * INLEN / ms / 1e3, 1)}))
