#!/usr/bin/env python3
"""tests/golden/make_golden.py -- generates the committed golden fixtures FROM THE REAL REFERENCE.

Runs only in the build container (needs oracle/_ref built from /root/reference by
oracle/Makefile).  The reference's own golden WAV pair is missing from the checkout and its
bench programs do not compile, so the fixtures are outputs of the reference
code itself on documented synthetic input (splitmix64 noise).

Outputs (small, committed):
  tests/golden/streams.npz   per case: per-call output counts + the full output stream
  tests/golden/tables.npz    low-pass real spectra, whole-step and polynomial banks
  tests/golden/inlen.json    r8b_inlen() (getInputRequiredForOutput) for n = 0..64 and some larger n
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refwrap as R  # noqa: E402

# name: (src, dst, maxin, chunk, n_in, tb, atten, seed)
CASES = {
    "cfg2_44k_96k": (44100.0, 96000.0, 1024, 1024, 6144, 2.0, 180.15, 7),
    "cfg3_96k_44k": (96000.0, 44100.0, 1024, 1000, 12000, 2.0, 180.15, 1),
    "cfg5_44k_2822k": (44100.0, 2822400.0, 256, 256, 1792, 2.0, 180.15, 1),
    "hbdown_176k_44k": (176400.0, 44100.0, 2048, 777, 24000, 2.0, 180.15, 1),
    "sacd_down_2822k_176k": (2822400.0, 176400.0, 4096, 4096, 65536, 2.0, 180.15, 2),
    "poly_44100_44101": (44100.0, 44101.0, 512, 300, 6000, 2.0, 180.15, 3),
    "up3_44k_132k": (44100.0, 132300.0, 512, 512, 4096, 2.0, 180.15, 4),
    "down3_48k_32k": (48000.0, 32000.0, 512, 511, 12000, 2.0, 180.15, 5),
    "ratio32_32k_48k": (32000.0, 48000.0, 512, 512, 6000, 2.0, 180.15, 6),
    "interm_44k_192k": (44100.0, 192000.0, 512, 512, 3072, 2.0, 180.15, 8),
    "res16_44k_48k": (44100.0, 48000.0, 512, 512, 5120, 2.0, 136.45, 9),
    "res16ir_48k_44k": (48000.0, 44100.0, 512, 100, 6000, 3.0, 109.56, 10),
    "impulse_44k_96k": (44100.0, 96000.0, 16384, 16384, 16384, 2.0, 180.15, 0),
}

LP_CASES = [(0.5, 2.0, 180.15, 2.0), (0.459375, 2.0, 180.15, 1.0), (0.5, 2.0, 180.15, 0.5),
            (1.0 / 3.0, 2.0, 180.15, 3.0), (0.5, 2.0, 136.45, 2.0), (0.5, 3.0, 109.56, 2.0)]
WS_BANKS = [(160, 180.15, False), (147, 180.15, False), (80, 136.45, False), (147, 109.56, True)]
POLY_BANKS = [(180.15, False), (136.45, False), (109.56, True)]


def main():
    streams = {}
    for name, (src, dst, maxin, chunk, n, tb, att, seed) in CASES.items():
        if seed == 0:
            x = np.zeros(n)
            x[0] = 1.0
        else:
            x = R.splitmix_uniform(seed, n)
        r = R.RefResampler(src, dst, maxin, tb, att)
        outs = [r.process(x[i:i + chunk]) for i in range(0, n, chunk)]
        y = np.concatenate(outs)
        if name == "impulse_44k_96k":
            y = y[:4096]
        streams[name + "/counts"] = np.array([len(o) for o in outs], dtype=np.int32)
        streams[name + "/y"] = y
        streams[name + "/params"] = np.array([src, dst, maxin, chunk, n, tb, att, seed])
        print(name, len(y))
    np.savez_compressed(os.path.join(HERE, "streams.npz"), **streams)

    tabs = {}
    for i, (nf, tb, att, g) in enumerate(LP_CASES):
        f = R.lpfilter_real_spectrum(nf, tb, att, g)
        tabs["lp%d/params" % i] = np.array([nf, tb, att, g, f["kernel_len"], f["block_len_bits"],
                                             f["latency"]])
        tabs["lp%d/H" % i] = f["H"]
    for i, (fr, att, third) in enumerate(WS_BANKS):
        b = R.fracbank(fr, 1, 2, att, third)
        tabs["ws%d/params" % i] = np.array([fr, att, float(third), b["filter_len"]])
        tabs["ws%d/table" % i] = b["table"]
    for i, (att, third) in enumerate(POLY_BANKS):
        b = R.fracbank(-1, 3, 8, att, third)
        fl = b["filter_len"]
        # undo the reference's SIMD pair shuffle (CDSPFracInterpolator.h:369-384) -> (c0,c1,c2) per tap
        t = b["table"].reshape(b["fracs"] + 1, fl // 2, 3, 2).transpose(0, 1, 3, 2).reshape(
            b["fracs"] + 1, fl * 3)
        tabs["poly%d/params" % i] = np.array([att, float(third), fl, b["fracs"]])
        # keep the fixture small: every 16th row plus the last
        rows = sorted(set(list(range(0, b["fracs"] + 1, 16)) + [b["fracs"]]))
        tabs["poly%d/rows" % i] = np.array(rows, dtype=np.int32)
        tabs["poly%d/table" % i] = t[rows]
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tabs)

    inlen = {}
    for name, (src, dst, maxin, chunk, n, tb, att, seed) in CASES.items():
        r = R.RefResampler(src, dst, maxin, tb, att)
        ns = list(range(0, 65)) + [100, 1000, 4096, 12345, 100000]
        inlen[name] = {"n": ns, "inlen": [r.input_required(k) for k in ns],
                       "max_out_len": r.maxout}
    with open(os.path.join(HERE, "inlen.json"), "w") as f:
        json.dump(inlen, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
