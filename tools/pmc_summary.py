#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV passes (tools/pmc.sh, tools/profile_all.sh): per kernel, the mean counter value per dispatch,
then -- where the counters needed are present -- derived ratios WITH their formulas, so that every percentage quoted
in DESIGN.md can be recomputed from this file.

Units (established by the calibration kernels of tools/ubench/pmc_calib.hip, printed below when collected): the SQ
*_CYCLES / ACTIVE_INST_* / WAIT_* counters count in units of FOUR shader cycles, summed over waves (per-wave
counters) or over CUs; a full-rate VALU instruction -- v_fma_f64 included on gfx950 -- occupies the vector ALU for
exactly one such unit, which is why SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU for kernels without quarter-rate
instructions (it is 4x SQ_INSTS_VALU for the v_rcp_f64 calibration kernel).  GRBM_GUI_ACTIVE is in cycles, summed
over the 8 XCDs."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "[pc]*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            m = re.search(r"(k_\w+)(<[^>]*>)?", name)
            if not m:
                continue
            short = m.group(1) + (m.group(2) or "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))


def mean(k, c):
    v = acc[k].get(c)
    return sum(v) / len(v) if v else None


for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-26s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    g = lambda c: mean(k, c)
    d = []
    waves, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
    if waves and wc:
        d.append(("wave lifetime, cycles", 4.0 * wc / waves, "4 * SQ_WAVE_CYCLES / SQ_WAVES"))
        for c, what in (("SQ_ACTIVE_INST_ANY", "an instruction in flight"), ("SQ_WAIT_ANY", "in s_waitcnt"),
                        ("SQ_WAIT_INST_ANY", "waiting to issue")):
            if g(c):
                d.append(("  of which " + what, g(c) / wc, "%s / SQ_WAVE_CYCLES" % c))
    gui = g("GRBM_GUI_ACTIVE")
    kc = gui / 8.0 if gui else None  # kernel duration in shader cycles
    if kc:
        d.append(("kernel duration, cycles", kc, "GRBM_GUI_ACTIVE / 8 XCDs"))
    if waves and g("SQ_INSTS_VALU"):
        d.append(("VALU instructions per wave", g("SQ_INSTS_VALU") / waves, "SQ_INSTS_VALU / SQ_WAVES"))
        f64 = [g(c) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                              "SQ_INSTS_VALU_TRANS_F64")]
        if all(x is not None for x in f64[:3]):
            tot = sum(x or 0.0 for x in f64)
            d.append(("  fp64 per wave (fma / add / mul)", tot / waves,
                      "(FMA_F64 %.0f + ADD_F64 %.0f + MUL_F64 %.0f) / SQ_WAVES" % tuple(x / waves for x in f64[:3])))
        for c, what in (("SQ_INSTS_SALU", "SALU"), ("SQ_INSTS_SMEM", "scalar loads"), ("SQ_INSTS_LDS", "LDS"),
                        ("SQ_INSTS_VMEM_RD", "vector loads"), ("SQ_INSTS_VMEM_WR", "vector stores")):
            if g(c) is not None:
                d.append(("%s instructions per wave" % what, g(c) / waves, "%s / SQ_WAVES" % c))
    if kc and g("SQ_ACTIVE_INST_VALU"):
        d.append(("vector ALU busy (all 1024 SIMDs)", 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * kc),
                  "4 * SQ_ACTIVE_INST_VALU / (1024 * kernel cycles)"))
        if g("SQ_INSTS_VALU"):
            d.append(("units of 4 cycles per VALU instruction", g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"),
                      "SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU"))
    if kc and g("SQ_BUSY_CU_CYCLES"):
        d.append(("CU time with a wave resident", g("SQ_BUSY_CU_CYCLES") / (256.0 * kc),
                  "SQ_BUSY_CU_CYCLES / (256 * kernel cycles)  [this one counts cycles: 0.98 for the calibration kernels]"))
    if kc and g("SQ_LDS_IDX_ACTIVE"):
        d.append(("LDS busy (all 256 CUs)", g("SQ_LDS_IDX_ACTIVE") / (256.0 * kc),
                  "SQ_LDS_IDX_ACTIVE / (256 * kernel cycles)"))
        if g("SQ_LDS_BANK_CONFLICT") is not None:
            d.append(("  of which bank-conflict replays", g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
                      "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"))
    if wc and g("SQ_WAIT_INST_LDS"):
        d.append(("wave time waiting on LDS", g("SQ_WAIT_INST_LDS") / wc, "SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES"))
    if wc and g("SQ_INST_CYCLES_VMEM_RD") is not None:
        d.append(("vector-memory issue share of wave time",
                  (g("SQ_INST_CYCLES_VMEM_RD") + (g("SQ_INST_CYCLES_VMEM_WR") or 0.0)) / wc,
                  "(SQ_INST_CYCLES_VMEM_RD + _WR) / SQ_WAVE_CYCLES"))
    if g("SQ_IFETCH") and g("SQ_INSTS_VALU"):
        d.append(("instruction fetches per wave", g("SQ_IFETCH") / waves if waves else 0.0, "SQ_IFETCH / SQ_WAVES"))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d.append(("HBM bytes per launch", (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0,
                  "(2 * FETCH_SIZE + WRITE_SIZE) KB  [FETCH_SIZE doubled on gfx950: MI355X_MICROARCH.md, HBM]"))
    if d:
        print("   -- derived")
        for what, v, how in d:
            print("   %-40s %14.4g   = %s" % (what, v, how))

# HBM traffic record for bench.py (rocprofv3 FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is
# doubled on gfx950 per MI355X_MICROARCH.md section HBM)
rec = {}
for k in acc:
    if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
        w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
        # (keyed by the device symbol, as bench.py's roofline.kernel and profiles/traffic.json are)
        rec[k] = {"kernel": k, "fetch_kb": f, "write_kb": w,
                      "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
with open(os.path.join(root, "traffic.json"), "w") as fh:
    json.dump(rec, fh, indent=1, sort_keys=True)
