#!/usr/bin/env python3
"""tools/headline_numbers.py [prefix] : the numbers DESIGN.md section 5 / README.md quote for BASELINE's cfg2, read off
profiles/<prefix>_cfg2_{bench_20.json,bench_400.json,kernel_stats.csv} (nothing is written; paste by hand)."""
import csv, json, os, re, sys
prefix = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for n in ("20", "400"):
    l = json.loads(open(os.path.join(root, "%s_cfg2_bench_%s.json" % (prefix, n))).read().strip().split("\n")[-1])
    r = l["roofline"]
    print("| %s | %.0f (%.4f) | %.0f (%.4f) | (%.4f) | %.4f | %.3f | %.3f |" % (
        "driver style, 5 + 20" if n == "20" else "40 + 400", l["value"], l["ms_per_step"], l["settled"]["value"],
        l["settled"]["ms_per_step"], l["other_placement"]["ms_per_step"], r["avg_kernel_ms"], r["frac"], r["path_frac"]))
for row in csv.DictReader(open(os.path.join(root, "%s_cfg2_kernel_stats.csv" % prefix))):
    if "k_convp" in row["Name"]:
        us = float(row["AverageNs"]) / 1e3
        print(re.search(r"k_\w+<[^>]*>", row["Name"]).group(0), "%.1f us x %s -> %.3f of 8 TB/s" % (us, row["Calls"], 426.392371e6 / (us * 1e-6) / 8e12))
