#!/usr/bin/env python3
"""tools/profile_collect.py <tag> <round prefix> : copies the summaries tools/profile_all.sh left under
gpurun_out/prof_<tag>/ into profiles/ (tracked) as <prefix>_<config>_{kernel_stats.csv,pmc_summary.txt,
bench_20.json,bench_400.json} and merges the HBM traffic records into profiles/traffic.json under
"<config>:<kernel name as bench.py reports it>"."""
import json, os, shutil, sys
tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
tpath = os.path.join(dst, "traffic.json")
try:
    traffic = json.load(open(tpath))
except (OSError, ValueError):
    traffic = {}
bench_cfg = {"cfg2": "cfg2", "cfg3": "cfg3", "cfg5": "cfg5"}
for cfg in sorted(os.listdir(src)):
    d = os.path.join(src, cfg)
    if not os.path.isdir(d):
        continue
    for f in ("kernel_stats.csv", "pmc_summary.txt", "bench_20.json", "bench_400.json", "rates.txt"):
        if os.path.exists(os.path.join(d, f)):
            shutil.copy(os.path.join(d, f), os.path.join(dst, "%s_%s_%s" % (prefix, cfg, f)))
    try:
        rec = json.load(open(os.path.join(d, "traffic.json")))
    except (OSError, ValueError):
        continue
    ckey = bench_cfg.get(cfg, cfg)
    for key in [key for key in traffic if key.startswith(ckey + ":")]:
        del traffic[key]
    for k, v in rec.items():
        v["profile"] = "%s_%s_pmc_summary.txt" % (prefix, cfg)
        traffic["%s:%s" % (ckey, v.get("kernel", k))] = v
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print("profiles/traffic.json:", sorted(traffic))
