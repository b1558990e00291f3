#!/usr/bin/env python3
"""tools/ab.py -- the one A/B driver for kernel experiments on the GPU box (replaces the one-off r2_*/r3_* scripts).

  python tools/ab.py --out gpurun_out/ab_park --reps 3 --steps 200 --config cfg2 \
      park1 park0:opt=park=0 old:lib=variants/old.so "diet:lib=variants/diet.so:opt=fold_tail=0"

Every variant is `name[:lib=<library>][:opt=<k=v>[,<k=v>...]][:args=<extra bench.py args>][:env=<K=V>[,<K=V>...]]`.  The variants run in
alternation (A B C A B C ...), `reps` times each, every run a fresh `python bench.py --steps S --warmup W --settle 0
--no-cpu` process on the same box, so that box-to-box differences (+-3 %) and clock drift cancel; the table gives
the median and the spread per variant.  Libraries come from tools/variant.sh (variants/<name>.so).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_variant(text):
    parts = text.split(":")
    v = {"name": parts[0], "lib": None, "opts": [], "args": [], "env": []}
    for p in parts[1:]:
        k, _, val = p.partition("=")
        if k == "lib":
            v["lib"] = val
        elif k == "opt":
            v["opts"] += val.split(",")
        elif k == "args":
            v["args"] += val.split()
        elif k == "env":
            v["env"] += val.split(",")
        else:
            raise SystemExit("variant field %r (lib= / opt= / args= / env=)" % p)
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default=None, help="bench.py --config (cfg2 / cfg3 / cfg5)")
    ap.add_argument("--bench-args", default="", help="extra bench.py arguments for every variant")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    variants = [parse_variant(t) for t in args.variants]
    rows = {v["name"]: [] for v in variants}
    log = open(os.path.join(args.out, "runs.jsonl"), "a")
    for rep in range(args.reps):
        for v in variants:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup",
                   str(args.warmup), "--settle", "0", "--no-cpu"]
            if args.config:
                cmd += ["--config", args.config]
            cmd += args.bench_args.split() + v["args"]
            for o in v["opts"]:
                cmd += ["--opt", o]
            env = dict(os.environ)
            for kv in v["env"]:
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            if v["lib"]:
                env["R8B_HIP_LIB"] = os.path.join(ROOT, v["lib"]) if not os.path.isabs(v["lib"]) else v["lib"]
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                print("%s rep %d FAILED rc %d: %s" % (v["name"], rep, r.returncode, r.stderr[-400:]), flush=True)
                continue
            d = json.loads(line[-1])
            rec = {"variant": v["name"], "rep": rep, "value": d["value"], "ms_per_step": d["ms_per_step"],
                   "kernels": d["roofline"]["kernels_ms_per_step"], "frac": d["roofline"]["frac"],
                   "path_frac": d["roofline"]["path_frac"]}
            log.write(json.dumps(rec) + "\n")
            log.flush()
            rows[v["name"]].append(rec)
            print("%-14s rep %d  %10.1f M in/s  %.4f ms/step  %s" % (v["name"], rep, d["value"], d["ms_per_step"],
                                                                    rec["kernels"]), flush=True)
    lines = ["%-14s %12s %10s %10s %8s  %s" % ("variant", "M in/s (med)", "ms/step", "min..max", "frac", "kernels ms (med)")]
    for v in variants:
        rs = rows[v["name"]]
        if not rs:
            continue
        ms = sorted(r["ms_per_step"] for r in rs)
        kn = {}
        for k in rs[0]["kernels"]:
            kn[k] = round(statistics.median(r["kernels"].get(k, 0.0) for r in rs), 4)
        lines.append("%-14s %12.1f %10.4f %4.4f..%.4f %8.4f  %s" % (
            v["name"], statistics.median(r["value"] for r in rs), statistics.median(ms), ms[0], ms[-1],
            statistics.median(r["frac"] for r in rs), kn))
    text = "\n".join(lines)
    print(text)
    with open(os.path.join(args.out, "summary.txt"), "a") as f:
        f.write("# %s\n%s\n" % (" ".join(sys.argv[1:]), text))


if __name__ == "__main__":
    main()
