#!/usr/bin/env python3
"""tools/isa_sections.py <mangled-kernel-substring> : static instruction-class counts of one kernel of
r8b_kernels.hip per barrier-separated section (device assembly via hipcc -S)."""
import collections, re, subprocess, sys
src = "/root/repo/r8brain-free-src_amd/csrc/r8b_kernels.hip"
extra = sys.argv[2:]
subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden",
                "-S", "--cuda-device-only", "-o", "/tmp/k_dev.s", src] + extra, check=True, stderr=subprocess.DEVNULL)
lines = open("/tmp/k_dev.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[1] in l and l.rstrip().split(":")[0].endswith("E"))
body = []
for l in lines[start:]:
    body.append(l)
    if "s_endpgm" in l:
        break
open("/tmp/kernel.s", "w").write("\n".join(body))
sec = 0
cnt = collections.defaultdict(collections.Counter)
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op == "s_barrier":
        sec += 1
        continue
    if op.startswith("v_") and "f64" in op: c = "fp64"
    elif op.startswith("v_"): c = "valu_other"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"): c = "wait"
    elif op.startswith("s_"): c = "salu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "scratch_", "buffer")): c = "vmem"
    else: c = "other"
    cnt[sec][c] += 1
print("%-4s %6s %10s %6s %6s %6s %6s" % ("sec", "fp64", "valu_other", "salu", "lds", "vmem", "wait"))
tot = collections.Counter()
for s in sorted(cnt):
    c = cnt[s]; tot.update(c)
    print("%-4d %6d %10d %6d %6d %6d %6d" % (s, c["fp64"], c["valu_other"], c["salu"], c["lds"], c["vmem"], c["wait"]))
print("tot  %6d %10d %6d %6d %6d %6d" % (tot["fp64"], tot["valu_other"], tot["salu"], tot["lds"], tot["vmem"], tot["wait"]))
