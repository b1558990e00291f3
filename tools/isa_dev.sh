#!/bin/bash
# tools/isa_dev.sh <mode> [LN UL] [extra flags...] : device assembly of the one pair-kernel instance of a development
# build (see tools/variant.sh) -> /tmp/kdev_<mode>.s, plus the static instruction-class counts per barrier section
mode=$1; LN=${2:-11}; UL=${3:-1}; shift; shift; shift
cd "$(dirname "$0")/../r8brain-free-src_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -DR8B_DEV_GEOMS \
  "-DR8B_CONVP_GEOMS(M)=M($LN,$UL)" "-DR8B_CONVP_GEOMS_BIG(M)=" "-DR8B_CONVP_GEOMS_DOWN(M)=" "-DR8B_CONVX_GEOMS(M)=" \
  "-DR8B_CONVX_GEOMS_DOWN(M)=" -DR8B_DEV_ONLY_MODE=$mode "$@" -S --cuda-device-only -o /tmp/kdev_$mode.s r8b_kernels.hip || exit 1
python3 - /tmp/kdev_$mode.s <<'PY'
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and "k_convp" in l and re.match(r"^_Z\w+:", l))
body = []
for l in lines[start:]:
    body.append(l)
    if "s_endpgm" in l:
        break
open(sys.argv[1].replace(".s", "_kernel.s"), "w").write("\n".join(body))
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
    elif op.startswith("s_load") or op.startswith("s_buffer_load"): c = "smem"
    elif op.startswith("s_"): c = "salu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "scratch_", "buffer")): c = "vmem"
    else: c = "other"
    cnt[sec][c] += 1
print("%-4s %6s %10s %6s %6s %6s %6s %6s" % ("sec", "fp64", "valu_other", "salu", "smem", "lds", "vmem", "wait"))
tot = collections.Counter()
for s in sorted(cnt):
    c = cnt[s]; tot.update(c)
    print("%-4d %6d %10d %6d %6d %6d %6d %6d" % (s, c["fp64"], c["valu_other"], c["salu"], c["smem"], c["lds"], c["vmem"], c["wait"]))
print("tot  %6d %10d %6d %6d %6d %6d %6d" % (tot["fp64"], tot["valu_other"], tot["salu"], tot["smem"], tot["lds"], tot["vmem"], tot["wait"]))
for l in lines:
    if "k_convp" in l and re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size)", l):
        print(l.strip())
PY
