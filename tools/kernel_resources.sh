#!/bin/bash
# tools/kernel_resources.sh [out.txt] : registers, scratch and LDS of EVERY kernel of the shipped library, from the
# compiler's own resource remarks (-Rpass-analysis=kernel-resource-usage) over a side build of the same translation
# units with the same flags (objects in csrc/_obj_res, no library kept).  One line per kernel, spilling ones first.
out=${1:-/dev/stdout}
case "$out" in /*) ;; *) out="$PWD/$out" ;; esac
cd "$(dirname "$0")/../r8brain-free-src_amd/csrc"
make -j8 OBJ=_obj_res OUT=/tmp/_r8b_res.so EXTRA_HIPFLAGS=-Rpass-analysis=kernel-resource-usage all > /tmp/_r8b_res.log 2>&1 || { tail -n 20 /tmp/_r8b_res.log; exit 1; }
rm -rf _obj_res /tmp/_r8b_res.so
python3 - /tmp/_r8b_res.log > "$out" <<'PY'
import re, subprocess, sys
rows, cur = {}, None
for line in open(sys.argv[1]):
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = rows.setdefault(m.group(1), {})
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|SGPRs Spill|VGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy)[^:]*: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
table = []
for n, d in zip(names, dem):
    r = rows[n]
    if "VGPRs" not in r:
        continue
    d = d.replace("void ", "").replace("r8bhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
    table.append((-(r.get("ScratchSize", 0)), d, r))
table.sort(key=lambda t: (t[0], t[1]))
print("# scratch = bytes per lane (VGPR spills and arrays the compiler left in memory); sspill = SGPRs spilled to VGPR lanes;")
print("# occ = waves per SIMD the REGISTERS allow (the pair kernels' 64 KB dynamic LDS array bounds them at two workgroups per CU)")
print("%-56s %5s %5s %8s %7s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "sspill", "occ"))
for _, d, r in table:
    print("%-56s %5d %5d %8d %7d %4d" % (d[:56], r["VGPRs"], r.get("TotalSGPRs", 0), r.get("ScratchSize", 0),
        r.get("SGPRs Spill", 0), r.get("Occupancy", 0)))
print("# %d kernels, %d with scratch" % (len(table), sum(1 for t in table if t[0] < 0)))
PY
