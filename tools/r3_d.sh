#!/bin/bash
out=gpurun_out/r3d; mkdir -p $out; rm -f $out/*
tools/ubench/_build/store_coalesce > $out/store.txt 2>&1
cat $out/store.txt
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
timeout 120 rocprofv3 --kernel-trace --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/$out/pmcs -- $R/tools/ubench/_build/store_coalesce > $R/$out/pmcs.log 2>&1
cd $R
python tools/pmc_summary.py $out/pmcs/.. 2>/dev/null | grep -A3 "k_store" | head -40
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3d/pmcs/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"][:60],row["Counter_Name"])].append(float(row["Counter_Value"]))
for k in sorted(acc): print(k, sum(acc[k])/len(acc[k]))
PY
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for v in abl0 abl32 abl33; do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
  R8B_TWO_ORDER=1 R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 2>>$out/err.log | line "$v natural order" >> $out/bench.txt 2>&1
done
R8B_TWO_ORDER=1 R8B_HIP_LIB=$PWD/variants/abl0.so timeout 120 python tools/dbg_parity.py 44100 96000 6 16384 3 >> $out/bench.txt 2>&1
cat $out/bench.txt
