#!/bin/bash
out=gpurun_out/r3m; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for rep in 1 2 3; do
for v in hot; do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 2>>$out/err.log | line "$v 20" >> $out/bench.txt 2>&1
done
done
R8B_HIP_LIB=$PWD/variants/hot.so timeout 120 python tools/dbg_parity.py 44100 96000 6 16384 3 >> $out/bench.txt 2>&1
cat $out/bench.txt
R8B_HIP_LIB=$PWD/variants/sthot.so timeout 120 python tools/stamps_probe.py 2>>$out/err.log | grep "wave [0]"
