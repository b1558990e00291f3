#!/bin/bash
out=gpurun_out/r3g; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for v in $(ls variants | sed 's/\.so//'); do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
done
cat $out/bench.txt
