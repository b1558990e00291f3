#!/bin/bash
out=gpurun_out/r3o; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-60s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for rep in 1 2; do
for v in tb10 tb10ns; do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 --tb 10 --atten 109.56 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
done
for v in ir16 ir16ns; do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 --atten 109.56 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
done
done
cat $out/bench.txt
