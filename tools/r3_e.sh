#!/bin/bash
out=gpurun_out/r3e; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for lds in 65600 53000 40000 32000; do
  R8B_FAKE_LDS=$lds R8B_HIP_LIB=$PWD/variants/m0.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 --src 44100 --dst 88200 2>>$out/err.log | line "m0 lds $lds" >> $out/bench.txt 2>&1
done
for lds in 65600 53000; do
  R8B_FAKE_LDS=$lds R8B_HIP_LIB=$PWD/variants/m4.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 2>>$out/err.log | line "m4 lds $lds" >> $out/bench.txt 2>&1
done
cat $out/bench.txt; tail -3 $out/err.log
