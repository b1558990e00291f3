#!/bin/bash
# (historical record of a round-3 experiment: it ran against the build of its own commit; variants, macros and the
# engine option "persist" it names were removed again -- DESIGN.md section 5, profiles/r03_experiments.txt)
# tools/r3_w.sh -- persistent form with the work queue chosen by the hardware's XCC id (and the statistics of how often
# blockIdx mod 8 is NOT the XCD)
out=gpurun_out/r3w; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-32s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for rep in 1 2; do
  R8B_HIP_LIB=$PWD/variants/lean.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --settle 0 2>>$out/err.log | line "lean" >> $out/bench.txt 2>&1
  for v in persist5 persist6; do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --settle 0 --opt persist=1 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
  done
done
cat $out/bench.txt; grep "k_convp persistent" $out/err.log | tail -2
