#!/bin/bash
# (historical record of a round-3 experiment: it ran against the build of its own commit; variants, macros and the
# engine option "persist" it names were removed again -- DESIGN.md section 5, profiles/r03_experiments.txt)
# tools/r3_u.sh -- wave priority by phase (s_setprio at the start of an item / before the last backward pass / before
# the interpolator): does asymmetry between the two workgroups of a CU matter?
out=gpurun_out/r3u; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
run() { local name=$1 lib=$2; shift; shift
  env "$@" R8B_HIP_LIB=$PWD/variants/$lib.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 2>>$out/err.log | line "$name" >> $out/bench.txt 2>&1; }
for rep in 1 2; do
  for v in lean p003 p033 p330 p300 p123 p321 p111; do run $v $v; done
done
cat $out/bench.txt; tail -3 $out/err.log
