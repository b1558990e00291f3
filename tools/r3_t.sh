#!/bin/bash
# (historical record of a round-3 experiment: it ran against the build of its own commit; variants, macros and the
# engine option "persist" it names were removed again -- DESIGN.md section 5, profiles/r03_experiments.txt)
# tools/r3_t.sh -- persistent workgroups with the second workgroup of every CU started late (R8B_STAGGER sleep periods)
out=gpurun_out/r3t; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
run() { local name=$1 lib=$2; shift; shift
  env "$@" R8B_HIP_LIB=$PWD/variants/$lib.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --opt persist=1 2>>$out/err.log | line "$name" >> $out/bench.txt 2>&1; }
for rep in 1 2; do
  run lean lean
  for s in 0 1 2 3 4 6; do run "persist stagger $s" persist R8B_STAGGER=$s; done
done
cat $out/bench.txt; tail -3 $out/err.log
