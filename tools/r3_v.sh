#!/bin/bash
# tools/r3_v.sh -- GPU tier + headline numbers of the new full build against variants/base.so (commit 1ef1b84)
out=gpurun_out/r3v; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-32s' % sys.argv[1], d['value'], d['ms_per_step'], d.get('first_window',{}).get('ms_per_step'), d['roofline']['kernels_ms_per_step'])" "$1"; }
for rep in 1 2; do
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --settle 0 2>>$out/err.log | line "base 400" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --settle 0 2>>$out/err.log | line "new 400" >> $out/bench.txt 2>&1
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 2>>$out/err.log | line "base 20 (settle 60)" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 2>>$out/err.log | line "new 20 (settle 60)" >> $out/bench.txt 2>&1
done
for c in cfg3 cfg5; do
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --settle 0 --config $c 2>>$out/err.log | line "base $c" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --settle 0 --config $c 2>>$out/err.log | line "new $c" >> $out/bench.txt 2>&1
done
cat $out/bench.txt
tail -3 $out/err.log
