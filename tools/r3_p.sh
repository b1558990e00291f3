#!/bin/bash
# (historical record of a round-3 experiment: it ran against the build of its own commit; variants, macros and the
# engine option "persist" it names were removed again -- DESIGN.md section 5, profiles/r03_experiments.txt)
# tools/r3_p.sh -- round 3, second session: GPU tier on the new build, then A/B of the headline configuration:
# variants/base.so (build of commit 1ef1b84), the new library with one workgroup per item (persist=0) and with
# persistent workgroups on work queues (default)
out=gpurun_out/r3p; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for rep in 1 2 3; do
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 2>>$out/err.log | line "base 400" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --opt persist=0 2>>$out/err.log | line "new persist=0 400" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 2>>$out/err.log | line "new persist=1 400" >> $out/bench.txt 2>&1
  R8B_PERSIST_PER_CU=3 timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 2>>$out/err.log | line "new persist=1 grid 3/CU 400" >> $out/bench.txt 2>&1
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 2>>$out/err.log | line "base 20" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 --opt persist=0 2>>$out/err.log | line "new persist=0 20" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 2>>$out/err.log | line "new persist=1 20" >> $out/bench.txt 2>&1
done
for c in cfg3 cfg5; do
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --config $c 2>>$out/err.log | line "base $c" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --config $c --opt persist=0 2>>$out/err.log | line "new persist=0 $c" >> $out/bench.txt 2>&1
  timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --config $c 2>>$out/err.log | line "new persist=1 $c" >> $out/bench.txt 2>&1
done
cat $out/bench.txt
tail -5 $out/err.log
