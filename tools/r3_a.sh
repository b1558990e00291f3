#!/bin/bash
# tools/r3_a.sh : round 3, first GPU pass -- GPU tier + smoke on the walker build, then A/B of the headline kernel
out=gpurun_out/r3a; mkdir -p $out; rm -f $out/*
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-60s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
run() { # <label> <lib or ''> args...
  lab=$1; lib=$2; shift; shift
  if [ -n "$lib" ]; then export R8B_HIP_LIB=$PWD/variants/$lib.so; else unset R8B_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu "$@" 2>>$out/err.log | line "$lab $*" >> $out/bench.txt 2>&1
}
for rep in 1 2; do
for v in "" base nopf; do
  for al in 1 0; do
    run "[$v]" "$v" --steps 20 --warmup 5 --opt align_groups=$al
    run "[$v]" "$v" --steps 400 --warmup 20 --opt align_groups=$al
  done
done
done
unset R8B_HIP_LIB
for cfg in "--config cfg3" "--config cfg5" "--src 44100 --dst 88200" "--src 88200 --dst 44100" "--src 44100 --dst 2822400 --block 1024 --channels 1024" "--src 44100 --dst 44101" "--src 48000 --dst 32000" "--channels 512" "--channels 600" "--channels 64"; do
  run "[full]" "" --steps 200 --warmup 20 $cfg
done
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default_20.json 2>>$out/err.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --steps 50 --warmup 10 --no-cpu > $R/$out/prof.log 2>&1
cd $R
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
tools/pmc4.sh r3a_cfg2 > $out/pmc.log 2>&1
cp gpurun_out/pmc_r3a_cfg2/traffic.json $out/traffic_cfg2.json 2>/dev/null
tail -3 $out/pytest.log; tail -1 $out/smoke.log; cat $out/bench.txt; cat $out/bench_default_20.json; cat $out/kernel_stats.csv | head -5; cat $out/pmc.log | tail -30
