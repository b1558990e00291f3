#!/bin/bash
# full-library check: GPU tier, smoke, driver-style bench line, side configurations
out=gpurun_out/r3k; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default_20.json 2>$out/bench_err.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
run() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>>$out/err.log | line "$*" >> $out/bench.txt 2>&1; }
run
run --config cfg3
run --config cfg5
run --src 44100 --dst 88200
run --src 88200 --dst 44100
run --src 44100 --dst 2822400 --block 1024 --channels 1024
run --src 44100 --dst 44101
run --src 48000 --dst 32000
run --src 176400 --dst 44100
run --src 44100 --dst 96000 --atten 109.56
tail -5 $out/pytest.log; tail -1 $out/smoke.log; cat $out/bench_default_20.json; cat $out/bench.txt
for ch in 256 1024 2048; do
  run --src 44100 --dst 44101 --channels $ch
  run --src 44100 --dst 44101 --channels $ch --opt poly_groups=0
done
tail -8 $out/bench.txt
