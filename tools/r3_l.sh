#!/bin/bash
out=gpurun_out/r3l; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-50s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for ch in 128 256 512 1024 2048; do
  timeout 120 python bench.py --no-cpu --steps 200 --warmup 20 --src 44100 --dst 44101 --channels $ch 2>>$out/err.log | line "poly ch $ch" >> $out/bench.txt 2>&1
done
cat $out/bench.txt
