#!/bin/bash
# tools/short_probe.sh <outdir>: cfg2-sized batch with the shorter filters of the reference's other presets
out=gpurun_out/$1; mkdir -p $out; rm -f $out/bench.txt
for a in "2.0 180.15" "2.0 136.45" "2.0 109.56" "5.0 109.56" "5.0 136.45" "10.0 109.56" "45.0 49.0"; do
  set -- $a
  for dir in "44100 96000" "96000 44100"; do
    set -- $a $dir
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu --tb $1 --atten $2 --src $3 --dst $4 $AB_ARGS 2>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tb %s att %s %s->%s' % ('$1','$2','$3','$4'), d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'], d['roofline']['frac'])" >> $out/bench.txt 2>&1
  done
done
cat $out/bench.txt
