#!/bin/bash
# tools/ab.sh <outdir> <variant names...> : bench every variants/<name>.so on the default workload
# (and the convolver-only 44100->88200 side workload), one line each
out=gpurun_out/$1; shift
mkdir -p $out
run() {
  lib=$1; shift
  R8B_HIP_LIB=$PWD/variants/$lib.so timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-12s %-40s' % ('$lib', '$*'), d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $out/bench.txt 2>&1
}
for v in "$@"; do
  run $v
  run $v --src 44100 --dst 88200
  run $v --src 96000 --dst 44100
done
cat $out/bench.txt
