#!/bin/bash
# tools/ab2.sh <outdir> <variant names...> : default workload only, 200 steps
out=gpurun_out/$1; shift
mkdir -p $out; rm -f $out/bench.txt
for v in "$@"; do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu $AB_ARGS 2>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-12s' % '$v', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $out/bench.txt 2>&1
done
cat $out/bench.txt
