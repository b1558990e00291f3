#!/bin/bash
# tools/matrix_probe.sh <outdir>: cfg2-sized batch (1024 ch x 16384) over ratios x filter settings; one
# line per run with the kernels' times -- finds topologies left on slow paths
out=gpurun_out/$1; mkdir -p $out; rm -f $out/bench.txt
# MAT_FILTERS="tb att;tb att;..." selects the filter settings
if [ -n "$MAT_FILTERS" ]; then IFS=';' read -ra FL <<< "$MAT_FILTERS"; else FL=("2.0 180.15" "2.0 109.56" "10.0 109.56" "45.0 49.0"); fi
for a in "${FL[@]}"; do
  for dir in "88200 44100" "176400 44100" "32000 48000" "48000 32000" "44100 132300" "96000 32000" "64000 48000" "44100 44101" "44100 176400" "48000 44100"; do
    set -- $a $dir
    timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu --tb $1 --atten $2 --src $3 --dst $4 $AB_ARGS 2>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tb %s att %s %s->%s' % ('$1','$2','$3','$4'), d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'], d['roofline']['path_frac'])" >> $out/bench.txt 2>&1
  done
done
cat $out/bench.txt
