#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out; rm -f $out/*
for v in $(ls variants | sed 's/\.so//'); do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python tools/stamps_probe.py $PARGS 2>>$out/err.log > $out/$v.txt
  cat $out/$v.txt
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 200 --warmup 20 $BARGS 2>>$out/err.log | tail -1 | cut -c1-200
done
