#!/bin/bash
out=gpurun_out/r3b; mkdir -p $out; rm -f $out/*
for v in b5 w5nopf w5; do
  for nch in 4 1024; do
    echo "== $v nch $nch" >> $out/dbg.txt
    R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python tools/dbg_parity.py 96000 44100 $nch 16384 3 >> $out/dbg.txt 2>&1
  done
done
for v in base w4; do
  echo "== $v cfg2 nch 6" >> $out/dbg.txt
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python tools/dbg_parity.py 44100 96000 6 16384 3 >> $out/dbg.txt 2>&1
done
cat $out/dbg.txt
R8B_HIP_LIB=$PWD/variants/base.so tools/pmc4.sh r3b_base > $out/pmc_base.txt 2>&1
R8B_HIP_LIB=$PWD/variants/w4.so tools/pmc4.sh r3b_w4 > $out/pmc_w4.txt 2>&1
cat $out/pmc_base.txt
