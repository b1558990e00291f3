#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out; rm -f $out/*
R8B_HIP_LIB=$PWD/variants/st3000.so timeout 120 python tools/stamps_probe.py 2>>$out/err.log > $out/st3000.txt
cat $out/st3000.txt
R8B_HIP_LIB=$PWD/variants/st0.so timeout 120 python tools/stamps_probe.py 44100 88200 2>>$out/err.log > $out/st0.txt
cat $out/st0.txt
