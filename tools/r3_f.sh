#!/bin/bash
out=gpurun_out/r3f; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for v in $(ls variants | sed 's/\.so//'); do
for lds in 65600 40000; do
  R8B_FAKE_LDS=$lds R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 --src 44100 --dst 88200 2>>$out/err.log | line "$v lds $lds" >> $out/bench.txt 2>&1
done
done
cat $out/bench.txt
export R8B_HIP_LIB=$PWD/variants/m0.so
R8B_FAKE_LDS=40000 tools/pmc4.sh r3f_m0_4wg --src 44100 --dst 88200 > $out/pmc_m0_4wg.txt 2>&1
R8B_FAKE_LDS=65600 tools/pmc4.sh r3f_m0_2wg --src 44100 --dst 88200 > $out/pmc_m0_2wg.txt 2>&1
awk '/^k_convp</,0' $out/pmc_m0_4wg.txt | grep -A40 derived
awk '/^k_convp</,0' $out/pmc_m0_2wg.txt | grep -A40 derived
