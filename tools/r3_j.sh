#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
for rep in 1 2; do
for v in $(ls variants | sed 's/\.so//'); do
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 300 --warmup 30 2>>$out/err.log | line "$v" >> $out/bench.txt 2>&1
  R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python bench.py --no-cpu --steps 20 --warmup 5 2>>$out/err.log | line "$v 20" >> $out/bench.txt 2>&1
done
done
for v in $(ls variants | sed 's/\.so//'); do
  R8B_DEBUG_TWO=1 R8B_HIP_LIB=$PWD/variants/$v.so timeout 120 python tools/dbg_parity.py 44100 96000 6 16384 3 >> $out/bench.txt 2>&1
done
cat $out/bench.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
R8B_HIP_LIB=$R/variants/q4.so timeout 120 rocprofv3 --kernel-trace --pmc TCP_TCC_WRITE_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$out/p1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu > $R/$out/p1.log 2>&1
cd $R
python tools/pmc_summary.py $out | head -8
