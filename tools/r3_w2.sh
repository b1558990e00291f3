#!/bin/bash
# tools/r3_w2.sh -- the register tail on the plain pair convolvers (2x up, 2x down in the spectrum, 8192-point 2x up / 3):
# GPU tier, a wide GPU fuzz, then A/B against variants/base.so (the library of the commit before) on one box
out=gpurun_out/r3w2; mkdir -p $out; rm -f $out/*
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
timeout 200 python tools/gpu_fuzz.py 250 9191 wide 2>&1 | tail -2
for args in "--src 44100 --dst 88200" "--src 88200 --dst 44100" "--src 48000 --dst 32000"; do
  for r in 1 2; do
    for lib in variants/base.so r8brain-free-src_amd/libr8bsrc_hip.so; do
      R8B_HIP_LIB=$PWD/$lib timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu $args 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$args', '$lib'[:13], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
    done
  done
done
