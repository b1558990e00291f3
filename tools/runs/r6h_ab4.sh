#!/bin/bash
# round 6, half-array form on the 4096 -> 8192-point geometry (two workgroups of 512 threads per CU) and behind the strided store
mkdir -p gpurun_out/r6h
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "half_array" 2>&1 | tail -4
run() { name=$1; shift; timeout 500 python tools/ab.py --out gpurun_out/r6h/$name --reps 3 --steps 200 "$@" full:opt=half=0 half > gpurun_out/r6h/$name.log 2>&1; echo "== $name"; tail -3 gpurun_out/r6h/$name.log; }
run r23 --bench-args "--src 48000 --dst 32000"
run up2tb1 --bench-args "--src 44100 --dst 88200 --tb 1"
run r23s --bench-args "--src 48000 --dst 32000 --tb 3 --atten 150"
run up2 --bench-args "--src 44100 --dst 88200"
