#!/bin/bash
# round 6, half-array form of the decimating 4096 -> 2048-point geometry (kernel modes 27 / 28): parity, then A/B
mkdir -p gpurun_out/r6h
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "half_array" 2>&1 | tail -4
run() { name=$1; shift; timeout 500 python tools/ab.py --out gpurun_out/r6h/$name --reps 3 --steps 200 "$@" full:opt=half=0 half > gpurun_out/r6h/$name.log 2>&1; echo "== $name"; tail -3 gpurun_out/r6h/$name.log; }
run down2 --bench-args "--src 88200 --dst 44100"
run hbdown --bench-args "--src 176400 --dst 44100"
run r32 --bench-args "--src 32000 --dst 48000 --tb 3"
