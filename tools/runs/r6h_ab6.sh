#!/bin/bash
# round 6, half-array forms with a complex kernel spectrum (minimum-phase chains; kernel modes 29 - 32): parity, then A/B
mkdir -p gpurun_out/r6h
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -k "half_array or minphase or MINPHASE" 2>&1 | tail -4
run() { name=$1; shift; timeout 500 python tools/ab.py --out gpurun_out/r6h/$name --reps 3 --steps 200 "$@" > gpurun_out/r6h/$name.log 2>&1; echo "== $name"; tail -4 gpurun_out/r6h/$name.log; }
run mp96 --bench-args "--phase 1" walk:opt=half_fused=0 oneblock:opt=half_fused=0,walk=0 half
run mp88 --bench-args "--src 44100 --dst 88200 --phase 1" full:opt=half=0 half
