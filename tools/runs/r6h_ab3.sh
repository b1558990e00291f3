#!/bin/bash
# round 6, half-array form with the interpolator fused in (kernel modes 23 / 25): parity, then A/B against the walk form
mkdir -p gpurun_out/r6h
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "half_array" 2>&1 | tail -4
run() { name=$1; shift; timeout 500 python tools/ab.py --out gpurun_out/r6h/$name --reps 3 --steps 200 "$@" > gpurun_out/r6h/$name.log 2>&1; echo "== $name"; tail -4 gpurun_out/r6h/$name.log; }
run hf_cfg2 walk oneblock:opt=walk=0 hfused:opt=half_fused=2
run hf_48k --bench-args "--src 44100 --dst 48000" base hfused:opt=half_fused=2
