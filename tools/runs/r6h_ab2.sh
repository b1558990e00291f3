#!/bin/bash
# round 6, half-array form: A/B over the configurations whose chain holds the 2048 -> 4096-point convolver-only block pair
mkdir -p gpurun_out/r6h
run() { name=$1; shift; timeout 400 python tools/ab.py --out gpurun_out/r6h/$name --reps 3 --steps 200 "$@" full half:opt=half=1 > gpurun_out/r6h/$name.log 2>&1; echo "== $name"; tail -3 gpurun_out/r6h/$name.log; }
run cfg5 --config cfg5
run cfg5x1024 --bench-args "--src 44100 --dst 2822400 --block 1024 --channels 1024"
run poly --bench-args "--src 44100 --dst 44101"
run up2_64 --bench-args "--src 44100 --dst 88200 --channels 64"
run up2_256 --bench-args "--src 44100 --dst 88200 --channels 256"
run up2_4096 --bench-args "--src 44100 --dst 88200 --channels 4096"
