#!/bin/bash
# tools/r3_y.sh -- A/B of the late history-tail copy (loads behind the last wave-local pass, stores behind the last backward pass)
for r in 1 2 3; do
AB_ARGS="" bash tools/ab2.sh r3y_c2_$r b2 d2 >/dev/null
AB_ARGS="--config cfg3" bash tools/ab2.sh r3y_c3_$r b3 d3 >/dev/null
done
for r in 1 2 3; do cat gpurun_out/r3y_c2_$r/bench.txt; done
for r in 1 2 3; do cat gpurun_out/r3y_c3_$r/bench.txt; done
