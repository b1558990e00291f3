#!/bin/bash
# tools/r3_y.sh -- A/B of one-kernel development builds (tools/variant.sh) on one box, alternating:
# w0 / x0: cfg2 / cfg3 with every twiddle base power fetched (-DR8B_TW_DERIVE=0); w1 / x1: first / last pass powers derived
for r in 1 2 3; do
AB_ARGS="" bash tools/ab2.sh r3y_c2_$r w0 w1 >/dev/null
AB_ARGS="--config cfg3" bash tools/ab2.sh r3y_c3_$r x0 x1 >/dev/null
done
for r in 1 2 3; do cat gpurun_out/r3y_c2_$r/bench.txt; done
for r in 1 2 3; do cat gpurun_out/r3y_c3_$r/bench.txt; done
