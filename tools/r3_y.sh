#!/bin/bash
# tools/r3_y.sh -- block-major workgroup order (-DR8B_BG_MAJOR=1): GPU tier on the full library built with it
# (variants/full1.so), then A/B of one-kernel builds on the same box (m0 / n0: pair-major as shipped, m1 / n1: block-major)
R8B_HIP_LIB=$PWD/variants/full1.so timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
AB_ARGS="" bash tools/ab2.sh r3y_c2_1 m0 m1 m0 m1 >/dev/null
AB_ARGS="--config cfg3" bash tools/ab2.sh r3y_c3_1 n0 n1 >/dev/null
cat gpurun_out/r3y_c2_1/bench.txt gpurun_out/r3y_c3_1/bench.txt
