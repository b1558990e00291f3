#!/bin/bash
# tools/r3_y.sh -- A/B of history-tail variants (b2 / b3: the tree before; q2 / q3: tail from the blocks' registers, the
# rest fetched beside the last block's samples; r2 / r3: the same with the blocks' "anything to do" test on two
# scalars pinned at kernel entry; development builds of cfg2 / cfg3)
for r in 1 2; do
AB_ARGS="" bash tools/ab2.sh r3y_c2_$r b2 q2 r2 >/dev/null
AB_ARGS="--config cfg3" bash tools/ab2.sh r3y_c3_$r b3 q3 r3 >/dev/null
done
for r in 1 2; do cat gpurun_out/r3y_c2_$r/bench.txt; done
for r in 1 2; do cat gpurun_out/r3y_c3_$r/bench.txt; done
R8B_HIP_LIB=$PWD/variants/tr2.so timeout 100 python tools/timeline_probe.py 2>&1 | grep -E "^lifetime|  block"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
