#!/bin/bash
# tools/pmc3.sh <tag> -- the wave-state counters only (one pass)
tag=$1; shift
R=$PWD
mkdir -p gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_$tag/p1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu "$@" > $R/gpurun_out/pmc_$tag/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_$tag/p2 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu "$@" > $R/gpurun_out/pmc_$tag/p2.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_$tag
