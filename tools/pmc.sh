#!/bin/bash
# tools/pmc.sh <tag> [bench args...] -- hardware-counter passes (rocprofv3 --pmc, kernel trace
# only) of the bench workload on the GPU box; one pass per counter group, CSVs under
# gpurun_out/pmc_<tag>/.
tag=$1; shift
R=$PWD
mkdir -p gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-cpu "$@" > $R/gpurun_out/pmc_$tag/p$i.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_$tag
