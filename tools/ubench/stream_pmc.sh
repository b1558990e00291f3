#!/bin/bash
# tools/ubench/stream_pmc.sh : hardware counters of the streaming kernels through the stand-alone harness
R=$PWD; out=$R/gpurun_out/stream_pmc; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- $R/tools/ubench/_build/stream_bench > $out/p$i.log 2>&1
done
cd $R && python tools/pmc_summary.py $out > $out/pmc_summary.txt 2>&1
find $out -name "*.csv" -size +200k -delete; rm -rf $out/p1 $out/p2 $out/p3 $out/p4 $out/p5
cat $out/pmc_summary.txt
