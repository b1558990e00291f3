#!/bin/bash
# tools/pmc4.sh <tag> [bench args...] -- hardware-counter passes (rocprofv3 --pmc with the kernel trace only; one
# pass per counter group) of a bench workload AND of the calibration kernels (tools/ubench/pmc_calib.hip), so that
# every percentage quoted in DESIGN.md can be recomputed from the summary: tools/pmc_summary.py prints the raw
# per-dispatch means, the calibration, and the derived ratios with their formulas.
tag=$1; shift
R=$PWD
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
PGRPS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL"
 "FETCH_SIZE" "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
)
i=0
for grp in "${PGRPS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-cpu "$@" > $out/p$i.log 2>&1
  if [ -x $R/tools/ubench/_build/pmc_calib ] && [ $i -le 3 ]; then
    timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/c$i -- $R/tools/ubench/_build/pmc_calib > $out/c$i.log 2>&1
  fi
done
cd $R
python tools/pmc_summary.py $out
