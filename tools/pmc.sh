#!/bin/bash
# tools/pmc.sh <tag> [core|mem|hbm|all] [bench args...] -- hardware-counter passes of a bench workload on the GPU box:
# one `rocprofv3 --pmc <group>` pass per counter group with the kernel trace only (never combined with other trace
# domains), CSVs under gpurun_out/pmc_<tag>/, then tools/pmc_summary.py (raw per-dispatch means, the calibration
# kernels of tools/ubench/pmc_calib.hip when built, and every derived ratio printed with its formula).
#   core : wave states, instruction mix incl. the fp64 split, LDS, vector memory, HBM bytes (FETCH_SIZE / WRITE_SIZE
#          in passes of their own, MI355X_MICROARCH.md section HBM), L1 <-> L2 request counts
#   mem  : the memory path -- TLB, L1 -> L2 round-trip latencies, L1 pending-queue stalls, L2 hits / tag stalls
#   hbm  : FETCH_SIZE and WRITE_SIZE only (two passes: what profiles/traffic.json is made of)
# (one script instead of the five generations pmc.sh .. pmc4.sh / pmc_mem.sh of rounds 1-3)
tag=$1; shift
set_=core
case "$1" in core|mem|hbm|all) set_=$1; shift;; esac
R=$PWD
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CORE=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL"
 "FETCH_SIZE" "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
)
MEM=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
)
HBM=("FETCH_SIZE" "WRITE_SIZE")
case $set_ in
  core) GRPS=("${CORE[@]}");;
  mem) GRPS=("${MEM[@]}");;
  hbm) GRPS=("${HBM[@]}");;
  all) GRPS=("${CORE[@]}" "${MEM[@]}");;
esac
i=0
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python $R/bench.py --steps 4 --warmup 2 --settle 0 --no-cpu "$@" > $out/p$i.log 2>&1
  if [ "$set_" != mem ] && [ "$set_" != hbm ] && [ -x $R/tools/ubench/_build/pmc_calib ] && [ $i -le 3 ]; then
    timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/c$i -- $R/tools/ubench/_build/pmc_calib > $out/c$i.log 2>&1
  fi
done
cd $R
python tools/pmc_summary.py $out
