#!/bin/bash
# (historical record of a round-3 experiment: it ran against the build of its own commit; variants, macros and the
# engine option "persist" it names were removed again -- DESIGN.md section 5, profiles/r03_experiments.txt)
# tools/r3_q.sh -- which part of the instruction diet costs time: development builds of the headline kernel
out=gpurun_out/r3q; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
run() { # name lib [env...]
  local name=$1 lib=$2; shift; shift
  env "$@" R8B_HIP_LIB=$PWD/variants/$lib.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 --opt persist=0 2>>$out/err.log | line "$name" >> $out/bench.txt 2>&1
}
for rep in 1 2; do
  R8B_HIP_LIB=$PWD/variants/base.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 2>>$out/err.log | line "base" >> $out/bench.txt 2>&1
  run "cur (loop code, persist=0)" cur
  run "nopl" nopl
  run "nopl norot" nopl R8B_NO_ROT=1
  run "nopl fwd" nopl_fwd
  run "nopl fwd norot" nopl_fwd R8B_NO_ROT=1
done
cat $out/bench.txt
# counters of the nopl build (instruction counts, wave cycles)
R=$PWD
cd /tmp && export TMPDIR=/tmp
PGRPS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
)
for v in nopl; do
  o=$R/gpurun_out/pmc_$v; mkdir -p $o
  i=0
  for grp in "${PGRPS[@]}"; do
    i=$((i+1))
    R8B_HIP_LIB=$R/variants/$v.so timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $o/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --opt persist=0 > $o/p$i.log 2>&1
  done
  cd $R; python tools/pmc_summary.py $o > $out/pmc_$v.txt 2>&1; cd /tmp
done
cd $R; tail -40 $out/pmc_nopl.txt
