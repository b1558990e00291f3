#!/bin/bash
# (historical record of a round-3 experiment: it ran against the build of its own commit; variants, macros and the
# engine option "persist" it names were removed again -- DESIGN.md section 5, profiles/r03_experiments.txt)
# tools/r3_r.sh -- lane assignment of the interpolator with fewer LDS bank clashes (lean), LDS reads pair by pair
# between the multiply-adds (pipe5 / pipe8: 5 / 8 pairs ahead); counters of lean and pipe5
out=gpurun_out/r3r; mkdir -p $out; rm -f $out/*
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s' % sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
run() { local name=$1 lib=$2; shift; shift
  env "$@" R8B_HIP_LIB=$PWD/variants/$lib.so timeout 120 python bench.py --no-cpu --steps 400 --warmup 40 2>>$out/err.log | line "$name" >> $out/bench.txt 2>&1; }
for rep in 1 2; do
  run base base
  run lean lean
  run pipe5 pipe5
  run pipe8 pipe8
done
run "lean 20" lean
cat $out/bench.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
PGRPS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
)
for v in lean pipe5; do
  o=$R/gpurun_out/pmc_$v; mkdir -p $o
  i=0
  for grp in "${PGRPS[@]}"; do
    i=$((i+1))
    R8B_HIP_LIB=$R/variants/$v.so timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $o/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-cpu > $o/p$i.log 2>&1
  done
  cd $R; python tools/pmc_summary.py $o > $out/pmc_$v.txt 2>&1; cd /tmp
done
cd $R; grep -E "wave lifetime|in flight|s_waitcnt|waiting to issue|kernel duration|LDS busy|bank-conflict|waiting on LDS|VALU instructions|SALU inst|LDS_DATA_FIFO|BANK_CONFLICT|IDX_ACTIVE" $out/pmc_lean.txt $out/pmc_pipe5.txt
