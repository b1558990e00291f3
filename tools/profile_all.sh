#!/bin/bash
# tools/profile_all.sh <tag> : rocprofv3 kernel stats + hardware counters for every hot kernel
# (BASELINE configs 2, 3, 5 and the side topologies), on the GPU box.  Writes under
# gpurun_out/prof_<tag>/<name>/; tools/profile_collect.py copies the summaries into profiles/.
# Counter groups: those of tools/pmc.sh core (one rocprofv3 --pmc pass per group, kernel trace only), the
# calibration kernels of tools/ubench/pmc_calib.hip once (configuration "calib").
tag=$1
R=$PWD
out=$R/gpurun_out/prof_$tag
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
# PROF_ONLY="name name ..." restricts the run to those configurations
prof() {
  name=$1; shift
  if [ -n "$PROF_ONLY" ] && ! echo " $PROF_ONLY " | grep -q " $name "; then return; fi
  mkdir -p $out/$name
  # driver-style bench line (20 steps) and the long one, no profiler attached
  (cd $R && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu "$@" > $out/$name/bench_20.json 2>/dev/null)
  (cd $R && timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu "$@" > $out/$name/bench_400.json 2>/dev/null)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/stats -- python $R/bench.py --steps 200 --warmup 40 --settle 0 --no-cpu "$@" > $out/$name/stats.log 2>&1
  i=0
  for grp in "${PGRPS[@]}"; do
    i=$((i+1))
    # PROF_LIGHT="name ...": those configurations collect the HBM byte counters only (groups 6 and 7)
    if [ -n "$PROF_LIGHT" ] && echo " $PROF_LIGHT " | grep -q " $name " && [ $i -ne 6 ] && [ $i -ne 7 ]; then continue; fi
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$name/p$i -- python $R/bench.py --steps 4 --warmup 2 --settle 0 --no-cpu "$@" > $out/$name/p$i.log 2>&1
  done
  (cd $R && python tools/pmc_summary.py $out/$name > $out/$name/pmc_summary.txt 2>&1)
  # keep only the summaries (the raw traces are large)
  find $out/$name -name "*_kernel_stats.csv" -exec cp {} $out/$name/kernel_stats.csv \;
  rm -rf $out/$name/stats $out/$name/p[0-9]*
}
if [ -z "$PROF_ONLY" ] || echo " $PROF_ONLY " | grep -q " calib "; then
  mkdir -p $out/calib
  i=0
  for grp in "${PGRPS[@]:0:3}"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/calib/c$i -- $R/tools/ubench/_build/pmc_calib > $out/calib/c$i.log 2>&1
  done
  $R/tools/ubench/_build/pmc_calib > $out/calib/rates.txt 2>&1
  (cd $R && python tools/pmc_summary.py $out/calib > $out/calib/pmc_summary.txt 2>&1)
  rm -rf $out/calib/c[0-9]*
fi
prof cfg2
prof cfg3 --config cfg3
prof cfg5 --config cfg5
prof cfg5x1024 --src 44100 --dst 2822400 --block 1024 --channels 1024
prof poly --src 44100 --dst 44101
prof hbdown --src 176400 --dst 44100
prof up2 --src 44100 --dst 88200
prof down2 --src 88200 --dst 44100
prof ir16 --src 44100 --dst 96000 --atten 109.56
prof tb10 --src 44100 --dst 96000 --tb 10 --atten 109.56
prof r23 --src 48000 --dst 32000
prof split --src 44100 --dst 88200 --tb 0.5
prof split23 --src 48000 --dst 32000 --tb 0.5
prof solo --src 96000 --dst 44100 --tb 0.5
prof solo13 --src 48000 --dst 16000 --tb 1
prof pair13 --src 48000 --dst 16000 --tb 2
prof minphase --src 44100 --dst 96000 --phase 1
prof solo192 --src 192000 --dst 44100 --tb 0.5
prof up3 --src 16000 --dst 48000
prof exact32 --src 32000 --dst 48000 --tb 0.5
prof r4844 --src 48000 --dst 44100
ls $out/*
