export PROF_ONLY="tb10 exact32"
export PROF_LIGHT="tb10 exact32"
bash tools/profile_all.sh r6 > gpurun_out/prof_r6b.log 2>&1
tail -3 gpurun_out/prof_r6b.log
