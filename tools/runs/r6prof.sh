export PROF_ONLY="cfg2 cfg3 cfg5 cfg5x1024 poly hbdown up2 down2 ir16 tb10 r23 split split23 solo solo13 pair13 minphase solo192 up3 exact32"
export PROF_LIGHT="cfg5x1024 poly up2 down2 ir16 tb10 r23 split split23 solo solo13 pair13 minphase solo192 up3 exact32"
bash tools/profile_all.sh r6 > gpurun_out/prof_r6.log 2>&1
tail -3 gpurun_out/prof_r6.log
