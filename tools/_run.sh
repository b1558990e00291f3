export R8B_HIP_LIB=$PWD/variants/walk3.so
python tools/dbg_parity.py 48000 32000 602 16384 4 2>&1 | tail -4
python tools/ab.py --out gpurun_out/ab_walk_r23 --reps 2 --steps 200 --bench-args "--src 48000 --dst 32000" walk:lib=variants/walk3.so nowalk:lib=variants/walk3.so:env=R8B_NO_WALK=1 2>&1 | tail -3
