# round 6, fourth soak: after the fused half-array forms took the In > Out chains with long input steps (masked lanes read nothing)
mkdir -p gpurun_out/r6soak4
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 6000 6231 wide > gpurun_out/r6soak4/wide_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 6000 6232 > gpurun_out/r6soak4/preset24_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_TB=1.5,4 R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 5000 6233 wide > gpurun_out/r6soak4/mid_half.log 2>&1) &
wait
tail -qn 1 gpurun_out/r6soak4/*.log
