# round 6, third soak: the final build, every half-array form (kernel modes 21 / 22 / 23 / 25 / 27 / 28) forced on the fuzz's
# three-channel batches
mkdir -p gpurun_out/r6soak3
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 6000 6221 wide > gpurun_out/r6soak3/wide_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 6000 6222 > gpurun_out/r6soak3/preset24_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_TB=1.5,4 R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 5000 6223 wide > gpurun_out/r6soak3/mid_half.log 2>&1) &
(timeout 900 python tools/gpu_fuzz.py 3000 6224 > gpurun_out/r6soak3/preset24_default.log 2>&1) &
wait
tail -qn 1 gpurun_out/r6soak3/*.log
