# round 6, second soak: the half-array forms (kernel modes 21 / 22 / 23 / 25) forced on the fuzz's three-channel batches
mkdir -p gpurun_out/r6soak2
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 5000 6211 wide > gpurun_out/r6soak2/wide_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 5000 6212 > gpurun_out/r6soak2/preset24_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_TB=0.5,1.6 R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 2500 6213 wide > gpurun_out/r6soak2/narrow_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_TB=1.5,3.5 R8B_FUZZ_OPTS="half=2 half_fused=2 walk=0" python tools/gpu_fuzz.py 4000 6214 wide > gpurun_out/r6soak2/mid_half.log 2>&1) &
(timeout 900 python tools/gpu_fuzz.py 3000 6215 wide > gpurun_out/r6soak2/wide_default.log 2>&1) &
wait
tail -qn 1 gpurun_out/r6soak2/*.log
