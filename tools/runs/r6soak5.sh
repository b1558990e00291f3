# round 6, fifth soak: after the fused 1:1 geometry's half-array form (kernel mode 33, BASELINE's cfg3); every form forced
mkdir -p gpurun_out/r6soak5
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 6000 6241 wide > gpurun_out/r6soak5/wide_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 6000 6242 > gpurun_out/r6soak5/preset24_half.log 2>&1) &
(timeout 900 env R8B_FUZZ_TB=1.5,4 R8B_FUZZ_OPTS="half=2 half_fused=2" python tools/gpu_fuzz.py 5000 6243 wide > gpurun_out/r6soak5/mid_half.log 2>&1) &
wait
tail -qn 1 gpurun_out/r6soak5/*.log
