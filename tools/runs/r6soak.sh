mkdir -p gpurun_out/r6soak
(timeout 800 python tools/gpu_fuzz.py 5000 6201 wide > gpurun_out/r6soak/wide_a.log 2>&1) &
(timeout 800 env R8B_FUZZ_OPTS="walk=2" python tools/gpu_fuzz.py 5000 6202 wide > gpurun_out/r6soak/wide_walk.log 2>&1) &
(timeout 800 env R8B_FUZZ_TB=0.5,1.2 python tools/gpu_fuzz.py 2500 6203 wide > gpurun_out/r6soak/narrow.log 2>&1) &
(timeout 800 python tools/gpu_fuzz.py 5000 6204 > gpurun_out/r6soak/preset24.log 2>&1) &
(timeout 800 env R8B_FUZZ_OPTS="fuse_hbconv=1 quad=1" python tools/gpu_fuzz.py 3000 6205 > gpurun_out/r6soak/offforms.log 2>&1) &
wait
tail -qn 1 gpurun_out/r6soak/*.log
