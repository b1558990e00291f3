set -x
mkdir -p gpurun_out/r6k
timeout 900 python -m pytest tests -m gpu -x -q -k "half_band_front or history or chunk" > gpurun_out/r6k/pytest_sel.log 2>&1; tail -5 gpurun_out/r6k/pytest_sel.log
timeout 600 python tools/ab.py --out gpurun_out/r6k/ab_176 --reps 3 --steps 200 --bench-args "--src 176400 --dst 44100" unfused:opt=fuse_hbconv=0 fused > gpurun_out/r6k/ab_176.txt 2>&1; cat gpurun_out/r6k/ab_176.txt
timeout 600 python tools/ab.py --out gpurun_out/r6k/ab_192 --reps 2 --steps 200 --bench-args "--src 192000 --dst 48000" unfused:opt=fuse_hbconv=0 fused > gpurun_out/r6k/ab_192.txt 2>&1; cat gpurun_out/r6k/ab_192.txt
