set -x
mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests -m gpu -x -q -k "history or soak or chunk or cascade" > gpurun_out/r6j/pytest_sel.log 2>&1; tail -3 gpurun_out/r6j/pytest_sel.log
timeout 600 python tools/ab.py --out gpurun_out/r6j/ab_cfg5 --reps 3 --steps 400 --config cfg5 old:lib=variants/r6_lv1.so new > gpurun_out/r6j/ab_cfg5.txt 2>&1; cat gpurun_out/r6j/ab_cfg5.txt
timeout 600 python tools/ab.py --out gpurun_out/r6j/ab_176 --reps 3 --steps 200 --bench-args "--src 176400 --dst 44100" old:lib=variants/r6_lv1.so new > gpurun_out/r6j/ab_176.txt 2>&1; cat gpurun_out/r6j/ab_176.txt
timeout 600 python tools/ab.py --out gpurun_out/r6j/ab_192 --reps 3 --steps 200 --bench-args "--src 192000 --dst 44100 --tb 0.5" old:lib=variants/r6_lv1.so new > gpurun_out/r6j/ab_192.txt 2>&1; cat gpurun_out/r6j/ab_192.txt
