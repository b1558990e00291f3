mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests -m gpu -x -q -k "split or one_channel or parked or solo or history or up3 or poly3 or long" > gpurun_out/r6m/pytest_sel.log 2>&1; tail -3 gpurun_out/r6m/pytest_sel.log
run() { n=$1; shift; timeout 600 python tools/ab.py --out gpurun_out/r6m/ab_$n --reps 3 --steps 200 --bench-args "$*" old:lib=variants/r6_lv1.so new > gpurun_out/r6m/ab_$n.txt 2>&1; echo "== $n"; tail -2 gpurun_out/r6m/ab_$n.txt; }
run split --src 44100 --dst 88200 --tb 0.5
run split23 --src 48000 --dst 32000 --tb 0.5
run solo --src 96000 --dst 44100 --tb 0.5
run solo13 --src 48000 --dst 16000 --tb 1
run up3 --src 16000 --dst 48000
run minsplit --src 44100 --dst 88200 --tb 0.5 --phase 1
