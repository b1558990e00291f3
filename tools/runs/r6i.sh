set -x
mkdir -p gpurun_out/r6i
timeout 900 python -m pytest tests -m gpu -x -q -k "long_filters or exact_block" > gpurun_out/r6i/pytest_sel.log 2>&1; tail -3 gpurun_out/r6i/pytest_sel.log
for r in "32000 48000 0.5" "64000 48000 0.5"; do set -- $r
timeout 600 python tools/ab.py --out gpurun_out/r6i/ab_$1_$3 --reps 2 --steps 60 --warmup 10 --bench-args "--src $1 --dst $2 --tb $3" old:lib=variants/r6_lv1.so new > gpurun_out/r6i/ab_$1_$3.txt 2>&1; cat gpurun_out/r6i/ab_$1_$3.txt
done
