#!/bin/bash
# tools/r3_x.sh -- GPU tier + smoke + the driver's bench line on the split build
out=gpurun_out/r3x; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_20.json 2>$out/err.log
python -c "
import json; d=json.loads(open('$out/bench_20.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'first', d.get('first_window'), 'roof', d['roofline']['frac'], d['roofline']['avg_kernel_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_vs_reference'])"
timeout 300 python bench.py --gpus 1 --no-cpu > $out/bench_400.json 2>>$out/err.log
python -c "
import json; d=json.loads(open('$out/bench_400.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'first', d.get('first_window'), 'roof', d['roofline']['frac'], d['roofline']['avg_kernel_ms'])"
for p in s16 f32; do timeout 120 python bench.py --no-cpu --steps 100 --warmup 20 --pcm $p --planar 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pcm planar $p', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"; done
