#!/bin/bash
# tools/r3_z2.sh -- final check of a build: GPU tier, smoke, the driver's default bench line (CPU leg included)
out=gpurun_out/r3z2; mkdir -p $out; rm -f $out/*
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
tail -2 $out/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 400 python bench.py > $out/bench_default.json 2>$out/err.log
python -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'first', d.get('first_window'), 'roof', d['roofline'], 'cpu', d['cpu_baseline'])"
