#!/bin/bash
# tools/r3_z.sh -- GPU tier + smoke, then kernel stats / bench lines / HBM byte counters of cfg2 and cfg3 on the final build
out=gpurun_out/r3z; mkdir -p $out; rm -f $out/*
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1
tail -2 $out/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
PROF_ONLY="cfg2 cfg3" PROF_LIGHT="cfg2 cfg3" bash tools/profile_all.sh r3e > $out/prof.log 2>&1
for c in cfg2 cfg3; do for b in bench_20 bench_400; do python -c "
import json; d=json.loads(open('gpurun_out/prof_r3e/$c/$b.json').read().strip().splitlines()[-1])
print('$c $b value', d['value'], 'ms', d['ms_per_step'], 'first', d.get('first_window', {}).get('value'), 'roof', d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['roofline'].get('traffic'))"; done; cat gpurun_out/prof_r3e/$c/pmc_summary.txt | tail -4; done
