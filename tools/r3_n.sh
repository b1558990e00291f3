#!/bin/bash
# full-library check + profiles of all configurations
out=gpurun_out/r3n; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default_20.json 2>$out/bench_err.log
tail -3 $out/pytest.log; tail -1 $out/smoke.log; cat $out/bench_default_20.json
tools/profile_all.sh r3 > $out/profile_all.log 2>&1
tail -5 $out/profile_all.log
