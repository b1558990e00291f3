#!/bin/bash
# tools/r2_g.sh : final pass of the session -- GPU tier, smoke, the driver-style bench line, a few side topologies
out=gpurun_out/r2g; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_default_20.json 2>$out/bench_err.log
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $out/bench.txt 2>&1
}
run --src 8000 --dst 32000
run --src 8000 --dst 192000
run --src 8000 --dst 176400
run --src 8000 --dst 44100
tail -2 $out/pytest.log; tail -1 $out/smoke.log; cat $out/bench_default_20.json; cat $out/bench.txt
