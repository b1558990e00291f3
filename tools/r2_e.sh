#!/bin/bash
out=gpurun_out/r2e2; mkdir -p $out; rm -f $out/bench.txt $out/err.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "matches_oracle or golden or chunk" > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $out/bench.txt 2>&1
}
run
run --opt pair_two=0
run --steps 20 --warmup 5
run --src 96000 --dst 44100
run --src 44100 --dst 88200
run --src 44100 --dst 2822400 --block 1024 --channels 64
cat $out/bench.txt; tail -3 $out/pytest.log
