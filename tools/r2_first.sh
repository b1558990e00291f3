#!/bin/bash
# round-2 first GPU pass: parity tier + pair kernel on/off timings (one line each)
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2a/pytest.log
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>gpurun_out/r2a/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> gpurun_out/r2a/bench.txt 2>&1
}
run
run --opt pair_conv=0
run --steps 20 --warmup 5
run --src 96000 --dst 44100
run --src 96000 --dst 44100 --opt pair_conv=0
run --src 44100 --dst 88200
run --src 44100 --dst 88200 --opt pair_conv=0
run --src 44100 --dst 2822400 --block 1024 --channels 64
run --src 44100 --dst 2822400 --block 1024 --channels 64 --opt pair_conv=0
run --src 44100 --dst 2822400 --block 1024 --channels 1024
cat gpurun_out/r2a/bench.txt; tail -5 gpurun_out/r2a/pytest.log
