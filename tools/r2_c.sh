#!/bin/bash
mkdir -p gpurun_out/r2c2; rm -f gpurun_out/r2c2/bench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "matches_oracle or golden" > gpurun_out/r2c2/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2c2/pytest.log
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>gpurun_out/r2c2/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> gpurun_out/r2c2/bench.txt 2>&1
}
run
run --opt pair_two=0
run --opt pair_two=0 --opt pair_mfma=1
run --steps 20 --warmup 5
cat gpurun_out/r2c2/bench.txt; tail -3 gpurun_out/r2c2/pytest.log
