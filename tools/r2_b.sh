#!/bin/bash
# pair kernel: vector vs matrix-core interpolation, quick parity on the GPU
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "matches_oracle or golden" > gpurun_out/r2b/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2b/pytest.log
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>gpurun_out/r2b/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> gpurun_out/r2b/bench.txt 2>&1
}
run
run --opt pair_mfma=0
run --steps 20 --warmup 5
cat gpurun_out/r2b/bench.txt; tail -5 gpurun_out/r2b/pytest.log
