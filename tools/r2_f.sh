#!/bin/bash
# tools/r2_f.sh : round-2 (third session) GPU pass -- full GPU tier, then the streaming-kernel topologies
out=gpurun_out/r2f; mkdir -p $out; rm -f $out/bench.txt $out/err.log
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $out/bench.txt 2>&1
}
run --src 44100 --dst 44101
run --src 48000 --dst 44111
run --src 176400 --dst 44100
run --src 44100 --dst 2822400 --block 1024 --channels 1024
run --src 44100 --dst 2822400 --block 1024 --channels 64
run --src 2822400 --dst 176400 --block 65536 --channels 256
run --src 44100 --dst 96000 --tb 45 --atten 49
run --src 96000 --dst 44100 --tb 45 --atten 49
run --src 32000 --dst 44100
run
timeout 200 python tools/minphase_probe.py > $out/minphase.txt 2>&1
cat $out/bench.txt; cat $out/minphase.txt; tail -3 $out/pytest.log
