#!/bin/bash
# tools/side_configs.sh : throughput of the non-headline topologies on the GPU box (one line each)
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-60s' % '$*', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
}
run --src 96000 --dst 44100
run --src 176400 --dst 44100
run --src 96000 --dst 48000
run --src 44100 --dst 44101
run --src 44100 --dst 88200
run --src 48000 --dst 32000
run --src 48000 --dst 32000 --opt fast_conv=0
run --src 32000 --dst 48000
run --src 44100 --dst 132300
run --src 96000 --dst 32000
run --src 44100 --dst 2822400 --block 1024 --channels 64
run --src 44100 --dst 2822400 --block 1024 --channels 1024
