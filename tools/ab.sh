#!/bin/bash
# tools/ab.sh [bench args] : runs the bench once per variants/*.so on the GPU box, prints one line each
for lib in variants/*.so; do
  R8B_HIP_LIB=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-28s' % '$lib', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
