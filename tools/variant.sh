#!/bin/bash
# tools/variant.sh <name> <hipcc -D flags...> : builds a kernel-tuning variant of the HIP library
# into variants/<name>.so (same host objects, different device flags) for A/B runs on the GPU box
# (R8B_HIP_LIB=$PWD/variants/<name>.so python bench.py ...).
set -e
name=$1; shift
cd "$(dirname "$0")/../r8brain-free-src_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden "$@" -c r8b_kernels.hip -o /tmp/k_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/r8b_design.o _obj/r8b_plan.o _obj/r8b_engine.o _obj/r8b_capi.o _obj/r8b_kernels_pcm.o /tmp/k_$name.o -o ../../variants/$name.so
