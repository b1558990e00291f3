#!/bin/bash
# tools/variant.sh <name> <hipcc -D flags...> : builds a kernel-tuning variant of the HIP library
# into variants/<name>.so (same host objects, different device flags) for A/B runs on the GPU box
# (R8B_HIP_LIB=$PWD/variants/<name>.so python bench.py ...).
# DEV=<mode> LN=<ln> UL=<ul> in the environment: a development build with ONE pair-kernel instance
# (k_convp<LN, UL, mode, 24>; default 11 / 1 = cfg2) and nothing else of the fast path -- a minute instead of ten.
set -e
name=$1; shift
cd "$(dirname "$0")/../r8brain-free-src_amd/csrc"
mkdir -p ../../variants
DEVFLAGS=()
if [ -n "$DEV" ]; then
  DEVFLAGS=(-DR8B_DEV_GEOMS "-DR8B_CONVP_GEOMS(M)=M(${LN:-11},${UL:-1})" "-DR8B_CONVP_GEOMS_BIG(M)=" "-DR8B_CONVP_GEOMS_DOWN(M)="
            "-DR8B_CONVX_GEOMS(M)=" "-DR8B_CONVX_GEOMS_DOWN(M)=" -DR8B_DEV_ONLY_MODE=$DEV)
fi
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden "${DEVFLAGS[@]}" "$@" -c r8b_kernels.hip -o /tmp/k_$name.o
PCMOBJ=_obj/r8b_kernels_pcm.o
if [ -n "$DEV" ]; then
  # (development builds: a stub instead of the PCM twins -- 2 MB instead of 15 on the way to the GPU box)
  g++ -std=c++17 -O1 -fPIC -fvisibility=hidden -I. -c ../../tools/pcm_stub.cpp -o /tmp/pcm_stub.o
  PCMOBJ=/tmp/pcm_stub.o
fi
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 _obj/r8b_design.o _obj/r8b_plan.o _obj/r8b_engine.o _obj/r8b_capi.o $PCMOBJ /tmp/k_$name.o -o ../../variants/$name.so
