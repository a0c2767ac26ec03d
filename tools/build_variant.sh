#!/bin/bash
# usage: tools/build_variant.sh NAME [hipcc flags...]  -> cuda-path-tracer_amd/ab/libpt_hip_NAME.so (an A/B build of the device
# library with extra compile flags; run it with PT_HIP_LIB or tools/ab_libs.sh ab/libpt_hip_NAME.so)
name=$1; shift
cd "$(dirname "$0")/../cuda-path-tracer_amd" || exit 1
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
    -I../include -Icsrc -Ihost "$@" -shared -o ab/libpt_hip_$name.so csrc/pt_hip.hip
