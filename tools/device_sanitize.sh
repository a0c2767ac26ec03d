#!/bin/bash
# Builds tools/device_sanitize (the device library's host code + the host library under AddressSanitizer and UBSan, the
# kernels as they are) here, where hipcc cross-compiles; run the binary on a machine with a GPU:
#   ASAN_OPTIONS=detect_leaks=0 tools/device_sanitize
# (leak detection off: the HIP runtime keeps its own allocations until exit)
set -e
cd "$(dirname "$0")/../cuda-path-tracer_amd"
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -Wno-unused-function $SAN \
    -I../include -Icsrc -Ihost -x hip csrc/pt_hip.hip -x hip ../tools/device_sanitize_main.cpp \
    -x hip host/bvh_builder.cpp -x hip host/scene.cpp -x hip host/assets.cpp -x hip host/envmap.cpp -x hip host/image.cpp \
    -lz -ldl -pthread -o ../tools/device_sanitize
echo built tools/device_sanitize
