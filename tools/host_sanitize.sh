#!/bin/bash
# Builds the host library's sources with ThreadSanitizer and with AddressSanitizer + UBSan into /tmp and runs the BVH
# builder (1 thread against 8 threads, outputs compared) under each.  CPU only.  usage: tools/host_sanitize.sh
set -e
cd "$(dirname "$0")/../cuda-path-tracer_amd"
for san in thread address,undefined; do
  out=/tmp/pt_host_sanitize_${san%%,*}
  g++ -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -pthread -fsanitize=$san -I../include -Ihost \
      ../tools/host_sanitize_main.cpp host/bvh_builder.cpp host/scene.cpp host/assets.cpp host/envmap.cpp host/image.cpp -lz -o $out
  echo "== -fsanitize=$san"
  $out
done
