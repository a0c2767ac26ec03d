#!/bin/bash
# usage: tools/ab_libs.sh "<bench args>" lib1.so lib2.so ... ; each library (path relative to cuda-path-tracer_amd/, "-" = default)
# twice, interleaved; prints Mrays/s and ms/frame
args=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset PT_HIP_LIB; else export PT_HIP_LIB=$PWD/cuda-path-tracer_amd/$lib; fi
    timeout -k 10 200 python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline $args 2>&1 | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s %8.1f Mrays/s %8.2f ms' % ('$lib', d['value'], d['ms_per_step']))"
  done
done
