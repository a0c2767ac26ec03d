#!/bin/bash
# third GPU pass: the two render modes on the 100.7 M-triangle scene and the other workloads, mode 2's service threshold on the
# dense workload
R=$PWD; out=$R/gpurun_out/r3c; mkdir -p $out
bash tools/sweep_tune.sh $out/modes_synth.txt "--scene synth --width 3840 --height 2160 --spp 32 --steps 2" "13=1" "13=2" || exit 1
bash tools/sweep_tune.sh $out/modes_dense.txt "--scene dragon_dense" "13=2,15=32" "13=2,15=40" "13=2,15=48" "13=2,15=52" "13=2,14=16" || exit 1
bash tools/sweep_tune.sh $out/modes_city.txt "--scene city --spp 256" "13=1" "13=2" || exit 1
bash tools/sweep_tune.sh $out/modes_dense64.txt "--scene dragon_dense --spp 64 --steps 6" "13=1" "13=2" || exit 1
