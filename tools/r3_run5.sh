#!/bin/bash
out=gpurun_out/r3e; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "tuning or golden or shipped or both_extend or whole_path or no_alloc or two_meshes or coincident" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/sweep_tune.sh $out/layout_dragon.txt "" "11=1" "11=2" "11=0" "11=1" "11=2" || exit 1
bash tools/sweep_tune.sh $out/layout_dense.txt "--scene dragon_dense" "11=1" "11=2" "11=1" "11=2" || exit 1
bash tools/sweep_tune.sh $out/layout_synth.txt "--scene synth --width 3840 --height 2160 --spp 32 --steps 2" "11=1" "11=2" "11=0" || exit 1
bash tools/sweep_tune.sh $out/layout_bunny.txt "--scene bunny" "11=1" "11=2" || exit 1
bash tools/sweep_tune.sh $out/layout_city.txt "--scene city --spp 256" "11=1" "11=2" || exit 1
