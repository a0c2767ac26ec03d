#!/bin/bash
# gather probe at the two kernels' measured mixes (arguments from tools/probe_mix.py on the counters already taken:
# headline workload 32 38 43, dense workload 35 47 42), then the 100.7 M-triangle profile
out=gpurun_out/r3_probe2; mkdir -p $out
for mix in "32 38 43" "35 47 42"; do
  f=$out/gather_probe_mixed_$(echo $mix | tr ' ' '_').txt
  echo "gather_probe2 $mix" > $f
  timeout -k 5 60 tools/gather_probe2 $mix >> $f 2>&1 || exit 1
  cat $f
done
bash tools/profile_round.sh r3_synth --scene synth --width 3840 --height 2160 --spp 32 > gpurun_out/r3_synth.out 2>&1 || { tail -20 gpurun_out/r3_synth.out; exit 1; }
tail -8 gpurun_out/r3_synth.out
