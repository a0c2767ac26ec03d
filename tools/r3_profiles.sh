#!/bin/bash
# the round's tracked profiles: headline workload (wavefront mode), dense workload and 100.7 M-triangle workload (whole-path mode)
bash tools/profile_round.sh r3_dragon > gpurun_out/r3_dragon.out 2>&1 || { tail -20 gpurun_out/r3_dragon.out; exit 1; }
tail -3 gpurun_out/r3_dragon.out
bash tools/profile_round.sh r3_dense --scene dragon_dense > gpurun_out/r3_dense.out 2>&1 || { tail -20 gpurun_out/r3_dense.out; exit 1; }
tail -3 gpurun_out/r3_dense.out
bash tools/profile_round.sh r3_synth --scene synth --width 3840 --height 2160 --spp 32 > gpurun_out/r3_synth.out 2>&1 || { tail -20 gpurun_out/r3_synth.out; exit 1; }
tail -3 gpurun_out/r3_synth.out
