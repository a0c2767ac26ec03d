#!/bin/bash
# the round's tracked profiles: headline workload (wavefront mode) and dense workload (whole-path mode)
bash tools/profile_round.sh r3_dragon k_extend_stream > gpurun_out/r3_dragon.out 2>&1 || { tail -20 gpurun_out/r3_dragon.out; exit 1; }
tail -3 gpurun_out/r3_dragon.out
bash tools/profile_round.sh r3_dense k_path_stream --scene dragon_dense > gpurun_out/r3_dense.out 2>&1 || { tail -20 gpurun_out/r3_dense.out; exit 1; }
tail -3 gpurun_out/r3_dense.out
