#!/bin/bash
# cache counters of the traversal kernel with node layouts 1 and 2, on the 100.7 M-triangle scene and the headline scene
R=$PWD; out=$R/gpurun_out/r3f; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
P="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
for lay in 1 2; do
  timeout -s KILL 400 rocprofv3 --pmc $P --output-format csv -d $out/synth_l$lay -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene synth --width 3840 --height 2160 --spp 32 --tune 11=$lay,13=1 > $out/synth_l$lay.log 2>&1 || { echo "synth layout $lay failed"; tail -5 $out/synth_l$lay.log; exit 1; }
  echo "== synth, node layout $lay"; python3 $R/tools/pmc_sum.py $out/synth_l$lay k_extend | tee $out/synth_l$lay.txt
  rm -rf $out/synth_l$lay
  timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d $out/dragon_l$lay -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --tune 11=$lay > $out/dragon_l$lay.log 2>&1 || { echo "dragon layout $lay failed"; tail -5 $out/dragon_l$lay.log; exit 1; }
  echo "== dragon, node layout $lay"; python3 $R/tools/pmc_sum.py $out/dragon_l$lay k_extend | tee $out/dragon_l$lay.txt
  rm -rf $out/dragon_l$lay
done
