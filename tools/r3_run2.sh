#!/bin/bash
# second GPU pass of the round: bounce alone with and without the normal table (one stream, kernel stats), the tuning
# re-sweep on the dense workload, and the cost of rendering slab after slab at 3840x2160
R=$PWD; out=$R/gpurun_out/r3b; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for v in "9=1" "9=1,17=0"; do
  tag=$(echo $v | tr '=,' '__')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -- python3 $R/bench.py --no-cpu-baseline --tune $v > $out/bench_$tag.log 2>&1 || { echo "stats run $v failed"; tail -5 $out/bench_$tag.log; exit 1; }
  python3 $R/tools/kstats.py $out/stats_$tag > $out/kernel_stats_$tag.txt
  echo "== --tune $v"; grep -E "k_bounce|k_generate|k_extend" $out/kernel_stats_$tag.txt | head -4; tail -1 $out/bench_$tag.log | grep -o '"ms_per_step": [0-9.]*'
  rm -rf $out/stats_$tag
done
cd $R
bash tools/sweep_tune.sh $out/dense_sweep.txt "--scene dragon_dense" "" "6=24" "6=32" "6=38" "6=50" "6=56" "9=1" "9=2" "9=4" "10=4" "10=5" "13=2" || exit 1
for lib in chunk64 chunk256; do
  line=$(PT_HIP_LIB=$R/cuda-path-tracer_amd/ab/libpt_hip_$lib.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --scene dragon_dense 2>/dev/null | tail -1) || exit 1
  echo "$lib $(echo "$line" | grep -o '"ms_per_step": [0-9.]*')" | tee -a $out/dense_sweep.txt
done
# 3840x2160, 64 spp: one slab (8.5 GB) against eight slabs of 8 samples
bash tools/sweep_tune.sh $out/slabs_4k.txt "--width 3840 --height 2160 --spp 64 --steps 3" "" "4=1061683200" "4=2123366400" || exit 1
