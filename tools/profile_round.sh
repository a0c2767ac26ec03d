#!/bin/bash
# usage: tools/profile_round.sh <tag> [bench args...]   (on the GPU box, from the repo root)
# The traversal kernel profiled is the one the library picks for the workload: k_extend_stream in wavefront mode,
# k_path_stream in whole-path mode (config.render_mode of the first run's bench line).
# 1. rocprofv3 --kernel-trace --stats of `bench.py <args>` (with the CPU leg: it counts the per-ray traversal means);
# 2. wavefront mode only: the same with one stream (extend launches not overlapped by anything);
# 3. the counter passes of tools/pmc_passes.sh; 4. the dependent-gather probes, the second at the kernel's measured mix.
# Everything lands in gpurun_out/<tag>/; tools/extend_traffic_json.py turns it into the summary bench.py reads.
tag=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -- python3 $R/bench.py "$@" > $R/gpurun_out/$tag/bench_default.log 2>&1 || { echo "stats run failed"; tail -5 $R/gpurun_out/$tag/bench_default.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/$tag/stats > $R/gpurun_out/$tag/kernel_stats.txt
rm -rf $R/gpurun_out/$tag/stats
cat $R/gpurun_out/$tag/kernel_stats.txt
kern=k_extend_stream
grep -q '"render_mode": 2' $R/gpurun_out/$tag/bench_default.log && kern=k_path_stream
echo "traversal kernel: $kern"
if [ "$kern" = "k_extend_stream" ]; then
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats1 -- python3 $R/bench.py --no-cpu-baseline --tune 9=1 "$@" > $R/gpurun_out/$tag/bench_one_stream.log 2>&1 || { echo "one-stream stats run failed"; tail -5 $R/gpurun_out/$tag/bench_one_stream.log; exit 1; }
  python3 $R/tools/kstats.py $R/gpurun_out/$tag/stats1 > $R/gpurun_out/$tag/kernel_stats_one_stream.txt
  rm -rf $R/gpurun_out/$tag/stats1
  cat $R/gpurun_out/$tag/kernel_stats_one_stream.txt
fi
cd $R
bash tools/pmc_passes.sh $tag/pmc $kern "$@" > gpurun_out/$tag/pmc_kernel.txt 2>&1 || { cat gpurun_out/$tag/pmc_kernel.txt; exit 1; }
python3 tools/pmc_sum.py gpurun_out/$tag/pmc k_bounce > gpurun_out/$tag/pmc_bounce.txt
python3 tools/pmc_sum.py gpurun_out/$tag/pmc k_generate > gpurun_out/$tag/pmc_generate.txt
rm -rf gpurun_out/$tag/pmc/p[0-9]
cat gpurun_out/$tag/pmc_kernel.txt
for n in 400 16000 1300000; do echo "records $n"; timeout -k 5 60 tools/gather_probe $n 2000 26624 || exit 1; done > gpurun_out/$tag/gather_probe.txt 2>&1
cat gpurun_out/$tag/gather_probe.txt
# the same fetches at the kernel's own mix: its share of busy lanes, of fetches that hit L1, of fetches that hit L2
mix=$(python3 tools/probe_mix.py gpurun_out/$tag/pmc_kernel.txt) || exit 1
echo "gather_probe2 $mix" > gpurun_out/$tag/gather_probe_mixed.txt
timeout -k 5 60 tools/gather_probe2 $mix >> gpurun_out/$tag/gather_probe_mixed.txt 2>&1 || exit 1
cat gpurun_out/$tag/gather_probe_mixed.txt
