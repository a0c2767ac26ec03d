#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box, from the repo root)
# 1. rocprofv3 --kernel-trace --stats of the default bench command; 2. the same with one stream (extend launches not
# overlapped by anything); 3. the counter passes of tools/pmc_passes.sh; 4. the dependent-gather probe.
# Everything lands in gpurun_out/<tag>/; tools/extend_traffic_json.py turns it into the summary bench.py reads.
tag=$1
R=$PWD
mkdir -p $R/gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -- python3 $R/bench.py > $R/gpurun_out/$tag/bench_default.log 2>&1 || { echo "stats run failed"; tail -5 $R/gpurun_out/$tag/bench_default.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/$tag/stats > $R/gpurun_out/$tag/kernel_stats.txt
cat $R/gpurun_out/$tag/kernel_stats.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats1 -- python3 $R/bench.py --no-cpu-baseline --tune 9=1 > $R/gpurun_out/$tag/bench_one_stream.log 2>&1 || { echo "one-stream stats run failed"; tail -5 $R/gpurun_out/$tag/bench_one_stream.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/$tag/stats1 > $R/gpurun_out/$tag/kernel_stats_one_stream.txt
cat $R/gpurun_out/$tag/kernel_stats_one_stream.txt
cd $R
bash tools/pmc_passes.sh $tag/pmc k_extend > gpurun_out/$tag/pmc_extend.txt 2>&1 || { cat gpurun_out/$tag/pmc_extend.txt; exit 1; }
python3 tools/pmc_sum.py gpurun_out/$tag/pmc k_bounce > gpurun_out/$tag/pmc_bounce.txt
python3 tools/pmc_sum.py gpurun_out/$tag/pmc k_generate > gpurun_out/$tag/pmc_generate.txt
cat gpurun_out/$tag/pmc_extend.txt
for n in 400 16000 1300000; do echo "records $n"; timeout -k 5 60 tools/gather_probe $n 2000 26624 || exit 1; done > gpurun_out/$tag/gather_probe.txt 2>&1
cat gpurun_out/$tag/gather_probe.txt
# the same fetches at extend's own mix: 41 of 64 lanes busy, 38 % from an L1-resident set, 41 % from an L2-resident one
timeout -k 5 60 tools/gather_probe2 41 38 41 > gpurun_out/$tag/gather_probe_mixed.txt 2>&1 || exit 1
cat gpurun_out/$tag/gather_probe_mixed.txt
