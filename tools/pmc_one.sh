#!/bin/bash
# usage: tools/pmc_one.sh <out-name> "<counters>" [bench args...]
# ONE rocprofv3 --pmc pass over a one-frame bench run; prints the sums for the extend, bounce and generate kernels.
name=$1; ctrs=$2; shift 2
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$name
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/$name/p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/$name/p.log 2>&1 || { echo "pass failed"; tail -5 $R/gpurun_out/$name/p.log; exit 1; }
for k in k_extend k_bounce k_generate; do echo "# $k"; python3 $R/tools/pmc_sum.py $R/gpurun_out/$name $k; done
