#!/bin/bash
# usage: tools/pmc_passes.sh <out-name> <kernel-substring> [bench args...]
# Runs the counter passes (one rocprofv3 run each, as the per-pass hardware limits require) over a one-frame bench run
# and prints the per-kernel sums.  Stops at the first pass that fails.
name=$1; pat=$2; shift 2
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$name
cd /tmp
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_SMEM"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  timeout -s KILL 420 rocprofv3 --pmc $p --output-format csv -d $R/gpurun_out/$name/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/$name/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/$name/p$i.log; exit 1; }
done
python3 $R/tools/pmc_sum.py $R/gpurun_out/$name "$pat"
