#!/bin/bash
# usage: tools/pmc_two.sh <out-name> <kernel-substring> [bench args...]   -- passes 1 and 3 of tools/pmc_passes.sh only
name=$1; pat=$2; shift 2
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$name
cd /tmp
passes=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $p --output-format csv -d $R/gpurun_out/$name/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/$name/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/$name/p$i.log; exit 1; }
done
python3 $R/tools/pmc_sum.py $R/gpurun_out/$name "$pat"
