#!/bin/bash
# A rank's share of the headline frame at 8, 4 and 2 GPUs (64, 128, 256 of the 512 samples), rendered by each render mode.
mkdir -p gpurun_out
out=gpurun_out/share_modes.txt
: > $out
for spp in 64 128 256; do
  for mode in 1 2; do
    line=$(timeout -k 10 200 python bench.py --spp $spp --steps 10 --warmup 2 --no-cpu-baseline --tune 13=$mode 2>/dev/null | grep '^{') || exit 1
    echo "spp=$spp mode=$mode $(echo "$line" | grep -o '"ms_per_step": [0-9.]*')" | tee -a $out
  done
done
