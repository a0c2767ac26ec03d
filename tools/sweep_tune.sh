#!/bin/bash
# usage: tools/sweep_tune.sh OUTFILE "bench args" tune1 tune2 ...   -- one bench.py run per --tune string, ms/frame each
out=$1; shift
args=$1; shift
mkdir -p "$(dirname "$out")"
for t in "$@"; do
    line=$(timeout -k 10 300 python bench.py --no-cpu-baseline $args --tune "$t" 2>/dev/null | tail -1) || { echo "$t FAILED" >> "$out"; exit 1; }
    echo "$t $(echo "$line" | grep -o '"ms_per_step": [0-9.]*') $(echo "$line" | grep -o '"value": [0-9.]*')" >> "$out"
done
cat "$out"
