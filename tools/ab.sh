#!/bin/bash
# usage: tools/ab.sh "<bench args A>" "<bench args B>" ... ; runs each config twice, interleaved, prints Mrays/s and ms/frame
for round in 1 2; do
  for cfg in "$@"; do
    timeout -k 10 200 python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>&1 | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s %8.1f Mrays/s %8.2f ms' % ('$cfg', d['value'], d['ms_per_step']))"
  done
done
