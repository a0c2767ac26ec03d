#!/bin/bash
# usage: tools/ab_env.sh <ENV_VAR> "<values>" [bench args...]   (on the GPU box, from the repo root)
# Runs bench.py once per value of an environment variable that a build under test reads and prints ms per frame.
var=$1; vals=$2; shift 2
for v in $vals; do
  out=$(env $var=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{') || { echo "$var=$v failed"; exit 1; }
  echo "$var=$v $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["film_check"]["mean"])')"
done
