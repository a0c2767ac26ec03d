#!/bin/bash
# usage: tools/ab_tunes.sh <out-dir-under-gpurun_out> <bench args> -- <tune1> <tune2> ...
# One short bench run per --tune setting ("-" = defaults), each under its own time limit; stops at the first failure.
out=$1; shift
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
mkdir -p gpurun_out/$out
for t in "$@"; do
  tag=${t//[=,]/_}
  if [ "$t" = "-" ]; then tune=(); else tune=(--tune "$t"); fi
  timeout -k 10 240 python bench.py --no-cpu-baseline "${args[@]}" "${tune[@]}" > gpurun_out/$out/$tag.log 2>&1 || { echo "run $t failed"; tail -5 gpurun_out/$out/$tag.log; exit 1; }
  python - "$t" gpurun_out/$out/$tag.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline") or {}
        print("%-16s ms/frame %8.2f  Mrays/s %9.1f  ext avg %.3f ms x %s" % (sys.argv[1], d["ms_per_step"], d["value"], r.get("avg_launch_ms", 0), r.get("launches")))
PY
done
