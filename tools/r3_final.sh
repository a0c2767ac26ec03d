#!/bin/bash
# the default bench line (counter block must be live), the other configurations, and the N > 1 path of bench.py started
# plainly with all ranks on this GPU
out=gpurun_out/r3_final; mkdir -p $out
timeout -k 10 300 python3 -u bench.py > $out/bench_default.log 2>$out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
python3 - $out/bench_default.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r = d["roofline"]
print(d["ms_per_step"], d["value"], "frac", r["frac"], "counters:", r["counters_source"], r["counters_missing_because"])
PY
bash tools/other_configs.sh $out/other_configs.jsonl || exit 1
PT_BENCH_REHEARSAL=1 timeout -k 10 600 python3 -u bench.py --gpus 2 --steps 2 --warmup 1 > $out/bench_two_ranks_one_gpu.log 2>&1 || { tail -5 $out/bench_two_ranks_one_gpu.log; exit 1; }
PT_BENCH_REHEARSAL=1 timeout -k 10 600 python3 -u bench.py --gpus 8 --steps 2 --warmup 1 > $out/bench_eight_ranks_one_gpu.log 2>&1 || { tail -5 $out/bench_eight_ranks_one_gpu.log; exit 1; }
grep -h -o '"n_gpus": [0-9]*, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $out/bench_*_ranks_one_gpu.log
