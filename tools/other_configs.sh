#!/bin/bash
# usage: tools/other_configs.sh <out.jsonl>   (on the GPU box, from the repo root)
# One bench line per configuration of the benchmark list other than the headline one, plus a rank's share of the
# headline frame at 8, 4 and 2 GPUs (64 / 128 / 256 of the 512 samples per pixel) on this one GPU.
out=$1
: > $out
run() { timeout -k 10 900 python3 bench.py "$@" 2>/dev/null | grep '^{' >> $out || { echo "failed: $*"; exit 1; }; }
run --scene balls --width 960 --height 540 --spp 32 --steps 20 --warmup 5
run --scene balls --spp 512 --steps 2 --warmup 1
run --scene bunny --spp 512 --steps 2 --warmup 1
run --scene city --spp 2048 --steps 2 --warmup 1
run --scene city --spp 256 --steps 4 --warmup 1
run --scene synth --instances 116 --width 3840 --height 2160 --spp 128 --steps 2 --warmup 1
run --scene dragon_dense --spp 512 --steps 2 --warmup 1
run --scene dragon_dense --spp 64 --steps 6 --warmup 2 --no-cpu-baseline
run --spp 64 --steps 20 --warmup 3 --no-cpu-baseline
run --spp 128 --steps 10 --warmup 2 --no-cpu-baseline
run --spp 256 --steps 6 --warmup 2 --no-cpu-baseline
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("%-70s %9.3f ms %9.1f Mrays/s" % (d["metric"][40:], d["ms_per_step"], d["value"]))
PY
