#!/bin/bash
out=gpurun_out/r3d; mkdir -p $out
timeout -k 10 840 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for sc in dragon dragon_dense city bunny balls; do
  timeout -k 10 300 python3 -u bench.py --scene $sc --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sc', d['ms_per_step'], d['value'], 'mode', d['config']['render_mode'])" || exit 1
done | tee $out/auto_modes.txt
