#!/bin/bash
# first GPU pass of the round: the GPU test-suite, then the normal-table A/B and the dense workload
out=gpurun_out/r3a; mkdir -p $out
timeout -k 10 840 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh "" "--tune 17=0" > $out/ab_normal_table.txt 2>&1 || exit 1
cat $out/ab_normal_table.txt
timeout -k 10 200 python3 -u bench.py --scene dragon_dense --steps 2 --warmup 1 > $out/dense.log 2>&1 || { tail -5 $out/dense.log; exit 1; }
tail -1 $out/dense.log
